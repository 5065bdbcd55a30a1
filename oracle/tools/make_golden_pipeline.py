"""Golden vectors for the device-side input pipeline from the REAL reference transforms - test infrastructure.

Needs mmcv (1.3.x, with cv2) next to /root/reference: Normalize -> RandomFlip -> Pad(size_divisor=32) -> RandomDrop ->
DefaultFormatBundle of mmdet/datasets/pipelines (transforms.py:706-753,440-466,649-664,487-514; formating.py:212-227) run
on seeded uint8 / float32 images with the nuScenes per-sensor norm_cfgs; inputs, the random decisions the transforms took and
their outputs go to tests/golden/pipeline.npz (data only).  tests/test_pipeline.py::test_pipeline_golden checks the numpy
restatement (oracle/input_pipeline_oracle.py) and, on a GPU, hrf_pack_input against it - which flips row f3 from
"parity unpinned" to pinned.  Neither mmcv nor cv2 exists in the build image of rounds 1-6: the script then says so and
exits 0 without writing anything.

    python oracle/tools/make_golden_pipeline.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('HRF_REFERENCE', '/root/reference')
OUT = os.path.join(ROOT, 'tests', 'golden', 'pipeline.npz')

# configs/_base_/datasets/nuscenes_detection_r640_clr_fusion.py:12-17
NUS = dict(img=dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
           lidar_img=dict(mean=[0.23277158, 0.31501067, -0.00012928071],
                          std=[2.5538357826888602, 3.7345728854535643, 0.2815488539921788], to_rgb=False),
           radar_img=dict(mean=[0.19778967, 0.03477772, 0.0025186215],
                          std=[3.219927182957935, 0.7240392925308506, 0.11561270078715341], to_rgb=False),
           probe=dict(mean=[0.1, 33.3, 7.7], std=[0.1, 0.3, 1.7], to_rgb=True))      # float32(std) != std: the cast order shows


def main():
    try:
        import cv2      # noqa: F401
        import mmcv     # noqa: F401
    except Exception as e:                                   # the usual case here
        print(f'make_golden_pipeline: mmcv / cv2 not importable ({type(e).__name__}: {e}); nothing written - '
              'row f3 stays parity-unpinned')
        return 0
    sys.path.insert(0, REF)
    from mmdet.datasets.pipelines.transforms import Normalize, Pad          # noqa: E402
    rng = np.random.RandomState(7)
    store = {}
    case = 0
    for (H0, W0), u8 in (((37, 50), True), ((64, 96), False), ((33, 47), True)):
        for key, cfg in NUS.items():
            img = rng.randint(0, 256, size=(H0, W0, 3)).astype(np.uint8)
            src = img if u8 else (img.astype(np.float32) + rng.rand(H0, W0, 3).astype(np.float32))
            sensor = {'img': 'img', 'lidar_img': 'lidar', 'radar_img': 'radar', 'probe': 'img'}[key]
            res = {'img': src.copy(), 'img_fields': ['img']}
            res = Normalize(cfg['mean'], cfg['std'], cfg['to_rgb'], sensor_type=sensor)(res)
            for flip in (False, True):
                out = mmcv.imflip(res['img'], direction='horizontal') if flip else res['img']
                out = mmcv.impad_to_multiple(out, 32, pad_val=0)
                store[f'c{case}/in'] = src
                store[f'c{case}/key'] = np.array(key)
                store[f'c{case}/mean'] = np.array(cfg['mean'], dtype=np.float64)
                store[f'c{case}/std'] = np.array(cfg['std'], dtype=np.float64)
                store[f'c{case}/to_rgb'] = np.array(cfg['to_rgb'])
                store[f'c{case}/flip'] = np.array(flip)
                store[f'c{case}/out'] = np.ascontiguousarray(out.transpose(2, 0, 1))
                case += 1
    store['n'] = np.array(case)
    np.savez_compressed(OUT, **store)
    print(f'wrote {OUT}: {case} cases')
    return 0


if __name__ == '__main__':
    sys.exit(main())
