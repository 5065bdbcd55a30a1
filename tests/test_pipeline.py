"""Device-side input pipeline: the numpy restatement of the reference's Normalize / RandomFlip / Pad /
RandomDrop / DefaultFormatBundle chain (oracle/input_pipeline_oracle.py - its Normalize rounding is NOT pinned against
mmcv/cv2, see its header) and the one-pass HIP kernel behind hrfuser_amd.DeviceInputPipeline, bit-exact vs that oracle."""
import numpy as np
import pytest
import torch

import helpers as T
import input_pipeline_oracle as P

CFGS = {   # configs/_base_/datasets/nuscenes_detection_r640_clr_fusion.py:12-17
    'img': dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True),
    'lidar_img': dict(mean=[0.23277158, 0.31501067, -0.00012928071],
                      std=[2.5538357826888602, 3.7345728854535643, 0.2815488539921788], to_rgb=False),
    'radar_img': dict(mean=[0.19778967, 0.03477772, 0.0025186215],
                      std=[3.219927182957935, 0.7240392925308506, 0.11561270078715341], to_rgb=False)}


def test_oracle_semantics():
    img = np.arange(2 * 3 * 3, dtype=np.float32).reshape(2, 3, 3)                 # H=2, W=3, BGR
    cfg = dict(mean=[1.0, 2.0, 3.0], std=[2.0, 4.0, 8.0], to_rgb=True)
    out = P.run_sample({'img': img}, {'img': cfg}, flip=False, drop={}, size_divisor=4)['img']
    assert out.shape == (3, 4, 4) and out.dtype == np.float32                      # CHW, padded to multiples of 4
    # channel 0 of the output is R = input channel 2, normalised with mean[0] / std[0]
    assert np.array_equal(out[0, :2, :3], (img[..., 2] - 1.0) * np.float32(0.5))
    assert np.array_equal(out[2, :2, :3], (img[..., 0] - 3.0) * np.float32(0.125))
    assert not out[:, 2:, :].any() and not out[:, :, 3:].any()                     # zero padding AFTER normalisation
    fl = P.run_sample({'img': img}, {'img': cfg}, flip=True, drop={}, size_divisor=4)['img']
    assert np.array_equal(fl[:, :2, :3], out[:, :2, :3][:, :, ::-1])               # flip inside the valid region only
    dr = P.run_sample({'img': img}, {'img': cfg}, flip=False, drop={'img': True}, size_divisor=4)['img']
    assert dr.shape == (3, 4, 4) and not dr.any()
    g = P.run_sample({'g': img[..., 0]}, {'g': dict(mean=[1.0], std=[2.0])}, False, {}, 4)['g']
    assert g.shape == (1, 4, 4)                                                    # 2-D image -> one channel


def test_normalize_constants_are_float32_first():
    """transforms.py:720-721 stores np.float32(mean) / np.float32(std); mmcv.imnormalize_ widens THOSE: the reciprocal is
    1 / float64(float32(std)), not 1 / float64(std) - different for std = 1.7, equal for the nuScenes constants"""
    cfg = dict(mean=[0.1], std=[1.7], to_rgb=False)
    x = np.full((1, 1, 1), 3.0, dtype=np.float32)
    out = P.imnormalize(x, cfg['mean'], cfg['std'], False)
    want = (np.float32(3.0) - np.float32(0.1)) * np.float32(1.0 / np.float64(np.float32(1.7)))
    assert out.dtype == np.float32 and out[0, 0, 0] == want
    assert np.float32(1.0 / np.float64(np.float32(1.7))) != np.float32(1.0 / 1.7)
    from hrfuser_amd import DeviceInputPipeline
    mean, stdinv = DeviceInputPipeline({'p': cfg})._constants('p', torch.device('cpu'))
    assert float(stdinv[0]) == float(np.float32(1.0 / np.float64(np.float32(1.7)))) and float(mean[0]) == float(np.float32(0.1))


def _golden_cases():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pipeline.npz')
    if not os.path.exists(path):
        pytest.skip('tests/golden/pipeline.npz absent: oracle/tools/make_golden_pipeline.py needs mmcv + cv2 (row f3 unpinned)')
    z = np.load(path)
    for i in range(int(z['n'])):
        yield (z[f'c{i}/in'], dict(mean=z[f'c{i}/mean'].tolist(), std=z[f'c{i}/std'].tolist(), to_rgb=bool(z[f'c{i}/to_rgb'])),
               bool(z[f'c{i}/flip']), z[f'c{i}/out'])


def test_pipeline_golden_oracle():
    """the numpy restatement against outputs of the REAL Normalize / imflip / impad_to_multiple (when the fixture exists)"""
    for src, cfg, flip, want in _golden_cases():
        got = P.run_sample({'s': src}, {'s': cfg}, flip=flip, drop={}, size_divisor=32)['s']
        assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_pipeline_golden_gpu():
    from hrfuser_amd import DeviceInputPipeline
    dev = T.use_backend('hip')
    for src, cfg, flip, want in _golden_cases():
        pipe = DeviceInputPipeline({'s': cfg})
        got = pipe({'s': torch.from_numpy(src[None]).to(dev)}, flip=torch.tensor([flip]).to(dev))['s']
        assert np.array_equal(got[0].cpu().numpy(), want)


def _batch(B, H0, W0, seed=0, u8=True):
    rng = np.random.default_rng(seed)
    cam = rng.integers(0, 256, (B, H0, W0, 3), dtype=np.uint8) if u8 else \
        rng.uniform(0, 255, (B, H0, W0, 3)).astype(np.float32)
    return {'img': cam, 'lidar_img': rng.normal(0, 3, (B, H0, W0, 3)).astype(np.float32),
            'radar_img': rng.normal(0, 2, (B, H0, W0, 3)).astype(np.float32)}


def run_kernel_vs_oracle(backend, B, H0, W0, u8):
    from hrfuser_amd import DeviceInputPipeline
    dev = T.use_backend(backend)
    try:
        batch = _batch(B, H0, W0, u8=u8)
        flips = np.array([b % 2 == 1 for b in range(B)])
        drops = {'img': np.zeros(B, bool), 'lidar_img': np.array([b == 0 for b in range(B)]),
                 'radar_img': np.array([b == B - 1 for b in range(B)])}
        ref = P.run_batch(batch, CFGS, flips, drops)
        pipe = DeviceInputPipeline(CFGS)
        got = pipe({k: torch.from_numpy(v).to(dev) for k, v in batch.items()}, flip=torch.from_numpy(flips).to(dev),
                   drop={k: torch.from_numpy(v).to(dev) for k, v in drops.items()})
        for k in batch:
            y = got[k]
            assert tuple(y.shape) == ref[k].shape
            assert y.is_contiguous(memory_format=torch.channels_last)
            assert np.array_equal(y.cpu().numpy(), ref[k]), k                       # bit-exact
        return got
    finally:
        T.use_backend('hip')


@pytest.mark.parametrize('u8', [True, False])
def test_pack_input_emul(u8):
    run_kernel_vs_oracle('emul', 3, 9, 13, u8)


@pytest.mark.gpu
def test_pack_input_gpu_fullsize():
    run_kernel_vs_oracle('hip', 2, 360, 640, True)       # -> (2, 3, 384, 640)


@pytest.mark.gpu
def test_backbone_reads_pipeline_output_in_place_gpu():
    """The pipeline's channels-last tensors go straight into the backbone (no layout copy): same features and the same
    input gradient as with NCHW-contiguous copies of the same values."""
    dev = T.use_backend('hip')
    net, _, _ = T.build_pair('t_nus', dev)
    net.train()
    got = run_kernel_vs_oracle('hip', 1, 60, 90, True)    # -> (1, 3, 64, 96)
    xs = [got[k].detach() for k in ('img', 'lidar_img', 'radar_img')]
    xa = [t.clone(memory_format=torch.preserve_format).requires_grad_(True) for t in xs]
    xb = [t.contiguous().clone().requires_grad_(True) for t in xs]
    assert not xa[0].is_contiguous() and xb[0].is_contiguous()
    ya = net(xa[0], xa[1:])
    sum(y.sum() for y in ya).backward()
    ga = [t.grad.clone() for t in xa]
    net.zero_grad(set_to_none=False)
    yb = net(xb[0], xb[1:])
    sum(y.sum() for y in yb).backward()
    for a, b in zip(ya, yb):
        assert T.relmax(a, b) < 1e-4
    for a, b in zip(ga, xb):
        assert T.relmax(a, b.grad) < 2e-3
