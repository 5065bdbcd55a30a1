"""Device-side input pipeline: what the reference does per sample on the host with numpy / cv2 between
the image decoders and the backbone - Normalize, RandomFlip, Pad(size_divisor=32), RandomDrop, DefaultFormatBundle
(mmdet/datasets/pipelines/transforms.py:706-753,440-466,649-664,487-514; formating.py:212-227; configured in
configs/_base_/datasets/nuscenes_detection_r640_clr_fusion.py:12-33) - as ONE kernel launch per sensor on the batch
(`hrf_pack_input`), writing channels-last storage that the HIP backbone consumes without a layout copy.

The random decisions (flip per sample, drop per sample and sensor) remain host-side inputs; Resize of the camera image
and the file decoders are not part of this module.
"""
import numpy as np
import torch

from . import _lib


class DeviceInputPipeline:
    """sensors: {key: dict(mean=[..], std=[..], to_rgb=bool)} in the order the detector takes them
    (img, lidar_img, radar_img[, gated_img]) - the `*_norm_cfg` dicts of the dataset config."""

    def __init__(self, sensors, size_divisor=32):
        self.sensors = {k: dict(v) for k, v in sensors.items()}
        self.size_divisor = int(size_divisor)
        self._const = {}

    def _constants(self, key, dev):
        ent = self._const.get((key, dev))
        if ent is None:
            c = self.sensors[key]
            # Normalize.__init__ stores float32 arrays (transforms.py:720-721); mmcv.imnormalize_ then forms float64(mean) and
            # 1 / float64(std) FROM THOSE: the config value is rounded to float32 FIRST
            mean = np.asarray(c['mean'], dtype=np.float32)
            stdinv = (1.0 / np.asarray(c['std'], dtype=np.float32).astype(np.float64)).astype(np.float32)
            ent = self._const[(key, dev)] = (torch.from_numpy(mean).to(dev), torch.from_numpy(stdinv).to(dev))
        return ent

    def __call__(self, batch, flip=None, drop=None):
        """batch: {key: (B, H0, W0, C) or (B, H0, W0) tensor on the GPU, float32 or uint8}; flip: (B,) bool tensor or None;
        drop: {key: (B,) bool tensor} or None -> {key: logical (B, C, Hp, Wp) float32 tensor, channels-last memory}."""
        L = _lib.lib()
        out = {}
        d = self.size_divisor
        for key, img in batch.items():
            if key not in self.sensors:
                raise KeyError(f'no norm_cfg for sensor {key!r}')
            if img.dim() == 3:
                img = img.unsqueeze(-1)                  # formating.py:224: 2-D images get a channel axis
            if img.dtype not in (torch.float32, torch.uint8):
                raise TypeError(f'{key}: float32 or uint8 images expected, got {img.dtype}')
            img = img.contiguous()
            B, H0, W0, C = img.shape
            mean, stdinv = self._constants(key, img.device)
            if mean.numel() != C:
                raise ValueError(f'{key}: {C} channels but mean/std have {mean.numel()} entries')
            Hp, Wp = -(-H0 // d) * d, -(-W0 // d) * d
            y = torch.empty(B, Hp, Wp, C, device=img.device, dtype=torch.float32)
            fl = flip.to(torch.uint8).contiguous() if flip is not None else None
            dr = drop[key].to(torch.uint8).contiguous() if (drop is not None and key in drop) else None
            L.hrf_pack_input(img, 1 if img.dtype == torch.uint8 else 0, B, H0, W0, C, mean, stdinv,
                             1 if self.sensors[key].get('to_rgb', False) else 0, fl, dr, y, Hp, Wp, _lib.stream_ptr())
            out[key] = y.permute(0, 3, 1, 2)
        return out
