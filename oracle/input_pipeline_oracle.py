"""CPU restatement (numpy) of the image side of the reference's training / test input pipeline - TEST INFRASTRUCTURE ONLY.

Follows, in the order configs/_base_/datasets/nuscenes_detection_r640_clr_fusion.py:18-33 applies them to one sample:
  Normalize      mmdet/datasets/pipelines/transforms.py:706-753  -> mmcv.imnormalize (per sensor mean/std, to_rgb)
  RandomFlip     transforms.py:440-466  -> mmcv.imflip(direction='horizontal') on EVERY img_field (camera, lidar, radar)
  Pad            transforms.py:649-664  -> mmcv.impad_to_multiple(size_divisor=32, pad_val=0): zeros bottom / right
  RandomDrop     transforms.py:487-514  -> a dropped sensor becomes all zeros (after padding)
  DefaultFormatBundle  formating.py:212-227 -> HWC -> CHW (a 2-D image gets a channel axis), stacked per batch
Resize (camera only, cv2 bilinear) and the file decoders stay on the host and are not restated: the pipeline starts
from the resized float32 / uint8 HWC images.  Random decisions are INPUTS (flip flag per sample, drop flag per sample and
sensor): the reference draws them from numpy / python RNG on the host.

PARITY UNPINNED for the Normalize arithmetic: mmcv (pinned by the reference's requirements: mmcv-full 1.3.17) implements
imnormalize with cv2.subtract / cv2.multiply on a float32 copy - float32 (x - float32(mean)) * float32(1 / float64(std)),
channel swap first when to_rgb - and neither mmcv nor cv2 exists in this image, so that rounding could not be checked
against the real library; flip, pad, drop and the layout change are exact data movement.
Only tests/ may import this.
"""
import numpy as np


def imnormalize(img, mean, std, to_rgb):
    """mmcv.image.photometric.imnormalize_ restated (see the header for what is not pinned)."""
    img = np.asarray(img).astype(np.float32)             # img.copy().astype(np.float32)
    if img.ndim == 2:
        img = img[..., None]
    # transforms.py:720-721: Normalize keeps np.float32(mean) / np.float32(std); imnormalize_ widens THOSE to float64
    mean32 = np.asarray(mean, dtype=np.float32).reshape(1, -1)
    stdinv32 = (1.0 / np.asarray(std, dtype=np.float32).astype(np.float64).reshape(1, -1)).astype(np.float32)
    if to_rgb:
        img = img[..., ::-1]                             # cv2.cvtColor(img, cv2.COLOR_BGR2RGB)
    out = (img - mean32).astype(np.float32)              # cv2.subtract
    return (out * stdinv32).astype(np.float32)           # cv2.multiply


def imflip_horizontal(img):
    return np.flip(img, axis=1)                          # mmcv.imflip(direction='horizontal')


def impad_to_multiple(img, divisor, pad_val=0):
    h, w = img.shape[:2]
    ph, pw = int(np.ceil(h / divisor)) * divisor, int(np.ceil(w / divisor)) * divisor
    out = np.full((ph, pw) + img.shape[2:], pad_val, dtype=img.dtype)
    out[:h, :w] = img
    return out


def run_sample(imgs, cfgs, flip, drop, size_divisor=32):
    """imgs: {key: HWC (or HW) array}; cfgs: {key: dict(mean, std, to_rgb)}; flip: bool; drop: {key: bool}
    -> {key: CHW float32}, in the reference's order of operations."""
    out = {}
    for key, img in imgs.items():
        c = cfgs[key]
        x = imnormalize(img, c['mean'], c['std'], c.get('to_rgb', False))
        if flip:
            x = imflip_horizontal(x)
        x = impad_to_multiple(x, size_divisor)
        if drop.get(key, False):
            x = np.zeros_like(x)
        out[key] = np.ascontiguousarray(x.transpose(2, 0, 1))
    return out


def run_batch(batch, cfgs, flips, drops, size_divisor=32):
    """batch: {key: [B] list/array of HWC}; flips: [B] bools; drops: {key: [B] bools} -> {key: (B, C, Hp, Wp) float32}."""
    B = len(flips)
    per = [run_sample({k: v[b] for k, v in batch.items()}, cfgs, bool(flips[b]),
                      {k: bool(d[b]) for k, d in drops.items()}, size_divisor) for b in range(B)]
    return {k: np.stack([p[k] for p in per]) for k in batch}
