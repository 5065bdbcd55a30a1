"""Detector-level caller of the hot path: what `TwoStageDetector` does around backbone + neck.

Mirrors mmdet/models/detectors/two_stage.py: `combine_mod_imgs` (:9-19), `extract_feat` (:76-84), the sensor-keyword
entry of `forward_train` (:156-160) and the test-time list unwrapping of `simple_test` (:211-220).  Under a real
mmdet installation nothing here is needed - `TwoStageDetector` builds `backbone` / `neck` through the registries that
`import hrfuser_amd` re-populates and calls them exactly like this class does; stand-alone (this image
has no mmcv) `FeatureExtractor` is that caller, and `ExtractTrainer` is the data-parallel training step over BOTH
engines in one hipGraph: backbone tape -> neck tape -> neck backward -> backbone backward -> gradient exchange ->
fused AdamW on the two flat parameter arenas.
"""
import torch
import torch.nn as nn

from . import _lib
from . import runtime as R
from .registry import BACKBONES, NECKS
from .trainer import Trainer


def combine_mod_imgs(lidar_img=None, radar_img=None, gated_img=None):
    """two_stage.py:9-19 - fixed sensor order lidar, radar, gated; None when no modality is given."""
    mod_imgs = [m for m in (lidar_img, radar_img, gated_img) if m is not None]
    return mod_imgs if mod_imgs else None


class FeatureExtractor(nn.Module):
    """backbone (+ neck) of a `TwoStageDetector` config: `FeatureExtractor(cfg.model.backbone, cfg.model.neck)`."""

    def __init__(self, backbone, neck=None):
        super().__init__()
        self.backbone = BACKBONES.build(backbone) if isinstance(backbone, dict) else backbone      # two_stage.py:41
        if neck is not None:
            self.neck = NECKS.build(neck) if isinstance(neck, dict) else neck                       # two_stage.py:43-44

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None                                      # base.py:27-29

    def extract_feat(self, img, mod_imgs=None):
        """two_stage.py:76-84."""
        if mod_imgs is not None:
            x = self.backbone(img, mod_imgs)
        else:
            x = self.backbone(img)             # camera-only backbones (HRFormer); HRFuser raises TypeError like the reference
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward(self, img, lidar_img=None, radar_img=None, gated_img=None):
        """The feature part of forward_train (two_stage.py:156-160): sensors by keyword."""
        return self.extract_feat(img, mod_imgs=combine_mod_imgs(lidar_img, radar_img, gated_img))

    def simple_test_feats(self, img, lidar_img=None, radar_img=None, gated_img=None):
        """The feature part of simple_test (two_stage.py:211-220): test pipelines wrap every modality in a list
        (one entry per augmentation); `mod_imgs[i] = mod_imgs[i][0]` unwraps the single-scale case."""
        mod_imgs = combine_mod_imgs(lidar_img, radar_img, gated_img)
        if mod_imgs:
            mod_imgs = [m[0] for m in mod_imgs]
        return self.extract_feat(img, mod_imgs=mod_imgs)


class ExtractTrainer:
    """Training step over backbone + neck on the explicit tapes (no torch.autograd), capturable into one hipGraph.

    Same contract as `Trainer` (synthetic loss L = sum_i <pyramid_i, cot_i>, flat-arena gradient all-reduce, fused
    AdamW with the reference's `paramwise_cfg` decay mask), with the neck's arena exchanged and stepped beside the
    backbone's."""

    def __init__(self, feats, group=None, world_size=1, **opt):
        assert feats.with_neck
        self.feats = feats
        self.tb = Trainer(feats.backbone, group=group, world_size=world_size, **opt)
        self.tn = Trainer(feats.neck, group=group, world_size=world_size, **opt)
        self.group, self.world = group, world_size
        self.graph = None

    def _step_impl(self, x, mods, cots):
        net, neck = self.feats.backbone, self.feats.neck
        eb, en = net._engine(), neck._engine()
        if not self.tb._ready:
            self.tb._setup(x.device)
            self.tn._setup(x.device)
        R.gpu_zero_(eb.flat_g)
        R.gpu_zero_(en.flat_g)
        cb, outs, _ = net._execute((x,) + tuple(mods), True)
        en.ready(x.device)
        en.begin_forward(True)
        cn = R.Ctx(neck, True, True)
        with torch.no_grad():
            srcs = [R.Act(o.t, True) for o in outs]              # the backbone's NHWC maps, in place
            pyr = neck._run(cn, srcs)
        for o, c in zip(pyr, cots):
            o.grad = R.gpu_clone(c)
        cn.run_backward()
        for o, s_ in zip(outs, srcs):
            o.grad = s_.grad
        cb.run_backward()
        L = _lib.lib()
        for tr, eng in ((self.tb, eb), (self.tn, en)):
            if self.world > 1 or tr.force:
                import torch.distributed as dist
                for a, b in tr.buckets(eng.flat_g.numel()):
                    dist.all_reduce(eng.flat_g[a:b], group=self.group)
            s = _lib.stream_ptr()
            L.hrf_adamw_tick(tr.state, tr.betas[0], tr.betas[1], s)
            L.hrf_adamw(eng.flat_p, eng.flat_g, tr.m, tr.v, tr.wd_mask, eng.flat_p.numel(), tr.lr, tr.betas[0],
                        tr.betas[1], tr.eps, tr.wd, tr.state, 1.0 / self.world, s)
        net.params_updated()
        neck.params_updated()
        return pyr

    def step(self, x, mods, cots):
        return self._step_impl(x, mods, cots)

    def capture(self, x, mods, cots, warmup=2):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_impl(x, mods, cots)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if self.world > 1 or self.tb.force:
            import os
            import time
            time.sleep(float(os.environ.get('HRF_CAPTURE_SETTLE', '1.0')))       # see Trainer.capture
        g = torch.cuda.CUDAGraph()
        with R.gc_paused(), torch.cuda.graph(g, capture_error_mode='thread_local'):
            self._graph_outs = self._step_impl(x, mods, cots)
        self.graph = g
        return g

    def replay(self):
        self.graph.replay()


def make_pyramid_cotangents(feats, x, mods, seed=5):
    """Fixed random cotangents (NHWC) for the pyramid outputs, shaped by a dry eval forward."""
    was_b, was_n = feats.backbone.training, feats.neck.training
    feats.eval()
    with torch.no_grad():
        ys = feats.extract_feat(x, list(mods))
    feats.backbone.train(was_b)
    feats.neck.train(was_n)
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(tuple(y.permute(0, 2, 3, 1).shape), generator=g).to(x.device) / y.numel() for y in ys]
