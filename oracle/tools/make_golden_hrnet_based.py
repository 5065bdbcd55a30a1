"""HRFuserHRNetBased (mmdet/models/backbones/hrfuser_hrnet_based.py:23-315) has no reference config; this container-only
script builds one in the spirit of the HRFuser-T config (HRNet-w18 trunk: BASIC blocks, widths 18/36/72/144, one module per
stage; modality stages B / C of BASIC blocks at width 18; the fusion blocks of the T config), checks
oracle.HRFuserHRNetOracle BIT-EXACT against the reference class imported through ref_loader (state-dict keys, eval and
train outputs, input and parameter gradients) and writes
  tests/golden/hrfuser_hrnet_cfg.json   the backbone kwargs (data)
  tests/golden/hrfuser_hrnet.npz        reference outputs (eval / train) at 2x64x96, fp64 gradient digests, state manifest."""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_loader as R                # noqa: E402
import hrfuser_oracle as O            # noqa: E402
from make_golden import OUT, disable_stochastic   # noqa: E402


def make_cfg():
    base = R.backbone_cfg('cascade_rcnn_hrfuser_t_1x_nus_r640_l_r_fusion_bn')
    ex = base['extra']
    basic = lambda nb: dict(num_modules=1, num_branches=nb, block='BASIC', num_blocks=(2,) * nb,
                            num_channels=(18, 36, 72, 144)[:nb])
    extra = dict(
        stage1=dict(num_modules=1, num_branches=1, block='BOTTLENECK', num_blocks=(2,), num_channels=(64,)),
        stage2=basic(2), stage3=basic(3), stage4=basic(4),
        LidarStageA=dict(num_modules=1, num_branches=1, block='BOTTLENECK', num_blocks=(2,), num_channels=(64,)),
        LidarStageB=basic(1), LidarStageC=basic(1), LidarStageD=None,
        ModFusionA=copy.deepcopy(ex['ModFusionA']), ModFusionB=copy.deepcopy(ex['ModFusionB']),
        ModFusionC=copy.deepcopy(ex['ModFusionC']), ModFusionD=None)
    return dict(type='HRFuserHRNetBased', extra=extra, norm_cfg=dict(type='BN', requires_grad=True, momentum=0.1),
                transformer_norm_cfg=dict(type='LN', eps=1e-6), norm_eval=False, num_fused_modalities=2)


def main():
    R.install()
    import importlib
    mod = importlib.import_module('mmdet.models.backbones.hrfuser_hrnet_based')
    cfg = make_cfg()
    kw = copy.deepcopy(cfg)
    kw.pop('type')
    ref = mod.HRFuserHRNetBased(**copy.deepcopy(kw))
    orc = O.HRFuserHRNetOracle(**copy.deepcopy(kw))
    assert list(ref.state_dict().keys()) == list(orc.state_dict().keys()), set(ref.state_dict()) ^ set(orc.state_dict())
    O.seeded_fill_(ref, 0)
    O.seeded_fill_(orc, 0)
    disable_stochastic(ref)
    disable_stochastic(orc)
    x, mods = O.seeded_inputs(2, 64, 96, [3, 3], seed=1)
    arrays = {}
    for mode in (False, True):
        ref.train(mode)
        orc.train(mode)
        sd0 = copy.deepcopy(ref.state_dict())
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya = ref(xa, [m.clone() for m in mods])
        yb = orc(xb, [m.clone() for m in mods])
        for a, b in zip(ya, yb):
            assert float((a - b).abs().max()) == 0.0
        g = torch.Generator().manual_seed(5)
        cots = [torch.randn(t.shape, generator=g) for t in ya]
        sum((t * c).sum() for t, c in zip(ya, cots)).backward()
        sum((t * c).sum() for t, c in zip(yb, cots)).backward()
        assert float((xa.grad - xb.grad).abs().max()) == 0.0
        for (n, p), (_, q) in zip(ref.named_parameters(), orc.named_parameters()):
            assert (p.grad is None) == (q.grad is None), n
            if p.grad is not None:
                assert float((p.grad - q.grad).abs().max()) == 0.0, n
        tag = 'train' if mode else 'eval'
        for i, t in enumerate(ya):
            arrays[f'B2_64x96/{tag}/out{i}'] = t.detach().numpy()
        ref.load_state_dict(sd0)
        orc.load_state_dict(sd0)
        ref.zero_grad(set_to_none=True)
        orc.zero_grad(set_to_none=True)
        print('mode', mode, 'oracle == reference bit for bit', [tuple(t.shape) for t in ya])
    manifest = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in ref.state_dict().items()]
    with open(os.path.join(OUT, 'hrfuser_hrnet_cfg.json'), 'w') as fh:
        json.dump({'cfg': json.loads(json.dumps(cfg, default=lambda o: list(o))), 'n_params': sum(p.numel() for p in ref.parameters()),
                   'entries': manifest}, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, 'hrfuser_hrnet.npz'), **arrays)
    print('written', len(arrays), 'arrays;', len(manifest), 'state entries')


if __name__ == '__main__':
    main()
