"""GroupNorm (norm_cfg = dict(type='GN', num_groups=...)): no reference config uses it, but every norm layer of the path is
built by mmcv.build_norm_layer(self.norm_cfg, ...) (hrnet.py:338-339,438,459,476; resnet.py:161-164; hrformer.py:269,278,281,
518,542,552; hrfuser_hrformer_based.py:388,397), which accepts it.  This container-only script builds the HRFuser-T nuScenes
backbone with GN(2 groups) in place of BN, checks oracle.HRFuserOracle BIT-EXACT against the reference class imported through
ref_loader (state-dict keys - the postfixed layers become gn1 / gn2 / gn3 -, eval and train outputs, input and parameter
gradients) and writes
  tests/golden/hrfuser_gn_cfg.json   the backbone kwargs + state manifest (data)
  tests/golden/hrfuser_gn.npz        reference outputs at 2x64x96 and fp64-free gradient digests,
and the same for the convolutional sibling HRFuserHRNetBased (BasicBlock trunk): tests/golden/hrfuser_hrnet_gn{_cfg.json,.npz}."""
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
    cfg = R.backbone_cfg('cascade_rcnn_hrfuser_t_1x_nus_r640_l_r_fusion_bn')
    cfg['norm_cfg'] = dict(type='GN', num_groups=2, requires_grad=True)
    return cfg


def make_hrnet_cfg():
    """the convolutional sibling (BasicBlock trunk: resnet.py:34-49 names its norms <abbr>1 / <abbr>2) with GN"""
    from make_golden_hrnet_based import make_cfg as hrnet_cfg
    cfg = hrnet_cfg()
    cfg['norm_cfg'] = dict(type='GN', num_groups=2, requires_grad=True)
    return cfg


def build_pair(cfg):
    R.install()
    kw = copy.deepcopy(cfg)
    kind = kw.pop('type')
    if kind == 'HRFuserHRNetBased':
        import importlib
        mod = importlib.import_module('mmdet.models.backbones.hrfuser_hrnet_based')
        return mod.HRFuserHRNetBased(**copy.deepcopy(kw)), O.HRFuserHRNetOracle(**copy.deepcopy(kw))
    return R.build_reference(copy.deepcopy(cfg)), O.HRFuserOracle(**kw)


def main():
    one(make_cfg(), 'hrfuser_gn')
    one(make_hrnet_cfg(), 'hrfuser_hrnet_gn')


def one(cfg, stem):
    ref, orc = build_pair(cfg)
    assert list(ref.state_dict().keys()) == list(orc.state_dict().keys()), set(ref.state_dict()) ^ set(orc.state_dict())
    assert any('.gn1.' in k for k in ref.state_dict()) and not any('.bn1.' in k for k in ref.state_dict())
    O.seeded_fill_(ref, 0)
    O.seeded_fill_(orc, 0)
    disable_stochastic(ref)
    disable_stochastic(orc)
    x, mods = O.seeded_inputs(2, 64, 96, [3, 3], seed=1)
    arrays = {}
    for mode in (False, True):
        ref.train(mode)
        orc.train(mode)
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
        arrays[f'B2_64x96/{tag}/dx'] = xa.grad.detach().numpy()
        arrays[f'B2_64x96/{tag}/gradnorms'] = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0
                                                        for _, p in ref.named_parameters()])
        ref.zero_grad(set_to_none=True)
        orc.zero_grad(set_to_none=True)
        print('mode', mode, 'oracle == reference bit for bit', [tuple(t.shape) for t in ya])
    manifest = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in ref.state_dict().items()]
    with open(os.path.join(OUT, stem + '_cfg.json'), 'w') as fh:
        json.dump({'cfg': json.loads(json.dumps(cfg, default=lambda o: list(o))), 'n_params': sum(p.numel() for p in ref.parameters()),
                   'entries': manifest}, fh, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(OUT, stem + '.npz'), **arrays)
    print(stem, 'written', len(arrays), 'arrays;', len(manifest), 'state entries')


if __name__ == '__main__':
    main()
