"""Golden vectors of the reference plain HRFormer backbone (container-only: imports /root/reference through
oracle/tools/ref_loader.py).  Checks oracle.HRFormerOracle bit-exact against mmdet's `HRFormer` built from
configs/hrformer/*.py (eval + train-mode outputs, input / parameter gradients) and writes
tests/golden/hrformer_cfgs.json + tests/golden/hrformer_t.npz."""
import copy
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, HERE)
import hrfuser_oracle as O          # noqa: E402
import ref_loader as R              # noqa: E402
from make_golden import jsonable, disable_stochastic      # noqa: E402

CFG = {'hrformer_t': 'cascade_rcnn_hrformer_t_1x_nus_r640', 'hrformer_t_bn': 'cascade_rcnn_hrformer_t_1x_nus_r640_bn',
       'hrformer_b_bn': 'cascade_rcnn_hrformer_b_1x_nus_r640_bn'}


def main():
    R.install()
    from mmdet.models.builder import BACKBONES
    cfgs = {}
    for tag, name in CFG.items():
        path = os.path.join(R.REF_ROOT, 'configs/hrformer', name + '.py')
        cfgs[tag] = copy.deepcopy(R.load_cfg(path)['model']['backbone'])
    with open(os.path.join(ROOT, 'tests', 'golden', 'hrformer_cfgs.json'), 'w') as fh:
        json.dump(jsonable(cfgs), fh, indent=1, sort_keys=True)
    out = {}
    for tag in ('hrformer_t_bn', 'hrformer_b_bn'):
        cfg = cfgs[tag]
        ref = BACKBONES.build(copy.deepcopy(cfg))
        c2 = copy.deepcopy(cfg)
        c2.pop('type')
        orc = O.HRFormerOracle(**c2)
        assert list(ref.state_dict().keys()) == list(orc.state_dict().keys()), 'state-dict keys differ'
        assert [tuple(v.shape) for v in ref.state_dict().values()] == [tuple(v.shape) for v in orc.state_dict().values()]
        rates_ref = [m.drop_path.drop_prob if hasattr(m.drop_path, 'drop_prob') else 0.0
                     for m in ref.modules() if type(m).__name__ == 'HRFormerBlock']
        rates_orc = [m.drop_path.p if isinstance(m.drop_path, O.DropPath) else 0.0
                     for m in orc.modules() if isinstance(m, O.HRFormerBlock)]
        assert np.allclose(rates_ref, rates_orc), 'stochastic-depth schedule differs'
        out[f'{tag}.drop_path_rates'] = np.asarray(rates_ref)
        O.seeded_fill_(ref, 0)
        O.seeded_fill_(orc, 0)
        disable_stochastic(ref)
        disable_stochastic(orc)
        if tag != 'hrformer_t_bn':
            continue                                # B: construction / schedule only (its forward is the same code)
        x, _ = O.seeded_inputs(2, 64, 96, [3], seed=1)
        for mode in ('eval', 'train'):
            ref.train(mode == 'train')
            orc.train(mode == 'train')
            sd0 = copy.deepcopy(ref.state_dict())
            xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
            ya, yb = ref(xa), orc(xb)
            g = torch.Generator().manual_seed(5)
            cots = [torch.randn(t.shape, generator=g) for t in ya]
            sum((t * c).sum() for t, c in zip(ya, cots)).backward()
            sum((t * c).sum() for t, c in zip(yb, cots)).backward()
            for i, (a, b) in enumerate(zip(ya, yb)):
                assert float((a - b).abs().max()) == 0.0, (mode, i)
                out[f'{tag}.{mode}.out{i}'] = a.detach().numpy()
            assert float((xa.grad - xb.grad).abs().max()) == 0.0
            out[f'{tag}.{mode}.dx'] = xa.grad.numpy()
            for (ka, pa), (kb, pb) in zip(ref.named_parameters(), orc.named_parameters()):
                assert ka == kb
                assert (pa.grad is None) == (pb.grad is None), ka
                if pa.grad is not None:
                    assert float((pa.grad - pb.grad).abs().max()) == 0.0, ka
            ref.zero_grad(); orc.zero_grad()
            ref.load_state_dict(sd0); orc.load_state_dict(sd0)
        print(tag, 'ok: reference == oracle bit-exact (eval + train, outputs and all gradients);',
              [tuple(t.shape) for t in ya])
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'hrformer_t.npz'), **out)


if __name__ == '__main__':
    main()
