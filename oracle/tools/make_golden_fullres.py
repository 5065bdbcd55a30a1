"""Full-size TRAINING fixtures from the REAL reference (container-only, needs /root/reference):
    python oracle/tools/make_golden_fullres.py            -> tests/golden/fullres_train.npz

For each BASELINE.json model (HRFuser-T nus 384x640, HRFuser-B nus 384x640, HRFuser-T stf 384x1248), train-mode
BatchNorm, Dropout / DropPath off, seeded parameters and inputs (hrfuser_oracle.seeded_fill_ / seeded_inputs):
  * forward digests of the four outputs (fp32 reference: sum, |sum|, max, 4096 strided samples);
  * fp64 gradient digests under the seeded random cotangents of the parity tests (generator seed 5, N(0,1)): per
    parameter (norm, sum) in named_parameters order, and digests of every input gradient.
Batch: 2 images (the per-GPU batch of every BASELINE config); HRFuser-B runs its fp64 pass on ONE image (the first of the
same two; a 2-image fp64 graph of the B model does not fit this container's memory) - the test feeds the same.
DATA only: nothing of the reference's text is stored."""
import copy
import gc
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_loader as R                # noqa: E402
import hrfuser_oracle as O            # noqa: E402
from make_golden import CFG, OUT, digest, disable_stochastic   # noqa: E402

SIZES = {'t_nus': (2, 384, 640, 2), 'b_nus': (2, 384, 640, 1), 't_stf': (2, 384, 1248, 2)}   # fwd batch, H, W, fp64-grad batch


def main():
    out = {}
    for tag, (B, H, W, Bg) in SIZES.items():
        cfg = R.backbone_cfg(CFG[tag])
        net = R.build_reference(copy.deepcopy(cfg))
        O.seeded_fill_(net, 0)
        disable_stochastic(net)
        mc = cfg.get('mod_in_channels', [3, 3])
        x, mods = O.seeded_inputs(B, H, W, mc, seed=1)
        net.train()
        sd0 = copy.deepcopy(net.state_dict())
        with torch.no_grad():
            ys = net(x.clone(), [m.clone() for m in mods])
        for i, y in enumerate(ys):
            d = digest(y)
            out[f'{tag}/train_B{B}/out{i}/samples'] = d.pop('samples')
            out[f'{tag}/train_B{B}/out{i}/meta'] = np.asarray([d['sum'], d['abssum'], d['max']] + d['shape'], dtype=np.float64)
        net.load_state_dict(sd0)
        del ys
        gc.collect()
        net64 = net.double()
        x64 = x[:Bg].double().requires_grad_(True)
        m64 = [m[:Bg].double().requires_grad_(True) for m in mods]
        ys = net64(x64, list(m64))
        g = torch.Generator().manual_seed(5)
        cots = [torch.randn(t.shape, generator=g).double() for t in ys]
        sum((t * c).sum() for t, c in zip(ys, cots)).backward()
        names, norms, sums = [], [], []
        for n, p in net64.named_parameters():
            if p.grad is None:
                continue
            names.append(n)
            norms.append(float(p.grad.norm()))
            sums.append(float(p.grad.sum()))
        key = f'{tag}/grad_B{Bg}'
        out[f'{key}/param_names'] = np.asarray(names)
        out[f'{key}/param_norm'] = np.asarray(norms)
        out[f'{key}/param_sum'] = np.asarray(sums)
        for nm, t in zip(['img'] + [f'mod{k}' for k in range(len(m64))], [x64] + m64):
            d = digest(t.grad)
            out[f'{key}/{nm}/samples'] = d.pop('samples')
            out[f'{key}/{nm}/meta'] = np.asarray([d['sum'], d['abssum'], d['max'], float(t.grad.norm())] + d['shape'], dtype=np.float64)
        for i, y in enumerate(ys):                       # fp64 outputs of the gradient run (train BN over Bg images)
            d = digest(y)
            out[f'{key}/out{i}/samples'] = d.pop('samples')
            out[f'{key}/out{i}/meta'] = np.asarray([d['sum'], d['abssum'], d['max']] + d['shape'], dtype=np.float64)
        print(tag, 'done:', len(names), 'parameter gradients', flush=True)
        del net, net64, ys, cots, x64, m64
        gc.collect()
    np.savez_compressed(os.path.join(OUT, 'fullres_train.npz'), **out)
    print('bytes', os.path.getsize(os.path.join(OUT, 'fullres_train.npz')))


if __name__ == '__main__':
    main()
