"""Generate the committed golden vectors under tests/golden/ from the REAL reference.

Container-only (needs /root/reference).  Run:  python oracle/tools/make_golden.py
Everything written here is DATA (inputs are regenerated from seeds; expected outputs /
gradient digests / key manifests are stored); no reference source text is stored.

Protocol:
  * parameters + BN running stats: `seeded_fill_(module, seed)` in sorted state-dict order
  * inputs: `seeded_inputs` / `torch.Generator(seed)` randn
  * Dropout / DropPath disabled (p = 0) - they cannot be RNG-matched
  * module-level goldens: reference run in fp64, stored as fp32
  * whole-net goldens: reference fp32 outputs (eval BN and train BN) at small non-square
    shapes + fp64 gradient digests; full-resolution digests (sum, |sum|, strided samples)
"""
import copy
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_loader as R                # noqa: E402
import hrfuser_oracle as O            # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'tests', 'golden')
CFG = {
    't_nus': 'cascade_rcnn_hrfuser_t_1x_nus_r640_l_r_fusion',
    't_nus_bn': 'cascade_rcnn_hrfuser_t_1x_nus_r640_l_r_fusion_bn',
    'b_nus': 'cascade_rcnn_hrfuser_b_1x_nus_r640_l_r_fusion',
    'b_nus_bn': 'cascade_rcnn_hrfuser_b_1x_nus_r640_l_r_fusion_bn',
    't_stf': 'cascade_rcnn_hrfuser_t_1x_stf_r1248_4mod',
    't_stf_bn': 'cascade_rcnn_hrfuser_t_1x_stf_r1248_4mod_bn',
}
NORM = dict(type='BN', requires_grad=True, momentum=0.1)
LN = dict(type='LN', eps=1e-6)


def disable_stochastic(net):
    for m in net.modules():
        if isinstance(m, nn.Dropout):
            m.p = 0.0
        if hasattr(m, 'drop_prob'):
            m.drop_prob = 0.0
        if isinstance(m, O.DropPath):
            m.p = 0.0


def jsonable(o):
    if isinstance(o, dict):
        return {k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    return o


def digest(t, nsamp=4096):
    f = t.detach().double().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(nsamp, f.numel())).long()
    return dict(shape=list(t.shape), sum=float(f.sum()), abssum=float(f.abs().sum()),
                max=float(f.abs().max()), samples=f[idx].float().numpy())


# ----------------------------------------------------------------------------- whole net
def whole_net():
    cfgs = {tag: R.backbone_cfg(name) for tag, name in CFG.items()}
    with open(os.path.join(OUT, 'backbone_cfgs.json'), 'w') as fh:
        json.dump(jsonable(cfgs), fh, indent=1, sort_keys=True)

    for tag in ('t_nus', 'b_nus', 't_stf'):
        cfg = cfgs[tag]
        net = R.build_reference(copy.deepcopy(cfg))
        manifest = [[k, list(v.shape), str(v.dtype).replace('torch.', '')] for k, v in net.state_dict().items()]
        with open(os.path.join(OUT, f'state_manifest_{tag}.json'), 'w') as fh:
            json.dump(dict(n_params=sum(p.numel() for p in net.parameters()), entries=manifest), fh)
        O.seeded_fill_(net, 0)
        disable_stochastic(net)
        mc = cfg.get('mod_in_channels', [3, 3])
        arrays = {}
        shapes = [(2, 64, 96), (1, 96, 64)] if tag == 't_nus' else [(2, 64, 96)]
        for (B, H, W) in shapes:
            key = f'B{B}_{H}x{W}'
            x, mods = O.seeded_inputs(B, H, W, mc, seed=1)
            net.eval()
            with torch.no_grad():
                ys = net(x.clone(), [m.clone() for m in mods])
            for i, y in enumerate(ys):
                arrays[f'{key}/eval/out{i}'] = y.numpy()
            # train-mode BN: fp32 outputs + fp64 gradient digests
            net.train()
            sd0 = copy.deepcopy(net.state_dict())
            ys = net(x.clone(), [m.clone() for m in mods])
            for i, y in enumerate(ys):
                arrays[f'{key}/train/out{i}'] = y.detach().numpy()
            net.load_state_dict(sd0)                     # undo running-stat update
            if B == 2:
                net64 = copy.deepcopy(net).double()
                net64.train()
                x64 = x.double().requires_grad_(True)
                m64 = [m.double().requires_grad_(True) for m in mods]
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
                arrays[f'{key}/grad/param_norm'] = np.asarray(norms)
                arrays[f'{key}/grad/param_sum'] = np.asarray(sums)
                arrays[f'{key}/grad/param_names'] = np.asarray(names)
                arrays[f'{key}/grad/x'] = x64.grad.float().numpy()
                for k, m in enumerate(m64):
                    arrays[f'{key}/grad/mod{k}'] = m.grad.float().numpy()
                # running-stat update check (momentum 0.1, unbiased var)
                arrays[f'{key}/train/bn1.running_mean'] = net64.state_dict()['bn1.running_mean'].float().numpy()
                arrays[f'{key}/train/bn1.running_var'] = net64.state_dict()['bn1.running_var'].float().numpy()
        np.savez_compressed(os.path.join(OUT, f'wholenet_{tag}.npz'), **arrays)
        print(tag, 'whole-net fixtures:', len(arrays))

    # full-resolution digests (eval, B=1; train, B=2 for t_nus)
    full = {}
    for tag, (H, W) in (('t_nus', (384, 640)), ('b_nus', (384, 640)), ('t_stf', (384, 1248))):
        cfg = cfgs[tag]
        net = R.build_reference(copy.deepcopy(cfg))
        O.seeded_fill_(net, 0)
        disable_stochastic(net)
        mc = cfg.get('mod_in_channels', [3, 3])
        net.eval()
        x, mods = O.seeded_inputs(1, H, W, mc, seed=1)
        with torch.no_grad():
            ys = net(x, mods)
        for i, y in enumerate(ys):
            d = digest(y)
            full[f'{tag}/eval_B1/out{i}/samples'] = d.pop('samples')
            full[f'{tag}/eval_B1/out{i}/meta'] = np.asarray([d['sum'], d['abssum'], d['max']] + d['shape'], dtype=np.float64)
        if tag == 't_nus':
            net.train()
            x, mods = O.seeded_inputs(2, H, W, mc, seed=1)
            with torch.no_grad():
                ys = net(x, mods)
            for i, y in enumerate(ys):
                d = digest(y)
                full[f'{tag}/train_B2/out{i}/samples'] = d.pop('samples')
                full[f'{tag}/train_B2/out{i}/meta'] = np.asarray([d['sum'], d['abssum'], d['max']] + d['shape'], dtype=np.float64)
        print(tag, 'full-res digests done')
    np.savez_compressed(os.path.join(OUT, 'fullres_digests.npz'), **full)


# ----------------------------------------------------------------------------- module level
def run_module(mod, inputs, call, train):
    """fp64 run; returns outputs (list) + grads wrt inputs and params under random cotangents."""
    mod = copy.deepcopy(mod).double()
    mod.train(train)
    ins = [t.double().requires_grad_(True) for t in inputs]
    outs = call(mod, ins)
    outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
    g = torch.Generator().manual_seed(7)
    cots = [torch.randn(o.shape, generator=g).double() for o in outs]
    sum((o * c).sum() for o, c in zip(outs, cots)).backward()
    res = {}
    for i, o in enumerate(outs):
        res[f'out{i}'] = o.detach().float().numpy()
    for i, t in enumerate(ins):
        res[f'gin{i}'] = t.grad.float().numpy()
    for n, p in mod.named_parameters():
        if p.grad is not None:
            res[f'gparam/{n}'] = p.grad.float().numpy()
    return res


def rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def module_level():
    R.install()
    from mmdet.models.backbones import hrformer as RF
    from mmdet.models.backbones import hrfuser_hrformer_based as RU
    from mmdet.models.backbones import resnet as RN
    arrays = {}

    def add(name, mod, inputs, call, modes=(False,)):
        O.seeded_fill_(mod, 3)
        disable_stochastic(mod)
        for train in modes:
            res = run_module(mod, inputs, call, train)
            for k, v in res.items():
                arrays[f'{name}/{"train" if train else "eval"}/{k}'] = v

    # local-window self attention, incl. non-divisible grids (pad asymmetry)
    for (c, h, H, W) in ((18, 1, 10, 13), (36, 2, 7, 7), (72, 4, 15, 8), (78, 2, 9, 16)):
        m = RF.LocalWindowSelfAttention(c, num_heads=h, window_size=7)
        add(f'lsa_c{c}_h{h}_{H}x{W}', m, [rand((2, H * W, c), 11)], lambda mod, i, H=H, W=W: mod(i[0], H, W))
    # multi-window cross attention
    for (c, h, H, W) in ((18, 1, 10, 13), (36, 2, 15, 8), (144, 8, 6, 10)):
        m = RU.MultiWindowCrossAttention(embed_dim=c, num_heads=h, window_size=7, proj_drop_rate=0.1)
        add(f'mwca_c{c}_h{h}_{H}x{W}', m, [rand((2, H * W, c), 12), rand((2, H * W, c), 13)],
            lambda mod, i, H=H, W=W: mod(i[0], i[1], H, W))
    # CrossFFN (train + eval BN)
    for (c, H, W) in ((18, 9, 11), (36, 6, 10)):
        m = RF.CrossFFN(c, 4 * c, c, norm_cfg=NORM)
        add(f'ffn_c{c}_{H}x{W}', m, [rand((2, H * W, c), 14)], lambda mod, i, H=H, W=W: mod(i[0], H, W), (False, True))
    # HRFormerBlock
    m = RF.HRFormerBlock(36, 36, num_heads=2, window_size=7, mlp_ratio=4, norm_cfg=NORM, transformer_norm_cfg=LN)
    add('block_c36_h2_9x12', m, [rand((2, 36, 9, 12), 15)], lambda mod, i: mod(i[0]), (False, True))
    # fusion blocks, M = 2 and 3
    for (c, h, M, H, W) in ((18, 1, 2, 10, 13), (36, 2, 3, 8, 15)):
        m = RU.HRFuserFusionBlock(c, c, num_heads=h, window_size=7, mlp_ratio=4, drop_path=0.2, norm_cfg=NORM,
                                  transformer_norm_cfg=LN, num_fused_modalities=M, proj_drop_rate=0.1)
        ins = [rand((2, c, H, W), 16)] + [rand((2, c, H, W), 17 + k) for k in range(M)]
        add(f'fusion_c{c}_M{M}_{H}x{W}', m, ins, lambda mod, i: mod(i[0], list(i[1:])), (False, True))
    # HR modules with 2/3/4 branches (cross-resolution exchange)
    chans, heads = (8, 16, 32, 64), (1, 2, 4, 8)
    for nb in (2, 3, 4):
        m = RF.HRFomerModule(nb, RF.HRFormerBlock, (1,) * nb, list(chans[:nb]), chans[:nb], heads[:nb], (7,) * nb,
                             (4,) * nb, True, drop_paths=[0.0], norm_cfg=NORM, transformer_norm_cfg=LN)
        ins = [rand((2, chans[i], 24 >> i, 40 >> i), 20 + i) for i in range(nb)]
        add(f'hrmodule_{nb}b', m, ins, lambda mod, i: mod(list(i)), (False, True))
    # Bottlenecks
    ds = nn.Sequential(nn.Conv2d(16, 64, 1, bias=False), nn.BatchNorm2d(64))
    m = RN.Bottleneck(16, 16, downsample=ds, norm_cfg=NORM)
    add('bottleneck_first', m, [rand((2, 16, 9, 10), 30)], lambda mod, i: mod(i[0]), (False, True))
    m = RN.Bottleneck(64, 16, norm_cfg=NORM)
    add('bottleneck_plain', m, [rand((2, 64, 9, 10), 31)], lambda mod, i: mod(i[0]), (False, True))
    np.savez_compressed(os.path.join(OUT, 'modules.npz'), **arrays)
    print('module fixtures:', len(arrays))


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    module_level()
    whole_net()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print('golden dir bytes:', tot)
