"""Pins the ORACLE (oracle/hrfuser_oracle.py) against golden vectors produced by the real
reference (oracle/tools/make_golden.py, run in the build container).  CPU only."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import hrfuser_oracle as O

NORM = dict(type='BN', requires_grad=True, momentum=0.1)
LN = dict(type='LN', eps=1e-6)


def relmax(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def disable_stochastic(net):
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, O.DropPath):
            m.p = 0.0


@pytest.fixture(scope='module')
def cfgs(golden_dir):
    with open(os.path.join(golden_dir, 'backbone_cfgs.json')) as fh:
        return json.load(fh)


def build(cfgs, tag):
    cfg = copy.deepcopy(cfgs[tag])
    cfg.pop('type')
    return O.HRFuserOracle(**cfg), cfg


@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_state_dict_manifest(cfgs, golden_dir, tag):
    net, _ = build(cfgs, tag)
    with open(os.path.join(golden_dir, f'state_manifest_{tag}.json')) as fh:
        man = json.load(fh)
    sd = net.state_dict()
    assert sum(p.numel() for p in net.parameters()) == man['n_params']
    assert set(sd) == {e[0] for e in man['entries']}
    for k, shape, dt in man['entries']:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == 'torch.' + dt, k


@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_wholenet_small(cfgs, golden_dir, tag):
    gold = np.load(os.path.join(golden_dir, f'wholenet_{tag}.npz'))
    net, cfg = build(cfgs, tag)
    O.seeded_fill_(net, 0)
    disable_stochastic(net)
    mc = cfg.get('mod_in_channels', [3, 3])
    keys = sorted({k.split('/')[0] for k in gold.files})
    for key in keys:
        B, hw = key.split('_')
        B, (H, W) = int(B[1:]), map(int, hw.split('x'))
        x, mods = O.seeded_inputs(B, H, W, mc, seed=1)
        net.eval()
        with torch.no_grad():
            ys = net(x, [m.clone() for m in mods])
        for i, y in enumerate(ys):
            assert relmax(y, gold[f'{key}/eval/out{i}']) < 1e-5
        sd0 = copy.deepcopy(net.state_dict())
        net.train()
        with torch.no_grad():
            ys = net(x, [m.clone() for m in mods])
        for i, y in enumerate(ys):
            assert relmax(y, gold[f'{key}/train/out{i}']) < 1e-5
        net.load_state_dict(sd0)


def test_wholenet_grads_fp64(cfgs, golden_dir):
    tag, key = 't_nus', 'B2_64x96'
    gold = np.load(os.path.join(golden_dir, f'wholenet_{tag}.npz'))
    net, cfg = build(cfgs, tag)
    O.seeded_fill_(net, 0)
    disable_stochastic(net)
    net = net.double().train()
    x, mods = O.seeded_inputs(2, 64, 96, [3, 3], seed=1)
    x = x.double().requires_grad_(True)
    mods = [m.double().requires_grad_(True) for m in mods]
    ys = net(x, list(mods))
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(t.shape, generator=g).double() for t in ys]
    sum((t * c).sum() for t, c in zip(ys, cots)).backward()
    names = list(gold[f'{key}/grad/param_names'])
    params = dict(net.named_parameters())
    got = [n for n, p in net.named_parameters() if p.grad is not None]
    assert got == names
    # the only params without gradient: quirk transition1[i][0]
    assert sorted(set(params) - set(got)) == ['transition1.0.1.bias', 'transition1.0.1.weight']
    norms = np.array([float(params[n].grad.norm()) for n in names])
    sums = np.array([float(params[n].grad.sum()) for n in names])
    gn, gs = gold[f'{key}/grad/param_norm'], gold[f'{key}/grad/param_sum']
    assert np.all(np.abs(norms - gn) <= 1e-8 * (1 + gn))
    assert np.all(np.abs(sums - gs) <= 1e-7 * (1 + gn))
    assert relmax(x.grad, gold[f'{key}/grad/x']) < 1e-6
    for k, m in enumerate(mods):
        assert relmax(m.grad, gold[f'{key}/grad/mod{k}']) < 1e-6
    sd = net.state_dict()
    assert relmax(sd['bn1.running_mean'], gold[f'{key}/train/bn1.running_mean']) < 1e-6
    assert relmax(sd['bn1.running_var'], gold[f'{key}/train/bn1.running_var']) < 1e-6


# ------------------------------------------------------------------ module level
def _mods():
    chans, heads = (8, 16, 32, 64), (1, 2, 4, 8)
    ds = lambda: torch.nn.Sequential(torch.nn.Conv2d(16, 64, 1, bias=False), torch.nn.BatchNorm2d(64))
    out = {}
    for (c, h, H, W) in ((18, 1, 10, 13), (36, 2, 7, 7), (72, 4, 15, 8), (78, 2, 9, 16)):
        out[f'lsa_c{c}_h{h}_{H}x{W}'] = (lambda c=c, h=h: O.LocalWindowSelfAttention(c, h),
                                         [((2, H * W, c), 11)], lambda m, i, H=H, W=W: m(i[0], H, W))
    for (c, h, H, W) in ((18, 1, 10, 13), (36, 2, 15, 8), (144, 8, 6, 10)):
        out[f'mwca_c{c}_h{h}_{H}x{W}'] = (lambda c=c, h=h: O.MultiWindowCrossAttention(c, h, 0.1),
                                          [((2, H * W, c), 12), ((2, H * W, c), 13)],
                                          lambda m, i, H=H, W=W: m(i[0], i[1], H, W))
    for (c, H, W) in ((18, 9, 11), (36, 6, 10)):
        out[f'ffn_c{c}_{H}x{W}'] = (lambda c=c: O.CrossFFN(c, 4 * c, NORM), [((2, H * W, c), 14)],
                                    lambda m, i, H=H, W=W: m(i[0], H, W))
    out['block_c36_h2_9x12'] = (lambda: O.HRFormerBlock(36, 2, 4, NORM, LN), [((2, 36, 9, 12), 15)],
                                lambda m, i: m(i[0]))
    for (c, h, M, H, W) in ((18, 1, 2, 10, 13), (36, 2, 3, 8, 15)):
        out[f'fusion_c{c}_M{M}_{H}x{W}'] = (
            lambda c=c, h=h, M=M: O.HRFuserFusionBlock(c, h, 4, NORM, LN, 0.2, M, 0.1),
            [((2, c, H, W), 16)] + [((2, c, H, W), 17 + k) for k in range(M)],
            lambda m, i: m(i[0], list(i[1:])))
    for nb in (2, 3, 4):
        out[f'hrmodule_{nb}b'] = (
            lambda nb=nb: O.HRFormerModule(list(chans[:nb]), (1,) * nb, heads[:nb], (4,) * nb, NORM, LN),
            [((2, chans[i], 24 >> i, 40 >> i), 20 + i) for i in range(nb)], lambda m, i: m(list(i)))
    out['bottleneck_first'] = (lambda: O.Bottleneck(16, 16, NORM, ds()), [((2, 16, 9, 10), 30)], lambda m, i: m(i[0]))
    out['bottleneck_plain'] = (lambda: O.Bottleneck(64, 16, NORM), [((2, 64, 9, 10), 31)], lambda m, i: m(i[0]))
    return out


MODS = _mods()


@pytest.mark.parametrize('name', sorted(MODS))
def test_module_level(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, 'modules.npz'))
    ctor, in_specs, call = MODS[name]
    modes = sorted({k.split('/')[1] for k in gold.files if k.startswith(name + '/')})
    assert modes
    for mode in modes:
        mod = ctor()
        O.seeded_fill_(mod, 3)
        disable_stochastic(mod)
        mod = mod.double().train(mode == 'train')
        ins = [torch.randn(s, generator=torch.Generator().manual_seed(seed)).double().requires_grad_(True)
               for s, seed in in_specs]
        outs = call(mod, ins)
        outs = list(outs) if isinstance(outs, (list, tuple)) else [outs]
        g = torch.Generator().manual_seed(7)
        cots = [torch.randn(o.shape, generator=g).double() for o in outs]
        sum((o * c).sum() for o, c in zip(outs, cots)).backward()
        pre = f'{name}/{mode}/'
        for i, o in enumerate(outs):
            assert relmax(o, gold[pre + f'out{i}']) < 1e-6, (name, mode, i)
        for i, t in enumerate(ins):
            assert relmax(t.grad, gold[pre + f'gin{i}']) < 1e-6
        n_checked = 0
        for n, p in mod.named_parameters():
            k = pre + 'gparam/' + n
            if p.grad is None:
                assert k not in gold.files
                continue
            gref = torch.as_tensor(gold[k]).double()
            # analytically-zero grads (k-bias, conv bias before train BN) -> absolute tolerance
            tol = 1e-6 * float(gref.abs().max()) + 1e-9
            assert float((p.grad - gref).abs().max()) <= tol + 1e-6 * float(gref.abs().max()), (name, mode, n)
            n_checked += 1
        assert n_checked > 0


def test_fullres_digest_t_nus_eval(cfgs, golden_dir):
    """config[0] of BASELINE.json: T backbone fwd, 1x3x384x640 + lidar + radar, CPU."""
    gold = np.load(os.path.join(golden_dir, 'fullres_digests.npz'))
    net, cfg = build(cfgs, 't_nus')
    O.seeded_fill_(net, 0)
    net.eval()
    x, mods = O.seeded_inputs(1, 384, 640, [3, 3], seed=1)
    with torch.no_grad():
        ys = net(x, mods)
    shapes = [(1, 18, 96, 160), (1, 36, 48, 80), (1, 72, 24, 40), (1, 144, 12, 20)]
    for i, y in enumerate(ys):
        assert tuple(y.shape) == shapes[i]
        meta = gold[f't_nus/eval_B1/out{i}/meta']
        f = y.double().reshape(-1)
        idx = torch.linspace(0, f.numel() - 1, min(4096, f.numel())).long()
        assert relmax(f[idx], gold[f't_nus/eval_B1/out{i}/samples']) < 1e-5
        assert abs(float(f.abs().sum()) - meta[1]) <= 1e-5 * meta[1]
