"""Plain camera-only HRFormer backbone: oracle pinned to the reference's golden vectors, product
parity vs the oracle (mmdet/models/backbones/hrformer.py:565-740 over hrnet.py:211-596)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import helpers as T
import hrfuser_oracle as O

GOLD = os.path.join(T.ROOT, 'tests', 'golden')


def _cfgs():
    with open(os.path.join(GOLD, 'hrformer_cfgs.json')) as fh:
        return json.load(fh)


def _oracle(tag, seed=0):
    c = copy.deepcopy(_cfgs()[tag])
    c.pop('type')
    orc = O.HRFormerOracle(**c)
    O.seeded_fill_(orc, seed)
    return orc


def _pair(tag, dev):
    from hrfuser_amd import build_backbone
    orc = _oracle(tag)
    net = build_backbone(copy.deepcopy(_cfgs()[tag]))
    net.load_state_dict(orc.state_dict())
    net.to(dev)
    return net, orc


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_oracle_matches_reference_golden(mode):
    gold = np.load(os.path.join(GOLD, 'hrformer_t.npz'))
    orc = _oracle('hrformer_t_bn')
    T.disable_stochastic(orc)
    orc.train(mode == 'train')
    x, _ = O.seeded_inputs(2, 64, 96, [3], seed=1)
    x.requires_grad_(True)
    ys = orc(x)
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(t.shape, generator=g) for t in ys]
    sum((t * c).sum() for t, c in zip(ys, cots)).backward()
    for i, y in enumerate(ys):
        assert np.array_equal(y.detach().numpy(), gold[f'hrformer_t_bn.{mode}.out{i}']), i
    assert np.array_equal(x.grad.numpy(), gold[f'hrformer_t_bn.{mode}.dx'])


def test_product_construction_matches_reference():
    from hrfuser_amd import build_backbone, HRFormer, HRFormerBlock, BACKBONES
    gold = np.load(os.path.join(GOLD, 'hrformer_t.npz'))
    assert BACKBONES.get('HRFormer') is HRFormer
    for tag in ('hrformer_t_bn', 'hrformer_b_bn'):
        net = build_backbone(copy.deepcopy(_cfgs()[tag]))
        orc = _oracle(tag)
        assert list(net.state_dict().keys()) == list(orc.state_dict().keys())
        assert [tuple(v.shape) for v in net.state_dict().values()] == [tuple(v.shape) for v in orc.state_dict().values()]
        # stochastic-depth schedule (hrformer.py:666-678) as built by the reference itself
        rates = [m.drop_path_prob for m in net.modules() if isinstance(m, HRFormerBlock)]
        assert np.allclose(rates, gold[f'{tag}.drop_path_rates'])
    with pytest.raises(AssertionError):
        HRFormer(extra=dict(stage1=dict()))                      # hrnet.py:296: all four stages are required


def _run(tag, B, H, W, train, backend, drop=False):
    dev = T.use_backend(backend)
    try:
        net, orc = _pair(tag, dev)
        if not drop:
            T.disable_stochastic(net, orc)
        net.train(train)
        orc.train(train)
        x, _ = O.seeded_inputs(B, H, W, [3], seed=1)
        if not train:
            with torch.no_grad():
                ya, yb = net(x.to(dev)), orc(x)
            assert isinstance(ya, list) and len(ya) == len(yb)
            for i, (a, b) in enumerate(zip(ya, yb)):
                assert a.shape == b.shape and T.relmax(a, b) < 1e-3, (i, T.relmax(a, b))
            return
        # forward gate against the UNPINNED fp64 oracle; gradient gate flip-free (the oracle runs take the product's ReLU
        # decisions, helpers.PinnedReLU) with the per-tensor rule of SURVEY 8c, as in test_parity_wholenet._fwd_bwd
        o64 = copy.deepcopy(orc).double()
        xa = x.clone().to(dev).requires_grad_(True)
        xb = x.double().requires_grad_(True)
        T.enable_relu_probe(net)
        ya = net(xa)
        with torch.no_grad():
            yfree = o64(xb.detach())
        for i, (a, b) in enumerate(zip(ya, yfree)):
            assert T.relmax(a, b) < 1e-3, (i, T.relmax(a, b))
        masks = T.relu_masks(net)
        with T.PinnedReLU(masks) as pin64:
            yb = o64(xb)
        o32 = copy.deepcopy(orc)
        xc = x.clone().requires_grad_(True)
        with T.PinnedReLU(masks):
            yc = o32(xc)
        g = torch.Generator().manual_seed(5)
        cots = [torch.randn(t.shape, generator=g) for t in yb]
        sum((t * c.to(dev)).sum() for t, c in zip(ya, cots)).backward()
        sum((t * c.double()).sum() for t, c in zip(yb, cots)).backward()
        sum((t * c).sum() for t, c in zip(yc, cots)).backward()
        print(f'[{tag} {B}x{H}x{W}] {pin64.sites} ReLU sites pinned, {pin64.flips} element(s) decided differently by fp64')
        e, e_ref = T.rel_l2(xa.grad, xb.grad), T.rel_l2(xc.grad, xb.grad)
        assert e <= max(1e-3, 3 * e_ref), (e, e_ref)
        T.tight_grad_gate(net.named_parameters(), o64.named_parameters(), o32.named_parameters(), 1e-3, f'{tag} train')
    finally:
        T.use_backend('hip')


def test_hrformer_emul_eval():
    _run('hrformer_t_bn', 1, 64, 64, False, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('train', [False, True])
def test_hrformer_gpu(train):
    _run('hrformer_t_bn', 2, 64, 96, train, 'hip')


@pytest.mark.gpu
def test_hrformer_gpu_fullres_eval():
    _run('hrformer_t_bn', 1, 384, 640, False, 'hip')


@pytest.mark.gpu
def test_hrformer_droppath_gpu(monkeypatch):
    """Stage DropPath (hrformer.py:373-380 via mmcv DropPath): with every sample kept, the product must equal the
    oracle whose DropPath multiplies by 1/keep - pins where and how the scale enters both residual paths."""
    from hrfuser_amd.backbone import Engine
    dev = T.use_backend('hip')
    cfg = copy.deepcopy(_cfgs()['hrformer_t_bn'])
    cfg['drop_path_rate'] = 0.3
    from hrfuser_amd import build_backbone
    c2 = copy.deepcopy(cfg)
    c2.pop('type')
    orc = O.HRFormerOracle(**c2)
    O.seeded_fill_(orc, 0)
    net = build_backbone(copy.deepcopy(cfg))
    net.load_state_dict(orc.state_dict())
    net.to(dev).train()
    orc.train()
    monkeypatch.setattr(Engine, 'droppath_scale',
                        lambda self, B, p: torch.full((B,), 1.0 / (1.0 - p), device=self.device))
    monkeypatch.setattr(O.DropPath, 'forward', lambda self, x: x / (1.0 - self.p) if self.training and self.p > 0 else x)
    x, _ = O.seeded_inputs(2, 64, 96, [3], seed=1)
    xa = x.clone().to(dev).requires_grad_(True)
    xb = x.clone().requires_grad_(True)
    T.enable_relu_probe(net)
    ya = net(xa)
    with T.PinnedReLU(T.relu_masks(net)):                  # flip-free: the oracle takes the product's ReLU decisions
        yb = orc(xb)
    for i, (a, b) in enumerate(zip(ya, yb)):
        assert T.relmax(a, b) < 1e-3, (i, T.relmax(a, b))
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(t.shape, generator=g) for t in yb]
    sum((t * c.to(dev)).sum() for t, c in zip(ya, cots)).backward()
    sum((t * c).sum() for t, c in zip(yb, cots)).backward()
    assert T.rel_l2(xa.grad, xb.grad) < 2e-3, T.rel_l2(xa.grad, xb.grad)


def test_stale_backward_raises_emul():
    """Every forward re-uses the engine's BatchNorm slots and step buffers: back-propagating an EARLIER forward after a
    later one must fail loudly, not produce gradients from the wrong statistics."""
    from hrfuser_amd import _lib
    dev = T.use_backend('emul')
    try:
        net, _ = _pair('hrformer_t_bn', dev)
        net.train()
        x, _ = O.seeded_inputs(1, 32, 32, [3], seed=1)
        x1 = x.clone().requires_grad_(True)
        y1 = net(x1)
        net(x.clone() * 0.5)
        with pytest.raises(_lib.HRFuserHipError, match='no longer the latest'):
            sum(t.sum() for t in y1).backward()
        y3 = net(x1)                                   # the latest forward still back-propagates
        sum(t.sum() for t in y3).backward()
        assert x1.grad is not None and torch.isfinite(x1.grad).all()
    finally:
        T.use_backend('hip')


@pytest.mark.gpu
def test_lane_timing_does_not_change_gradients_gpu():
    """tools/race_check.py on the plain HRFormer and on HRFuser-T: one eager training step under lane-timing perturbations (one leaf
    lane, idle launches in front of the blocks of one width, no grouping) gives the same gradients.  Regression test of the
    transition1 race (two lanes writing one gradient buffer, one overwriting, one accumulating)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'tools', 'race_check.py'), 'hrformer_t_bn', 't_nus_bn'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'RACE CHECK OK' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
