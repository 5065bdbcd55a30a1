"""SURVEY 8f-4 (last part): HRFuserHRNetBased (hrfuser_hrnet_based.py:23-315) - the fusion dataflow over a convolutional
HRNet trunk.  The oracle class is bit-exact against the reference class (oracle/tools/make_golden_hrnet_based.py, which also
wrote the fixtures used here); the product is checked against the oracle through the C ABI."""
import copy
import json
import os

import numpy as np
import pytest
import torch

import hrfuser_oracle as O
from helpers import (ROOT, PinnedReLU, disable_stochastic, enable_relu_probe, rel_l2, relmax, relu_masks, tight_grad_gate,
                     use_backend)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def _cfg():
    with open(os.path.join(GOLD, 'hrfuser_hrnet_cfg.json')) as fh:
        return json.load(fh)


def _pair(dev):
    from hrfuser_amd import build_backbone
    meta = _cfg()
    kw = copy.deepcopy(meta['cfg'])
    kw.pop('type')
    orc = O.HRFuserHRNetOracle(**kw)
    O.seeded_fill_(orc, 0)
    net = build_backbone(copy.deepcopy(meta['cfg']))
    net.load_state_dict(orc.state_dict())
    net.to(dev)
    disable_stochastic(net, orc)
    return net, orc, meta


def test_oracle_matches_reference_golden_and_manifest():
    net, orc, meta = _pair(torch.device('cpu'))
    sd = net.state_dict()
    assert list(sd.keys()) == [e[0] for e in meta['entries']] == list(orc.state_dict().keys())
    assert sum(p.numel() for p in net.parameters()) == meta['n_params']
    for k, shape, dt in meta['entries']:
        assert list(sd[k].shape) == shape and str(sd[k].dtype) == 'torch.' + dt, k
    gold = np.load(os.path.join(GOLD, 'hrfuser_hrnet.npz'))
    x, mods = O.seeded_inputs(2, 64, 96, [3, 3], seed=1)
    for mode in ('eval', 'train'):
        orc.train(mode == 'train')
        sd0 = copy.deepcopy(orc.state_dict())
        with torch.no_grad():
            ys = orc(x.clone(), [m.clone() for m in mods])
        for i, y in enumerate(ys):
            assert float((y - torch.as_tensor(gold[f'B2_64x96/{mode}/out{i}'])).abs().max()) == 0.0     # bit-exact restatement
        orc.load_state_dict(sd0)


def _run(train, backend):
    dev = use_backend(backend)
    net, orc, meta = _pair(dev)
    net.train(train)
    orc.train(train)
    B, H, W = (2, 64, 96) if backend == 'hip' else ((2, 64, 64) if train else (1, 32, 32))   # train: >= 8 samples per BatchNorm
    x, mods = O.seeded_inputs(B, H, W, [3, 3], seed=1)
    xa = x.clone().to(dev).requires_grad_(True)
    enable_relu_probe(net)
    ya = net(xa, [m.to(dev) for m in mods])
    if backend == 'hip':
        gold = np.load(os.path.join(GOLD, 'hrfuser_hrnet.npz'))
        for i, y in enumerate(ya):
            assert relmax(y, torch.as_tensor(gold[f'B2_64x96/{"train" if train else "eval"}/out{i}'])) < 1e-3
    masks = relu_masks(net)
    refs = []
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(t.shape, generator=g) for t in ya]
    for dt in (torch.float64, torch.float32):
        o = copy.deepcopy(orc).to(dt)
        xb = x.to(dt).requires_grad_(True)
        with PinnedReLU(masks):
            ys = o(xb, [m.to(dt) for m in mods])
        sum((t * c.to(dt)).sum() for t, c in zip(ys, cots)).backward()
        refs.append((o, ys, xb))
    o64, yb, xb = refs[0]
    for p, q in zip(ya, yb):
        assert relmax(p, q) < 1e-3
    sum((t * c.to(dev)).sum() for t, c in zip(ya, cots)).backward()
    e, e_ref = rel_l2(xa.grad, xb.grad), rel_l2(refs[1][2].grad, xb.grad)
    assert e <= max(1e-3, 3 * e_ref), (e, e_ref)
    tight_grad_gate(net.named_parameters(), o64.named_parameters(), refs[1][0].named_parameters(), 1e-3, f'hrnet-based train={train}')


@pytest.mark.parametrize('train', [False, True])
def test_hrnet_based_emul(train):
    _run(train, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('train', [False, True])
def test_hrnet_based_gpu(train):
    _run(train, 'hip')
