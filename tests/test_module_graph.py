"""Captured graphs at the module boundary: `backbone(img, mods)` + `loss.backward()` through torch.autograd -
what mmdet's TwoStageDetector.extract_feat does (two_stage.py:76-84) - replays one forward and one backward hipGraph from the
third call of an input signature on.  The replayed passes must equal the eager ones: outputs, input gradients, every parameter
gradient (accumulating into .grad like eager), running statistics; other signatures / modes keep working beside it."""
import os

import pytest
import torch

import hrfuser_oracle as O
from helpers import build_pair, use_backend


def _iter(net, x, mods, cots, zero=True):
    if zero:
        net.zero_grad(set_to_none=False)
    xa = x.clone().requires_grad_(True)
    ma = [m.clone().requires_grad_(True) for m in mods]
    ys = net(xa, list(ma))
    sum((y * c).sum() for y, c in zip(ys, cots)).backward()
    return [y.detach().clone() for y in ys], [xa.grad.clone()] + [m.grad.clone() for m in ma], \
        {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


@pytest.mark.gpu
def test_module_graph_equals_eager_gpu():
    dev = use_backend('hip')
    os.environ.pop('HRF_MODULE_GRAPH', None)
    net, _, cfg = build_pair('t_nus_bn', dev)
    ref, _, _ = build_pair('t_nus_bn', dev)
    x, mods = O.seeded_inputs(2, 64, 96, [3, 3], seed=1)
    x, mods = x.to(dev), [m.to(dev) for m in mods]
    net.train()
    ref.train()
    with torch.no_grad():
        ys = ref(x, list(mods))
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(y.shape, generator=g).to(dev) for y in ys]
    # eager reference: the same four iterations with module graphs switched off
    os.environ['HRF_MODULE_GRAPH'] = '0'
    try:
        ref.load_state_dict(net.state_dict())
        eager = [_iter(ref, x, mods, cots) for _ in range(4)]
    finally:
        os.environ.pop('HRF_MODULE_GRAPH', None)
    assert not any(e.fwd is not None for k, e in ref.__dict__.get('_hrf_graphs', {}).items() if k != '_setup')
    runs = [_iter(net, x, mods, cots) for _ in range(4)]          # 2 eager warm-ups, capture + replay, replay
    ents = [e for k, e in net.__dict__['_hrf_graphs'].items() if k != '_setup']
    assert len(ents) == 1 and ents[0].fwd is not None and ents[0].bwd is not None and not ents[0].failed
    gmax = max(float(v.abs().max()) for v in eager[0][2].values())
    for it in (2, 3):
        for a, b in zip(runs[it][0], eager[it][0]):
            assert _rel(a, b) < 1e-5
        for a, b in zip(runs[it][1], eager[it][1]):
            assert _rel(a, b) < 1e-4
        for k, b in eager[it][2].items():
            assert float((runs[it][2][k] - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-3 * gmax), (it, k)
    # running statistics advanced identically (4 momentum updates each)
    sa, sb = net.state_dict(), ref.state_dict()
    for k in sa:
        if 'running_' in k:
            assert _rel(sa[k], sb[k]) < 1e-5, k
        if 'num_batches_tracked' in k:
            assert int(sa[k]) == int(sb[k]) == 4, (k, int(sa[k]), int(sb[k]))
    # gradients ACCUMULATE across replays like eager .grad does
    single = runs[3][2]
    _iter(net, x, mods, cots, zero=True)
    acc = _iter(net, x, mods, cots, zero=False)[2]
    os.environ['HRF_MODULE_GRAPH'] = '0'
    try:
        for _ in range(2):                                         # (the reference net sees the same six training passes)
            _iter(ref, x, mods, cots)
    finally:
        os.environ.pop('HRF_MODULE_GRAPH', None)
    for k, b in single.items():
        assert float((acc[k] - 2 * b).abs().max()) <= 2e-4 * max(float(b.abs().max()), 1e-3 * gmax), k
    # another signature beside it: eval / no_grad forward (own entry, forward graph only), then the training key again
    net.eval()
    ref.eval()
    with torch.no_grad():
        outs = [net(x, list(mods)) for _ in range(4)]
        os.environ['HRF_MODULE_GRAPH'] = '0'
        try:
            want = ref(x, list(mods))
        finally:
            os.environ.pop('HRF_MODULE_GRAPH', None)
    for a, b in zip(outs[-1], want):
        assert _rel(a, b) < 1e-5
    net.train()
    again = _iter(net, x, mods, cots)
    for a, b in zip(again[0], eager[3][0]):
        assert _rel(a, b) < 1e-5                                  # train-mode outputs do not depend on the running statistics
    ents = [e for k, e in net.__dict__['_hrf_graphs'].items() if k != '_setup']
    assert len(ents) == 2 and sum(e.bwd is not None for e in ents) == 1


@pytest.mark.gpu
def test_module_graph_stale_backward_raises_gpu():
    """backward of a replayed forward after ANOTHER forward of the module overwrote the BatchNorm slots must raise, as the
    eager route does."""
    from hrfuser_amd._lib import HRFuserHipError
    dev = use_backend('hip')
    net, _, _ = build_pair('t_nus_bn', dev)
    x, mods = O.seeded_inputs(1, 64, 96, [3, 3], seed=1)
    x, mods = x.to(dev), [m.to(dev) for m in mods]
    net.train()
    for _ in range(3):
        ys = net(x.clone().requires_grad_(True), list(mods))
        sum(y.sum() for y in ys).backward()
    ys = net(x.clone().requires_grad_(True), list(mods))          # replayed forward
    net(x.clone().requires_grad_(True), list(mods))               # ... and another one before its backward
    with pytest.raises((HRFuserHipError, RuntimeError)):
        sum(y.sum() for y in ys).backward()


@pytest.mark.gpu
def test_module_graph_two_training_signatures_alternate_gpu():
    """a second TRAINING signature (other B/H/W -> other window count) rebinds the engine's slot arena /
    segment table / index map and grows the Dropout / DropPath pools; the captured graphs of the first signature carry raw
    pointers to the old ones.  Every entry now keeps its buffers alive and gets tables of its own: alternating replays of both
    signatures (with allocator churn in between) must keep producing the eager gradients."""
    dev = use_backend('hip')
    os.environ.pop('HRF_MODULE_GRAPH', None)
    net, _, _ = build_pair('t_nus_bn', dev)
    ref, _, _ = build_pair('t_nus_bn', dev)
    ref.load_state_dict(net.state_dict())
    net.train()
    ref.train()
    sigs = []
    for (B, H, W) in ((2, 64, 96), (1, 96, 64)):
        x, mods = O.seeded_inputs(B, H, W, [3, 3], seed=1 + B)
        x, mods = x.to(dev), [m.to(dev) for m in mods]
        with torch.no_grad():
            ys = ref(x, list(mods))
        g = torch.Generator().manual_seed(5 + B)
        cots = [torch.randn(y.shape, generator=g).to(dev) for y in ys]
        os.environ['HRF_MODULE_GRAPH'] = '0'
        try:
            want = _iter(ref, x, mods, cots)
        finally:
            os.environ.pop('HRF_MODULE_GRAPH', None)
        sigs.append((x, mods, cots, want))
    for x, mods, cots, _ in sigs:                                  # capture both (2 eager warm-ups + capture each)
        for _ in range(3):
            _iter(net, x, mods, cots)
    ents = [e for k, e in net.__dict__['_hrf_graphs'].items() if k != '_setup']
    assert len(ents) == 2 and all(e.fwd is not None and e.bwd is not None and not e.failed for e in ents)
    gmax = max(float(v.abs().max()) for v in sigs[0][3][2].values())
    for rnd in range(3):
        for x, mods, cots, want in sigs:
            junk = [torch.full((1 << 20,), float('nan'), device=dev) for _ in range(8)]     # allocator churn: freed blocks get reused
            del junk
            torch.cuda.empty_cache()
            got = _iter(net, x, mods, cots)
            for a, b in zip(got[0], want[0]):
                assert _rel(a, b) < 1e-5, rnd
            for a, b in zip(got[1], want[1]):
                assert _rel(a, b) < 1e-4, rnd
            for k, b in want[2].items():
                assert bool(torch.isfinite(got[2][k]).all()), (rnd, k)
                assert float((got[2][k] - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-3 * gmax), (rnd, k)
    # an eager step of a THIRD shape in between (rebinds the engine's tables again), then the first signature once more
    x3, m3 = O.seeded_inputs(1, 64, 64, [3, 3], seed=9)
    ys = net(x3.to(dev).requires_grad_(True), [m.to(dev) for m in m3])
    sum(y.sum() for y in ys).backward()
    x, mods, cots, want = sigs[0]
    got = _iter(net, x, mods, cots)
    for k, b in want[2].items():
        assert float((got[2][k] - b).abs().max()) <= 1e-4 * max(float(b.abs().max()), 1e-3 * gmax), k


@pytest.mark.gpu
def test_random_pools_do_not_grow_with_alternating_shapes_gpu():
    """Engine._rng_take keyed a call site's slot by (site, draw) only and appended a new span whenever the
    size changed - alternating shapes grew the Dropout / DropPath pools every step.  Slots are keyed by the size too."""
    dev = use_backend('hip')
    os.environ['HRF_MODULE_GRAPH'] = '0'
    try:
        net, _, _ = build_pair('t_nus_bn', dev, stochastic=True)
        net.train()
        eng = net._engine()
        sizes = []
        for it in range(6):
            B, H, W = ((2, 64, 96), (1, 96, 64))[it % 2]
            x, mods = O.seeded_inputs(B, H, W, [3, 3], seed=1)
            ys = net(x.to(dev).requires_grad_(True), [m.to(dev) for m in mods])
            sum(y.sum() for y in ys).backward()
            sizes.append(dict(eng._rng_plan))
        assert sizes[2] == sizes[3] == sizes[4] == sizes[5], sizes       # both shapes seen: the plan is final
        assert sum(sizes[-1].values()) > 0
    finally:
        os.environ.pop('HRF_MODULE_GRAPH', None)
