"""Multi-problem launches (csrc/hrf_group.h, runtime.Strand): the equal-shape layers of the camera stream's finest branch and
the modality streams are issued as ONE launch.  CPU suite: the queue / merge logic of hrf_group_begin / hrf_group_end on the
emulator, and the lock-step scheduler against the serial one (results must be IDENTICAL: merging never changes arithmetic).
GPU: the same equality on the real kernels, plus the launch count of a training step."""
import os

import pytest
import torch

import hrfuser_oracle as O
from helpers import build_pair, use_backend
from hrfuser_amd import _lib


def _two_convs(L, dev, group):
    g = torch.Generator().manual_seed(0)
    outs = []
    if group:
        L.hrf_group_begin()
    for k in range(3):
        B, H, W, Cin, Cout = 2, 9, 11, 18, 72
        x = torch.randn(B, H, W, Cin, generator=g).to(dev)
        w = torch.randn(Cout, Cin, 1, 1, generator=g).to(dev)
        b = torch.randn(Cout, generator=g).to(dev)
        y = torch.full((B, H, W, Cout), float('nan'), device=dev)
        L.hrf_conv_fwd(x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, w, b, 1, 1, Cout, y, Cout, 0, None, None, 0,
                       0, None, None, None, None, None, None, 0.0, _lib.stream_ptr())
        outs.append((x, w, b, y))
    # a call of another shape in the same bracket: issued on its own, in order
    x2 = torch.randn(1, 5, 7, 36, generator=g).to(dev)
    w2 = torch.randn(36, 36, 1, 1, generator=g).to(dev)
    y2 = torch.full((1, 5, 7, 36), float('nan'), device=dev)
    L.hrf_conv_fwd(x2, 5 * 7 * 36, 7 * 36, 36, 1, 1, 5, 7, 36, w2, None, 1, 1, 36, y2, 36, 0, None, None, 0,
                   0, None, None, None, None, None, None, 0.0, _lib.stream_ptr())
    if group:
        L.hrf_group_end(_lib.stream_ptr())
    return [o[3] for o in outs] + [y2], outs, (x2, w2)


def _queue_merge(backend):
    dev = use_backend(backend)
    L = _lib.lib()
    c0 = [L.hrf_group_count(k) for k in range(3)]
    ys_g, outs, (x2, w2) = _two_convs(L, dev, True)
    c1 = [L.hrf_group_count(k) for k in range(3)]
    ys_s, _, _ = _two_convs(L, dev, False)
    cap = L.hrf_group_count(3)                             # problems per launch of this build (product: 1, emulator tests: 4)
    assert c1[2] - c0[2] == 1                              # one hrf_group_end
    assert c1[1] - c0[1] == 4                              # four calls' launches ...
    assert c1[0] - c0[0] == (2 if cap >= 3 else 4)         # ... in two launches (3 merged + 1 alone) when compiled for it
    for a, b in zip(ys_g, ys_s):
        assert torch.equal(a, b)                           # merging never changes arithmetic
    for (x, w, b, y) in outs:
        ref = torch.einsum('bhwc,oc->bhwo', x.double().cpu(), w.double().cpu()[:, :, 0, 0]) + b.double().cpu()
        assert float((y.double().cpu() - ref).abs().max()) < 1e-4
    assert not bool(torch.isnan(ys_g[-1]).any())


def test_group_queue_merge_emul():
    _queue_merge('emul')


@pytest.mark.gpu
def test_group_queue_merge_gpu():
    _queue_merge('hip')


def _step(net, x, mods, dev, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        net.zero_grad(set_to_none=False)
        xa = x.clone().to(dev).requires_grad_(True)
        ma = [m.clone().to(dev).requires_grad_(True) for m in mods]
        ys = net(xa, list(ma))
        g = torch.Generator().manual_seed(5)
        cots = [torch.randn(t.shape, generator=g) for t in ys]
        sum((t * c.to(dev)).sum() for t, c in zip(ys, cots)).backward()
        grads = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        return [y.detach().clone() for y in ys], [xa.grad.clone()] + [m.grad.clone() for m in ma], grads
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _lockstep_equals_serial(backend, tag, B, H, W):
    """The same forward + backward with merged launches on lock-step strands, with lock-step strands but no merging, and
    with the bodies run one after the other: outputs identical; gradients identical up to the order of atomic additions."""
    dev = use_backend(backend)
    net, orc, cfg = build_pair(tag, dev)
    mc = cfg.get('mod_in_channels', [3, 3])
    x, mods = O.seeded_inputs(B, H, W, mc, seed=1)
    net.train()
    L = _lib.lib()
    c0 = [L.hrf_group_count(k) for k in range(2)]
    ya, ga, pa = _step(net, x, mods, dev, {'HRF_LOCKSTEP': '1', 'HRF_GROUP': '1'})
    c1 = [L.hrf_group_count(k) for k in range(2)]
    yb, gb, pb = _step(net, x, mods, dev, {'HRF_LOCKSTEP': '0'})
    c2 = [L.hrf_group_count(k) for k in range(2)]
    yc, gc, pc = _step(net, x, mods, dev, {'HRF_LOCKSTEP': '1', 'HRF_GROUP': '0'})
    merged, carried = c1[0] - c0[0], c1[1] - c0[1]
    print(f'[{tag}] merged launches {merged} carrying {carried} calls; serial pass issued {c2[0] - c1[0]} through the group path')
    if L.hrf_group_count(3) > 1:
        assert carried > merged > 0, (merged, carried)      # the sensor streams' equal layers really shared launches
    else:
        assert carried == merged == 0                       # a build for one problem per launch never brackets
    assert c2[0] == c1[0]                                    # the serial schedule never brackets
    for a, b, c in zip(ya, yb, yc):
        assert torch.equal(a, b) and torch.equal(a, c)
    rel = lambda u, v: float((u.double() - v.double()).abs().max() / (v.double().abs().max() + 1e-30))
    for a, b, c in zip(ga, gb, gc):
        assert rel(a, b) < 1e-5 and rel(c, b) < 1e-5
    gmax = max(float(v.abs().max()) for v in pb.values())
    for k in pb:
        den = max(float(pb[k].abs().max()), 1e-3 * gmax)
        assert float((pa[k] - pb[k]).abs().max()) / den < 1e-4, k
        assert float((pc[k] - pb[k]).abs().max()) / den < 1e-4, k


def test_lockstep_equals_serial_emul():
    _lockstep_equals_serial('emul', 't_nus_bn', 1, 32, 64)


@pytest.mark.gpu
@pytest.mark.parametrize('tag,B,H,W', [('t_nus_bn', 2, 64, 96), ('t_stf_bn', 1, 64, 128)])
def test_lockstep_equals_serial_gpu(tag, B, H, W):
    _lockstep_equals_serial('hip', tag, B, H, W)


def test_leaf_balancer_keeps_variants_together():
    """runtime._balance: every leaf exactly once; leaves with the same key (one kernel variant of the grouped weight-gradient
    launches) travel in chunks of `chunk`, i.e. a variant with n problems is spread over at most ceil(n / chunk) lanes; keyless
    leaves are placed one by one; the lane loads stay balanced (longest-processing-time-first over the chunks)."""
    from hrfuser_amd.runtime import _balance
    log = []
    mk = lambda name: (lambda: log.append(name))
    items = [(10.0 + i, mk(f'a{i}'), ('conv_w', 72, 18)) for i in range(19)] + \
            [(3.0, mk(f'b{i}'), ('conv_w', 288, 72)) for i in range(5)] + \
            [(50.0, mk('fold')), (7.0, mk('rpb'), None)] + [(1.0, mk(f'd{i}'), ('dw_w', 2)) for i in range(9)]
    parts = _balance(items, 4, chunk=8)
    assert len(parts) == 4
    lanes = []
    for p in parts:
        log.clear()
        for fn in p:
            fn()
        lanes.append(list(log))
    flat = [n for lane in lanes for n in lane]
    assert sorted(flat) == sorted([f'a{i}' for i in range(19)] + [f'b{i}' for i in range(5)] + ['fold', 'rpb'] + [f'd{i}' for i in range(9)])
    for prefix, n in (('a', 19), ('b', 5), ('d', 9)):
        used = [sum(1 for x in lane if x.startswith(prefix)) for lane in lanes]
        assert sum(used) == n and sum(1 for u in used if u) <= -(-n // 8), (prefix, used)
    loads = []
    cost = {f'a{i}': 10.0 + i for i in range(19)}
    cost.update({f'b{i}': 3.0 for i in range(5)}); cost.update({'fold': 50.0, 'rpb': 7.0}); cost.update({f'd{i}': 1.0 for i in range(9)})
    for lane in lanes:
        loads.append(sum(cost[x] for x in lane))
    assert max(loads) <= 0.5 * sum(loads)                      # no lane carries more than half of the phase
    # chunk = 1: the round-3 behaviour (every leaf on its own)
    parts1 = _balance(items, 4, chunk=1)
    assert sum(len(p) for p in parts1) == len(items)
