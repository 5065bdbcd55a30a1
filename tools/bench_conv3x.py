"""Front-end 3x3 convolutions: the round-5 engines (conv3_engine / conv_engine, OIHW weights) against the packed-weight engine
(csrc/conv3x_engine.hip) at the shapes of the HRFuser-T / -B training step, isolated and graph-timed, with a cross-check of the
two results.  python tools/bench_conv3x.py [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrfuser_amd import _lib                                   # noqa: E402
from hrfuser_amd.profiling import _graph_time                  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
g = torch.Generator(device='cuda').manual_seed(1)
R = lambda *sh: torch.randn(*sh, device=dev, generator=g)
sp = _lib.stream_ptr
KC = _lib.STAT_COPIES
P = _lib._ptr


def pack(w, direction):
    Cout, Cin = w.shape[:2]
    wp = torch.empty(L.hrf_conv3x_pack_size(Cout, Cin, direction), device=dev)
    jobs = (_lib.Conv3xPackJob * 1)()
    jobs[0] = _lib.Conv3xPackJob(P(w), P(wp), Cout, Cin, direction)
    L.hrf_conv3x_pack(jobs, 1, sp())
    return wp


def moments(C, count):
    mean, var = torch.randn(C, device=dev) * 0.3, torch.rand(C, device=dev) * 0.8 + 0.4
    rows = torch.stack([mean, var + mean ** 2]).double() * count / KC
    return rows[None].repeat(KC, 1, 1).reshape(-1).contiguous()


def fin_of(C, count):
    t = dict(stats=moments(C, count), gamma=torch.rand(C, device=dev) + 0.5, beta=R(C) * 0.3, rm=R(C), rv=torch.rand(C, device=dev) + 0.5,
             scale=torch.zeros(C, device=dev), shift=torch.zeros(C, device=dev), mean=torch.zeros(C, device=dev), invstd=torch.zeros(C, device=dev))
    fin = _lib.BnFin(P(t['stats']), P(t['gamma']), P(t['beta']), P(t['rm']), P(t['rv']), P(t['scale']), P(t['shift']), P(t['mean']),
                     P(t['invstd']), float(count), 1e-5, 0.1, 0, 0, C)
    return fin, t


def bfin_of(C, count):
    t = dict(gstats=(torch.randn(KC * 2 * C, device=dev) * 0.2).double() * count / KC, gamma=torch.rand(C, device=dev) + 0.5,
             mean=R(C) * 0.3, invstd=torch.rand(C, device=dev) + 0.7, dgamma=R(C), dbeta=R(C), cA=torch.zeros(C, device=dev),
             cB=torch.zeros(C, device=dev), cC=torch.zeros(C, device=dev))
    bf = _lib.BnBFin(P(t['gstats']), P(t['gamma']), P(t['mean']), P(t['invstd']), P(t['dgamma']), P(t['dbeta']), P(t['cA']), P(t['cB']),
                     P(t['cC']), float(count), 1, 0, C)
    return bf, t


def rel(a, b):
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


rows = []


def fwd_case(name, B, H, W, Cin, Cout, tf, stride=1):
    x, w = R(B, H, W, Cin), R(Cout, Cin, 3, 3) * 0.1
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    fin, ft = fin_of(Cin, B * H * W) if tf else (None, None)
    st = (H * W * Cin, W * Cin, Cin, 1)
    y0, y1 = torch.empty(B, Ho, Wo, Cout, device=dev), torch.empty(B, Ho, Wo, Cout, device=dev)
    s0, s1 = (torch.zeros(KC * 2 * Cout, dtype=torch.float64, device=dev) for _ in range(2))
    wp = pack(w, 0)
    old = lambda y=y0, s=s0: L.hrf_conv_fwd(x, *st, B, H, W, Cin, w, None, 3, stride, Cout, y, Cout, 0, None, None, 0, tf, None, None, None, s, fin,
                                            None, 0.0, sp())
    new = lambda y=y1, s=s1: L.hrf_conv_fwd_packed(x, *st, B, H, W, Cin, w, None, 3, stride, Cout, y, Cout, 0, None, None, 0, tf, None, None, None,
                                                   s, fin, None, 0.0, wp, sp())
    old(); new()
    torch.cuda.synchronize()
    err = rel(y1, y0)
    serr = rel(s1.view(KC, -1).sum(0), s0.view(KC, -1).sum(0))
    flops = 2.0 * 9 * Cin * Cout * B * Ho * Wo
    rows.append(dict(name=name, old_us=_graph_time(old) * 1e6, new_us=_graph_time(new) * 1e6, gflop=flops / 1e9, err=err, stat_err=serr))


def bwd_case(name, B, H, W, Cin, Cout, stride, epi):
    """dX of a Cin -> Cout convolution on an (H, W) input grid"""
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    du, yraw, w = R(B, Ho, Wo, Cout), R(B, Ho, Wo, Cout), R(Cout, Cin, 3, 3) * 0.1
    bf, bt = bfin_of(Cout, B * Ho * Wo)
    st = (H * W * Cin, W * Cin, Cin, 1)
    xr, sc, sh = R(B, H, W, Cin), torch.rand(Cin, device=dev) + 0.5, R(Cin) * 0.3
    d0, d1 = R(B, H, W, Cin), None
    d1 = d0.clone()
    s0, s1 = (torch.zeros(KC * 2 * Cin, dtype=torch.float64, device=dev) for _ in range(2))
    wp = pack(w, 1)
    tail = lambda s: (0, 1, xr, Cin, sc, sh, 1, s) if epi else (1, 0, None, 0, None, None, 0, None)
    old = lambda: L.hrf_conv_bwd_data(du, Cout, 0, yraw, bt['cA'], bt['cB'], bt['cC'], bf, w, 3, stride, Cout, B, H, W, Cin, d0, *st,
                                      *tail(s0), sp())
    new = lambda: L.hrf_conv_bwd_data_packed(du, Cout, 0, yraw, bt['cA'], bt['cB'], bt['cC'], bf, w, 3, stride, Cout, B, H, W, Cin, d1, *st,
                                             *tail(s1), wp, sp())
    old(); new()
    torch.cuda.synchronize()
    err = rel(d1, d0)
    serr = rel(s1.view(KC, -1).sum(0), s0.view(KC, -1).sum(0)) if epi else 0.0
    flops = 2.0 * 9 * Cin * Cout * B * Ho * Wo
    rows.append(dict(name=name, old_us=_graph_time(old) * 1e6, new_us=_graph_time(new) * 1e6, gflop=flops / 1e9, err=err, stat_err=serr))


if os.environ.get('C3X_SMALL'):
    # the shapes of the 2x64x96 test network
    fwd_case('fwd 64->64 s1 2x16x24', 2, 16, 24, 64, 64, 2)
    bwd_case('bwd 64->64 s1 2x16x24', 2, 16, 24, 64, 64, 1, True)
    bwd_case('bwd 64->64 s2 2x32x48', 2, 32, 48, 64, 64, 2, True)
    bwd_case('bwd 256->18 s1 2x16x24', 2, 16, 24, 256, 18, 1, False)
    bwd_case('bwd 256->36 s2 2x16x24', 2, 16, 24, 256, 36, 2, False)
    bwd_case('bwd 256->36 s2 2x16x24 epi', 2, 16, 24, 256, 36, 2, True)
    bwd_case('bwd 256->18 s1 2x16x24 epi', 2, 16, 24, 256, 18, 1, True)
for nb in [int(v) for v in os.environ.get('C3X_B', '').split(',') if v]:
    fwd_case(f'fwd 64->64 s1 {nb}x96x160', nb, 96, 160, 64, 64, 2)
fwd_case('fwd 64->64 s1 2x96x160 (Bottleneck conv2)', 2, 96, 160, 64, 64, 2)
fwd_case('fwd 64->64 s2 2x192x320 (stem conv2)', 2, 192, 320, 64, 64, 2, 2)
fwd_case('fwd 256->36 s2 2x96x160 (transition)', 2, 96, 160, 256, 36, 0, 2)
fwd_case('fwd 18->36 s2 2x96x160 (fuse down)', 2, 96, 160, 18, 36, 2, 2)
bwd_case('bwd 64->64 s1 2x96x160', 2, 96, 160, 64, 64, 1, True)
bwd_case('bwd 64->64 s2 2x192x320 (stem conv2)', 2, 192, 320, 64, 64, 2, True)
bwd_case('bwd 256->18 s1 2x96x160 (transition)', 2, 96, 160, 256, 18, 1, False)
bwd_case('bwd 256->36 s2 2x96x160 (transition)', 2, 96, 160, 256, 36, 2, False)
fwd_case('fwd 256->36.. n/a: 256->64 s1 2x96x160 (4 slabs)', 2, 96, 160, 256, 64, 0)
fwd_case('fwd 64->64 s1 2x96x312 (STF)', 2, 96, 312, 64, 64, 2)
bwd_case('bwd 256->78 s1 2x96x160 (B transition)', 2, 96, 160, 256, 78, 1, False)
for r in rows:
    r['old_tflops'] = r['gflop'] / r['old_us'] * 1e3
    r['new_tflops'] = r['gflop'] / r['new_us'] * 1e3
    print(f"{r['name']:52s} old {r['old_us']:7.1f} us {r['old_tflops']:6.1f} TF | new {r['new_us']:7.1f} us {r['new_tflops']:6.1f} TF "
          f"({r['new_tflops'] / 157.3:.2f} of peak) | rel err {r['err']:.1e} stats {r['stat_err']:.1e}", flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
