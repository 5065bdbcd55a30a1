"""Depthwise 3x3 kernels at the CrossFFN shapes of the HRFuser-T step: one-channel lanes (hrf_debug_knob(40, 1)) against the
float4-lane kernels with 8-row / 4-row tiles (40 = 2 / 3), graph-timed, with the operand pieces (moments, finalize-on-load +
GELU) switched off one at a time.  python tools/bench_dw.py [out.json]"""
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


MODES = {1: 'lane1', 2: 'lane4 th8', 3: 'lane4 th4'}
rows = []


def fwd_case(B, H, W, C):
    x, w, bias = R(B, H, W, C), R(C, 1, 3, 3) * 0.3, R(C)
    fin, ft = fin_of(C, B * H * W)
    sc, sh = torch.rand(C, device=dev) + 0.5, R(C) * 0.3
    ys = {}
    for what, (tf, f, sca, sha, want) in {'fin+GELU+moments': (3, fin, None, None, True), 'fin+GELU': (3, fin, None, None, False),
                                         'GELU+moments': (3, None, sc, sh, True), 'plain+moments': (0, None, None, None, True),
                                         'plain': (0, None, None, None, False)}.items():
        row = dict(name=f'dw_fwd {B}x{H}x{W}x{C} {what}', mbytes=2 * x.numel() * 4 / 1e6)
        for mode, tag in MODES.items():
            L.hrf_debug_knob(40, mode)
            y = torch.empty_like(x)
            st = torch.zeros(KC * 2 * C, dtype=torch.float64, device=dev)
            fn = lambda: L.hrf_dwconv_fwd(x, B, H, W, C, w, bias, 1, tf, sca, sha, y, st if want else None, f, sp())
            fn()
            torch.cuda.synchronize()
            ys[mode] = (y, st.view(KC, -1).sum(0))
            row[tag] = _graph_time(fn) * 1e6
        row['err'] = max(rel(ys[m][0], ys[1][0]) for m in (2, 3))
        row['stat_err'] = max(rel(ys[m][1], ys[1][1]) for m in (2, 3)) if want else 0.0
        rows.append(row)
    L.hrf_debug_knob(40, 1)


def bwd_case(B, H, W, C, wg=True):
    du, yraw, xr, w = R(B, H, W, C), R(B, H, W, C), R(B, H, W, C), R(C, 1, 3, 3) * 0.3
    bf, bt = bfin_of(C, B * H * W)
    sc, sh = torch.rand(C, device=dev) + 0.5, R(C) * 0.3
    n = 10 * C
    outs = {}
    row = dict(name=f'dw_bwd_data{"_weight" if wg else ""} {B}x{H}x{W}x{C} bfin+GELU\'+moments', mbytes=4 * du.numel() * 4 / 1e6)
    for mode, tag in MODES.items():
        L.hrf_debug_knob(40, mode)
        dx = torch.empty_like(du)
        st = torch.zeros(KC * 2 * C, dtype=torch.float64, device=dev)
        scr = torch.zeros(KC * n, device=dev)
        if wg:
            fn = lambda: L.hrf_dwconv_bwd_data_weight(du, yraw, bt['cA'], bt['cB'], bt['cC'], bf, w, B, H, W, C, dx, xr, sc, sh, 2, st, scr,
                                                      scr[9 * C:], n, sp())
        else:
            fn = lambda: L.hrf_dwconv_bwd_data(du, yraw, bt['cA'], bt['cB'], bt['cC'], bf, w, 1, B, H, W, C, dx, 0, 1, xr, sc, sh, 2, st, sp())
        fn()
        torch.cuda.synchronize()
        outs[mode] = (dx, st.view(KC, -1).sum(0), scr.view(KC, -1).sum(0))
        row[tag] = _graph_time(fn) * 1e6
    row['err'] = max(rel(outs[m][0], outs[1][0]) for m in (2, 3))
    row['stat_err'] = max(rel(outs[m][1], outs[1][1]) for m in (2, 3))
    row['dw_err'] = max(rel(outs[m][2], outs[1][2]) for m in (2, 3)) if wg else 0.0
    rows.append(row)
    L.hrf_debug_knob(40, 1)


for shape in [(2, 96, 160, 72), (2, 48, 80, 144), (2, 24, 40, 288), (2, 12, 20, 576)]:
    fwd_case(*shape)
if os.environ.get('DW_BWD', '1') != '0':
    for shape in [(2, 96, 160, 72), (2, 48, 80, 144), (2, 24, 40, 288), (2, 12, 20, 576)]:
        bwd_case(*shape)
    bwd_case(2, 96, 160, 72, wg=False)
for r in rows:
    print(f"{r['name']:64s} " + ' '.join(f"{t} {r[t]:6.2f} us ({r['mbytes'] / r[t]:4.2f} TB/s)" for t in MODES.values())
          + f" | err {r['err']:.1e} stats {r['stat_err']:.1e}" + (f" dw {r['dw_err']:.1e}" if 'dw_err' in r else ''), flush=True)
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
