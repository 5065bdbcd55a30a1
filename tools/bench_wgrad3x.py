"""3x3 weight gradients of the front end: the pixel-major kernel (hrf_conv_bwd_weight, atomics) against the LDS-staged kernel
(hrf_conv_bwd_weight_s: csrc/wgrad3x_engine.hip, slabs + fold) per problem, isolated and graph-timed, for several block counts
(hrf_debug_knob 32).  python tools/bench_wgrad3x.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrfuser_amd import _lib                                   # noqa: E402
from hrfuser_amd.profiling import _graph_time                  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
R = lambda *sh: torch.randn(*sh, device=dev)
sp = _lib.stream_ptr


def case(name, B, H, W, Cin, Cout, stride, tf):
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    du, yraw, x = R(B, Ho, Wo, Cout), R(B, Ho, Wo, Cout), R(B, H, W, Cin)
    co = [R(Cout) for _ in range(3)]
    sc, sh = (torch.rand(Cin, device=dev) + 0.5, R(Cin) * 0.3) if tf else (None, None)
    st = (H * W * Cin, W * Cin, Cin, 1)
    dw0, dw1 = torch.zeros(Cout, Cin, 3, 3, device=dev), torch.zeros(Cout, Cin, 3, 3, device=dev)
    old = lambda: L.hrf_conv_bwd_weight(du, Cout, 0, yraw, *co, x, *st, B, H, W, Cin, 3, stride, Cout, tf, sc, sh, None, dw0, None, sp())
    old()
    torch.cuda.synchronize()
    t_old = _graph_time(old) * 1e6
    gf = 2.0 * 9 * Cin * Cout * B * Ho * Wo / 1e9
    out = [f'{name:38s} old {t_old:6.1f} us {gf / t_old * 1e3:5.1f} TF |']
    for blocks in (256, 384, 512, 768):
        L.hrf_debug_knob(32, blocks)
        nsc = L.hrf_conv_bwd_weight_scratch(*st, B, H, W, Cin, 3, stride, Cout, tf, 0)
        scratch = torch.empty(nsc, device=dev)
        new = lambda: L.hrf_conv_bwd_weight_s(du, Cout, 0, yraw, *co, x, *st, B, H, W, Cin, 3, stride, Cout, tf, sc, sh, None, dw1, None,
                                              scratch, sp())
        dw0.zero_(); dw1.zero_()
        old(); new()
        torch.cuda.synchronize()
        err = float((dw1 - dw0).abs().max() / dw0.abs().max())
        t = _graph_time(new) * 1e6
        out.append(f' {blocks}: {t:5.1f} us {gf / t * 1e3:5.1f} TF ({nsc * 4 / 1e6:.0f} MB, err {err:.0e})')
    L.hrf_debug_knob(32, 0)
    print(''.join(out), flush=True)


case('64->64 s1 2x96x160 tf2', 2, 96, 160, 64, 64, 1, 2)
case('64->64 s2 2x192x320 tf2', 2, 192, 320, 64, 64, 2, 2)
case('256->18 s1 2x96x160', 2, 96, 160, 256, 18, 1, 0)
case('256->36 s2 2x96x160', 2, 96, 160, 256, 36, 2, 0)
