"""Isolated (graph-batched, HIP-event timed) durations of the dense weight-gradient kernel over its pixel-split cap
(debug knob 3) and the plain-store debug mode, on the small and the dominant problems of the HRFuser-T step.

    python tools/bench_wgrad.py
"""
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


def problem(B, H, W, Cin, Cout, bnb, tf_mode):
    x, dy, yr = R(B, H, W, Cin), R(B, H, W, Cout), R(B, H, W, Cout)
    dw = torch.zeros(Cout, Cin, device=dev)
    s, t = R(Cin), R(Cin)
    c = [R(Cout) for _ in range(3)] if bnb else [None] * 3

    def call():
        L.hrf_conv_bwd_weight(dy, Cout, 0, yr if bnb else None, *c, x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 1, 1, Cout,
                              tf_mode, s if tf_mode else None, t if tf_mode else None, None, dw, None, sp())
    return call, (x, dy, yr, dw, s, t, c)


cases = [('72->72 @24x40', (2, 24, 40, 72, 72, False, 0)), ('288->72 @24x40 gelu bnb', (2, 24, 40, 288, 72, True, 3)),
         ('144->144 @12x20', (2, 12, 20, 144, 144, False, 0)), ('72->18 @96x160 gelu bnb', (2, 96, 160, 72, 18, True, 3)),
         ('144->36 @48x80 gelu bnb', (2, 48, 80, 144, 36, True, 3)), ('64->256 @96x160 relu bnb', (2, 96, 160, 64, 256, True, 2))]
for name, args in cases:
    call, keep = problem(*args)
    row = []
    for cap in (0, 1, 2, 4, 8, 16, 32, 64):
        L.hrf_debug_knob(3, cap)
        row.append(f'cap{cap}={_graph_time(call) * 1e6:6.1f}')
    L.hrf_debug_knob(3, 0)
    L.hrf_debug_knob(1, 1)
    row.append(f'plain={_graph_time(call) * 1e6:6.1f}')
    L.hrf_debug_knob(1, 0)
    print(f'{name:28s}', ' '.join(row), flush=True)


def problem3(B, H, W, Cin, Cout, stride, bnb, tf_mode):
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    x, dy, yr = R(B, H, W, Cin), R(B, Ho, Wo, Cout), R(B, Ho, Wo, Cout)
    dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
    s, t = R(Cin), R(Cin)
    c = [R(Cout) for _ in range(3)] if bnb else [None] * 3

    def call():
        L.hrf_conv_bwd_weight(dy, Cout, 0, yr if bnb else None, *c, x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 3, stride, Cout,
                              tf_mode, s if tf_mode else None, t if tf_mode else None, None, dw, None, sp())
    return call, (x, dy, yr, dw, s, t, c)


print('3x3 weight gradients: n\' = ci*9+tap variant (knob 1 = 2) vs tap-blocked')
for name, args in [('64->64 s1 @96x160 relu bnb', (2, 96, 160, 64, 64, 1, True, 2)), ('64->64 s2 @192x320 relu bnb', (2, 192, 320, 64, 64, 2, True, 2)),
                   ('256->18 s1 @96x160 bnb', (2, 96, 160, 256, 18, 1, True, 0)), ('256->36 s2 @96x160 bnb', (2, 96, 160, 256, 36, 2, True, 0)),
                   ('18->18 s2 @96x160 bnb', (2, 96, 160, 18, 18, 2, True, 0)), ('36->72 s2 @48x80 bnb', (2, 48, 80, 36, 72, 2, True, 0))]:
    call, keep = problem3(*args)
    L.hrf_debug_knob(1, 2)
    t_old = _graph_time(call) * 1e6
    L.hrf_debug_knob(1, 0)
    row = [f'old={t_old:6.1f}', f'new={_graph_time(call) * 1e6:6.1f}']
    for cap in (16, 32, 64, 128):
        L.hrf_debug_knob(3, cap)
        row.append(f'cap{cap}={_graph_time(call) * 1e6:6.1f}')
    L.hrf_debug_knob(3, 0)
    print(f'{name:28s}', ' '.join(row), flush=True)
