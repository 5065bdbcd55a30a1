"""Isolated durations of the weight gradient of wide 1x1 problems: pixel-major kernel (hrf_debug_knob(8, 1)) vs the LDS-tiled
kernel (csrc/wgrad_tiled.hip), one problem per launch and 12 problems per grouped launch, priced against the fp32 MFMA peak.

    python tools/bench_wgrad_tiled.py
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
    db = torch.zeros(Cout, device=dev)
    s, t = R(Cin), R(Cin)
    c = [R(Cout) for _ in range(3)] if bnb else [None] * 3

    def call():
        L.hrf_conv_bwd_weight(dy, Cout, 0, yr if bnb else None, *c, x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 1, 1, Cout,
                              tf_mode, s if tf_mode else None, t if tf_mode else None, None, dw, db, sp())
    return call, (x, dy, yr, dw, db, s, t, c)


cases = [('312->78 @96x160 gelu bnb', (2, 96, 160, 312, 78, True, 3)), ('78->312 @96x160 bnb', (2, 96, 160, 78, 312, True, 0)),
         ('78->234 @96x160', (2, 96, 160, 78, 234, False, 0)), ('78->78 @96x160', (2, 96, 160, 78, 78, False, 0)),
         ('64->256 @96x160 relu bnb', (2, 96, 160, 64, 256, True, 2)), ('256->64 @96x160 bnb', (2, 96, 160, 256, 64, True, 1)),
         ('624->156 @48x80 gelu bnb', (2, 48, 80, 624, 156, True, 3)), ('288->72 @24x40 gelu bnb', (2, 24, 40, 288, 72, True, 3)),
         ('2496->624 @12x20 gelu bnb', (2, 12, 20, 2496, 624, True, 3))]
NG = 12
for name, args in cases:
    B, H, W, Cin, Cout = args[:5]
    gf = 2.0 * B * H * W * Cin * Cout / 1e9
    calls = [problem(*args) for _ in range(NG)]

    def grouped():
        L.hrf_wgrad_group_begin()
        for c, _ in calls:
            c()
        L.hrf_wgrad_group_end(sp())
    row = []
    for knob, tag in ((1, 'pixel-major'), (0, 'tiled')):
        L.hrf_debug_knob(8, knob)
        t1 = _graph_time(calls[0][0]) * 1e6
        tg = _graph_time(grouped) * 1e6
        row.append(f'{tag}: alone {t1:7.1f} us {gf / t1 * 1e3:5.1f} TF/s | {NG} grouped {tg:7.1f} us {NG * gf / tg * 1e3:5.1f} TF/s')
    L.hrf_debug_knob(8, 0)
    print(f'{name:28s}', '   '.join(row), flush=True)
