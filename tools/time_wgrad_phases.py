"""Phase timestamps (wall_clock64, 100 MHz) of workgroup 17 (wave 0) of wgrad_dense_kernel on HRFuser-B / -T 1x1 shapes, plus the
graph-timed launch duration.  Needs a library with the stamps compiled in (never the product build):
    HRF_EXTRA_FLAGS=-DHRF_WG_TIMING python -m hrfuser_amd.build_ext --force      (or HRF_TIMING_LIB=<path of such a build>)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrfuser_amd import _lib                                   # noqa: E402

if os.environ.get('HRF_TIMING_LIB'):
    _lib.LIB_PATH = os.environ['HRF_TIMING_LIB']
from hrfuser_amd.profiling import _graph_time                  # noqa: E402

L = _lib.lib()
dll = L._dll
dev = torch.device('cuda:0')
R = lambda *sh: torch.randn(*sh, device=dev)
sp = _lib.stream_ptr


def wgrad(B, H, W, Cin, Cout, tf, bnb):
    dy, yraw, x = R(B, H, W, Cout), R(B, H, W, Cout), R(B, H, W, Cin)
    co = [R(Cout) for _ in range(3)] if bnb else [None] * 3
    sc, sh = R(Cin), R(Cin)
    rs = R(B * H * W, 2) if tf == 4 else None
    dw = torch.zeros(Cout, Cin, 1, 1, device=dev)
    return lambda: L.hrf_conv_bwd_weight(dy, Cout, 0, yraw if bnb else None, *co, x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 1, 1, Cout,
                                         tf, sc if tf else None, sh if tf else None, rs, dw, None, sp())


CASES = [('B fc1 dW 312x78  LN(x), BN-bwd(dy)     2x96x160', wgrad(2, 96, 160, 78, 312, 4, True)),
         ('B fc3 dW 78x312  GELU(BN(x)), BN-bwd   2x96x160', wgrad(2, 96, 160, 312, 78, 3, True)),
         ('B qkv dW 234x78  LN(x)                 2x96x160', wgrad(2, 96, 160, 78, 234, 4, False)),
         ('B out dW 78x78   plain                 2x96x160', wgrad(2, 96, 160, 78, 78, 0, False)),
         ('B fc1 dW 624x156 LN(x), BN-bwd         2x48x80 ', wgrad(2, 48, 80, 156, 624, 4, True)),
         ('B fc3 dW 312x1248 GELU(BN(x)), BN-bwd  2x24x40 ', wgrad(2, 24, 40, 1248, 312, 3, True)),
         ('B fc3 dW 624x2496 GELU(BN(x)), BN-bwd  2x12x20 ', wgrad(2, 12, 20, 2496, 624, 3, True)),
         ('T fc3 dW 18x72   GELU(BN(x)), BN-bwd   2x96x160', wgrad(2, 96, 160, 72, 18, 3, True)),
         ('T fc3 dW 72x288  GELU(BN(x)), BN-bwd   2x24x40 ', wgrad(2, 24, 40, 288, 72, 3, True)),
         ('T q   dW 72x72   plain                 2x24x40 ', wgrad(2, 24, 40, 72, 72, 0, False)),
         ('T fc1 dW 288x72  LN(x), BN-bwd         2x24x40 ', wgrad(2, 24, 40, 72, 288, 4, True)),
         ('T fc3 dW 36x144  GELU(BN(x)), BN-bwd   2x48x80 ', wgrad(2, 48, 80, 144, 36, 3, True)),
         ('T q   dW 144x144 plain                 2x12x20 ', wgrad(2, 12, 20, 144, 144, 0, False)),
         ('T fuse dW 18x18  plain, BN-bwd         2x48x80 ', wgrad(2, 48, 80, 18, 18, 0, True))]
if os.environ.get('HRF_WG_GROUPED'):
    # as the step issues them: queued between hrf_wgrad_group_begin / _end (16 pixel splits at most instead of 128)
    def grouped(fn):
        def run():
            L.hrf_wgrad_group_begin()
            fn()
            L.hrf_wgrad_group_end(sp())
        return run
    CASES = [(n + ' [grouped]', grouped(f)) for n, f in CASES]
for name, fn in CASES:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    line = f'{name} '
    if hasattr(dll, 'hrf_wgrad_stamps'):
        buf = (ctypes.c_longlong * 16)()
        dll.hrf_wgrad_stamps(buf)
        t = list(buf)
        d = lambda a, b: (t[b] - t[a]) / 100.0
        line += f'| setup {d(0, 1):5.2f} pixel loop {d(1, 2):6.2f} merge {d(2, 3):5.2f} atomics {d(3, 4):5.2f} tail {d(4, 5):5.2f} | block {d(0, 5):6.2f} us '
    us = _graph_time(fn) * 1e6
    print(line + f'| launch {us:7.2f} us', flush=True)
