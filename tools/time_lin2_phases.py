"""Phase timestamps (wall_clock64, 100 MHz) of workgroup 100 of the lin2 kernels on HRFuser-B's shapes, plus graph-timed
launch durations of the same calls.  Needs a library with the stamps compiled in (never the product build):
    HRF_EXTRA_FLAGS=-DHRF_L2_TIMING python -m hrfuser_amd.build_ext --force
or a copy of such a build passed as HRF_TIMING_LIB=<path>."""
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
has_stamps = hasattr(dll, 'hrf_lin2_stamps')
dev = torch.device('cuda:0')
R = lambda *sh: torch.randn(*sh, device=dev)
sp = _lib.stream_ptr
L.hrf_debug_knob(28, 1)                                        # always the LDS-tiled engine


def stamps():
    if not has_stamps:
        return None
    buf = (ctypes.c_longlong * 64)()
    torch.cuda.synchronize()
    dll.hrf_lin2_stamps(buf)
    return list(buf)


def report(name, fn, S, passes):
    torch.cuda.synchronize()
    for _ in range(3):
        fn()
    t = stamps()
    us = _graph_time(fn) * 1e6
    line = f'{name:58s} {us:7.2f} us'
    if t is not None:
        d = lambda a, b: (t[b] - t[a]) / 100.0
        steps = [d(2 + s, 3 + s) for s in range(min(S, 40))]
        line += (f' | tables {d(0, 1):5.2f} fill {d(1, 2):5.2f} steps {sum(steps):6.2f} (first {steps[0]:.2f} mid {steps[len(steps) // 2]:.2f} last {steps[-1]:.2f})')
        last = 2 + min(S, 40)
        for p in range(passes):
            b = 44 + 4 * p
            line += f' | pass{p}: loads {d(last, b):5.2f} to-lds {d(b, b + 1):5.2f} rows {d(b + 1, b + 2):5.2f} moments {d(b + 2, b + 3):5.2f}'
            last = b + 3
        line += f' | block {d(0, last):6.2f} us'
    print(line, flush=True)


def fwd(B, H, W, K, N, tf, stats=True):
    x, w, y = R(B, H, W, K), R(N, K, 1, 1) * 0.05, R(B, H, W, N)
    sc, sh = R(K), R(K)
    rs = R(B * H * W, 2) if tf == 4 else None
    st = torch.zeros(16 * N, dtype=torch.float64, device=dev) if stats else None
    return lambda: L.hrf_conv_fwd(x, H * W * K, W * K, K, 1, B, H, W, K, w, None, 1, 1, N, y, N, 0, None, None, 0, tf,
                                  sc if tf else None, sh if tf else None, rs, st, None, None, 0.0, sp())


def bwd(B, H, W, K, N, bnb, epi):
    """dX[M][N] from dY[M][K]: conv Cin = N, Cout = K"""
    dy, yraw, w, dx = R(B, H, W, K), R(B, H, W, K), R(K, N, 1, 1) * 0.05, R(B, H, W, N)
    co = [R(K) for _ in range(3)] if bnb else [None] * 3
    xraw, sc, sh = R(B, H, W, N), R(N), R(N)
    st = torch.zeros(16 * N, dtype=torch.float64, device=dev) if epi else None
    return lambda: L.hrf_conv_bwd_data(dy, K, 0, yraw if bnb else None, *co, None, w, 1, 1, K, B, H, W, N, dx, H * W * N, W * N, N, 1, 0,
                                       1 if epi else 0, xraw if epi else None, N if epi else 0, sc if epi else None, sh if epi else None,
                                       2 if epi else 0, st, sp())


def steps(K):
    return (K + 15) // 16


def passes(N):
    return 1 if N <= 80 else 2


CASES = [
    ('fwd  fc1  78->312  LN on load, moments     2x96x160', fwd(2, 96, 160, 78, 312, 4), 78, 312),
    ('fwd  fc3 312->78   BN+GELU on load, moments 2x96x160', fwd(2, 96, 160, 312, 78, 3), 312, 78),
    ('fwd  qkv  78->234  LN on load              2x96x160', fwd(2, 96, 160, 78, 234, 4, False), 78, 234),
    ('fwd  plain 78->312 no transform, no moments 2x96x160', fwd(2, 96, 160, 78, 312, 0, False), 78, 312),
    ('bwd  fc3 dX K=78 -> N=312  BN-bwd + GELU\' + moments', bwd(2, 96, 160, 78, 312, True, True), 78, 312),
    ('bwd  fc1 dX K=312 -> N=78  BN-bwd                   ', bwd(2, 96, 160, 312, 78, True, False), 312, 78),
    ('bwd  plain K=78 -> N=312                            ', bwd(2, 96, 160, 78, 312, False, False), 78, 312),
    ('fwd  fc1 156->624 LN, moments               2x48x80', fwd(2, 48, 80, 156, 624, 4), 156, 624),
    ('bwd  fc1 dX K=624 -> N=156 BN-bwd           2x48x80', bwd(2, 48, 80, 624, 156, True, False), 624, 156),
]
for name, fn, K, N in CASES:
    report(name, fn, steps(K), passes(N))
