"""Where the time of the small row-GEMM launches goes: lin_fwd / lin_bwd_data at the CrossFFN fc3 shape (2x96x160, 72 <-> 18
channels) with the transform, the moment emission and the epilogue switched off one at a time.
python tools/bench_lin.py [out.json]"""
import json
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
B, H, W = 2, 96, 160
x72, y18, w3 = R(B, H, W, 72), R(B, H, W, 18), R(18, 72, 1, 1)
s72, t72 = R(72), R(72)
st18 = torch.zeros(16 * 18, dtype=torch.float64, device=dev)
st72 = torch.zeros(16 * 72, dtype=torch.float64, device=dev)
c18 = [R(18) for _ in range(3)]
dy18, dx72 = R(B, H, W, 18), R(B, H, W, 72)


def fwd(tf, stats):
    return lambda: L.hrf_conv_fwd(x72, H * W * 72, W * 72, 72, 1, B, H, W, 72, w3, None, 1, 1, 18, y18, 18, 0, None, None, 0, tf,
                                  s72 if tf else None, t72 if tf else None, None, stats, None, None, 0.0, sp())


def bwd(bnb, epi, stats):
    co = c18 if bnb else [None] * 3
    return lambda: L.hrf_conv_bwd_data(dy18, 18, 0, y18 if bnb else None, *co, None, w3, 1, 1, 18, B, H, W, 72, dx72, H * W * 72, W * 72,
                                       72, 1, 0, 1 if epi else 0, x72 if epi else None, 72 if epi else 0, s72 if epi else None,
                                       t72 if epi else None, 2 if epi else 0, stats if epi else None, sp())


rows = {}
for name, fn in [('lin_fwd 72->18 plain', fwd(0, None)), ('lin_fwd + moments', fwd(0, st18)), ('lin_fwd + affine', fwd(1, None)),
                 ('lin_fwd + affine+GELU', fwd(3, None)), ('lin_fwd + affine+GELU + moments', fwd(3, st18)),
                 ('lin_bwd 18->72 plain', bwd(False, False, None)), ('lin_bwd + BN-bwd on load', bwd(True, False, None)),
                 ('lin_bwd + BN-bwd + GELU\' epilogue', bwd(True, True, None)), ('lin_bwd + BN-bwd + epilogue + moments', bwd(True, True, st72))]:
    rows[name] = _graph_time(fn) * 1e6
    print(f'{name:42s} {rows[name]:6.2f} us', flush=True)
for C in (18, 72):                         # the floor of a launch in a graph: a trivial streaming kernel over the same rows
    a, b = R(B, H, W, C), R(B, H, W, C)
    name = f'scale_add 2x96x160x{C} (3 tensors)'
    rows[name] = _graph_time(lambda: L.hrf_scale_add(a, None, 1.0, None, 1, b, None, a, B * H * W, C, sp())) * 1e6
    print(f'{name:42s} {rows[name]:6.2f} us')
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
