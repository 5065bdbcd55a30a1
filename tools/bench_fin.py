"""Cost of the BatchNorm finalize-on-load prologue: isolated durations of the hot consumer kernels of the CrossFFN chain at
the branch-0 size with the hrf_bn_fin_t / hrf_bn_bfin_t form vs. precomputed scale/shift (coefficients) in memory.

    python tools/bench_fin.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for _d in ('tests', 'oracle'):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), _d))
from hrfuser_amd import _lib                                   # noqa: E402
from hrfuser_amd.profiling import _graph_time                  # noqa: E402
from test_kernels import make_fin, make_bfin                   # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
R = lambda *sh: torch.randn(*sh, device=dev)
sp = _lib.stream_ptr
B, H, W = 2, 96, 160
P = B * H * W
x72, y18, y72 = R(B, H, W, 72), R(B, H, W, 18), R(B, H, W, 72)
w3, wd, bd = R(18, 72, 1, 1), R(72, 1, 3, 3), R(72)
s72, t72 = R(72), R(72)
st18 = torch.zeros(32 * 18, dtype=torch.float64, device=dev)
st72 = torch.zeros(32 * 72, dtype=torch.float64, device=dev)
fin72, keep1 = make_fin(L, 72, P, dev, g)
bfin18, keep2 = make_bfin(L, 18, P, dev, g)
bfin72, keep3 = make_bfin(L, 72, P, dev, g)
c18 = [R(18) for _ in range(3)]
c72 = [R(72) for _ in range(3)]
dy18, dx72, dy72 = R(B, H, W, 18), R(B, H, W, 72), R(B, H, W, 72)


def t(name, fn_pre, fn_fin):
    a, b = _graph_time(fn_pre) * 1e6, _graph_time(fn_fin) * 1e6
    print(f'{name:34s} precomputed {a:6.2f} us   on-load {b:6.2f} us   delta {b - a:+5.2f}', flush=True)


t('lin_fwd fc3 72->18 (BN+GELU on load)',
  lambda: L.hrf_conv_fwd(x72, H * W * 72, W * 72, 72, 1, B, H, W, 72, w3, None, 1, 1, 18, y18, 18, 0, None, None, 0, 3, s72, t72, None, st18, None, None, 0.0, sp()),
  lambda: L.hrf_conv_fwd(x72, H * W * 72, W * 72, 72, 1, B, H, W, 72, w3, None, 1, 1, 18, y18, 18, 0, None, None, 0, 3, s72, t72, None, st18, fin72, None, 0.0, sp()))
t('dw_fwd 72 (BN+GELU on load)',
  lambda: L.hrf_dwconv_fwd(x72, B, H, W, 72, wd, bd, 1, 3, s72, t72, y72, st72, None, sp()),
  lambda: L.hrf_dwconv_fwd(x72, B, H, W, 72, wd, bd, 1, 3, s72, t72, y72, st72, fin72, sp()))
t('lin_bwd_data fc3 18->72 (epi)',
  lambda: L.hrf_conv_bwd_data(dy18, 18, 0, y18, *c18, None, w3, 1, 1, 18, B, H, W, 72, dx72, H * W * 72, W * 72, 72, 1, 0, 1, x72, 72, s72, t72, 2, st72, sp()),
  lambda: L.hrf_conv_bwd_data(dy18, 18, 0, y18, *c18, bfin18, w3, 1, 1, 18, B, H, W, 72, dx72, H * W * 72, W * 72, 72, 1, 0, 1, x72, 72, s72, t72, 2, st72, sp()))
t('dw_bwd_data 72 (epi)',
  lambda: L.hrf_dwconv_bwd_data(dy72, y72, *c72, None, wd, 1, B, H, W, 72, dx72, 0, 1, x72, s72, t72, 2, st72, sp()),
  lambda: L.hrf_dwconv_bwd_data(dy72, y72, *c72, bfin72, wd, 1, B, H, W, 72, dx72, 0, 1, x72, s72, t72, 2, st72, sp()))
fin18, keep4 = make_fin(L, 18, P, dev, g)
s18, t18, o18 = R(18), R(18), R(B, H, W, 18)
t('affine_act_res 18 (tail)',
  lambda: L.hrf_affine_act_res(y18, s18, t18, None, None, None, dy18, None, 0, 0, 0, o18, P, 18, None, 0.0, None, None, sp()),
  lambda: L.hrf_affine_act_res(y18, s18, t18, None, None, None, dy18, None, 0, 0, 0, o18, P, 18, None, 0.0, fin18, None, sp()))
scr = torch.zeros(8 * 720, device=dev)
t('dw_bwd_data 72: plain vs +weight gradient',
  lambda: L.hrf_dwconv_bwd_data(dy72, y72, *c72, None, wd, 1, B, H, W, 72, dx72, 0, 1, x72, s72, t72, 2, st72, sp()),
  lambda: L.hrf_dwconv_bwd_data_weight(dy72, y72, *c72, None, wd, B, H, W, 72, dx72, x72, s72, t72, 2, st72, scr, scr[648:], 720, sp()))
t('dw_bwd_data 72: plain vs +weight gradient, no stats',
  lambda: L.hrf_dwconv_bwd_data(dy72, y72, *c72, None, wd, 1, B, H, W, 72, dx72, 0, 1, x72, s72, t72, 2, None, sp()),
  lambda: L.hrf_dwconv_bwd_data_weight(dy72, y72, *c72, None, wd, B, H, W, 72, dx72, x72, s72, t72, 2, None, scr, scr[648:], 720, sp()))
