"""What the BatchNorm moments of the fused attention block's FFN head cost: hrf_attn_block_fwd at 2x96x160x18 re-issued with and
without `stats1` (644 workgroups x 144 fp64 atomics into 4 copies).   python tools/ab_fwd_moments.py"""
import os
import sys

import torch

os.environ.setdefault('HRF_LANES', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hrfuser_amd.backbone as B                               # noqa: E402
from hrfuser_amd import _lib, profiling                        # noqa: E402
from hrfuser_amd.testing import BlockHarness                   # noqa: E402

NORM = dict(type='BN', requires_grad=True, momentum=0.1)
LN = dict(type='LN', eps=1e-6)
dev = torch.device('cuda:0')
C, heads, H, W = (int(sys.argv[1]), int(sys.argv[2]), 48, 80) if len(sys.argv) > 2 else (18, 1, 96, 160)
blk = B.HRFormerBlock(C, C, heads, norm_cfg=NORM, transformer_norm_cfg=LN)
hn = BlockHarness(blk, lambda ctx, b, x: b.run(ctx, x[0])).to(dev)
hn.train()
x = torch.randn(2, C, H, W, device=dev, requires_grad=True)
real = _lib.lib
base = real()
prof = profiling.ProfLib(base, timing=False)
hn(x)
torch.cuda.synchronize()
_lib.lib = lambda: prof
try:
    hn(x)
    torch.cuda.synchronize()
finally:
    _lib.lib = real
rec = [(n, a, args) for n, a, args in prof.records if n == 'hrf_attn_block_fwd']
assert rec, [n for n, _, _ in prof.records]
name, a, args = rec[0]
st = args[0]                                                   # the hrf_attn_block_t the engine passed
fn = getattr(base, name)
t_with = profiling._graph_time(lambda: fn(st, _lib.stream_ptr())) * 1e6
saved = st.stats1
st.stats1 = None
t_without = profiling._graph_time(lambda: fn(st, _lib.stream_ptr())) * 1e6
st.stats1 = saved
print(f'attn_block_fwd 2x{H}x{W}x{C}: with moments {t_with:.2f} us, without {t_without:.2f} us, moments cost {t_with - t_without:.2f} us')
