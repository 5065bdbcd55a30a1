"""Launches the hot kernels of the training step eagerly (4 times each) on their branch-0 (96x160, 2 images) shapes so
that rocprofv3 --pmc can attribute hardware counters (FETCH_SIZE / WRITE_SIZE / SQ_*) to them:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_kernels.py

(tools/prof_r02.sh drives it; profiles/README.md lists the collected numbers and the gfx950 corrections applied)."""
import os
import sys

import torch

os.environ.setdefault('HRF_LANES', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrfuser_amd import _lib                                   # noqa: E402
import hrfuser_amd.backbone as BB                              # noqa: E402
from hrfuser_amd.testing import BlockHarness                   # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
R = lambda *sh: torch.randn(*sh, device=dev)
sp = _lib.stream_ptr
B, H, W = 2, 96, 160
P = B * H * W
# dominant signature of the step (bench.py roofline.dominant_shape): CrossFFN fc3 weight gradient, 72 -> 18 channels,
# GELU(BN(.)) operand, BatchNorm-backward on dY
x72, dy18, yr18 = R(B, H, W, 72), R(B, H, W, 18), R(B, H, W, 18)
dw3 = torch.zeros(18, 72, device=dev)
s72, c18 = R(72), [R(18) for _ in range(3)]
# the 3x3 64 -> 64 convolution of the Bottlenecks: forward, data gradient, weight gradient
Cin = Cout = 64
x, dy, yr = R(B, H, W, Cin), R(B, H, W, Cout), R(B, H, W, Cout)
dw = torch.zeros(Cout, Cin, 3, 3, device=dev)
sc, sh, cA, cB, cC = R(Cin), R(Cin), R(Cout), R(Cout), R(Cout)
w, y, dx = R(Cout, Cin, 3, 3), R(B, H, W, Cout), R(B, H, W, Cin)
st = torch.zeros(32 * Cout, dtype=torch.float64, device=dev)
# CrossFFN depthwise 3x3 on the 72-channel hidden map and the fc3 forward / data gradient
wd, bd, y72 = R(72, 1, 3, 3), R(72), R(B, H, W, 72)
st72 = torch.zeros(32 * 72, dtype=torch.float64, device=dev)
w3, y18 = R(18, 72, 1, 1), R(B, H, W, 18)
dx72, dwd = R(B, H, W, 72), torch.zeros(_lib.STAT_COPIES * 10 * 72, device=dev)
st18 = torch.zeros(32 * 18, dtype=torch.float64, device=dev)
# one HRFormerBlock at the branch-0 size: the fused attention block kernels (forward, backward) and the slot fold
blk = BB.HRFormerBlock(18, 18, 1, norm_cfg=dict(type='BN', requires_grad=True, momentum=0.1), transformer_norm_cfg=dict(type='LN', eps=1e-6))
hn = BlockHarness(blk, lambda ctx, b, xs: b.run(ctx, xs[0])).to(dev)
hn.train()
xb = torch.randn(B, 18, H, W, device=dev, requires_grad=True)
gb = torch.randn(B, 18, H, W, device=dev)
# HRFuser-B's CrossFFN GEMMs at the 96x160 branch (round 3: the LDS-tiled lin2 engine): fc1 forward 78 -> 312 with LayerNorm on load
# and BatchNorm moments, the fc3 data gradient (K = 78 -> N = 312: BatchNorm backward on load, GELU' epilogue, moments), the
# fc1 weight gradient 312 x 78
bx78, bw1, by312 = R(B, H, W, 78), R(312, 78, 1, 1) * 0.05, R(B, H, W, 312)
brs, bg78 = R(P, 2), R(78)
bst312 = torch.zeros(32 * 312, dtype=torch.float64, device=dev)
bdy78, byr78, bw3, bdx312, bxr312 = R(B, H, W, 78), R(B, H, W, 78), R(78, 312, 1, 1) * 0.05, R(B, H, W, 312), R(B, H, W, 312)
bc78, bs312 = [R(78) for _ in range(3)], R(312)
bdw1, bdy312, byr312, bc312 = torch.zeros(312, 78, device=dev), R(B, H, W, 312), R(B, H, W, 312), [R(312) for _ in range(3)]
# round 6: the packed-weight front-end engine (csrc/conv3x_engine.hip: forward, stride-1 and stride-2 data gradient) and the LDS-staged
# 3x3 weight gradient (csrc/wgrad3x_engine.hip) on the same 64 -> 64 problem
def _pack(wt, direction):
    co_, ci_ = wt.shape[:2]
    wp_ = torch.empty(L.hrf_conv3x_pack_size(co_, ci_, direction), device=dev)
    jobs = (_lib.Conv3xPackJob * 1)()
    jobs[0] = _lib.Conv3xPackJob(wt.data_ptr(), wp_.data_ptr(), co_, ci_, direction)
    L.hrf_conv3x_pack(jobs, 1, sp())
    return wp_


wpf, wpb = _pack(w, 0), _pack(w, 1)
H2, W2 = 2 * H, 2 * W
dx2, x2 = R(B, H2, W2, Cin), R(B, H2, W2, Cin)
nsc = L.hrf_conv_bwd_weight_scratch(H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 3, 1, Cout, 2, 0)
wscr = torch.empty(max(1, nsc), device=dev)
for it in range(4):
    L.hrf_conv_fwd_packed(x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, w, None, 3, 1, Cout, y, Cout, 0, None, None, 0, 2, sc, sh, None, st, None, None, 0.0, wpf, sp())
    L.hrf_conv_bwd_data_packed(dy, Cout, 0, yr, cA, cB, cC, None, w, 3, 1, Cout, B, H, W, Cin, dx, H * W * Cin, W * Cin, Cin, 1, 0, 1, x, Cin, sc, sh, 1, st, wpb, sp())
    L.hrf_conv_bwd_data_packed(dy, Cout, 0, yr, cA, cB, cC, None, w, 3, 2, Cout, B, H2, W2, Cin, dx2, H2 * W2 * Cin, W2 * Cin, Cin, 1, 0, 1, x2, Cin, sc, sh, 1, st, wpb, sp())
    if nsc > 0:
        L.hrf_conv_bwd_weight_s(dy, Cout, 0, yr, cA, cB, cC, x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 3, 1, Cout, 2, sc, sh, None, dw, None, wscr, sp())
    L.hrf_conv_fwd(bx78, H * W * 78, W * 78, 78, 1, B, H, W, 78, bw1, None, 1, 1, 312, by312, 312, 0, None, None, 0, 4, bg78, bg78, brs, bst312, None, None, 0.0, sp())
    L.hrf_conv_bwd_data(bdy78, 78, 0, byr78, *bc78, None, bw3, 1, 1, 78, B, H, W, 312, bdx312, H * W * 312, W * 312, 312, 1, 0, 1, bxr312, 312, bs312, bs312, 2, bst312, sp())
    L.hrf_conv_bwd_weight(bdy312, 312, 0, byr312, *bc312, bx78, H * W * 78, W * 78, 78, 1, B, H, W, 78, 1, 1, 312, 4, bg78, bg78, brs, bdw1, None, sp())
    L.hrf_conv_bwd_weight(dy18, 18, 0, yr18, *c18, x72, H * W * 72, W * 72, 72, 1, B, H, W, 72, 1, 1, 18, 3, s72, s72, None, dw3, None, sp())
    L.hrf_conv_bwd_weight(dy, Cout, 0, yr, cA, cB, cC, x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 3, 1, Cout, 2, sc, sh, None, dw, None, sp())
    L.hrf_conv_fwd(x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, w, None, 3, 1, Cout, y, Cout, 0, None, None, 0, 2, sc, sh, None, st, None, None, 0.0, sp())
    L.hrf_conv_bwd_data(dy, Cout, 0, yr, cA, cB, cC, None, w, 3, 1, Cout, B, H, W, Cin, dx, H * W * Cin, W * Cin, Cin, 1, 0, 1, x, Cin, sc, sh, 1, st, sp())
    L.hrf_dwconv_fwd(x72, B, H, W, 72, wd, bd, 1, 3, s72, s72, y72, st72, None, sp())
    # round 6: the rest of the CrossFFN chain of the 18-channel branch (the "streaming" kernels) - the depthwise data +
    # weight gradient (BatchNorm backward on load, GELU' epilogue, moments, dW / db) and the fc3 data gradient 18 -> 72
    L.hrf_dwconv_bwd_data_weight(y72, x72, s72, s72, s72, None, wd, B, H, W, 72, dx72, x72, s72, s72, 2, st72, dwd, dwd[9 * 72:], 10 * 72, sp())
    L.hrf_conv_bwd_data(dy18, 18, 0, yr18, *c18, None, w3, 1, 1, 18, B, H, W, 72, dx72, H * W * 72, W * 72, 72, 1, 0, 1, x72, 72, s72, s72, 2, st72, sp())
    L.hrf_conv_fwd(x72, H * W * 72, W * 72, 72, 1, B, H, W, 72, w3, None, 1, 1, 18, y18, 18, 0, None, None, 0, 3, s72, s72, None, st18, None, None, 0.0, sp())
    hn(xb)[0].backward(gb)
torch.cuda.synchronize()
