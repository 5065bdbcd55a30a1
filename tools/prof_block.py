"""Isolated timing of ONE HRFormerBlock (fused attention block + CrossFFN) per width at the HRFuser-T grid sizes:
run under `rocprofv3 --kernel-trace --stats` (serial stream, no lanes) to read per-kernel durations."""
import os, sys, time, torch
os.environ.setdefault('HRF_LANES', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hrfuser_amd.backbone as B
import hrfuser_amd.runtime as R
from hrfuser_amd.testing import BlockHarness
NORM = dict(type='BN', requires_grad=True, momentum=0.1)
LN = dict(type='LN', eps=1e-6)
dev = torch.device('cuda:0')
cases = [(18, 1, 96, 160), (36, 2, 48, 80), (72, 4, 24, 40), (144, 8, 12, 20)]
iters = int(os.environ.get('ITERS', '30'))
for C, h, H, W in cases:
    blk = B.HRFormerBlock(C, C, h, norm_cfg=NORM, transformer_norm_cfg=LN)
    with torch.no_grad():
        for p in blk.parameters():
            p.normal_(0, 0.1)
    hn = BlockHarness(blk, lambda ctx, b, x: b.run(ctx, x[0])).to(dev)
    hn.train()
    x = torch.randn(2, C, H, W, device=dev, requires_grad=True)
    g = torch.randn(2, C, H, W, device=dev)
    for it in range(iters + 3):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        y = hn(x)[0]
        y.backward(g)
    torch.cuda.synchronize()
    print(f'C={C} heads={h} {H}x{W}: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms per fwd+bwd (eager, host-paced)')
