"""Phase timestamps (wall_clock64, 100 MHz) of one workgroup of attn_block_bwd_kernel<18,1> at the s4 grid.
Needs a library built with the stamps compiled in:  HRF_EXTRA_FLAGS=-DHRF_AB_TIMING python -m hrfuser_amd.build_ext --force"""
import os, sys, torch, ctypes
os.environ.setdefault('HRF_LANES', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hrfuser_amd.backbone as B
import hrfuser_amd.runtime as R
from hrfuser_amd import _lib
if os.environ.get('HRF_TIMING_LIB'):
    _lib.LIB_PATH = os.environ['HRF_TIMING_LIB']
from hrfuser_amd.testing import BlockHarness
NORM = dict(type='BN', requires_grad=True, momentum=0.1); LN = dict(type='LN', eps=1e-6)
dev = torch.device('cuda:0')
C, h, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 18, int(sys.argv[2]) if len(sys.argv) > 2 else 1, 96, 160
if C == 36: H, W = 48, 80
blk = B.HRFormerBlock(C, C, h, norm_cfg=NORM, transformer_norm_cfg=LN)
hn = BlockHarness(blk, lambda ctx, b, x: b.run(ctx, x[0])).to(dev); hn.train()
x = torch.randn(2, C, H, W, device=dev, requires_grad=True); g = torch.randn(2, C, H, W, device=dev)
ts = torch.zeros(16, dtype=torch.int64, device=dev)
L = _lib.lib()
orig = L._fns['hrf_attn_block_bwd']
def patched(a, s):
    a.out_rowstat = ts.data_ptr()
    return orig(a, s)
L._fns['hrf_attn_block_bwd'] = patched
for it in range(5):
    y = hn(x)[0]; y.backward(g)
torch.cuda.synchronize()
t = ts.cpu().tolist()
names = ['start', 'zeroed', 'loads issued+staged', 'xhat/dy1 staged', 'ffn bwd (dn2, LN2, dW1)', 'dy+dO', 'projections', 'attention', 'dn+LN1+outputs', 'weight grads', 'end']
print('tick = 10 ns')
for k in range(1, 11):
    print(f'{names[k]:32s} {(t[k] - t[k-1]) * 10 / 1000:.2f} us')
print('total', (t[10] - t[0]) * 10 / 1000, 'us')
