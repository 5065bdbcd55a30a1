"""Where does a conv3w step go?  Times the 256->256 3x3 at 2x96x160 with parts of the step disabled (results are
wrong on purpose; knob 25): 0 full, 1 no MFMA, 2 no fragment LDS reads, 4 no weight refill, combinations."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hrfuser_amd import _lib
L = _lib.lib(); s = _lib.stream_ptr()
B, H, W, C = 2, 96, 160, 256
x = torch.randn(B, H, W, C, device='cuda'); w = torch.randn(C, C, 3, 3, device='cuda') * 0.01
wp = torch.empty(9 * C * C, device='cuda'); y = torch.empty(B, H, W, C, device='cuda')
L.hrf_conv3_pack(w, C, C, 0, wp, s)
for dbg in (0, 1, 2, 4, 3, 6, 7, 0):
    L.hrf_debug_knob(25, dbg)
    for _ in range(3):
        L.hrf_conv3_packed(x, C, wp, None, y, C, 0, B, H, W, C, C, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        L.hrf_conv3_packed(x, C, wp, None, y, C, 0, B, H, W, C, C, s)
    torch.cuda.synchronize()
    print('dbg', dbg, 'us/launch %.1f' % ((time.perf_counter() - t0) / 20 * 1e6), flush=True)
