"""Launches the neck's dominant kernel (conv3w_kernel<4>: 3x3 256->256 at 2x96x160) 4 times for rocprofv3 --pmc passes
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, FETCH_SIZE,
WRITE_SIZE - separate passes, tools/prof_pmc_conv3w.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrfuser_amd import _lib

L = _lib.lib()
s = _lib.stream_ptr()
B, H, W, C = 2, 96, 160, 256
x = torch.randn(B, H, W, C, device='cuda')
w = torch.randn(C, C, 3, 3, device='cuda') * 0.02
wp = torch.empty(9 * C * C, device='cuda')
y = torch.empty(B, H, W, C, device='cuda')
L.hrf_conv3_pack(w, C, C, 0, wp, s)
for _ in range(4):
    L.hrf_conv3_packed(x, C, wp, None, y, C, 0, B, H, W, C, C, s)
torch.cuda.synchronize()
