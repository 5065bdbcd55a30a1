"""One representative GROUPED weight-gradient launch of the dominant kernel variant of the training step
(wgrad_dense_kernel<4, 4, true, 1, true>: 3x3 convolutions whose input is ReLU(BN(.)) and whose output feeds a
BatchNorm), issued 4 times so that rocprofv3 --pmc can attribute FETCH_SIZE / WRITE_SIZE to it:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_grouped.py

The group = what one lane of the deferred weight-gradient phase queues for this variant in HRFuser-T (bench.py reports
3.75 problems per launch): stem conv2 (64->64 stride 2 from 192x320), the two layer1 Bottleneck 3x3 convolutions
(64->64 at 96x160) and one new-branch transition (36->72 stride 2 from 48x80)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrfuser_amd import _lib

L = _lib.lib()
dev = torch.device('cuda:0')
R = lambda *sh: torch.randn(*sh, device=dev)
B = 2
PROBLEMS = [(192, 320, 64, 64, 2), (96, 160, 64, 64, 1), (96, 160, 64, 64, 1), (48, 80, 36, 72, 2)]   # H, W, Cin, Cout, stride
bufs = []
alg = 0
for (H, W, Cin, Cout, s) in PROBLEMS:
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    bufs.append(dict(x=R(B, H, W, Cin), dy=R(B, Ho, Wo, Cout), yr=R(B, Ho, Wo, Cout), c=[R(Cout) for _ in range(3)],
                     sc=R(Cin), sh=R(Cin), dw=torch.zeros(Cout, Cin, 3, 3, device=dev)))
    alg += 4 * (B * H * W * Cin + 2 * B * Ho * Wo * Cout + 9 * Cin * Cout)
print('algorithmic bytes of the launch', alg)
for it in range(4):
    L.hrf_wgrad_group_begin()
    for (H, W, Cin, Cout, s), b in zip(PROBLEMS, bufs):
        L.hrf_conv_bwd_weight(b['dy'], Cout, 0, b['yr'], *b['c'], b['x'], H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin, 3, s,
                              Cout, 2, b['sc'], b['sh'], None, b['dw'], None, _lib.stream_ptr())
    L.hrf_wgrad_group_end(_lib.stream_ptr())
torch.cuda.synchronize()
