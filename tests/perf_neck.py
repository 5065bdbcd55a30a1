"""Time the HRFPN neck on cuda:0 at the BASELINE.json T/B shapes: HIP engine vs. the same math in
eager PyTorch-ROCm (the oracle module moved to the GPU = what the reference would run there).  Prints JSON lines."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ may use the oracle (eager comparison leg)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from hrfuser_amd import HRFPN          # noqa: E402


def timed(fn, n=20, warm=15):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    only = os.environ.get('NECK_ONLY', '')           # 'hip' | 'eager' | '' ; NECK_TAG = 'T' | 'B' | ''
    tags = os.environ.get('NECK_TAG', '')
    import hrfpn_oracle as N             # eager comparison leg only (not the product path)
    import hrfuser_oracle as O
    dev = torch.device('cuda:0')
    for tag, chans in (('T', [18, 36, 72, 144]), ('B', [78, 156, 312, 624]), ('T', [18, 36, 72, 144])):
        if tags and tag != tags:
            continue
        B, H, W = 2, 96, 160
        g = torch.Generator().manual_seed(0)
        xs = [torch.randn(B, c, H >> i, W >> i, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
              for i, c in enumerate(chans)]
        orc = N.HRFPNOracle(in_channels=chans, out_channels=256)
        O.seeded_fill_(orc, 1)
        net = HRFPN(in_channels=chans, out_channels=256)
        net.load_state_dict(orc.state_dict())
        net.to(dev)
        orc.to(dev)
        res = {'neck': tag, 'B': B, 'grid': [H, W]}
        for name, m in (('hip', net), ('eager', orc)):
            if only and name != only:
                continue
            m.eval()
            with torch.no_grad():
                res[f'{name}_fwd_ms'] = round(timed(lambda: m(xs)), 3)
            m.train()
            xr = [t.detach().requires_grad_(True) for t in xs]

            def step():
                ys = m(xr)
                torch.autograd.backward(list(ys), [torch.ones_like(y) for y in ys])
            res[f'{name}_fwd_bwd_ms'] = round(timed(step), 3)
        flop = 2 * B * H * W * (sum(chans) * 256 + 9 * 256 * 256 * sum(4.0 ** -i for i in range(5)))
        res['fwd_gflop'] = round(flop / 1e9, 2)
        if 'hip_fwd_ms' in res:
            res['hip_fwd_tflops'] = round(flop / res['hip_fwd_ms'] / 1e9, 1)
        print(json.dumps(res), flush=True)


if __name__ == '__main__':
    main()
