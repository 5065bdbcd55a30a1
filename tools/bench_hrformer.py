"""Camera-only A/B baseline: training step of the plain HRFormer-T backbone on the same kernels, same
batch / resolution / optimizer as bench.py's HRFuser-T step (hipGraph replay).  Prints one JSON line."""
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrfuser_amd import build_backbone            # noqa: E402
from hrfuser_amd.trainer import Trainer           # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'hrformer_t_bn'
    B, H, W = 2, 384, 640
    dev = torch.device('cuda:0')
    with open(os.path.join(ROOT, 'tests', 'golden', 'hrformer_cfgs.json')) as fh:
        cfg = json.load(fh)[tag]
    cfg['drop_path_rate'] = 0.0 if 'no_dp' in sys.argv else cfg.get('drop_path_rate', 0.0)
    net = build_backbone(copy.deepcopy(cfg)).to(dev).train()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, 3, H, W, generator=g).to(dev)
    net.eval()
    with torch.no_grad():
        ys = net(x)
    net.train()
    cots = [torch.randn(tuple(y.permute(0, 2, 3, 1).shape), generator=g).to(dev) / y.numel() for y in ys]
    tr = Trainer(net)
    tr.capture(x, [], cots)
    for _ in range(5):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        tr.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 30 * 1e3
    print(json.dumps({'workload': f'{tag} camera-only backbone train step (fwd+bwd+AdamW), {B}x3x{H}x{W}, hipGraph',
                      'ms_per_step': round(ms, 3), 'images_per_sec': round(B / ms * 1e3, 2),
                      'params': sum(p.numel() for p in net.parameters())}))


if __name__ == '__main__':
    main()
