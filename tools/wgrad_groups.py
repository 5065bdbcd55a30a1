"""Per-variant table of the grouped weight-gradient launches of one training step (in-situ HIP events of the library:
hrf_wgrad_group_report), priced against the fp32 MFMA peak.   python tools/wgrad_groups.py [t_nus_bn|b_nus_bn|t_stf_bn]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrfuser_amd import build_backbone, profiling               # noqa: E402
from hrfuser_amd.trainer import Trainer, make_cotangents        # noqa: E402
tag = sys.argv[1] if len(sys.argv) > 1 else 't_nus_bn'
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'backbone_cfgs.json')))[tag]
torch.manual_seed(0)
net = build_backbone(cfg).to(dev).train()
W = 1248 if 'stf' in tag else 640
x = torch.randn(2, 3, 384, W, device=dev)
mods = [torch.randn(2, c, 384, W, device=dev) for c in cfg.get('mod_in_channels', [3, 3])]
cots = make_cotangents(net, x, mods)
tr = Trainer(net)
for _ in range(2):
    tr.step(x, mods, cots)
rows = profiling.grouped_wgrad_report(tr, x, mods, cots, steps=3)
tot = sum(r['time_per_step_ms'] for r in rows)
print(f'{tag}: {sum(r["launches_per_step"] for r in rows):.0f} grouped launches per step, {tot:.3f} ms (in situ)')
print(f'{"kernel":46s} {"n":>4s} {"prob":>5s} {"us":>7s} {"ms/step":>8s} {"GF":>7s} {"MB":>7s} {"TF/s":>6s} {"GB/s":>6s}  heaviest')
for r in rows:
    us = r['avg_launch_us']
    print(f'{r["kernel"]:46s} {r["launches_per_step"]:4.0f} {r["problems_per_launch"]:5.1f} {us:7.1f} {r["time_per_step_ms"]:8.3f} '
          f'{r["flops_per_launch"] / 1e9:7.2f} {r["bytes_per_launch"] / 1e6:7.1f} {r["flops_per_launch"] / us / 1e6:6.1f} '
          f'{r["bytes_per_launch"] / us / 1e3:6.0f}  {r["heaviest_problem"]}')
