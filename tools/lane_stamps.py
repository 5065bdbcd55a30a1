"""Ground truth about lane starts WITHOUT a profiler: one captured training step with GPU timestamps (hrf_stamp, 100 MHz) at
every fork / join of the lanes (HipModule.enable_lane_stamps), replayed a few times; prints for every fork of the last
replay when each sibling lane really started and finished relative to the fork point.
    python tools/lane_stamps.py [model] > gpurun_out/lane_stamps.txt"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from hrfuser_amd import build_backbone  # noqa: E402
from hrfuser_amd.trainer import Trainer, make_cotangents  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 't_nus_bn'
with open(os.path.join(ROOT, 'tests', 'golden', 'backbone_cfgs.json')) as fh:
    cfg = json.load(fh)[tag]
dev = torch.device('cuda:0')
torch.manual_seed(1234)
net = build_backbone(copy.deepcopy(cfg)).to(dev)
net.train()
H, W = 384, (1248 if tag.startswith('t_stf') else 640)
mc = cfg.get('mod_in_channels', [3, 3])
g = torch.Generator().manual_seed(100)
x = torch.randn(2, 3, H, W, generator=g).to(dev)
mods = [torch.randn(2, c, H, W, generator=g).to(dev) for c in mc]
cots = make_cotangents(net, x, mods)
tr = Trainer(net)
st = net.enable_lane_stamps()
tr.capture(x, mods, cots)
for _ in range(8):
    tr.replay()
torch.cuda.synchronize()
t = st.buf[:len(st.marks)].cpu().tolist()
t0 = min(t)
forks, joins = {}, {}
for (what, (k, lane, stage)), v in zip(st.marks, t):
    (forks if what == 'fork' else joins).setdefault(k, {})[lane] = ((v - t0) / 100.0, stage)
print(f'{len(st.marks)} stamps, step span {(max(t) - t0) / 100.0:.1f} us')
for k in sorted(forks):
    f = forks[k]
    p, stage = f['parent']
    lanes = sorted(i for i in f if i != 'parent')
    j = joins.get(k, {})
    jp = j.get('parent', (None,))[0]
    row = '  '.join(f'L{i} +{f[i][0] - p:.0f}..{(j[i][0] - p) if i in j else float("nan"):.0f}' for i in lanes)
    print(f'fork {k:3d} after {stage:16s} at {p:8.1f} us  parent joins at +{(jp - p) if jp is not None else float("nan"):.0f}:  {row}')
