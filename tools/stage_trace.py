"""One captured HRFuser training step with stage stamps, replayed a few times - to be run under
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o st -- python3 tools/stage_trace.py [model]
tools/stage_trace_report.py then cuts the kernel trace of the last replay at the stamp kernels (one per stage boundary) and
prints, per stage, the kernels per hardware queue with their start offsets: the critical chain of a stage is readable from it."""
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
stf = tag.startswith('t_stf')
H, W = 384, (1248 if stf else 640)
mc = cfg.get('mod_in_channels', [3, 3])
g = torch.Generator().manual_seed(100)
x = torch.randn(2, 3, H, W, generator=g).to(dev)
mods = [torch.randn(2, c, H, W, generator=g).to(dev) for c in mc]
cots = make_cotangents(net, x, mods)
tr = Trainer(net)
st = net.enable_stage_stamps()
tr.capture(x, mods, cots)
for _ in range(6):
    tr.replay()
torch.cuda.synchronize()
print(json.dumps([(d, n, round(t, 1)) for d, n, t in st.read()]))
