"""ONE eager HRFuser training step (no hipGraph, HRF_LANES=0: every launch on one stream, dispatch order = program order) with the
stage stamps on, so that a rocprofv3 --pmc pass can attribute hardware counters to the STAGES of the backbone:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o s -- python3 tools/pmc_stages.py [model]
    python tools/pmc_stages_report.py <fetch counter csv> <write counter csv> out.json

The stamp kernels (one per stage boundary) cut the dispatch sequence; tools/pmc_stages_report.py sums the counters between them.
(A whole step under --pmc crashed the profiler in round 1 with the multi-stream schedule: profiles/r01_pmc_step_crash.log.)"""
import copy
import json
import os
import sys

os.environ.setdefault('HRF_LANES', '0')
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
tr.step(x, mods, cots)                       # warm-up (allocator, random pools) without stamps
torch.cuda.synchronize()
st = net.enable_stage_stamps()
tr.step(x, mods, cots)                       # the measured step: the report uses the dispatches after the FIRST stamp kernel
torch.cuda.synchronize()
print(json.dumps([(d, n, round(t, 1)) for d, n, t in st.read()]))
