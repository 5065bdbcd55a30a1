"""SURVEY 8e on ONE GPU: the SyncBN configuration (norm_cfg SyncBN, configs[2]) with a forced one-rank RCCL group -
the batched exchanges (runtime.Ctx.parallel / flush_sync / flush_bwd) and the gradient-bucket all-reduces run through
real RCCL collectives, eagerly AND inside the captured hipGraph, and are checked against the fp64 oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import copy, os, sys, torch
ROOT = %r
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2] if len(sys.argv) > 2 else '29641', RANK='0', WORLD_SIZE='1', HRF_FORCE_COLLECTIVES='1')
import torch.distributed as dist
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)                 # 'nccl' is RCCL on ROCm
import hrfuser_oracle as O
from helpers import PinnedReLU, build_pair, enable_relu_probe, relu_masks, rel_l2, relmax, tight_grad_gate, use_backend
from hrfuser_amd.trainer import Trainer
use_backend('hip')
TAG = sys.argv[1] if len(sys.argv) > 1 else 't_nus'
net, orc, cfg = build_pair(TAG, dev)                            # norm_cfg type SyncBN
assert cfg['norm_cfg']['type'] == 'SyncBN'
net.train(); orc.train()
enable_relu_probe(net)
B, H, W = 2, 64, 96
x, mods = O.seeded_inputs(B, H, W, cfg.get('mod_in_channels', [3, 3]), seed=1)
xd, md = x.to(dev), [m.to(dev) for m in mods]
g = torch.Generator().manual_seed(5)
cots = [torch.randn((B, H // 4 >> i, W // 4 >> i, c), generator=g) for i, c in enumerate(cfg['extra']['stage4']['num_channels'])]
cd = [c.to(dev) for c in cots]
tr = Trainer(net, lr=0.0, weight_decay=0.0, group=dist.group.WORLD, world_size=1)    # lr 0: every step sees the same weights
assert tr.force
outs = tr.step(xd, md, cd)
ncoll = tr.collectives_per_step
eng = net._engine()
g_eager = eng.flat_g.clone()
y_eager = [o.t.clone() for o in outs]
masks = relu_masks(net)
# ---- oracle (fp64 / fp32, ReLU decisions pinned to the product's)
refs = []
for dt in (torch.float64, torch.float32):
    o = copy.deepcopy(orc).to(dt).train()
    with PinnedReLU(masks):
        ys = o(x.to(dt), [m.to(dt) for m in mods])
    sum((y.permute(0, 2, 3, 1) * c.to(dt)).sum() for y, c in zip(ys, cots)).backward()
    refs.append((o, ys))
o64, ys = refs[0]
for p, q in zip(y_eager, ys):
    assert relmax(p, q.permute(0, 2, 3, 1).detach()) < 1e-3
tight_grad_gate(net.named_parameters(), o64.named_parameters(), refs[1][0].named_parameters(), 1e-3, TAG + ' SyncBN over 1-rank RCCL, eager')
# ---- the same step captured into ONE hipGraph (collectives inside) and replayed
net.__dict__['_relu_probe'] = False
tr.capture(xd, md, cd)
for _ in range(3):
    tr.replay()
torch.cuda.synchronize()
assert rel_l2(eng.flat_g, g_eager) < 1e-5, rel_l2(eng.flat_g, g_eager)
for p, q in zip(tr._graph_outs, y_eager):
    assert relmax(p.t, q) < 1e-5
lane_comms = os.environ.get('HRF_SYNC_LANE_COMMS', '0') == '1'   # experiment: unbatched exchanges on per-lane communicators
assert 0 < ncoll <= (800 if lane_comms else {'t_nus': 240, 'b_nus': 300, 't_stf': 300}[TAG]), ncoll   # 330-374 BatchNorms x 2 directions, batched + gradient buckets
print('SYNCBN_GPU_OK', TAG, 'collectives_per_step', ncoll, tr.sync_schedule)
sys.stdout.flush()
dist.barrier()
os._exit(0)
'''


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_syncbn_forced_rccl_eager_and_graph_gpu(tag):
    """t_nus; b_nus (BatchNorms wider than HRF_FIN_MAXC take the stand-alone packed finalize forms); t_stf (M = 3: the widest
    exchange batches)."""
    port = {'t_nus': '29641', 'b_nus': '29643', 't_stf': '29645'}[tag]
    r = subprocess.run([sys.executable, '-c', WORKER % ROOT, tag, port], capture_output=True, text=True, timeout=1500)
    sys.stdout.write(r.stdout[-3000:])
    assert 'SYNCBN_GPU_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.gpu
def test_syncbn_forced_rccl_flush_mode_gpu():
    """HRF_WGRAD=flush (weight-gradient leaves issued early on side lanes) with a gradient exchange: the folds and bucket
    all-reduces of the exchange rounds must wait for the side lanes (they read the arena while flushed
    leaves were still writing it) - same oracle gate, leaves flushed every 8."""
    env = dict(os.environ, HRF_WGRAD='flush', HRF_WGRAD_FLUSH='8', HRF_GRAD_OVERLAP='4')
    r = subprocess.run([sys.executable, '-c', WORKER % ROOT, 't_nus', '29647'], capture_output=True, text=True, timeout=1500, env=env)
    sys.stdout.write(r.stdout[-3000:])
    assert 'SYNCBN_GPU_OK' in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
