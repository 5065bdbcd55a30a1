"""Where the gradient all-reduces sit inside the weight-gradient phase of the CAPTURED step: GPU timestamps
(hrf_stamp) at the end of every leaf group on the main lane and around every bucket's all-reduce on the communication lane, on a
forced one-rank RCCL group (one GPU per box).   python tools/exchange_overlap.py [b_nus_bn|t_nus_bn] [rounds]"""
import json, os, sys
os.environ.setdefault('HRF_FORCE_COLLECTIVES', '1')
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29711', RANK='0', WORLD_SIZE='1')
if len(sys.argv) > 2:
    os.environ['HRF_GRAD_OVERLAP'] = sys.argv[2]
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrfuser_amd import build_backbone                          # noqa: E402
import hrfuser_amd.runtime as R                                 # noqa: E402
from hrfuser_amd.trainer import Trainer, make_cotangents        # noqa: E402
tag = sys.argv[1] if len(sys.argv) > 1 else 'b_nus_bn'
dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'backbone_cfgs.json')))[tag]
torch.manual_seed(0)
net = build_backbone(cfg).to(dev).train()
W = 1248 if 'stf' in tag else 640
x = torch.randn(2, 3, 384, W, device=dev)
mods = [torch.randn(2, c, 384, W, device=dev) for c in cfg.get('mod_in_channels', [3, 3])]
cots = make_cotangents(net, x, mods)
tr = Trainer(net, group=dist.group.WORLD, world_size=1)
st = net.__dict__['_exchange_stamps'] = R.StageStamps(dev, 64)
st0 = R.StageStamps(dev, 4)


def mark_start(ctx):
    pass


tr.step(x, mods, cots)
tr.step(x, mods, cots)
st.reset()
tr.capture(x, mods, cots, warmup=0)            # (no eager step inside: every mark below belongs to the captured step)
marks = list(st.marks)
rows = []
for _ in range(5):
    tr.replay()
    torch.cuda.synchronize()
    t = st.buf[:len(marks)].cpu().tolist()
    rows.append(t)
t = rows[-1]
t0 = min(t)
n = net._engine().flat_g.numel()
print(f'{tag}: {4 * n / 2**20:.0f} MB of gradients, {len(tr.buckets(n))} buckets, {tr.overlap_rounds(n)} leaf groups (HRF_GRAD_OVERLAP={os.environ.get("HRF_GRAD_OVERLAP", "auto")}); '
      f'{tr.collectives_per_step} collectives per step, schedule: {tr.sync_schedule}')
print('microseconds since the first stamp of the weight-gradient phase (captured step, replay 5; 100 MHz GPU counter):')
for (what, name), v in sorted(zip(marks, t), key=lambda p: p[1]):
    print(f'  {(v - t0) / 100.0:9.1f}  {what:16s} {name}')
last_leaf = max(v for (w, _), v in zip(marks, t) if w == 'leaves_done')
first_ar = min(v for (w, _), v in zip(marks, t) if w == 'allreduce_begin')
print(f'first all-reduce starts {(last_leaf - first_ar) / 100.0:.1f} us BEFORE the last weight-gradient leaf group is done' if first_ar < last_leaf
      else 'no overlap: the first all-reduce starts after the last leaf group')
tr.check()
torch.cuda.synchronize()
sys.stdout.flush()
os._exit(0)
