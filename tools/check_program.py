"""Replay-program validation: K recorded-program steps must reproduce K eager steps (same init, same
inputs, dropout disabled), then host / wall time of program replay vs hipGraph replay."""
import sys, os, copy, json, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrfuser_amd import build_backbone
from hrfuser_amd.trainer import Trainer, make_cotangents

dev = torch.device('cuda:0')
cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'backbone_cfgs.json')))['t_nus_bn']


def make(drop):
    torch.manual_seed(0)
    net = build_backbone(copy.deepcopy(cfg)).to(dev)
    net.train()
    if not drop:
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if hasattr(m, 'drop_path_prob'):
                m.drop_path_prob = 0.0
    return net


g = torch.Generator().manual_seed(1)
x = torch.randn(2, 3, 384, 640, generator=g).to(dev)
mods = [torch.randn(2, 3, 384, 640, generator=g).to(dev) for _ in range(2)]

# ---- numerics: 3 warm-up (eager) + 1 recorded + 3 replayed  ==  7 eager steps
na, nb = make(False), make(False)
cots = make_cotangents(na, x, mods)
ta, tb = Trainer(na), Trainer(nb)
for _ in range(6):
    ta.step(x, mods, cots)
tb.capture_program(x, mods, cots, warmup=2)        # 2 warm-up + 1 recorded = 3 steps
for _ in range(3):
    tb.replay_program()
torch.cuda.synchronize()
pa, pb = na._engine().flat_p, nb._engine().flat_p
rel = float((pa - pb).abs().max() / pa.abs().max())
rm = max(float((a.running_mean - b.running_mean).abs().max()) for a, b in zip(na.modules(), nb.modules())
         if isinstance(a, torch.nn.modules.batchnorm._BatchNorm))
# nondeterminism baseline: a second eager run (fp32 atomics reorder sums)
ne = make(False)
te = Trainer(ne)
for _ in range(6):
    te.step(x, mods, cots)
torch.cuda.synchronize()
pe = ne._engine().flat_p
print('eager vs eager rel max diff %.3e' % float((pa - pe).abs().max() / pa.abs().max()))
print('program info (launches, streams, events):', tb.prog_info)
print('params after 6 steps: eager vs program rel max diff %.3e ; running_mean max diff %.3e' % (rel, rm))
base = float((pa - pe).abs().max() / pa.abs().max())
assert rel < max(1e-4, 5 * base), (rel, base)

# ---- timing (dropout on, like bench.py)
nc = make(True)
tc = Trainer(nc)
tc.capture_program(x, mods, cots)
for _ in range(5):
    tc.replay_program()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    tc.replay_program()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('program: host %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
nd = make(True)
td = Trainer(nd)
td.capture(x, mods, cots)
for _ in range(5):
    td.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    td.replay()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('hipGraph: host %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
