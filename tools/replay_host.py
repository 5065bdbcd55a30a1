"""Host-side cost of replaying the captured training step (hipGraphLaunch is host-bound for multi-stream graphs)."""
import sys, os, copy, json, time, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
from hrfuser_amd import build_backbone
from hrfuser_amd.trainer import Trainer, make_cotangents
dev=torch.device('cuda:0')
cfg=json.load(open(os.path.join(ROOT,'tests','golden','backbone_cfgs.json')))['t_nus_bn']
torch.manual_seed(0)
net=build_backbone(copy.deepcopy(cfg)).to(dev); net.train()
x=torch.randn(2,3,384,640,device=dev); mods=[torch.randn(2,3,384,640,device=dev) for _ in range(2)]
cots=make_cotangents(net,x,mods)
tr=Trainer(net); tr.capture(x,mods,cots)
for _ in range(5): tr.replay()
torch.cuda.synchronize()
hs=[]; ts=[]
for _ in range(10):
    torch.cuda.synchronize(); t0=time.perf_counter(); tr.replay(); t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    hs.append((t1-t0)*1e3); ts.append((t2-t0)*1e3)
print('host replay() call ms:', [round(v,2) for v in hs])
print('total incl sync ms   :', [round(v,2) for v in ts])
# back-to-back replays (pipelined)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(20): tr.replay()
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print('20 replays: host %.2f ms each, total %.2f ms each'%((t1-t0)/20*1e3,(t2-t0)/20*1e3))
