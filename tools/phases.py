"""Wall time of the phases of one captured training step (forward / data-gradient chain / weight-gradient phase)."""
import sys, os, copy, json, time, torch
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
from hrfuser_amd import build_backbone
from hrfuser_amd.trainer import Trainer, make_cotangents
import hrfuser_amd.runtime as R
dev=torch.device('cuda:0')
cfg=json.load(open(os.path.join(ROOT,'tests','golden','backbone_cfgs.json')))['t_nus_bn']
torch.manual_seed(0)
net=build_backbone(copy.deepcopy(cfg)).to(dev); net.train()
x=torch.randn(2,3,384,640,device=dev); mods=[torch.randn(2,3,384,640,device=dev) for _ in range(2)]
cots=make_cotangents(net,x,mods)
tr=Trainer(net)
def timeg(fn, n=20):
    s=torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
tr.step(x,mods,cots)
def fwd_only():
    ctx,outs,_=net._execute((x,)+tuple(mods), False)
def fwd_rec():
    ctx,outs,_=net._execute((x,)+tuple(mods), True)
def fwd_bwd():
    eng=net._engine(); eng.flat_g.zero_()
    ctx,outs,_=net._execute((x,)+tuple(mods), True)
    for o,c in zip(outs,cots): o.grad=c.clone()
    ctx.run_backward()
print('train fwd only      %.2f ms'%timeg(fwd_only))
os.environ['HRF_DEBUG_SKIP_WGRAD']='1'
print('fwd+bwd no wgrad    %.2f ms'%timeg(fwd_bwd))
os.environ['HRF_DEBUG_SKIP_WGRAD']='0'
print('fwd+bwd with wgrad  %.2f ms'%timeg(fwd_bwd))
print('full step           %.2f ms'%timeg(lambda: tr._step_impl(x,mods,cots)))
for lanes in ('0',):
    os.environ['HRF_LANES']=lanes
    print('HRF_LANES=0 train fwd only %.2f ms'%timeg(fwd_only))
    os.environ['HRF_DEBUG_SKIP_WGRAD']='1'
    print('HRF_LANES=0 fwd+bwd no wgrad %.2f ms'%timeg(fwd_bwd))
    os.environ['HRF_DEBUG_SKIP_WGRAD']='0'
    print('HRF_LANES=0 fwd+bwd with wgrad %.2f ms'%timeg(fwd_bwd))
os.environ['HRF_LANES']='1'
for k in ('4','16'):
    os.environ['HRF_WGRAD_LANES']=k
    print(f'WGRAD_LANES={k} fwd+bwd %.2f ms'%timeg(fwd_bwd))
