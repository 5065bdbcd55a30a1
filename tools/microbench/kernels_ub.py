import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hrfuser_amd import _lib
from hrfuser_amd.profiling import _graph_time
L=_lib.lib(); dev=torch.device('cuda:0')
R=lambda *sh: torch.randn(*sh,device=dev)
def T(name, fn): print(f'{name:72s} {_graph_time(fn)*1e6:8.1f} us', flush=True)
sp=_lib.stream_ptr
shapes=[(2,96,160,18,18,0,False),(2,96,160,18,54,4,False),(2,96,160,18,72,4,True),(2,96,160,72,18,3,True),(2,48,80,36,144,4,True),(2,48,80,144,36,3,True),(2,24,40,72,288,4,True),(2,12,20,144,576,4,True),(2,12,20,576,144,3,True),(2,96,160,64,256,2,True),(2,96,160,256,64,2,True)]
for (B,H,W,Cin,Cout,tf,bnb) in shapes:
    x=R(B,H,W,Cin); dy=R(B,H,W,Cout); yr=R(B,H,W,Cout); dw=torch.zeros(Cout,Cin,device=dev); db=torch.zeros(Cout,device=dev)
    sc=R(Cin); sh=R(Cin); rs=R(B*H*W,2); cA=R(Cout); cB=R(Cout); cC=R(Cout)
    co=(cA,cB,cC) if bnb else (None,None,None)
    for nobias in (0,):
        fn=lambda: L.hrf_conv_bwd_weight(dy,Cout,0,yr if bnb else None,*co,x,H*W*Cin,W*Cin,Cin,1,B,H,W,Cin,1,1,Cout,tf,sc if tf else None,sh if tf else None,rs if tf==4 else None,dw,None if nobias else db,sp())
        for dbg in (0,):
            T(f'wgrad dbg={dbg} nobias={nobias} {B}x{H}x{W} {Cin}->{Cout} tf{tf} bnb{int(bnb)}', fn)
L.hrf_debug_knob(5,0)
# lin fwd / bwd_data: new (knob4=0) vs old (knob4=1)
def conv(B,H,W,Cin,Cout,tf,stats,res):
    x=R(B,H,W,Cin); w=R(Cout,Cin,1,1); b=R(Cout); y=R(B,H,W,Cout); st=torch.zeros(32*Cout,dtype=torch.float64,device=dev) if stats else None
    sc=R(Cin); sh=R(Cin); rs=R(B*H*W,2); rr=R(B,H,W,Cout) if res else None
    return lambda: L.hrf_conv_fwd(x,H*W*Cin,W*Cin,Cin,1,B,H,W,Cin,w,b,1,1,Cout,y,Cout,0,rr,None,Cout,tf,sc if tf else None,sh if tf else None,rs if tf==4 else None,st,None,None,0.0,sp())
def bwdd(B,H,W,Cin,Cout,bnb,epi):
    dy=R(B,H,W,Cout); yr=R(B,H,W,Cout); w=R(Cout,Cin,1,1); dx=R(B,H,W,Cin); xr=R(B,H,W,Cin); sc=R(Cin); sh=R(Cin)
    cA=R(Cout); cB=R(Cout); cC=R(Cout); st=torch.zeros(32*Cin,dtype=torch.float64,device=dev)
    co=(cA,cB,cC) if bnb else (None,None,None)
    if epi: return lambda: L.hrf_conv_bwd_data(dy,Cout,0,yr if bnb else None,*co,w,1,1,Cout,B,H,W,Cin,dx,H*W*Cin,W*Cin,Cin,1,0,1,xr,Cin,sc,sh,2,st,sp())
    return lambda: L.hrf_conv_bwd_data(dy,Cout,0,yr if bnb else None,*co,w,1,1,Cout,B,H,W,Cin,dx,H*W*Cin,W*Cin,Cin,1,1,0,None,0,None,None,0,None,sp())
for (B,H,W,Cin,Cout,tf) in [(2,96,160,18,54,4),(2,96,160,18,18,0),(2,96,160,18,72,4),(2,96,160,72,18,3),(2,48,80,36,144,4),(2,48,80,144,36,3),(2,24,40,288,72,3),(2,12,20,144,576,4),(2,12,20,576,144,3),(2,96,160,64,256,2),(2,96,160,256,64,2)]:
    for k4 in (1,0):
        L.hrf_debug_knob(4,k4)
        tag='LIN' if k4==0 else 'OLD'
        T(f'{tag} fwd {B}x{H}x{W} {Cin}->{Cout} tf{tf} stats res', conv(B,H,W,Cin,Cout,tf,True,True))
        T(f'{tag} fwd {B}x{H}x{W} {Cin}->{Cout} tf{tf} plain', conv(B,H,W,Cin,Cout,tf,False,False))
        T(f'{tag} bwd_data {B}x{H}x{W} {Cin}<-{Cout} bnb epi1', bwdd(B,H,W,Cin,Cout,True,True))
        T(f'{tag} bwd_data {B}x{H}x{W} {Cin}<-{Cout} plain acc', bwdd(B,H,W,Cin,Cout,False,False))
L.hrf_debug_knob(4,0)
