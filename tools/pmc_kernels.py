"""Launches the hot kernels of the training step eagerly (4 times each) on their branch-0 shapes so that
rocprofv3 --pmc can attribute hardware counters (FETCH_SIZE / WRITE_SIZE / SQ_*) to them:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_kernels.py

(see profiles/README.md for the collected numbers and the gfx950 corrections applied)."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hrfuser_amd import _lib
L=_lib.lib(); dev=torch.device('cuda:0')
R=lambda *sh: torch.randn(*sh,device=dev)
sp=_lib.stream_ptr
B,H,W=2,96,160
# wgrad 3x3 64->64
Cin=Cout=64
x=R(B,H,W,Cin); dy=R(B,H,W,Cout); yr=R(B,H,W,Cout); dw=torch.zeros(Cout,Cin,3,3,device=dev); sc=R(Cin); sh=R(Cin); cA=R(Cout); cB=R(Cout); cC=R(Cout)
w=R(Cout,Cin,3,3); y=R(B,H,W,Cout); st=torch.zeros(32*Cout,dtype=torch.float64,device=dev)
dx=R(B,H,W,Cin)
# attention branch 0
C=18; P=B*H*W; qkv=R(P,3*C); o=R(P,C); Tt=R(169,1); bq=R(3*C); dqkv=R(P,3*C); dT=torch.zeros(169,1,device=dev); dbq=torch.zeros(3*C,device=dev)
# dense wgrad 18->72 LN bnb ; lin fwd 18->72
x18=R(B,H,W,18); dy72=R(B,H,W,72); yr72=R(B,H,W,72); dw2=torch.zeros(72,18,device=dev); rs=R(P,2); s18=R(18); c72=[R(72) for _ in range(3)]; w2=R(72,18,1,1); y72=R(B,H,W,72); st72=torch.zeros(32*72,dtype=torch.float64,device=dev)
x72=R(B,H,W,72); dy18=R(B,H,W,18); yr18=R(B,H,W,18); dw3=torch.zeros(18,72,device=dev); s72=R(72); c18=[R(18) for _ in range(3)]
for it in range(4):
    # dominant signature of the step: fc3 weight gradient (GELU(BN(.)) operand, BatchNorm-backward on dY)
    L.hrf_conv_bwd_weight(dy18,18,0,yr18,*c18,x72,H*W*72,W*72,72,1,B,H,W,72,1,1,18,3,s72,s72,None,dw3,None,sp())
    L.hrf_conv_bwd_weight(dy,Cout,0,yr,cA,cB,cC,x,H*W*Cin,W*Cin,Cin,1,B,H,W,Cin,3,1,Cout,2,sc,sh,None,dw,None,sp())
    L.hrf_conv_fwd(x,H*W*Cin,W*Cin,Cin,1,B,H,W,Cin,w,None,3,1,Cout,y,Cout,0,None,None,0,2,sc,sh,None,st,None,None,0.0,sp())
    L.hrf_conv_bwd_data(dy,Cout,0,yr,cA,cB,cC,None,w,3,1,Cout,B,H,W,Cin,dx,H*W*Cin,W*Cin,Cin,1,0,1,x,Cin,sc,sh,1,st,sp())
    L.hrf_window_attn_fwd(qkv,3*C,0,qkv,3*C,C,qkv,3*C,2*C,bq[C:2*C],bq[2*C:],Tt,o,C,B,H,W,C,1,sp())
    L.hrf_window_attn_bwd(qkv,3*C,0,qkv,3*C,C,qkv,3*C,2*C,bq[C:2*C],bq[2*C:],Tt,o,C,dqkv,3*C,0,dqkv,3*C,C,dqkv,3*C,2*C,dbq[C:2*C],dbq[2*C:],dT,0,B,H,W,C,1,sp())
    L.hrf_conv_bwd_weight(dy72,72,0,yr72,*c72,x18,H*W*18,W*18,18,1,B,H,W,18,1,1,72,4,s18,s18,rs,dw2,None,sp())
    L.hrf_conv_fwd(x18,H*W*18,W*18,18,1,B,H,W,18,w2,None,1,1,72,y72,72,0,None,None,0,4,s18,s18,rs,st72,None,None,0.0,sp())
torch.cuda.synchronize()
