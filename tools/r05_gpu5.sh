# round 5, GPU call 5: eval-mode fused CrossFFN - parity + eval forward time (A/B by HRF_FFN_EVAL) + per-kernel times
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05_5; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ffn_eval.py tests/test_parity_wholenet.py tests/test_hrformer.py -m gpu -x -q -k "ffn_eval or eval or fullres_digest" > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
for M in t_nus_bn b_nus_bn t_stf_bn; do
for v in 1 0; do
  HRF_FFN_EVAL=$v python bench.py --model $M --no-cpu-baseline --no-neck --no-eager --no-roofline --steps 10 --warmup 3 > $O/bench_${M}_ffn$v.json 2>> $O/bench.err
  python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_${M}_ffn$v.json').read().strip().splitlines()[-1])
print('$M HRF_FFN_EVAL=$v', 'ms_per_step', d['ms_per_step'], 'fwd_ms_per_img', d.get('fwd_ms_per_img'))
PY
done; done
python - <<'PY' 2>&1 | tee -a $O/summary.txt
# isolated times of the new kernel at the model's shapes
import torch, sys, os
sys.path.insert(0, 'tests')
from hrfuser_amd import _lib
from hrfuser_amd.profiling import _graph_time
import test_ffn_eval as T
L = _lib.lib()
dev = torch.device('cuda:0')
for (B,H,W,C) in [(2,96,160,18),(2,48,80,36),(2,24,40,72),(2,12,20,144),(2,96,160,78),(2,48,80,156),(2,96,312,18)]:
    Hd=4*C
    t = lambda *s: torch.randn(*s, device=dev)
    x=t(B,H,W,C); out=torch.empty_like(x)
    a=_lib.FfnEval(); a.B,a.H,a.W,a.C,a.hidden=B,H,W,C,Hd
    bufs=[t(C),t(C),t(Hd,C)*0.3,t(Hd),t(Hd),t(Hd),t(Hd,9)*0.3,t(Hd),t(Hd),t(Hd),t(C,Hd)*0.2,t(C),t(C),t(C)]
    a.x=x.data_ptr(); a.ln_g,a.ln_b,a.ln_eps=bufs[0].data_ptr(),bufs[1].data_ptr(),1e-6
    a.w1,a.b1,a.s1,a.t1=[b.data_ptr() for b in bufs[2:6]]
    a.wd,a.bd,a.s2,a.t2=[b.data_ptr() for b in bufs[6:10]]
    a.w3,a.b3,a.s3,a.t3=[b.data_ptr() for b in bufs[10:14]]
    a.out=out.data_ptr()
    dt=_graph_time(lambda: L.hrf_ffn_eval(a, _lib.stream_ptr()))
    fl=2.0*B*H*W*C*Hd*2 + 2.0*B*H*W*Hd*9
    print(f'ffn_eval B{B} {H}x{W} C{C}: {dt*1e6:7.1f} us  {fl/dt/1e12:6.2f} TFLOP/s  {(2*B*H*W*C*4)/dt/1e9:7.1f} GB/s (x in + out)')
PY
