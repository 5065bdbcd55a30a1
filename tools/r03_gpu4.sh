#!/bin/bash
# round-3 GPU pass 4: LDS-tiled lin2 engine - kernel parity, HRFuser-B / T A-B against the register-only row GEMM
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03f
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "lin2" > $O/t_lin2.log 2>&1; echo "rc $?" >> $O/t_lin2.log
tail -n 3 $O/t_lin2.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-neck --no-eager"
run() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'], d['step_ms']['median'], d['fwd_ms_per_img'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run b_lin2 timeout 900 python bench.py $B --model b_nus_bn --dump-kernels $O/kern_b_lin2.json
run b_old env HRF_KNOBS=28=2 timeout 900 python bench.py $B --model b_nus_bn --no-roofline
run t_lin2 timeout 600 python bench.py $B --no-roofline
run t_old env HRF_KNOBS=28=2 timeout 600 python bench.py $B --no-roofline
run t_lin2b timeout 600 python bench.py $B --no-roofline
mkdir -p $O/trace; rocprofv3 --kernel-trace --output-format csv -d $O/trace -o st -- python3 tools/stage_trace.py > $O/stage_trace.out 2>&1; python tools/stage_trace_report.py $(ls $O/trace/*/st_kernel_trace.csv | head -1) "fwd stage3" "bwd stage3" > $O/stage_report.txt 2>&1; rm -rf $O/trace
timeout 1500 python -m pytest tests/test_module_graph.py tests/test_neck.py tests/test_parity_blocks.py -x -q -m gpu -k "module_graph or backbone_into or wide" > $O/t_misc.log 2>&1; echo "rc $?" >> $O/t_misc.log
timeout 1200 python -m pytest tests/test_parity_wholenet.py -x -q -m gpu -k "b_nus and (small or medium)" > $O/t_b.log 2>&1; echo "rc $?" >> $O/t_b.log
for f in t_misc t_b; do echo == $f; tail -n 5 $O/$f.log; done
