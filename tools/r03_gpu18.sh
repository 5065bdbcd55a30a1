#!/bin/bash
# deep-batch variants of the register row GEMM (few rows, long contraction): parity + A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "conv" > $O/t_conv.log 2>&1; echo "rc $?" >> $O/t_conv.log; tail -n 3 $O/t_conv.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-neck --no-eager"
run() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'], d['step_ms']['median'], d['fwd_ms_per_img'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run b_deep timeout 900 python bench.py $B --model b_nus_bn --dump-kernels $O/kern_b.json
run b_off env HRF_KNOBS=20=0 timeout 900 python bench.py $B --model b_nus_bn --no-roofline
run t_deep timeout 600 python bench.py $B --no-roofline --steps 30
run t_off env HRF_KNOBS=20=0 timeout 600 python bench.py $B --no-roofline --steps 30
run t_deep2 timeout 600 python bench.py $B --no-roofline --steps 30
run stf_deep timeout 600 python bench.py $B --no-roofline --model t_stf_bn
run stf_off env HRF_KNOBS=20=0 timeout 600 python bench.py $B --no-roofline --model t_stf_bn
