#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03l
mkdir -p $O
export TMPDIR=/tmp
HRF_TIMING_LIB=$PWD/scratch/libhrf_l2timing.so timeout 500 python tools/time_lin2_phases.py > $O/l2phases.txt 2>&1
cat $O/l2phases.txt | cut -c1-400
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
run b_new timeout 900 python bench.py $B --model b_nus_bn --dump-kernels $O/kern_b.json
run t_new timeout 600 python bench.py $B --no-roofline
run t_old env HRF_KNOBS=28=2 timeout 600 python bench.py $B --no-roofline
run t_new2 timeout 600 python bench.py $B --no-roofline
