#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03m
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels.py -x -q -m gpu -k "pointwise" > $O/t_pw.log 2>&1; echo "rc $?" >> $O/t_pw.log
tail -n 3 $O/t_pw.log
B="--steps 20 --warmup 5 --no-cpu-baseline --no-neck --no-eager"
run() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'], d['step_ms']['median'], d['fwd_ms_per_img'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run b_new timeout 900 python bench.py $B --model b_nus_bn --dump-kernels $O/kern_b.json
run t_new timeout 600 python bench.py $B --dump-kernels $O/kern_t.json
run stf_new timeout 600 python bench.py $B --model t_stf_bn --no-roofline
timeout 1200 python -m pytest tests/test_parity_wholenet.py -x -q -m gpu -k "small" > $O/t_small.log 2>&1; echo "rc $?" >> $O/t_small.log
tail -n 3 $O/t_small.log
