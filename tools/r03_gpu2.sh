#!/bin/bash
# round-3 second GPU pass: same-box A/B of the r02 tree, the serial schedule and the bundling variants; new tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03b
mkdir -p $O
export TMPDIR=/tmp
B="--steps 30 --warmup 5 --no-cpu-baseline --no-neck --no-eager --no-roofline"
run() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; tail -c 600 $O/$name.json | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['step_ms']['median'], d['fwd_ms_per_img'])
except Exception as e: print('$name ERR', e)"; }
if [ -d ab_old ]; then (cd ab_old && run old1 timeout 600 python bench.py $B); fi
run serial env HRF_LOCKSTEP=0 timeout 600 python bench.py $B
run group timeout 600 python bench.py $B
run nokeep env HRF_KEEP_FIRST=0 timeout 600 python bench.py $B
run st_tr env HRF_BUNDLE_WHAT=stems,trans timeout 600 python bench.py $B
run st env HRF_BUNDLE_WHAT=stems timeout 600 python bench.py $B
run stages_nokeep env HRF_BUNDLE_WHAT=stages HRF_KEEP_FIRST=0 timeout 600 python bench.py $B
run lock_nogroup env HRF_GROUP=0 timeout 600 python bench.py $B
if [ -d ab_old ]; then (cd ab_old && run old2 timeout 600 python bench.py $B); fi
timeout 1200 python -m pytest tests/test_module_graph.py tests/test_bench_launch.py -x -q -m gpu > $O/t_new.log 2>&1; echo "rc $?" >> $O/t_new.log
timeout 900 python -m pytest tests/test_parity_wholenet.py -x -q -m gpu -k "norm_eval or pre_neck" > $O/t_ne.log 2>&1; echo "rc $?" >> $O/t_ne.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
timeout 1500 python -m pytest tests/test_syncbn_gpu.py tests/test_neck.py tests/test_stochastic.py -x -q -m gpu > $O/t_sync.log 2>&1; echo "rc $?" >> $O/t_sync.log
for f in t_new t_ne t_sync; do echo == $f; tail -n 4 $O/$f.log; done
