#!/bin/bash
# round-3 third GPU pass: same-box A/B of the r02 tree vs the product build with one problem per launch; remaining new tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03d
mkdir -p $O
export TMPDIR=/tmp
B="--steps 30 --warmup 5 --no-cpu-baseline --no-neck --no-eager --no-roofline"
run() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'], d['step_ms']['median'], d['fwd_ms_per_img'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
if [ -d ab_old ]; then (cd ab_old && run old1 timeout 600 python bench.py $B); fi
run new1 timeout 600 python bench.py $B
run new_lockstep env HRF_LOCKSTEP=1 timeout 600 python bench.py $B
if [ -d ab_old ]; then (cd ab_old && run old2 timeout 600 python bench.py $B); fi
run new2 timeout 600 python bench.py $B
timeout 1200 python -m pytest tests/test_module_graph.py tests/test_bench_launch.py tests/test_grouping.py -x -q -m gpu > $O/t_new.log 2>&1; echo "rc $?" >> $O/t_new.log
timeout 900 python -m pytest tests/test_neck.py -x -q -m gpu > $O/t_neck.log 2>&1; echo "rc $?" >> $O/t_neck.log
for f in t_new t_neck; do echo == $f; tail -n 6 $O/$f.log; done
