#!/bin/bash
# weight-gradient leaves: ONE (or few) late flush onto low-priority lanes instead of the phase at the end / a flush per stage
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03r
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
run base timeout 600 python bench.py $B
for n in 60 100 150 200 260; do run flush$n env HRF_WGRAD=flush HRF_WGRAD_FLUSH=$n timeout 600 python bench.py $B; done
run base2 timeout 600 python bench.py $B
run flush150_prio0 env HRF_WGRAD=flush HRF_WGRAD_FLUSH=150 HRF_SIDE_PRIORITY=0 timeout 600 python bench.py $B
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "rc $?" >> $O/smoke.log; tail -n 3 $O/smoke.log
timeout 900 python -m pytest tests/test_groupnorm.py -x -q -m gpu > $O/t_gn.log 2>&1; echo "rc $?" >> $O/t_gn.log; tail -n 3 $O/t_gn.log
