#!/bin/bash
# lane-plan A/B: at most 4 concurrently active HIP streams (= hardware queues)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03h
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
run kf env HRF_KEEP_FIRST=1 timeout 600 python bench.py $B
run ml1 env HRF_MOD_LANES=1 timeout 600 python bench.py $B
run kf_ml1 env HRF_KEEP_FIRST=1 HRF_MOD_LANES=1 timeout 600 python bench.py $B
run kf_ml1_cam env HRF_KEEP_FIRST=1 HRF_MOD_LANES=1 HRF_CAM_FIRST=1 timeout 600 python bench.py $B
run cam env HRF_CAM_FIRST=1 timeout 600 python bench.py $B
run max3 env HRF_MAX_LANES=3 timeout 600 python bench.py $B
run kf_max3 env HRF_KEEP_FIRST=1 HRF_MAX_LANES=3 timeout 600 python bench.py $B
run base2 timeout 600 python bench.py $B
