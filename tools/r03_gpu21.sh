#!/bin/bash
# stream priorities: the coarse branches of an HRModule (the thin, critical chains) on high-priority streams (HRF_THIN_PRIO=k: branches >= k)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03s
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
run prio1 env HRF_THIN_PRIO=1 timeout 600 python bench.py $B
run prio2 env HRF_THIN_PRIO=2 timeout 600 python bench.py $B
run base2 timeout 600 python bench.py $B
run prio1b env HRF_THIN_PRIO=1 timeout 600 python bench.py $B
run prio2b env HRF_THIN_PRIO=2 timeout 600 python bench.py $B
run stf_base timeout 600 python bench.py $B --model t_stf_bn
run stf_prio1 env HRF_THIN_PRIO=1 timeout 600 python bench.py $B --model t_stf_bn
run b_base timeout 600 python bench.py $B --model b_nus_bn --steps 20
run b_prio1 env HRF_THIN_PRIO=1 timeout 600 python bench.py $B --model b_nus_bn --steps 20
