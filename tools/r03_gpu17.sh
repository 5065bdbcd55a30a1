#!/bin/bash
# lane plan A/B: the camera stage's coarse branches share one stream while modality stages run beside it (HRF_CAM_LANES)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03p
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
run t_base timeout 600 python bench.py $B
run t_cam2 env HRF_CAM_LANES=2 timeout 600 python bench.py $B
run t_cam1 env HRF_CAM_LANES=1 timeout 600 python bench.py $B
run t_base2 timeout 600 python bench.py $B
run t_cam2b env HRF_CAM_LANES=2 timeout 600 python bench.py $B
run stf_base timeout 600 python bench.py $B --model t_stf_bn
run stf_cam2 env HRF_CAM_LANES=2 timeout 600 python bench.py $B --model t_stf_bn
run stf_cam1 env HRF_CAM_LANES=1 timeout 600 python bench.py $B --model t_stf_bn
run b_base timeout 600 python bench.py $B --model b_nus_bn --steps 20
run b_cam2 env HRF_CAM_LANES=2 timeout 600 python bench.py $B --model b_nus_bn --steps 20
HRF_CAM_LANES=2 timeout 900 python -m pytest tests/test_parity_wholenet.py -x -q -m gpu -k "train_small and t_nus" > $O/t_par.log 2>&1; echo "rc $?" >> $O/t_par.log; tail -n 3 $O/t_par.log
