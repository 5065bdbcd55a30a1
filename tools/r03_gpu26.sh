#!/bin/bash
# does the CAPTURE ORDER of the nodes decide when a lane starts?  HRF_LOCKSTEP=1 runs sibling bodies as coroutines: their launches are
# captured round-robin instead of lane by lane
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
O=$PWD/gpurun_out/r03x
rm -rf $O; mkdir -p $O
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
run lockstep env HRF_LOCKSTEP=1 timeout 600 python bench.py $B
run base2 timeout 600 python bench.py $B
run lockstep2 env HRF_LOCKSTEP=1 timeout 600 python bench.py $B
mkdir -p $O/trace
(cd /tmp && env HRF_LOCKSTEP=1 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o st -- python3 $GRAFT_REPO_ROOT/tools/stage_trace.py > $O/stage_trace.out 2>&1)
python tools/stage_trace_report.py $(find $O/trace -name 'st_kernel_trace.csv' | head -1) "fwd stage3" > $O/stage_lockstep.txt 2>&1
rm -rf $O/trace
grep -E "^fwd|^bwd|^step" $O/stage_lockstep.txt | head -12
awk '/^fwd stage3/{f=1} f{print} /^fwd transitions_c/{exit}' $O/stage_lockstep.txt | sed -n 2,40p | cut -c1-90
