#!/bin/bash
# run-to-run spread of the default step on ONE box, and whether the stream -> hardware-queue assignment (shifted by K throwaway
# streams created before the lanes) explains it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03t
mkdir -p $O
export TMPDIR=/tmp
B="--steps 30 --warmup 5 --no-cpu-baseline --no-neck --no-eager --no-roofline"
run() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'], d['step_ms']['median'], d['step_ms']['p10'], d['step_ms']['p90'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
for r in 1 2 3 4 5 6; do run base_$r timeout 600 python bench.py $B; done
for k in 1 2 3; do for r in 1 2 3; do run skew${k}_$r env HRF_STREAM_SKEW=$k timeout 600 python bench.py $B; done; done
