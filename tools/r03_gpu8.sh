#!/bin/bash
# round-3 re-entry pass: bench lines of HEAD (T / B / STF) with per-kernel tables, then the full GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03i
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python bench.py --dump-kernels $O/kern_t.json > $O/bench_t.json 2> $O/bench_t.err
timeout 900 python bench.py --model b_nus_bn --no-cpu-baseline --no-neck --no-eager --steps 20 --warmup 5 --dump-kernels $O/kern_b.json > $O/bench_b.json 2> $O/bench_b.err
timeout 900 python bench.py --model t_stf_bn --no-cpu-baseline --no-neck --no-eager --steps 30 --warmup 5 --dump-kernels $O/kern_stf.json > $O/bench_stf.json 2> $O/bench_stf.err
for f in bench_t bench_b bench_stf; do python - $O/$f.json $f <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(sys.argv[2], d['value'], d['ms_per_step'], d['step_ms']['median'], d['fwd_ms_per_img'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
done
timeout 2400 python -m pytest tests -x -q -m gpu > $O/t_gpu.log 2>&1; echo "rc $?" >> $O/t_gpu.log
tail -n 8 $O/t_gpu.log
