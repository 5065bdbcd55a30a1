#!/bin/bash
# round-3 first GPU pass: new scheduler + multi-problem launches - quick parity subset, then bench A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r03a
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_grouping.py tests/test_attn_block_abi.py -x -q -m gpu > $O/t_group.log 2>&1; echo "rc $?" >> $O/t_group.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-neck --no-eager > $O/bench_group.json 2> $O/bench_group.err
HRF_GROUP=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/bench_nogroup.json 2> $O/bench_nogroup.err
HRF_LOCKSTEP=0 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/bench_serial.json 2> $O/bench_serial.err
timeout 1500 python -m pytest tests/test_parity_blocks.py tests/test_parity_wholenet.py -x -q -m gpu -k "not fullres" > $O/t_parity.log 2>&1; echo "rc $?" >> $O/t_parity.log
tail -3 $O/t_group.log $O/t_parity.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], d['value'], d['ms_per_step'], d['step_ms'], (d.get('roofline') or {}).get('library_launches_per_step'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
