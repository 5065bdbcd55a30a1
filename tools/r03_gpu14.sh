#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
for v in 4 18; do
  echo "== wave_gemm_tl unroll $v (wave_tgemm 4)" >> $O/ab_phases.txt
  HRF_TIMING_LIB=$PWD/scratch/libhrf_ab_tl$v.so timeout 300 python tools/time_ab_phases.py >> $O/ab_phases.txt 2>&1
  HRF_TIMING_LIB=$PWD/scratch/libhrf_ab_tl$v.so timeout 300 python tools/time_ab_phases.py 36 2 >> $O/ab_phases.txt 2>&1
done
grep -v "^tick" $O/ab_phases.txt | grep -v amdgpu
B="--steps 30 --warmup 5 --no-cpu-baseline --no-neck --no-eager --no-roofline"
run() { name=$1; shift; ( "$@" ) > $O/$name.json 2> $O/$name.err; python - $O/$name.json $name <<'PY'
import sys,json
try:
    d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1]); print(sys.argv[2], d['ms_per_step'], d['step_ms']['median'], d['fwd_ms_per_img'])
except Exception as e: print(sys.argv[2], 'ERR', e)
PY
}
run t1 timeout 600 python bench.py $B
run t2 timeout 600 python bench.py $B
run stf timeout 600 python bench.py $B --model t_stf_bn
timeout 900 python -m pytest tests/test_attn_block_abi.py tests/test_parity_blocks.py -x -q -m gpu > $O/t_ab.log 2>&1; echo "rc $?" >> $O/t_ab.log
tail -n 3 $O/t_ab.log
