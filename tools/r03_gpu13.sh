#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
for v in 1 4 13; do
  echo "== wave_tgemm unroll $v" >> $O/ab_phases.txt
  HRF_TIMING_LIB=$PWD/scratch/libhrf_ab_u$v.so timeout 300 python tools/time_ab_phases.py >> $O/ab_phases.txt 2>&1
  HRF_TIMING_LIB=$PWD/scratch/libhrf_ab_u$v.so timeout 300 python tools/time_ab_phases.py 36 2 >> $O/ab_phases.txt 2>&1
done
cat $O/ab_phases.txt
timeout 1500 python -m pytest tests/test_groupnorm.py tests/test_kernels.py -x -q -m gpu -k "groupnorm or gn_ or pointwise" > $O/t_gn.log 2>&1; echo "rc $?" >> $O/t_gn.log
tail -n 5 $O/t_gn.log
