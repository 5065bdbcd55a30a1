#!/bin/bash
# full GPU suite (no -x: one failure must not hide the rest)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03j
mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -q -m gpu --durations=15 > $O/t_gpu.log 2>&1; echo "rc $?" >> $O/t_gpu.log
tail -n 30 $O/t_gpu.log
