#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r03g
mkdir -p $O/trace
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o st -- python3 tools/stage_trace.py > $O/stage_trace.out 2>&1
CSV=$(find $O/trace -name 'st_kernel_trace.csv' | head -1)
python tools/stage_trace_report.py $CSV "fwd stage3" "bwd stage3" "fwd stems" "fwd transitions_a" > $O/stage_report.txt 2>&1
head -40 $O/stage_report.txt
rm -rf $O/trace
