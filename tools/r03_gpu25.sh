#!/bin/bash
# where do the lanes of stage 3 land with exactly 4 streams (camera b0 on main, b1, b2, ONE stream for both modality stages)?
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
O=$PWD/gpurun_out/r03w
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for cfg in "default" "HRF_KEEP_FIRST=1 HRF_MOD_LANES=1"; do
  tag=$(echo "$cfg" | tr ' =' '__')
  mkdir -p $O/trace
  (cd /tmp && env $([ "$cfg" = default ] || echo $cfg) rocprofv3 --kernel-trace --output-format csv -d $O/trace -o st -- python3 $GRAFT_REPO_ROOT/tools/stage_trace.py > $O/stage_trace_$tag.out 2>&1)
  python tools/stage_trace_report.py $(find $O/trace -name 'st_kernel_trace.csv' | head -1) "fwd stage3" > $O/stage_$tag.txt 2>&1
  rm -rf $O/trace
  echo "== $cfg"; grep -E "^fwd|^bwd|^step" $O/stage_$tag.txt | head -30
done
