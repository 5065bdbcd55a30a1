#!/bin/bash
# per-stage PMC traffic: one eager single-stream step per counter (bounded by a timeout: a whole-step --pmc pass crashed the tool in round 1)
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
O=$PWD/gpurun_out/r03u
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o s -- python3 $GRAFT_REPO_ROOT/tools/pmc_stages.py > $O/fetch.log 2>&1; echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o s -- python3 $GRAFT_REPO_ROOT/tools/pmc_stages.py > $O/write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
F=$(find $O/fetch -name '*counter_collection.csv' | head -1); W=$(find $O/write -name '*counter_collection.csv' | head -1)
echo "csv: $F $W"
python tools/pmc_stages_report.py "$F" "$W" $O/r03_stage_hbm_traffic.json 2>&1 | tail -30
tail -3 $O/fetch.log | cut -c1-300
rm -rf $O/fetch $O/write
