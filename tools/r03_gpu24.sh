#!/bin/bash
# per-stage PMC traffic for HRFuser-B and the 4-modality STF config (same method as tools/r03_gpu23.sh)
set -u
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
O=$PWD/gpurun_out/r03v
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for tag in b_nus_bn t_stf_bn; do
  cd /tmp
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$tag -o s -- python3 $GRAFT_REPO_ROOT/tools/pmc_stages.py $tag > $O/fetch_$tag.log 2>&1; echo "$tag fetch rc=$?"
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$tag -o s -- python3 $GRAFT_REPO_ROOT/tools/pmc_stages.py $tag > $O/write_$tag.log 2>&1; echo "$tag write rc=$?"
  cd $GRAFT_REPO_ROOT
  F=$(find $O/fetch_$tag -name '*counter_collection.csv' | head -1); W=$(find $O/write_$tag -name '*counter_collection.csv' | head -1)
  python tools/pmc_stages_report.py "$F" "$W" $O/r03_stage_hbm_traffic_${tag%_bn}.json 2>&1 | tail -3
  rm -rf $O/fetch_$tag $O/write_$tag
done
