# MEASURED fabric traffic per STAGE: bash tools/prof_stages_pmc.sh r04 [model] -> gpurun_out/pmc_stages_r04/r04_stage_hbm_traffic[_model].json
# (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes, --kernel-trace only, over ONE eager single-stream step: tools/pmc_stages.py)
R=${1:?round tag}; M=${2:-t_nus_bn}
cd "${GRAFT_REPO_ROOT:?run through gpurun}"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_stages_$R; mkdir -p $OUT; cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o s -- python3 $GRAFT_REPO_ROOT/tools/pmc_stages.py $M > $OUT/$c.out 2> $OUT/$c.err; echo "$c rc=$?"
done
# third pass (round 6): matrix-pipe busy cycles + active cycles of the same step (MFMA-busy PER STAGE)
rm -rf $OUT/SQ
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/SQ -o s -- python3 $GRAFT_REPO_ROOT/tools/pmc_stages.py $M > $OUT/SQ.out 2> $OUT/SQ.err; echo "SQ rc=$?"
cd $GRAFT_REPO_ROOT
sfx=""; [ "$M" != "t_nus_bn" ] && sfx="_${M%_bn}"
python tools/pmc_stages_report.py $(find $OUT/FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $OUT/WRITE_SIZE -name '*counter_collection.csv' | head -1) $OUT/${R}_stage_hbm_traffic$sfx.json $OUT/FETCH_SIZE.out "$(find $OUT/SQ -name '*counter_collection.csv' | head -1)"
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE $OUT/SQ
