# kernel trace of a short bench run -> timeline analysis of one steady-state step (tools/step_timeline.py)
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-roofline --no-neck --no-eager $BENCH_EXTRA > $OUT/bench.log 2>&1
echo trace rc=$?
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
F=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/step_timeline.py $F $OUT/step_timeline.json > $OUT/step_timeline.txt 2>&1
cat $OUT/step_timeline.txt
cp $F $OUT/kernel_trace.csv; rm -rf $OUT/trace
