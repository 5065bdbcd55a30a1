# final neck artefacts for profiles/: perf lines (HIP vs eager), serial-lane kernel trace (isolated kernel times)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/neck_final
python3 tests/perf_neck.py > gpurun_out/neck_final/r01_neck_bench.jsonl 2> gpurun_out/neck_final/perf.err
export HRF_LANES=0
bash tools/prof_neck.sh > gpurun_out/neck_final/prof_serial.log 2>&1
cp gpurun_out/prof_neck/neck_kernel_trace_summary.csv gpurun_out/neck_final/r01_neck_kernel_trace_serial_T.csv
unset HRF_LANES
bash tools/prof_neck.sh > gpurun_out/neck_final/prof_lanes.log 2>&1
cp gpurun_out/prof_neck/neck_kernel_trace_summary.csv gpurun_out/neck_final/r01_neck_kernel_trace_lanes_T.csv
cat gpurun_out/neck_final/r01_neck_bench.jsonl
