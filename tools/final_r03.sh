# round-3 artifact collection on ONE box: profiles (kernel trace + PMC), the bench lines, the timeline, the stage trace
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/final_r03; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
bash tools/prof_r03.sh > $O/prof.log 2>&1
cp gpurun_out/prof_r03/r03_hbm_traffic.json profiles/ 2>/dev/null
bash tools/prof_pmc_sq_r03.sh > $O/prof_sq.log 2>&1
cp gpurun_out/pmc_sq_r03/r03_pmc_sq.json $O/ 2>/dev/null
python bench.py --dump-kernels $O/r03_kernels_graph_timed.json > $O/r03_bench_line.json 2> $O/bench.err
python bench.py --model b_nus_bn --no-cpu-baseline --no-neck --no-eager --steps 20 --warmup 5 --dump-kernels $O/r03_kernels_b_nus.json > $O/r03_bench_b_nus.json 2>> $O/bench.err
python bench.py --model t_stf_bn --no-cpu-baseline --no-neck --no-eager --steps 30 --warmup 5 --dump-kernels $O/r03_kernels_t_stf.json > $O/r03_bench_t_stf.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/r03_bench_forced_rccl.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 HRF_SYNC_LANE_COMMS=1 python bench.py --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/r03_bench_forced_rccl_lane_comms.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 python bench.py --model b_nus_bn --no-cpu-baseline --no-neck --no-eager --no-roofline --steps 20 --warmup 5 > $O/r03_bench_forced_rccl_b_nus.json 2>> $O/bench.err
# the launcher form the driver uses for N > 1, on one rank
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/r03_bench_torchrun_n1.json 2>> $O/bench.err
python bench.py --gpus 2 --steps 2 --warmup 1 > $O/r03_bench_gpus2_on_one_gpu.txt 2>&1; echo "rc $?" >> $O/r03_bench_gpus2_on_one_gpu.txt
bash tools/prof_timeline.sh > /dev/null 2>&1
cp gpurun_out/timeline/step_timeline.txt $O/r03_step_timeline.txt; cp gpurun_out/timeline/step_timeline.json $O/r03_step_timeline.json
cp gpurun_out/prof_r03/r03_* gpurun_out/prof_r03/bench_under_rocprof.log $O/
mkdir -p $O/trace; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o st -- python3 $GRAFT_REPO_ROOT/tools/stage_trace.py > $GRAFT_REPO_ROOT/$O/stage_trace.out 2>&1); python tools/stage_trace_report.py $(find $O/trace -name 'st_kernel_trace.csv' | head -1) "fwd stage3" "bwd stage3" > $O/r03_stage_trace.txt 2>&1; rm -rf $O/trace
for f in $O/r03_bench_*.json; do echo $f; tail -1 $f | cut -c1-260; done
cat $O/r03_bench_gpus2_on_one_gpu.txt | tail -3
