# round-2 artifact collection on ONE box: profiles (kernel trace + PMC), the bench lines, the timeline
cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r02; rm -rf $O; mkdir -p $O
bash tools/prof_r02.sh > $O/prof.log 2>&1
cp gpurun_out/prof_r02/r02_hbm_traffic.json profiles/ 2>/dev/null
python bench.py --dump-kernels $O/r02_kernels_graph_timed.json > $O/r02_bench_line.json 2> $O/bench.err
python bench.py --model b_nus_bn --no-cpu-baseline --no-neck --no-eager --steps 20 --warmup 5 --dump-kernels $O/r02_kernels_b_nus.json > $O/r02_bench_b_nus.json 2>> $O/bench.err
python bench.py --model t_stf_bn --no-cpu-baseline --no-neck --no-eager --steps 30 --warmup 5 --dump-kernels $O/r02_kernels_t_stf.json > $O/r02_bench_t_stf.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 python bench.py --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/r02_bench_forced_rccl.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 HRF_SYNC_BATCH=0 python bench.py --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/r02_bench_forced_rccl_unbatched.json 2>> $O/bench.err
bash tools/prof_timeline.sh > /dev/null 2>&1
cp gpurun_out/timeline/step_timeline.txt gpurun_out/timeline/step_timeline.json $O/
cp gpurun_out/prof_r02/r02_* gpurun_out/prof_r02/bench_under_rocprof.log $O/
for f in $O/r02_bench_*.json; do echo $f; tail -1 $f | cut -c1-260; done
