# Round artifact collection on ONE box: bash tools/final_round.sh r04 -> gpurun_out/final_r04/ (copy what is to be judged into
# profiles/): the bench lines (headline, HRFuser-B, STF, forced SyncBN schedules), rocprofv3 kernel trace + PMC passes, the step
# timeline, the profiler-free lane stamps and the stage trace.
set -u
R=${1:?round tag, e.g. r04}
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
O=gpurun_out/final_$R; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
# round 6: the PMC passes run FIRST and their summaries go into profiles/ of this box's copy, so that the bench line's `traffic`,
# `traffic_ratio`, per-stage MFMA-busy / bytes name files of THIS round (the same files are committed from gpurun_out/ afterwards)
bash tools/prof_round.sh $R > $O/prof.log 2>&1
cp gpurun_out/prof_$R/${R}_* gpurun_out/prof_$R/bench_under_rocprof.log $O/ 2>/dev/null
cp gpurun_out/prof_$R/${R}_hbm_traffic.json profiles/ 2>/dev/null
bash tools/prof_stages_pmc.sh $R > $O/prof_stages.log 2>&1
cp gpurun_out/pmc_stages_$R/${R}_stage_hbm_traffic.json $O/ 2>/dev/null; cp gpurun_out/pmc_stages_$R/${R}_stage_hbm_traffic.json profiles/ 2>/dev/null
python bench.py --dump-kernels $O/${R}_kernels_graph_timed.json > $O/${R}_bench_line.json 2> $O/bench.err
python bench.py --model b_nus_bn --no-cpu-baseline --no-neck --no-eager --steps 20 --warmup 5 --dump-kernels $O/${R}_kernels_b_nus.json > $O/${R}_bench_b_nus.json 2>> $O/bench.err
python bench.py --model t_stf_bn --no-cpu-baseline --no-neck --no-eager --steps 30 --warmup 5 --dump-kernels $O/${R}_kernels_t_stf.json > $O/${R}_bench_t_stf.json 2>> $O/bench.err
# the reference's own per-GPU training batch for HRFuser-T (3 images: cascade_rcnn_hrfuser_t_1x_nus_r640_l_r_fusion.py:49)
python bench.py --batch 3 --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/${R}_bench_batch3.json 2>> $O/bench.err
# the SyncBN configuration on ONE GPU (forced one-rank group): collective main-lane schedule, one communicator per lane, peer-to-peer exchange
HRF_FORCE_COLLECTIVES=1 HRF_SYNC_P2P=0 python bench.py --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/${R}_bench_forced_rccl.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 HRF_SYNC_P2P=0 HRF_SYNC_LANE_COMMS=1 python bench.py --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/${R}_bench_forced_rccl_lane_comms.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 HRF_SYNC_P2P=1 python bench.py --no-cpu-baseline --no-neck --no-eager --no-roofline > $O/${R}_bench_forced_p2p.json 2>> $O/bench.err
HRF_FORCE_COLLECTIVES=1 HRF_SYNC_P2P=1 python bench.py --model b_nus_bn --no-cpu-baseline --no-neck --no-eager --no-roofline --steps 20 --warmup 5 > $O/${R}_bench_forced_p2p_b_nus.json 2>> $O/bench.err
# two ranks sharing this GPU (gloo control plane, IPC inboxes): the launcher, the auto-selected peer-to-peer exchange, the sync_ab arms
python bench.py --gpus 2 --backend gloo --steps 3 --warmup 1 --height 128 --width 192 --no-roofline --sync-ab-timeout 600 > $O/${R}_bench_two_ranks_one_gpu.json 2>> $O/bench.err
python bench.py --gpus 2 --steps 2 --warmup 1 > $O/${R}_bench_gpus2_on_one_gpu.txt 2>&1; echo "rc $?" >> $O/${R}_bench_gpus2_on_one_gpu.txt
python tools/lane_stamps.py > $O/${R}_lane_stamps.txt 2>> $O/bench.err
bash tools/prof_pmc_sq.sh $R > $O/prof_sq.log 2>&1
cp gpurun_out/pmc_sq_$R/${R}_pmc_sq.json $O/ 2>/dev/null
bash tools/prof_timeline.sh > /dev/null 2>&1
cp gpurun_out/timeline/step_timeline.txt $O/${R}_step_timeline.txt; cp gpurun_out/timeline/step_timeline.json $O/${R}_step_timeline.json
bash tools/stage_trace.sh final_$R/st t_nus_bn -- "fwd stage3" "bwd stage3" "fwd stage4" > /dev/null 2>&1; cp $O/st/stage_trace.txt $O/${R}_stage_trace.txt 2>/dev/null; rm -rf $O/st
# gradient all-reduce inside the weight-gradient phase (HRFuser-B: four bucket groups; HRFuser-T forced to four)
python tools/exchange_overlap.py b_nus_bn > $O/${R}_grad_exchange_overlap_b_nus.txt 2>> $O/bench.err
python tools/exchange_overlap.py t_nus_bn 4 > $O/${R}_grad_exchange_overlap_t_nus_forced4.txt 2>> $O/bench.err
# weight gradients: per-variant table of the grouped launches, isolated pixel-major vs LDS-tiled kernel
python tools/wgrad_groups.py t_nus_bn > $O/${R}_wgrad_groups_t_nus.txt 2>> $O/bench.err
python tools/wgrad_groups.py b_nus_bn > $O/${R}_wgrad_groups_b_nus.txt 2>> $O/bench.err
python tools/bench_wgrad_tiled.py > $O/${R}_wgrad_tiled_microbench.txt 2>> $O/bench.err
# round 6 microbenchmarks: front-end 3x3 engines old vs packed, 3x3 weight gradient pixel-major vs LDS-staged, row-GEMM decomposition
# (what moments / transforms / epilogues cost), same-line atomics by copies and scope
python tools/bench_conv3x.py $O/${R}_conv3x_microbench.json 2>> $O/bench.err | grep -v amdgpu > $O/${R}_conv3x_microbench.txt
python tools/bench_wgrad3x.py 2>> $O/bench.err | grep -v amdgpu > $O/${R}_wgrad3x_microbench.txt
python tools/bench_lin.py $O/${R}_lin_decomposition.json 2>> $O/bench.err | grep -v amdgpu > $O/${R}_stream_kernels_decomposition.txt
python tools/bench_dw.py $O/${R}_dw_decomposition.json 2>> $O/bench.err | grep -v amdgpu >> $O/${R}_stream_kernels_decomposition.txt
./tools/microbench/atomics_scope > $O/${R}_atomics_scope.txt 2>> $O/bench.err
python tools/stream_report.py $R $O > /dev/null 2>> $O/bench.err
python tools/eval_kernels.py t_nus_bn $O/${R}_eval_kernels.json 2>> $O/bench.err | grep -v amdgpu > $O/${R}_eval_kernels.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${R}_smoke.txt 2>&1
# the full GPU suite with its slowest calls
timeout 1400 python -m pytest tests -m gpu -x -q > $O/${R}_gpu_suite_durations.txt 2>&1; echo "gpu suite rc $?" >> $O/${R}_gpu_suite_durations.txt
for f in $O/${R}_bench_*.json; do echo $f; tail -1 $f | cut -c1-260; done
tail -3 $O/${R}_gpu_suite_durations.txt
