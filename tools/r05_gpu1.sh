# round 5, GPU call 1: parity of the rewritten kernels + A/B of the fused attention backward (8 waves, role split) against round 4
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05_1; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_attn_block_abi.py tests/test_parity_blocks.py tests/test_p2p_exchange.py -m gpu -x -q > $O/pytest_a.log 2>&1; echo "pytest_a rc $?" | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv_gpu or wgrad" > $O/pytest_b.log 2>&1; echo "pytest_b rc $?" | tee -a $O/summary.txt
for v in cur ab_minw4 r04; do
  L=""; [ $v != cur ] && L="HRF_LIB_PATH=$PWD/build_ab/$v.so"
  for M in t_nus_bn t_stf_bn; do
    env $L python bench.py --model $M --no-cpu-baseline --no-neck --no-eager --steps 30 --warmup 8 --dump-kernels $O/kernels_${v}_$M.json > $O/bench_${v}_$M.json 2>> $O/bench.err
    python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_${v}_$M.json').read().strip().splitlines()[-1])
k=json.load(open('$O/kernels_${v}_$M.json'))
ab=[(s['shape'],round(s['avg_launch_us'],1),s['launches_per_step']) for s in k['signatures'] if 'attn_block_bwd' in s['shape']]
print('$v $M', d['ms_per_step'], 'fwd_ms_per_img', d.get('fwd_ms_per_img'), ab)
PY
  done
done
