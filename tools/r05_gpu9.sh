set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05_9; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
for v in ab_roles0 ab_roles4; do
  HRF_LIB_PATH=$PWD/build_ab/$v.so timeout 900 python -m pytest tests/test_attn_block_abi.py tests/test_parity_blocks.py -m gpu -x -q > $O/pytest_$v.log 2>&1; echo "pytest $v rc $?" | tee -a $O/summary.txt; tail -1 $O/pytest_$v.log | tee -a $O/summary.txt
done
for v in cur ab_roles0 ab_roles4; do
  L=""; [ $v != cur ] && L="HRF_LIB_PATH=$PWD/build_ab/$v.so"
  for M in t_nus_bn t_stf_bn; do
    env $L python bench.py --model $M --no-cpu-baseline --no-neck --no-eager --steps 30 --warmup 8 --dump-kernels $O/kernels_${v}_$M.json > $O/bench_${v}_$M.json 2>> $O/bench.err
    python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_${v}_$M.json').read().strip().splitlines()[-1])
k=json.load(open('$O/kernels_${v}_$M.json'))
ab=[(s['shape'].replace('attn_block_','').replace('B=2,',''),round(s['avg_launch_us'],1),s['launches_per_step']) for s in k['signatures'] if 'attn_block_bwd' in s['shape']]
print('$v $M', d['ms_per_step'], 'fwd_ms_per_img', d.get('fwd_ms_per_img'), ab)
PY
  done
done
