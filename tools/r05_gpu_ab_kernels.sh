# A/B of two library builds: step time + isolated times of selected kernel families.  bash tools/r05_gpu_ab_kernels.sh OUT PATTERN lib1 [lib2 ...]  ("cur" = in-tree)
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/$1; PAT=$2; shift 2
rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in "$@"; do
  L=""; [ $v != cur ] && L="HRF_LIB_PATH=$PWD/build_ab/$v.so"
  env $L python bench.py --model ${MODEL:-t_nus_bn} --no-cpu-baseline --no-neck --no-eager --steps 30 --warmup 8 --dump-kernels $O/k_${v}.json > $O/b_${v}.json 2>> $O/bench.err
  python - <<PY | tee -a $O/summary.txt
import json,re
d=json.loads(open('$O/b_${v}.json').read().strip().splitlines()[-1])
k=json.load(open('$O/k_${v}.json'))
rows=[(s['shape'][:70],round(s['avg_launch_us'],1),s['launches_per_step']) for s in k['signatures'] if re.search(r'$PAT', s['shape']+' '+s['kernel'])][:14]
print('$v rep$rep', d['ms_per_step'], 'fwd', d.get('fwd_ms_per_img'), rows)
PY
done; done
