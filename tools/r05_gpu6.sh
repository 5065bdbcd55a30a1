set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05_6; rm -rf $O; mkdir -p $O
for M in t_nus_bn t_stf_bn; do
for v in "HRF_FFN_EVAL=0" "HRF_FFN_EVAL_MAXC=18" "HRF_FFN_EVAL_MAXC=36" "HRF_FFN_EVAL=0" "HRF_FFN_EVAL_MAXC=18"; do
  env $v python bench.py --model $M --no-cpu-baseline --no-neck --no-eager --no-roofline --steps 5 --warmup 2 > $O/b.json 2>> $O/bench.err
  python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$M $v', 'fwd_ms_per_img', d.get('fwd_ms_per_img'))
PY
done; done
