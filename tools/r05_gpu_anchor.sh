# A/B of the stage-level fork anchor (HRF_STAGE_ANCHOR): step time + lane stamps.   bash tools/r05_gpu_anchor.sh OUT
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for v in 0 1; do
  HRF_STAGE_ANCHOR=$v python bench.py --model ${MODEL:-t_nus_bn} --no-cpu-baseline --no-neck --no-eager --no-roofline --steps 30 --warmup 8 > $O/b_$v.json 2>> $O/bench.err
  python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open('$O/b_$v.json').read().strip().splitlines()[-1])
print('anchor=$v rep$rep', d['ms_per_step'], 'fwd', d.get('fwd_ms_per_img'))
PY
done; done
for v in 0 1; do HRF_STAGE_ANCHOR=$v python tools/lane_stamps.py ${MODEL:-t_nus_bn} > $O/lane_stamps_$v.txt 2>> $O/bench.err; done
