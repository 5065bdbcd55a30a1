# A/B of environment settings: step time (+ lane stamps).   bash tools/r05_gpu_envab.sh OUT "NAME:ENV=V ENV2=V" ...
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/$1; shift; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs python bench.py --model ${MODEL:-t_nus_bn} --no-cpu-baseline --no-neck --no-eager --no-roofline --steps 30 --warmup 8 > $O/b_$name.json 2>> $O/bench.err
  python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1])
print('$name rep$rep', d['ms_per_step'], 'fwd', d.get('fwd_ms_per_img'))
PY
done; done
if [ "${STAMPS:-1}" = 1 ]; then
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  env $envs python tools/lane_stamps.py ${MODEL:-t_nus_bn} > $O/lane_stamps_$name.txt 2>> $O/bench.err
done; fi
