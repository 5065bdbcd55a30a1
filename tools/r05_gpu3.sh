# round 5, GPU call 3: full GPU suite + headline bench line with stage table
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r05_3; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
python bench.py --no-cpu-baseline --no-neck --no-eager --dump-kernels $O/kernels_t_nus.json > $O/bench_t_nus.json 2> $O/bench.err; echo "bench rc $?" | tee -a $O/summary.txt
python - <<PY | tee -a $O/summary.txt
import json
d=json.loads(open('$O/bench_t_nus.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'fwd_ms_per_img', d.get('fwd_ms_per_img'))
r=d.get('roofline') or {}
print('roofline', r.get('kernel'), r.get('frac'), r.get('selection','')[:100]); print('ranking', r.get('family_ranking_ms')); print('resources', r.get('resources'))
for s in (d.get('stage_roofline') or {}).get('stages', []): print(s['name'], s.get('fwd_ms'), s.get('bwd_ms'), s.get('fwd_flops_frac'), s.get('bwd_flops_frac'))
PY
timeout 1400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest_gpu rc $?" | tee -a $O/summary.txt; tail -25 $O/pytest_gpu.log | tee -a $O/summary.txt
