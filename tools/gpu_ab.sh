# One-box A/B of the training step: bash tools/gpu_ab.sh OUT "ENV1" "ENV2" ... (each ENV = space-separated VAR=val list or "-"),
# every variant timed twice, interleaved (box-to-box spread is 2-4 %, so only same-box pairs count).  Optional MODEL=b_nus_bn.
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/$1; shift
mkdir -p $O
M=${MODEL:-t_nus_bn}
for rep in 1 2; do
  k=0
  for v in "$@"; do
    k=$((k+1))
    [ "$v" = "-" ] && v=""
    env $v python bench.py --model $M --no-cpu-baseline --no-neck --no-eager --no-roofline --steps ${STEPS:-40} --warmup 10 > $O/ab_${k}_$rep.json 2>> $O/ab.err
    echo "variant $k [$v] rep $rep: $(python -c "import json,sys; d=json.loads(open('$O/ab_${k}_$rep.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], 'fwd_ms_per_img', d.get('fwd_ms_per_img'))" 2>&1)"
  done
done
