# A/B of runtime knobs (environment only, same build, same box): prints ms/step per setting; settings from "$@" or defaults
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python bench.py ${BENCH_ARGS:---steps 60 --warmup 10} --no-cpu-baseline --no-neck --no-eager --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"; }
for rep in $(seq 1 ${REPS:-2}); do
  for s in "${@:-HRF_X=0}"; do run $s; done
done
