# A/B of a compile-time constant on ONE box: ab_const.sh <file> <sed-pattern-with-@V@> <values...>.  Runs on the gpurun box's
# scratch copy of the tree only; the edited file and the shipped library are restored on every exit path.
# Prefer HRF_EXTRA_FLAGS=-DNAME=value builds copied to scratch/ (tools/time_lin2_phases.py) when the constant is a macro.
set -eu
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the scratch copy of the tree)}"
F=$1; PAT=$2; shift 2
cp $F /tmp/ab_orig
restore() { cp /tmp/ab_orig $F; python -c "from hrfuser_amd import build_ext; build_ext.build()" > /dev/null 2>&1 || true; }
trap restore EXIT
for rep in 1 2; do for v in "$@"; do
  cp /tmp/ab_orig $F
  sed -i "$(echo "$PAT" | sed "s/@V@/$v/g")" $F
  python -c "from hrfuser_amd import build_ext; build_ext.build()" > /dev/null 2>&1
  echo -n "value $v : "
  python bench.py ${BENCH_ARGS:---steps 60 --warmup 10} --no-cpu-baseline --no-neck --no-eager --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['median'])"
done; done
