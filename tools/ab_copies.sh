# A/B on ONE box: step time and finalize-on-load cost with 16 vs 8 replicated accumulator copies.  Runs on the gpurun box's
# scratch copy of the tree only; the tracked header and the shipped library are restored on every exit path.
set -eu
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the scratch copy of the tree)}"
H=include/hrfuser_hip.h
cp $H /tmp/ab_copies_orig.h
restore() { cp /tmp/ab_copies_orig.h $H; python -c "from hrfuser_amd import build_ext; build_ext.build()" > /dev/null 2>&1 || true; }
trap restore EXIT
for n in ${COPIES_LIST:-16 8}; do
  sed -i "s/^#define HRF_STAT_COPIES .*/#define HRF_STAT_COPIES $n/" $H
  python -c "from hrfuser_amd import build_ext; build_ext.build()" > /dev/null 2>&1
  echo "== copies $n"
  python tools/bench_fin.py 2>&1 | tail -5
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-neck --no-eager --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], d['step_ms'])"
done
