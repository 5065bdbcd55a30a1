# A/B on ONE box of the replicated-accumulator count over the three BASELINE models.  COPIES_LIST="4 8" bash tools/ab_copies_models.sh
set -eu
cd "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the scratch copy of the tree)}"
H=include/hrfuser_hip.h
cp $H /tmp/ab_copies_orig.h
restore() { cp /tmp/ab_copies_orig.h $H; python -c "from hrfuser_amd import build_ext; build_ext.build()" > /dev/null 2>&1 || true; }
trap restore EXIT
for n in ${COPIES_LIST:-4 8}; do
  sed -i "s/^#define HRF_STAT_COPIES .*/#define HRF_STAT_COPIES $n/" $H
  python -c "from hrfuser_amd import build_ext; build_ext.build()" > /dev/null 2>&1
  for m in t_nus_bn b_nus_bn t_stf_bn; do
    python bench.py --model $m --steps 30 --warmup 8 --no-cpu-baseline --no-neck --no-eager --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('copies $n $m ms_per_step', d['ms_per_step'], 'fwd', d.get('fwd_ms_per_img'))"
  done
done
