# rocprofv3 kernel trace of one captured training step cut at the stage stamps: bash tools/stage_trace.sh OUTDIR [model] -- stage names...
set -u
cd "${GRAFT_REPO_ROOT:?}"
O=$GRAFT_REPO_ROOT/gpurun_out/$1; M=${2:-t_nus_bn}; shift; shift; [ "${1:-}" = "--" ] && shift
mkdir -p $O/trace
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/trace -o st -- python3 $GRAFT_REPO_ROOT/tools/stage_trace.py $M > $O/stage_trace.out 2>&1)
python tools/stage_trace_report.py $(find $O/trace -name 'st_kernel_trace.csv' | head -1) $O/stage_trace.out "$@" > $O/stage_trace.txt 2>&1
rm -rf $O/trace
head -30 $O/stage_trace.txt
