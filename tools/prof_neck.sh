# rocprofv3 kernel trace of the neck (HIP leg, T shapes by default): per-kernel summary -> gpurun_out/prof_neck/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_neck
rm -rf $OUT; mkdir -p $OUT
export NECK_ONLY=${NECK_ONLY:-hip} NECK_TAG=${NECK_TAG:-T}
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o n -- python3 $GRAFT_REPO_ROOT/tests/perf_neck.py > $OUT/bench_neck_under_rocprof.log 2>&1
echo trace rc=$?
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections,os,re
OUT='gpurun_out/prof_neck'
def short(n):
    n=n.replace('(anonymous namespace)::',''); n=re.sub(r'\(.*','',n); return n.replace('void ','')
tr=glob.glob(OUT+'/trace/**/*kernel_trace.csv', recursive=True)
rows=list(csv.DictReader(open(tr[0])))
agg=collections.defaultdict(lambda:[0,0])
for r in rows:
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    k=short(r['Kernel_Name'])+' grid=%s wg=%s'%(r.get('Grid_Size_X','?'),r.get('Workgroup_Size_X','?'))
    agg[k][0]+=1; agg[k][1]+=d
with open(OUT+'/neck_kernel_trace_summary.csv','w') as fh:
    fh.write('kernel,calls,total_ns,avg_ns\n')
    for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): fh.write(f'"{k}",{v[0]},{v[1]},{v[1]/v[0]:.1f}\n')
import shutil; shutil.rmtree(OUT+'/trace', ignore_errors=True)
PY
head -40 $OUT/neck_kernel_trace_summary.csv; tail -2 $OUT/bench_neck_under_rocprof.log
