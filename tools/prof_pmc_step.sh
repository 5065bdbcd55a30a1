cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export HRF_LANES=0
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r01
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/sf -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $OUT/step_fetch.log 2>&1; echo fetch rc=$?
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/sw -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $OUT/step_write.log 2>&1; echo write rc=$?
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections,json,re,shutil
OUT='gpurun_out/prof_r01'
def fam(n):
    n=n.replace('(anonymous namespace)::',''); n=re.sub(r'\(.*','',n).replace('void ','')
    return n
res={}
for tag,cn in (('sf','FETCH_SIZE'),('sw','WRITE_SIZE')):
    fs=glob.glob(OUT+f'/{tag}/**/*counter_collection.csv', recursive=True)
    if not fs: print('no',tag); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get('Counter_Name')!=cn: continue
        k=fam(r['Kernel_Name'])
        if 'at::' in k or 'rocclr' in k: continue
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    res[cn]={k:{'calls':v[0],'sum_kb':v[1]} for k,v in agg.items()}
json.dump(res, open(OUT+'/r01_pmc_step_raw.json','w'), indent=1)
for cn,d in res.items():
    print(cn, len(d), sorted(((v['sum_kb'],k,v['calls']) for k,v in d.items()), reverse=True)[:8])
for d in ('sf','sw'): shutil.rmtree(OUT+'/'+d, ignore_errors=True)
PY
tail -3 $OUT/step_fetch.log | cut -c1-300
