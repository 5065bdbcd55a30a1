cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_grouped; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_grouped.py > $OUT/pmc_fetch.log 2>&1; echo fetch rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 $GRAFT_REPO_ROOT/tools/pmc_grouped.py > $OUT/pmc_write.log 2>&1; echo write rc=$?
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections,json,re,shutil
OUT='gpurun_out/prof_grouped'
def short(n):
    n=n.replace('(anonymous namespace)::',''); n=re.sub(r'\(.*','',n); return n.replace('void ','')
res={}
for tag,cn in (('pmc_fetch','FETCH_SIZE'),('pmc_write','WRITE_SIZE')):
    fs=glob.glob(OUT+f'/{tag}/**/*counter_collection.csv', recursive=True)
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get('Counter_Name')!=cn: continue
        k=short(r['Kernel_Name'])
        if 'wgrad_dense' not in k: continue
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    res[cn]={k:{'launches':v[0],'avg':v[1]/v[0]} for k,v in agg.items()}
alg=[l for l in open(OUT+'/pmc_fetch.log') if l.startswith('algorithmic bytes')]
res['algorithmic']=int(alg[0].split()[-1]) if alg else None
json.dump(res, open(OUT+'/r01_pmc_grouped_raw.json','w'), indent=1)
print(json.dumps(res, indent=1))
for d in ('pmc_fetch','pmc_write'): shutil.rmtree(OUT+'/'+d, ignore_errors=True)
PY
