cd "${GRAFT_REPO_ROOT:?run through gpurun}"; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_conv3w; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_conv3w.py > $OUT/$tag.log 2>&1; echo "$grp rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections,json,shutil,os
OUT='gpurun_out/pmc_conv3w'
res=collections.defaultdict(lambda:[0,0.0])
for f in glob.glob(OUT+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv3w_kernel' not in r['Kernel_Name']: continue
        k=r['Counter_Name']; res[k][0]+=1; res[k][1]+=float(r['Counter_Value'])
out={k:v[1]/v[0] for k,v in res.items()}
json.dump(out, open(OUT+'/r01_pmc_conv3w.json','w'), indent=1)
print(json.dumps(out, indent=1))
for d in os.listdir(OUT):
    if os.path.isdir(OUT+'/'+d): shutil.rmtree(OUT+'/'+d, ignore_errors=True)
PY
