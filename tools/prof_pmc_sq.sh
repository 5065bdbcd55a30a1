R=${1:?round tag, e.g. r04}; export HRF_ROUND=$R
# SQ counters (MFMA busy, LDS conflicts, waves) of the hot kernels at their branch-0 shapes (tools/pmc_kernels.py),
# one rocprofv3 --pmc pass per counter group, --kernel-trace only
cd "${GRAFT_REPO_ROOT:?run through gpurun}"; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$R; rm -rf $OUT; mkdir -p $OUT; cd /tmp
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU" \
           "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_LEVEL_VMEM SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_WAIT_ANY"; do
  tag=$(echo $grp | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernels.py > $OUT/$tag.log 2>&1; echo "$grp rc=$?"
done
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
python3 - <<'PY'
import csv,glob,collections,json,shutil,os,re
import os as _os
R=_os.environ['HRF_ROUND']
OUT='gpurun_out/pmc_sq_'+R
def short(n):
    n=n.replace('(anonymous namespace)::',''); n=re.sub(r'\(.*','',n); return n.replace('void ','')
res=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
for f in glob.glob(OUT+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=short(r['Kernel_Name'])
        if 'at::' in k or 'rocclr' in k: continue
        c=r['Counter_Name']; res[k][c][0]+=1; res[k][c][1]+=float(r['Counter_Value'])
out={}
for k,d in res.items():
    v={c:x[1]/x[0] for c,x in d.items()}
    # derived: MFMA pipe busy fraction = MFMA busy cycles / (SIMDs x active cycles); GRBM_GUI_ACTIVE sums the 8 XCDs
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v and v['GRBM_GUI_ACTIVE']>0:
        v['mfma_busy_frac']=round(v['SQ_VALU_MFMA_BUSY_CYCLES']/1024.0/(v['GRBM_GUI_ACTIVE']/8.0),4)
    # round 6 (streaming kernels): waves resident per CU averaged over the launch, share of wave-cycles spent waiting on ANY
    # instruction dependency (s_waitcnt / data hazards) resp. issuing, vector-memory instructions in flight per CU
    if v.get('SQ_WAVE_CYCLES',0)>0 and v.get('GRBM_GUI_ACTIVE',0)>0:
        v['waves_per_cu_avg']=round(v['SQ_WAVE_CYCLES']*4.0/256.0/(v['GRBM_GUI_ACTIVE']/8.0),2)   # SQ_WAVE_CYCLES counts in quad-cycles
    if v.get('SQ_WAVE_CYCLES',0)>0 and 'SQ_WAIT_INST_ANY' in v:
        v['wait_inst_frac']=round(v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES'],4)
        v['active_inst_frac']=round(v.get('SQ_ACTIVE_INST_ANY',0)/v['SQ_WAVE_CYCLES'],4)
    if v.get('SQ_INST_LEVEL_VMEM',0)>0 and v.get('GRBM_GUI_ACTIVE',0)>0:
        v['vmem_in_flight_per_cu']=round(v['SQ_INST_LEVEL_VMEM']/256.0/(v['GRBM_GUI_ACTIVE']/8.0),2)
    if 'SQ_LDS_BANK_CONFLICT' in v and v.get('SQ_LDS_IDX_ACTIVE',0)>0:
        v['lds_conflict_frac']=round(v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE'],4)
    out[k]=v
json.dump(out, open(OUT+'/'+R+'_pmc_sq.json','w'), indent=1)
for k,v in out.items(): print(k, {c:(round(x,4) if x<10 else round(x)) for c,x in v.items()})
for d in os.listdir(OUT):
    if os.path.isdir(OUT+'/'+d): shutil.rmtree(OUT+'/'+d, ignore_errors=True)
PY
