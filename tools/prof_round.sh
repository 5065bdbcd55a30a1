# per-round profiles: bash tools/prof_round.sh r04 -> gpurun_out/prof_r04/: kernel trace + stats of the bench command, PMC fabric
# traffic of the hot kernels (separate --pmc passes, --kernel-trace only; gfx950 corrections: profiles/README.md)
R=${1:?round tag, e.g. r04}; export HRF_ROUND=$R
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o $R -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-neck --no-eager > $OUT/bench_under_rocprof.log 2>&1
echo trace rc=$?
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernels.py > $OUT/pmc_fetch.log 2>&1; echo fetch rc=$?
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o w -- python3 $GRAFT_REPO_ROOT/tools/pmc_kernels.py > $OUT/pmc_write.log 2>&1; echo write rc=$?
cd "${GRAFT_REPO_ROOT:?run through gpurun}"
python3 - <<'PY'
import csv,glob,collections,os,json,re,shutil
import os as _os
R=_os.environ['HRF_ROUND']
OUT='gpurun_out/prof_'+R
def short(n):
    n=n.replace('(anonymous namespace)::',''); n=re.sub(r'\(.*','',n); return n.replace('void ','')
for f in glob.glob(OUT+'/trace/**/*kernel_stats.csv', recursive=True): os.replace(f, OUT+'/'+R+'_kernel_stats.csv')
tr=glob.glob(OUT+'/trace/**/*kernel_trace.csv', recursive=True)
if tr:
    rows=list(csv.DictReader(open(tr[0])))
    agg=collections.defaultdict(lambda:[0,0])
    for r in rows:
        d=int(r['End_Timestamp'])-int(r['Start_Timestamp']); k=short(r['Kernel_Name']); agg[k][0]+=1; agg[k][1]+=d
    with open(OUT+'/'+R+'_kernel_trace_summary.csv','w') as fh:
        fh.write('kernel,calls,total_ns,avg_ns\n')
        for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]): fh.write(f'"{k}",{v[0]},{v[1]},{v[1]/v[0]:.1f}\n')
    print('trace dispatches',len(rows))
res={}
for tag,cn in (('pmc_fetch','FETCH_SIZE'),('pmc_write','WRITE_SIZE')):
    fs=glob.glob(OUT+f'/{tag}/**/*counter_collection.csv', recursive=True)
    if not fs: print('no',tag); continue
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(fs[0])):
        if r.get('Counter_Name')!=cn: continue
        k=short(r['Kernel_Name'])
        if 'at::' in k or 'rocclr' in k: continue
        agg[k][0]+=1; agg[k][1]+=float(r['Counter_Value'])
    res[cn]={k:{'calls':v[0],'avg_per_launch':v[1]/v[0]} for k,v in agg.items()}
json.dump(res, open(OUT+'/'+R+'_pmc_raw.json','w'), indent=1)
for cn,d in res.items():
    for k,v in d.items(): print(cn,k,v)
# fabric traffic per launch: FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide
# coalesced reads (MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B), hence 2 x FETCH + WRITE
F, Wr = res.get('FETCH_SIZE', {}), res.get('WRITE_SIZE', {})
SHAPES = {'wgrad_dense_kernel<2, 5, true, 2, 0>': ('conv_bwd_weight[B=2,H=96,W=160,Cin=72,Cout=18,KH=1,stride=1,tf_mode=3,bnb=1]', 13276224),
          # HRFuser-B fc1 forward 78 -> 312 (x + rowstat + y), fc3 data gradient (dy, yraw, xraw, dx), fc1 weight gradient (dy, yraw, x)
          'lin2_fwd_kernel<Tile<2, 10, 2>, 4>': ('conv_fwd[B=2,H=96,W=160,Cin=78,Cout=312,KH=1,stride=1,tf_mode=4]', 30720 * (78 + 2 + 312) * 4),
          'lin2_bwd_data_kernel<Tile<2, 10, 2>, true>': ('conv_bwd_data[B=2,H=96,W=160,Cin=312,Cout=78,KH=1,stride=1,epi=1,accumulate=0,bnb=1]', 30720 * (78 + 78 + 312 + 312) * 4),
          'wgrad_dense_kernel<5, 5, true, 0, 0>': ('conv_bwd_weight[B=2,H=96,W=160,Cin=78,Cout=312,KH=1,stride=1,tf_mode=4,bnb=1]', 30720 * (312 + 312 + 78 + 2) * 4),
          # round 6: the fused attention block (algorithmic bytes of hrfuser_amd.profiling.work_model), the packed-weight 3x3 engine
          # (x + y + weights; dy + yraw + xraw + dx + weights) and the LDS-staged 3x3 weight gradient (dy + yraw + x + dw)
          'attn_block_bwd_kernel<18, 1, 4, true, false, false>': ('attn_block_bwd[B=2,H=96,W=160,C=18,heads=1,cross=0,w1=1]', 39414416),
          'attn_block_fwd_kernel<18, 1>': ('attn_block_fwd[B=2,H=96,W=160,C=18,heads=1,cross=0,w1=1]', 13281408),
          'conv3x_kernel<0, 2>': ('conv_fwd_packed[B=2,H=96,W=160,Cin=64,Cout=64,KH=3,stride=1,tf_mode=2]', (30720 * 128 + 36864) * 4),
          'conv3x_kernel<1, 2>': ('conv_bwd_data_packed[B=2,H=96,W=160,Cin=64,Cout=64,KH=3,stride=1,epi=1,accumulate=0,bnb=1]', (30720 * 256 + 36864) * 4),
          'conv3x_kernel<2, 2>': ('conv_bwd_data_packed[B=2,H=192,W=320,Cin=64,Cout=64,KH=3,stride=2,epi=1,accumulate=0,bnb=1]', (30720 * 128 + 122880 * 128 + 36864) * 4),
          'wgrad3x_kernel<1, false>': ('conv_bwd_weight_s[B=2,H=96,W=160,Cin=64,Cout=64,KH=3,stride=1,tf_mode=2,bnb=1]', (30720 * 192 + 36864) * 4)}
traffic = {'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over tools/pmc_kernels.py (each hot kernel '
                     'launched eagerly on its branch-0 shape, 2x96x160); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch',
           'kernels': {}, 'shapes': {}}
for k in sorted(set(F) & set(Wr)):
    b = (2 * F[k]['avg_per_launch'] + Wr[k]['avg_per_launch']) * 1024
    traffic['kernels'][k] = {'bytes_per_launch': round(b), 'fetch_kb': F[k]['avg_per_launch'], 'write_kb': Wr[k]['avg_per_launch']}
    if k in SHAPES:
        tag, alg = SHAPES[k]
        traffic['shapes'][tag] = {'bytes_per_launch': round(b), 'algorithmic_bytes': alg, 'ratio': round(b / alg, 3), 'kernel': k}
json.dump(traffic, open(OUT+'/'+R+'_hbm_traffic.json','w'), indent=1)
for d in ('trace','pmc_fetch','pmc_write'): shutil.rmtree(OUT+'/'+d, ignore_errors=True)
PY
tail -1 $OUT/bench_under_rocprof.log | cut -c1-300
