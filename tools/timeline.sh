cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections,re
OUT='gpurun_out/tl'
def short(n):
    n=n.replace('(anonymous namespace)::',''); n=re.sub(r'\(.*','',n); return n.replace('void ','')
tr=glob.glob(OUT+'/trace/**/*kernel_trace.csv', recursive=True)
rows=list(csv.DictReader(open(tr[0])))
print('cols',list(rows[0].keys()))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),short(r['Kernel_Name']),r.get('Queue_Id','?'),r.get('Stream_Id','?')) for r in rows]
ev.sort()
# step boundaries: adamw_kernel
ad=[i for i,e in enumerate(ev) if e[2].startswith('adamw_kernel')]
print('adamw idx',ad[-4:], 'n',len(ev))
lo=ad[-2]+1; hi=ad[-1]+1
st=ev[lo:hi]
t0=min(e[0] for e in st); t1=max(e[1] for e in st)
print('last step: kernels',len(st),'span ms',(t1-t0)/1e6,'busy sum ms',sum(e[1]-e[0] for e in st)/1e6)
# concurrency
pts=[]
for s,e,_,_,_ in st: pts.append((s,1)); pts.append((e,-1))
pts.sort(); cur=0; last=t0; hist=collections.Counter()
for t,d in pts:
    hist[cur]+=t-last; last=t; cur+=d
tot=sum(hist.values())
print('concurrency histogram (frac of span):', {k:round(v/tot,3) for k,v in sorted(hist.items())})
# per queue busy
q=collections.defaultdict(int)
for s,e,_,qi,si in st: q[(qi,si)]+=e-s
print('per queue/stream busy ms', {k:round(v/1e6,2) for k,v in q.items()})
# coarse timeline: 1 ms buckets: top kernels
nb=int((t1-t0)/1e6)+1
for b in range(nb):
    a0=t0+b*1e6; a1=a0+1e6
    agg=collections.Counter(); cnt=0
    for s,e,k,_,_ in st:
        ov=min(e,a1)-max(s,a0)
        if ov>0: agg[k.split('<')[0]]+=ov; cnt+=1
    print(f'[{b:2d} ms] n={cnt:4d} busy={sum(agg.values())/1e6:5.2f}  '+', '.join(f'{k}:{v/1e3:.0f}' for k,v in agg.most_common(6)))
# idle gaps (no kernel running)
gaps=hist.get(0,0)
print('idle (no kernel running) ms', gaps/1e6)
# top kernels in step
agg=collections.defaultdict(lambda:[0,0])
for s,e,k,_,_ in st: agg[k][0]+=1; agg[k][1]+=e-s
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:25]: print(f'{k[:70]:70s} n={v[0]:4d} avg={v[1]/v[0]/1e3:7.1f}us tot={v[1]/1e6:6.2f}ms')
PY
