"""Timeline analysis of ONE captured training step from a rocprofv3 kernel trace (csv):

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline ...
    python tools/step_timeline.py out/**/t_kernel_trace.csv

Steps are delimited by `adamw_kernel` (last kernel of a step).  Prints, for a steady-state step: wall time, union busy time,
time-weighted concurrency histogram, per-family busy time and the time each family spends running ALONE (concurrency 1 =
the serial part of the dependency chain), and the idle gaps."""
import collections
import csv
import json
import re
import sys


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)
    return n.replace('void ', '')


def main(path, which=-2, out=None):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])))
    rows.sort()
    ends = [e for s, e, k in rows if k.startswith('adamw_kernel')]
    if len(ends) < 3:
        print('not enough steps in the trace')
        return
    t0, t1 = ends[which - 1], ends[which]
    step = [(s, e, k) for s, e, k in rows if s >= t0 and e <= t1 + 1]
    wall = (t1 - t0) / 1e3
    ev = []
    for i, (s, e, k) in enumerate(step):
        ev.append((s, 1, i))
        ev.append((e, -1, i))
    ev.sort()
    active = set()
    hist = collections.Counter()
    alone = collections.Counter()
    busy = collections.Counter()
    last = t0
    gaps = []
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            c = len(active)
            hist[min(c, 8)] += dt
            if c == 0:
                gaps.append(dt)
            if c == 1:
                alone[re.sub(r'<.*', '', step[next(iter(active))][2])] += dt
            for j in active:
                busy[re.sub(r'<.*', '', step[j][2])] += dt / c
        last = t
        if d == 1:
            active.add(i)
        else:
            active.discard(i)
    fam = collections.Counter()
    cnt = collections.Counter()
    for s, e, k in step:
        f = re.sub(r'<.*', '', k)
        fam[f] += e - s
        cnt[f] += 1
    res = {'wall_us': wall, 'kernels': len(step), 'sum_kernel_us': sum(fam.values()) / 1e3,
           'busy_union_us': (t1 - t0 - hist[0]) / 1e3, 'idle_us': hist[0] / 1e3, 'n_gaps': len(gaps),
           'concurrency_hist_us': {str(k): v / 1e3 for k, v in sorted(hist.items())},
           'families': {f: {'launches': cnt[f], 'in_step_us': fam[f] / 1e3, 'wall_share_us': busy[f] / 1e3, 'alone_us': alone[f] / 1e3}
                        for f, _ in busy.most_common(40)}}
    print(f"step wall {wall:.0f} us, {len(step)} kernels, sum of durations {res['sum_kernel_us']:.0f} us, idle {res['idle_us']:.0f} us in {len(gaps)} gaps")
    print('concurrency histogram (us):', {k: round(v) for k, v in res['concurrency_hist_us'].items()})
    print(f"{'family':34s} {'n':>5s} {'in-step us':>10s} {'wall share':>10s} {'alone us':>9s}")
    for f, v in res['families'].items():
        print(f"{f:34s} {v['launches']:5d} {v['in_step_us']:10.0f} {v['wall_share_us']:10.0f} {v['alone_us']:9.0f}")
    # phases: first backward kernel / deferred phase markers
    names = [k for _, _, k in step]
    def first(pred):
        for s, e, k in step:
            if pred(k):
                return (s - t0) / 1e3
        return None
    res['t_first_bwd_us'] = first(lambda k: 'bwd' in k or 'wgrad' in k)
    res['t_first_wgrad_us'] = first(lambda k: k.startswith('wgrad_dense'))
    res['t_fold_slots_us'] = first(lambda k: k.startswith('fold_slots'))
    print('first backward kernel at', res['t_first_bwd_us'], 'first wgrad_dense at', res['t_first_wgrad_us'], 'fold_slots at', res['t_fold_slots_us'])
    # coarse time series: per 500 us bin the mean concurrency and the family with the largest share
    nb = int((t1 - t0) // 500000) + 1
    binc = [0.0] * nb
    binf = [collections.Counter() for _ in range(nb)]
    for s, e, k in step:
        f = re.sub(r'<.*', '', k)
        b0, b1 = int((s - t0) // 500000), int((e - t0) // 500000)
        for b in range(b0, min(b1, nb - 1) + 1):
            lo, hi = max(s, t0 + b * 500000), min(e, t0 + (b + 1) * 500000)
            if hi > lo:
                binc[b] += (hi - lo) / 500000
                binf[b][f] += (hi - lo)
    res['bins_500us'] = [{'t_ms': b * 0.5, 'concurrency': round(binc[b], 2), 'top': [f"{f}:{v / 1e3:.0f}" for f, v in binf[b].most_common(3)]} for b in range(nb)]
    for r in res['bins_500us']:
        print(f"{r['t_ms']:5.1f} ms  conc {r['concurrency']:4.2f}  {' '.join(r['top'])}")
    if out:
        json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(sys.argv[1], out=sys.argv[2] if len(sys.argv) > 2 else None)
