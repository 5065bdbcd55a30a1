"""Cut a rocprofv3 kernel trace (csv) of tools/stage_trace.py at the stamp kernels and report per stage:
    python tools/stage_trace_report.py <kernel_trace.csv> <stdout of stage_trace.py> [stage-name-substring ...]
For each stage of the LAST replayed step: wall time, number of kernels, sum of kernel durations, and (for the stages named on
the command line) every kernel with queue id, start offset (us) and duration - the dependency chain is readable from it."""
import csv
import re
import sys



def short(n):
    n = n.replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)
    return n.replace('void ', '')


def stamp_names(outfile):
    """(direction, stage) of every stamp of one step, in issue order: the JSON line tools/stage_trace.py printed."""
    import json
    for line in reversed(open(outfile).read().splitlines()):
        if line.startswith('[['):
            return [(d, n) for d, n, _ in json.loads(line)]
    raise SystemExit(f'no stamp list in {outfile}')


def main(path, outfile, show):
    names = stamp_names(outfile)
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '?'), r.get('Stream_Id', '?')))
    rows.sort()
    stamps = [i for i, r in enumerate(rows) if r[2].startswith('stamp_kernel')]
    per = len(names)                               # fwd marks + bwd marks + weight_gradients + step_end
    if len(stamps) < per:
        print('not enough stamps', len(stamps))
        return
    last = stamps[-per:]
    t0 = rows[last[0]][0]
    print(f'step (first to last stamp): {(rows[last[-1]][0] - t0) / 1e3:.1f} us')
    for k in range(1, len(last)):
        a, b = rows[last[k - 1]][1], rows[last[k]][0]
        d, n = names[k]
        label = f'{d} {n}' if d != 'bwd' or n == 'weight_gradients' else f'bwd {names[k - 1][1]}'
        ks = [r for r in rows if r[0] >= a and r[1] <= b and not r[2].startswith('stamp_kernel')]
        busy = sum(r[1] - r[0] for r in ks)
        queues = sorted(set(r[3] for r in ks))
        print(f'{label:28s} wall {(b - a) / 1e3:8.1f} us  kernels {len(ks):4d}  sum {busy / 1e3:8.1f} us  queues {len(queues)}')
        if any(s in label for s in show):
            for r in ks:
                print(f'      q{r[3]:>3s} s{r[4]:>3s} +{(r[0] - a) / 1e3:8.1f} {(r[1] - r[0]) / 1e3:7.1f} us  {r[2][:70]}')


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
