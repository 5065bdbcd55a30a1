"""Per-stage fabric traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_stages.py:
    python tools/pmc_stages_report.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <stdout of pmc_stages.py>
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md)."""
import csv
import json
import re
import sys

NAMES = None            # [(direction, stage)] of every stamp of the step, from the JSON line tools/pmc_stages.py printed


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*', '', n).replace('void ', '')


def per_stage(path, counter):
    rows = {}
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') != counter:
            continue
        d = int(r['Dispatch_Id'])
        rows[d] = (short(r['Kernel_Name']), rows.get(d, ('', 0.0))[1] + float(r['Counter_Value']))
    seq = [rows[d] for d in sorted(rows)]
    stamps = [i for i, (k, _) in enumerate(seq) if k.startswith('stamp_kernel')]
    names = NAMES
    per = len(names)
    if len(stamps) < per:
        raise SystemExit(f'{path}: {len(stamps)} stamp kernels, expected {per}')
    last = stamps[-per:]
    out = {}
    for k in range(1, len(last)):
        d, n = names[k]
        label = f'{d} {n}' if d != 'bwd' or n == 'weight_gradients' else f'bwd {names[k - 1][1]}'
        seg = seq[last[k - 1] + 1:last[k]]
        out[label] = {'launches': len(seg), 'kb': sum(v for _, v in seg)}
    return out


def main(fetch, write, out, stdout_file):
    global NAMES
    for line in reversed(open(stdout_file).read().splitlines()):
        if line.startswith('[['):
            NAMES = [(d, n) for d, n, _ in json.loads(line)]
            break
    if NAMES is None:
        raise SystemExit(f'no stamp list in {stdout_file}')
    F, Wr = per_stage(fetch, 'FETCH_SIZE'), per_stage(write, 'WRITE_SIZE')
    res = {'method': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes over ONE eager, single-stream training step '
                     '(tools/pmc_stages.py), dispatches cut at the stage-stamp kernels; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024',
           'stages': {}}
    tot = 0
    for k in F:
        b = (2 * F[k]['kb'] + Wr.get(k, {'kb': 0})['kb']) * 1024
        tot += b
        res['stages'][k] = {'launches': F[k]['launches'], 'fetch_kb': round(F[k]['kb'], 1), 'write_kb': round(Wr.get(k, {'kb': 0})['kb'], 1),
                            'MB': round(b / 1e6, 1)}
        print(f'{k:30s} {F[k]["launches"]:5d} launches  {b / 1e6:9.1f} MB')
    res['step_MB'] = round(tot / 1e6, 1)
    print('step', res['step_MB'], 'MB')
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:5])
