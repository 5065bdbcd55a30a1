"""Per-stage fabric traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/pmc_stages.py:
    python tools/pmc_stages_report.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <stdout of pmc_stages.py> [<sq csv>]
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md).
Round 6: every backward segment is split into its DATA-gradient launches and its WEIGHT-gradient launches (by kernel name: the eager
step issues the weight gradients inline), and an optional third pass (SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE) gives the matrix-pipe
busy fraction per stage: sum of MFMA busy cycles / 1024 SIMDs over sum of (GRBM_GUI_ACTIVE / 8 XCDs) of the segment's launches."""
import csv
import json
import re
import sys

NAMES = None            # [(direction, stage)] of every stamp of the step, from the JSON line tools/pmc_stages.py printed


def short(n):
    n = n.replace('(anonymous namespace)::', '')
    return re.sub(r'\(.*', '', n).replace('void ', '')


WGRAD = ('wgrad_', 'dw_bwd_wgt', 'conv_bwd_wgt', 'fold_', 'rpb_grad', 'wgrad3')     # launches of the weight-gradient work


def is_wgrad(kernel):
    return any(kernel.startswith(p) for p in WGRAD)


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
        out[label] = {'launches': len(seg), 'kb': sum(v for _, v in seg), 'kb_wgrad': sum(v for kn, v in seg if is_wgrad(kn)),
                      'launches_wgrad': sum(1 for kn, _ in seg if is_wgrad(kn))}
    return out


def main(fetch, write, out, stdout_file, sq=None):
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
        bw = (2 * F[k]['kb_wgrad'] + Wr.get(k, {'kb_wgrad': 0})['kb_wgrad']) * 1024
        res['stages'][k] = {'launches': F[k]['launches'], 'fetch_kb': round(F[k]['kb'], 1), 'write_kb': round(Wr.get(k, {'kb': 0})['kb'], 1),
                            'MB': round(b / 1e6, 1), 'wgrad_launches': F[k]['launches_wgrad'], 'wgrad_MB': round(bw / 1e6, 1),
                            'data_MB': round((b - bw) / 1e6, 1)}
        print(f'{k:30s} {F[k]["launches"]:5d} launches  {b / 1e6:9.1f} MB  (weight gradients: {F[k]["launches_wgrad"]} launches, {bw / 1e6:.1f} MB)')
    if sq:
        M, G = per_stage(sq, 'SQ_VALU_MFMA_BUSY_CYCLES'), per_stage(sq, 'GRBM_GUI_ACTIVE')
        res['method'] += ('; mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over sum GRBM_GUI_ACTIVE / 8 XCDs of the launches of the '
                          'segment (third pass; every launch runs alone on the chip in that step), _data: without the weight-gradient launches')
        for k in res['stages']:
            if k in M and k in G and G[k]['kb'] > 0:
                res['stages'][k]['mfma_busy'] = round(M[k]['kb'] / 1024.0 / (G[k]['kb'] / 8.0), 4)
                gd, md = G[k]['kb'] - G[k]['kb_wgrad'], M[k]['kb'] - M[k]['kb_wgrad']
                if gd > 0:
                    res['stages'][k]['mfma_busy_data'] = round(md / 1024.0 / (gd / 8.0), 4)
                if G[k]['kb_wgrad'] > 0:
                    res['stages'][k]['mfma_busy_wgrad'] = round(M[k]['kb_wgrad'] / 1024.0 / (G[k]['kb_wgrad'] / 8.0), 4)
                res['stages'][k]['active_us_single_stream'] = round(G[k]['kb'] / 8.0 / 2.4e3, 1)       # at 2.4 GHz: an upper bound of the rate
    res['step_MB'] = round(tot / 1e6, 1)
    print('step', res['step_MB'], 'MB')
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:6])
