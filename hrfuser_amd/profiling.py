"""In-process measurement helpers for bench.py: per-launch HIP-event timing of every C-ABI call,
an algorithmic work model (FLOPs / compulsory bytes) per launch, and the roofline of the
dominant kernel.  Pure measurement plumbing - nothing here computes results."""
import ctypes
import math
import os
import time

import torch

from . import _lib
from .runtime import gc_paused as _gc_paused


def _pick_nt(C):
    T = (C + 15) // 16
    best, cost = 2, 1 << 30
    for nt in (2, 3, 4):
        c = ((T + nt - 1) // nt) * nt
        if c <= cost:
            cost, best = c, nt
    return best


def _out_hw(H, W, KH, s):
    p = KH // 2
    return (H + 2 * p - KH) // s + 1, (W + 2 * p - KH) // s + 1


def work_model(name, a):
    """-> (kernel key as rocprof names it, algorithmic FLOPs, compulsory HBM bytes) of one launch."""
    f4 = 4.0
    if name in ('hrf_conv_fwd', 'hrf_conv_fwd_split', 'hrf_conv_fwd_packed'):       # (_split: K slices + fixed-order reduce + moments, priced as ONE call)
        Ho, Wo = _out_hw(a['H'], a['W'], a['KH'], a['stride'])
        M, K = a['B'] * Ho * Wo, a['KH'] ** 2 * a['Cin']
        by = f4 * (a['B'] * a['H'] * a['W'] * a['Cin'] + a['Cout'] * K + M * a['Cout'] * (1 + (a['res'] is not None) + (a['res2'] is not None)))
        if name == 'hrf_conv_fwd_packed':
            return 'conv3x_kernel<0, 2>', 2.0 * M * a['Cout'] * K, by
        if a['KH'] == 1 and a['stride'] == 1 and a['sC'] == 1 and a['Cin'] >= 4 and a['Cout'] >= 4:
            return 'lin_fwd_kernel', 2.0 * M * a['Cout'] * K, by
        return f"conv_fwd_kernel<{_pick_nt(a['Cout'])},{a['KH']},{a['tf_mode']}>", 2.0 * M * a['Cout'] * K, by
    if name in ('hrf_conv_bwd_data', 'hrf_conv_bwd_data_packed'):
        Ho, Wo = _out_hw(a['H'], a['W'], a['KH'], a['stride'])
        M, K = a['B'] * a['H'] * a['W'], a['KH'] ** 2 * a['Cout']
        by = f4 * (a['B'] * Ho * Wo * a['Cout'] * (2 if a['cA'] is not None else 1) + a['Cin'] * K
                   + M * a['Cin'] * (2 if a['epi'] else 1 + bool(a['accumulate'])))
        # algorithmic FLOPs: every (output pixel of the convolution, tap) pair once - a stride-2 problem has a quarter of the
        # pairs per INPUT pixel (rounds 1-5 priced 9 taps per input pixel there: 4x too many)
        fl = 2.0 * a['B'] * Ho * Wo * a['Cin'] * K
        if name == 'hrf_conv_bwd_data_packed':
            return f"conv3x_kernel<{a['stride']}, 2>", fl, by
        if a['KH'] == 1 and a['stride'] == 1 and a['sC'] == 1 and a['Cin'] >= 4 and a['Cout'] >= 4:
            return 'lin_bwd_data_kernel', fl, by
        return f"conv_bwd_data_kernel<{_pick_nt(a['Cin'])},{a['KH']},{int(a['cA'] is not None)}>", fl, by
    if name == 'hrf_conv_bwd_weight_s':
        Ho, Wo = _out_hw(a['H'], a['W'], a['KH'], a['stride'])
        Mp, Np = a['B'] * Ho * Wo, a['KH'] ** 2 * a['Cin']
        by = f4 * (Mp * a['Cout'] * (2 if a['cA'] is not None else 1) + a['B'] * a['H'] * a['W'] * a['Cin'] + a['Cout'] * Np)
        return f"wgrad3x_kernel<{a['stride']}>", 2.0 * Mp * a['Cout'] * Np, by
    if name == 'hrf_conv_bwd_weight':
        Ho, Wo = _out_hw(a['H'], a['W'], a['KH'], a['stride'])
        Mp, Np = a['B'] * Ho * Wo, a['KH'] ** 2 * a['Cin']
        by = f4 * (Mp * a['Cout'] * (2 if a['cA'] is not None else 1) + a['B'] * a['H'] * a['W'] * a['Cin'] + a['Cout'] * Np)
        dense = a['sC'] == 1 and (a['KH'] == 3 or a['stride'] == 1)
        return ('wgrad_dense_kernel' if dense else 'conv_bwd_wgt_kernel'), 2.0 * Mp * a['Cout'] * Np, by
    if name in ('hrf_dwconv_fwd', 'hrf_dwconv_bwd_weight'):
        Ho, Wo = _out_hw(a['H'], a['W'], 3, a['stride'])
        n_in, n_out = a['B'] * a['H'] * a['W'] * a['C'], a['B'] * Ho * Wo * a['C']
        if name == 'hrf_dwconv_fwd':
            return f"dw_fwd_kernel<{a['stride']}>", 18.0 * n_out, f4 * (n_in + n_out)
        return f"dw_bwd_wgt_kernel<{a['stride']}>", 20.0 * n_out, f4 * (n_in + n_out * (2 if a['cA'] is not None else 1))
    if name == 'hrf_dwconv_bwd_data':
        Ho, Wo = _out_hw(a['H'], a['W'], 3, a['stride'])
        n_in, n_out = a['B'] * a['H'] * a['W'] * a['C'], a['B'] * Ho * Wo * a['C']
        return (f"dw_bwd_data_kernel<{a['stride']}>", 18.0 * n_in,
                f4 * (n_out * (2 if a['cA'] is not None else 1) + n_in * (2 if a['epi'] else 1)))
    if name == 'hrf_dwconv_bwd_data_weight':               # stride 1, epi 1: data + weight gradient in one pass
        n = a['B'] * a['H'] * a['W'] * a['C']
        return 'dw_bwd_data_kernel<1,wg>', 38.0 * n, f4 * (n * (2 if a['cA'] is not None else 1) + 2 * n)
    if name in ('hrf_window_attn_fwd', 'hrf_window_attn_bwd'):
        nwin = a['B'] * math.ceil(a['H'] / 7) * math.ceil(a['W'] / 7)
        D = a['C'] // a['heads']
        P = a['B'] * a['H'] * a['W']
        unit = 2.0 * 49 * 49 * D * nwin * a['heads']          # one 49x49xD contraction per (window, head)
        if name == 'hrf_window_attn_fwd':
            return f"attn_fwd_kernel<{D}>", 2 * unit, f4 * P * a['C'] * 4
        return f"attn_bwd_kernel<{D}>", 5 * unit, f4 * P * a['C'] * 7
    if name in ('hrf_attn_block_fwd', 'hrf_attn_block_bwd'):
        C, heads = a['C'], a['heads']
        rows = float(a['B']) * a['H'] * a['W']
        nwin = a['B'] * math.ceil(a['H'] / 7) * math.ceil(a['W'] / 7)
        cross, ffn = bool(a['cross']), bool(a['w1'])
        attn = 2 * 2.0 * 49 * 49 * (C // heads) * nwin * heads               # q k^T and p v on the 49-token windows
        fwd = 2.0 * rows * C * C * 4 + attn + (2.0 * rows * C * 4 * C if ffn else 0.0)
        wts = 4.0 * (4 * C * C + (4 * C * C if ffn else 0))
        if name == 'hrf_attn_block_fwd':
            by = f4 * rows * C * (2 + 2 * cross + (4 if ffn else 0)) + wts
            return f'attn_block_fwd_kernel<{C}, {heads}>', fwd, by
        slot = 4 * C * C + (4 * C * C if ffn else 0)
        by = f4 * (rows * C * (4 + 3 * cross + (8 if ffn else 0)) + nwin * (slot + heads * 49 * 49)) + wts
        return f'attn_block_bwd_kernel<{C}, {heads}>', 3.0 * fwd, by
    if name == 'hrf_rpb_grad':
        return 'rpb_grad_kernel', 0.0, f4 * a['nwin'] * a['heads'] * 49 * 49
    rc = None
    if 'rows' in a and 'C' in a:
        rc = float(a['rows']) * a['C']
    if name == 'hrf_affine_act_res':
        return 'affine_act_res_kernel', 4 * rc, f4 * rc * (2 + (a['res'] is not None) + (a['y2'] is not None))
    if name == 'hrf_act_bwd':
        n = 2 + (a['out'] is not None) + (a['y1'] is not None and a['st1'] is not None or a['mode'] == 1) \
            + (a['y2'] is not None) + (a['y3'] is not None)
        return 'act_bwd_kernel', 6 * rc, f4 * rc * n
    if name == 'hrf_scale_add':
        return 'scale_add_kernel', 3 * rc, f4 * rc * (2 + (a['mask'] is not None) + (a['res'] is not None) + (a['res2'] is not None))
    if name == 'hrf_ln_stats':
        return 'ln_stats_kernel', 4 * rc, f4 * rc
    if name == 'hrf_ln_bwd':
        return 'ln_bwd_kernel', 12 * rc, f4 * rc * (3 + bool(a['accumulate']))
    if name == 'hrf_fuse_sum':
        n = float(a['B']) * a['H'] * a['W'] * a['C']
        nt = sum(1 for k in range(4) if a[f'type{k}'])
        return 'fuse_sum_kernel', 4 * n * nt, f4 * n * (1 + nt)
    if name == 'hrf_bilinear_up_bwd':
        n = float(a['B']) * a['H'] * a['W'] * a['C']
        return 'bilinear_up_bwd_kernel', 2 * n, f4 * (n + 3.0 * a['B'] * a['Hs'] * a['Ws'] * a['C'])
    if name == 'hrf_adamw':
        return 'adamw_kernel', 12.0 * a['n'], f4 * a['n'] * 7
    return name.replace('hrf_', '') + '_kernel', 0.0, 0.0


class ProfLib:
    """Wraps the loaded library: brackets every C-ABI launch with HIP events recorded on the launch
    stream (torch's current stream - the stream the kernels are enqueued on)."""

    def __init__(self, lib, timing=True):
        self._lib = lib
        self.require_cuda = lib.require_cuda
        self.protos = lib.protos
        self.records = []
        self.timing = timing

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        names = [n for _, n in self._lib.protos[name]]

        def describe(args):
            d = dict(zip(names, args))
            p = d.get('p')
            if isinstance(p, ctypes.Structure):           # struct-argument entry points: the shape lives in the struct
                for f, _ in p._fields_:
                    v = getattr(p, f)
                    if f in ('B', 'H', 'W', 'C', 'heads', 'hidden'):
                        d[f] = v
                if hasattr(p, 'xq'):                       # (hrf_attn_block_t; hrf_ffn_eval_t has neither)
                    d['cross'] = int(p.xq != p.xkv)
                    d['w1'] = int(bool(p.w1))
            return d

        if name in _lib._RAW_RETURN or name in ('hrf_wgrad_group_begin', 'hrf_wgrad_group_end', 'hrf_debug_knob', 'hrf_group_begin',
                                                'hrf_group_end'):
            return fn                                     # queries / plumbing: not launches, and their value matters

        def call(*args):
            if not self.timing:
                fn(*args)
                self.records.append((name, describe(args), args))
                return
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*args)
            e1.record()
            self.records.append((name, dict(zip(names, args)), e0, e1))
        return call


def _signature(name, a):
    return (name,) + tuple(v for v in a.values() if isinstance(v, (int, float)) and not isinstance(v, bool))


def shape_tag(name, a):
    """Human-readable signature of one launch: entry point + the shape/mode integers that select the
    kernel variant (pointer arguments reduced to present/absent).  Also the key of profiles/*traffic*.json."""
    keep = ('B', 'H', 'W', 'C', 'Cin', 'Cout', 'KH', 'stride', 'rows', 'heads', 'tf_mode', 'epi', 'mode', 'accumulate', 'cross', 'w1',
            'nwin')
    parts = [f'{k}={a[k]}' for k in keep if k in a and isinstance(a[k], (int, float))]
    if 'cA' in a:
        parts.append(f"bnb={int(a['cA'] is not None)}")
    return name.replace('hrf_', '') + '[' + ','.join(parts) + ']'


def _graph_time(fn, reps=20, replays=5):
    """GPU-side average duration of one launch: `reps` back-to-back launches captured into a hipGraph
    on the launch stream, replayed and bracketed by HIP events (no host launch overhead inside)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with _gc_paused(), torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * replays)


def profile_step(trainer, x, mods, cots, steps=1):
    """Per-kernel table of one training step -> {key: [launches, seconds, flops, bytes, steps]}.

    Pass 1 records every C-ABI call of an eager step (arguments only).  Pass 2 re-issues one
    representative call per distinct (entry point, shape) signature inside a captured hipGraph and
    times it with HIP events on the launch stream, so the per-launch duration is measured on the GPU
    (an event pair around a single eager launch mostly measures the ~10 us host/ctypes gap).
    Re-issuing accumulating kernels perturbs gradients: call this after the timed region only."""
    real = _lib.lib
    base = real()
    prof = ProfLib(base, timing=False)
    trainer.step(x, mods, cots)                    # eager warm-up (allocator, caches)
    torch.cuda.synchronize()
    _lib.lib = lambda: prof
    saved = {k: os.environ.get(k) for k in ('HRF_LANES', 'HRF_GROUP')}
    os.environ['HRF_LANES'] = '0'                  # record on one stream; replay order is irrelevant
    os.environ['HRF_GROUP'] = '0'                  # one record per C-ABI call: the table prices single-problem launches
    try:
        for _ in range(steps):
            trainer.step(x, mods, cots)
        torch.cuda.synchronize()
    finally:
        _lib.lib = real
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    sigs = {}
    for name, a, args in prof.records:
        sg = _signature(name, a)
        ent = sigs.get(sg)
        if ent is None:
            sigs[sg] = [1, name, a, args]
        else:
            ent[0] += 1
    table = {}
    sig_rows = []
    for sg, (cnt, name, a, args) in sigs.items():
        fn = getattr(base, name)
        sptr = [None]

        def call(fn=fn, args=args):
            fn(*args[:-1], _lib.stream_ptr())
        try:
            dt = _graph_time(call)
        except Exception:
            torch.cuda.synchronize()
            continue
        key, fl, by = work_model(name, a)
        sig_rows.append({'shape': shape_tag(name, a), 'kernel': key, 'launches_per_step': cnt // steps,
                         'avg_launch_us': dt * 1e6, 'time_per_step_ms': dt * cnt / steps * 1e3,
                         'flops_per_launch': fl, 'bytes_per_launch': by})
        t = table.setdefault(key, [0, 0.0, 0.0, 0.0])
        t[0] += cnt
        t[1] += dt * cnt
        t[2] += fl * cnt
        t[3] += by * cnt
    for t in table.values():
        t.append(steps)
    sig_rows.sort(key=lambda r: -r['time_per_step_ms'])
    profile_step.last_signatures = sig_rows       # per-(entry point, shape) rows of the same measurement
    return table


# forward work per image and stage (GFLOP, eager-equivalent fp32 MB = leaf-op bytes + attention-matrix bytes),
# grouped by the regions the engine executes one after the other (the camera stem / layer1 and the modality stems / layer_a
# run bundled; stage k of the camera runs beside modality stage b / c).
_B2 = {
    't_nus': {'stem_cam': (1.35, 196), 'layer1_cam': (4.40, 322), 'stem_mod': (2.69, 155), 'layer_a_mod': (8.81, 645), 'transitions': (6.26, 158),
              'fusion_a': (0.53, 191), 'stage2': (0.73, 279), 'stage_b': (0.78, 358), 'fusion_b': (0.74, 225), 'stage3': (3.14, 1003),
              'stage_c': (2.35, 1075), 'fusion_c': (0.94, 242), 'stage4': (2.70, 734)},
    'b_nus': {'stem_cam': (1.35, 196), 'layer1_cam': (4.40, 322), 'stem_mod': (2.69, 155), 'layer_a_mod': (8.81, 645), 'transitions': (34.73, 368),
              'fusion_a': (6.99, 693), 'stage2': (10.31, 1074), 'stage_b': (10.36, 1378), 'fusion_b': (10.49, 815), 'stage3': (62.60, 5160),
              'stage_c': (41.45, 5514), 'fusion_c': (13.89, 876), 'stage4': (42.02, 2841)},
    't_stf': {'stem_cam': (2.62, 458), 'layer1_cam': (8.59, 629), 'stem_mod': (7.45, 449), 'layer_a_mod': (25.76, 1886), 'transitions': (16.36, 424),
              'fusion_a': (1.35, 466), 'stage2': (1.41, 542), 'stage_b': (2.30, 1049), 'fusion_b': (1.91, 552), 'stage3': (6.13, 1955),
              'stage_c': (6.89, 3146), 'fusion_c': (2.41, 594), 'stage4': (5.28, 1434)},
}
_REGIONS = [('stems', ('stem_cam', 'layer1_cam', 'stem_mod', 'layer_a_mod')), ('transitions', ('transitions',)),
            ('fusion_a', ('fusion_a',)), ('stage2+stage_b', ('stage2', 'stage_b')), ('fusion_b', ('fusion_b',)),
            ('stage3+stage_c', ('stage3', 'stage_c')), ('fusion_c', ('fusion_c',)), ('stage4', ('stage4',))]


def _stage_pmc(tag):
    """stage name -> {'MB': ...} of the committed per-stage PMC pass (HRFuser-T nus only), or None"""
    if tag.replace('_bn', '') != 't_nus':
        return None
    path, name = _newest(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles'), 'stage_hbm_traffic.json')
    if path is None:
        return None
    import json
    with open(path) as fh:
        st = json.load(fh).get('stages')
    if st is not None:
        st = dict(st)
        st['_file'] = name
    return st


def stage_table(trainer, x, mods, cots, tag, peak_f, peak_b, replays=10):
    """Per-stage durations of the CAPTURED training step and their roofline fractions (north star: "achieved fraction of
    HBM and MFMA roofline reported per stage").  The step is captured once more with GPU timestamps at the stage boundaries
    (hrf_stamp: readable after a hipGraph replay, unlike HIP events), replayed, and the medians of the stamp differences are
    priced against SURVEY App. B-2's per-stage FLOPs / eager-equivalent bytes.  Forward rows use the stage's forward work;
    backward rows its data-gradient work (= 1x forward), the weight gradients (= 1x forward of ALL stages) run as one
    deferred phase with its own row."""
    net = trainer.net
    work = _B2.get(tag.replace('_bn', ''))
    old_graph, old_outs = trainer.graph, getattr(trainer, '_graph_outs', None)
    st = net.enable_stage_stamps()
    try:
        trainer.capture(x, mods, cots, warmup=1)
        runs = []
        for _ in range(replays):
            trainer.replay()
            torch.cuda.synchronize()
            runs.append(st.read())
    finally:
        net.enable_stage_stamps(False)
        trainer.graph, trainer._graph_outs = old_graph, old_outs
    names = [(d, n) for d, n, _ in runs[0]]
    med = []
    for i in range(len(names)):
        v = sorted(r[i][2] for r in runs)
        med.append(v[len(v) // 2])
    t = {nm: v for nm, v in zip(names, med)}
    B = x.shape[0]
    order = [n for d, n in names if d == 'fwd']
    fwd, bwd = {}, {}
    for i in range(1, len(order)):
        fwd[order[i]] = t[('fwd', order[i])] - t[('fwd', order[i - 1])]
        bwd[order[i]] = t[('bwd', order[i - 1])] - t[('bwd', order[i])]
    turn = t[('bwd', order[-1])] - t[('fwd', order[-1])]           # cotangent hand-over between the passes
    wg = t[('bwd', 'weight_gradients')] - t[('bwd', 'start')]
    end = t.get(('step', 'step_end'))
    rows = []

    def frac(gflop, mb, us):
        if work is None or us <= 0:
            return None, None
        return round(B * gflop * 1e9 / (us * 1e-6) / peak_f, 4), round(B * mb * 1e6 / (us * 1e-6) / peak_b, 4)
    tot_g = tot_mb = 0.0
    regions = list(_REGIONS)
    if 'fusion_a' not in order:
        # the fusion blocks run at the head of the following stage's branch lanes (no boundary of their own, backbone.
        # _fuse_stage): the stamp interval named after the stage contains them
        merged, pend = [], None
        for reg, parts in regions:
            if reg.startswith('fusion_'):
                pend = (reg, parts)
            elif pend is not None and reg.startswith('stage'):
                merged.append((reg, pend[1] + parts, pend[0] + '+' + reg, (pend[0], reg)))
                pend = None
            else:
                merged.append((reg, parts))
        regions = merged
    pmc_keys = {}
    for ent in regions:
        reg, parts = ent[0], ent[1]
        label = ent[2] if len(ent) > 2 else reg
        pmc_keys[label] = ent[3] if len(ent) > 3 else None
        keys = [k for k in order if k == reg or (reg == 'transitions' and k.startswith('transitions_'))]
        if not keys:
            continue
        f_us, b_us = sum(fwd[k] for k in keys), sum(bwd[k] for k in keys)
        g = sum(work[p][0] for p in parts) if work else 0.0
        mb = sum(work[p][1] for p in parts) if work else 0.0
        tot_g, tot_mb = tot_g + g, tot_mb + mb
        ff, fb = frac(g, mb, f_us)
        bf, bb = frac(g, mb, b_us)
        rows.append({'name': label, 'fwd_ms': round(f_us / 1e3, 4), 'bwd_ms': round(b_us / 1e3, 4), 'fwd_gflop': round(B * g, 2),
                     'fwd_eager_MB': round(B * mb, 1), 'fwd_flops_frac': ff, 'fwd_bytes_frac': fb, 'bwd_flops_frac': bf, 'bwd_bytes_frac': bb})
    # MEASURED fabric traffic per stage (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over one eager single-stream step, cut at the same
    # stamps: tools/pmc_stages.py -> profiles/r03_stage_hbm_traffic.json; a separate run - the source is stated)
    pmc = _stage_pmc(tag)
    if pmc is not None:
        for row in rows:
            if pmc_keys.get(row['name']) and all(('fwd ' + k) in pmc for k in pmc_keys[row['name']]):
                keys = list(pmc_keys[row['name']])                # a merged row over a PMC pass that still had the fusion stamps
            else:
                reg = pmc_keys[row['name']][1] if pmc_keys.get(row['name']) else row['name']
                keys = [k for k in (list(order) + ['transitions_a', 'transitions_b', 'transitions_c'])
                        if k == reg or (reg == 'transitions' and k.startswith('transitions_'))]
                keys = list(dict.fromkeys(keys))
            fmb = sum(pmc.get('fwd ' + k, {}).get('MB', 0.0) for k in keys)
            bmb = sum(pmc.get('bwd ' + k, {}).get('MB', 0.0) for k in keys)
            row['fwd_pmc_MB'], row['bwd_pmc_MB_incl_weight_gradients'] = round(fmb, 1), round(bmb, 1)
            row['fwd_pmc_hbm_frac'] = round(fmb * 1e6 / (row['fwd_ms'] * 1e-3) / peak_b, 4) if row['fwd_ms'] else None
            if any('data_MB' in pmc.get('bwd ' + k, {}) for k in keys):      # (round 6 file: weight gradients split out by kernel name)
                dmb = sum(pmc.get('bwd ' + k, {}).get('data_MB', 0.0) for k in keys)
                row['bwd_pmc_MB_data_gradient'] = round(dmb, 1)
                row['bwd_pmc_hbm_frac'] = round(dmb * 1e6 / (row['bwd_ms'] * 1e-3) / peak_b, 4) if row['bwd_ms'] else None
            # matrix-pipe busy fraction of the stage's launches (each alone on the chip in the PMC step): launch-time weighted
            for d, key, out_key in (('fwd ', 'mfma_busy', 'fwd_mfma_busy'), ('bwd ', 'mfma_busy_data', 'bwd_mfma_busy_data_gradient')):
                ws = [(pmc.get(d + k, {}).get(key), pmc.get(d + k, {}).get('active_us_single_stream', 0.0)) for k in keys]
                ws = [(v, w) for v, w in ws if v is not None and w > 0]
                if ws:
                    row[out_key] = round(sum(v * w for v, w in ws) / sum(w for _, w in ws), 4)
    wf, wb = frac(tot_g, tot_mb, wg)
    rows.append({'name': 'weight_gradients (deferred phase, all stages)', 'fwd_ms': None, 'bwd_ms': round(wg / 1e3, 4),
                 'fwd_gflop': round(B * tot_g, 2), 'fwd_eager_MB': round(B * tot_mb, 1), 'bwd_flops_frac': wf, 'bwd_bytes_frac': wb})
    step_pmc = None
    if pmc is not None:
        tot = sum(v.get('MB', 0.0) for k, v in pmc.items() if k != '_file')
        step_ms = (end if end is not None else t[('bwd', 'weight_gradients')]) / 1e3
        step_pmc = {'MB': round(tot, 1), 'hbm_frac': round(tot * 1e6 / (step_ms * 1e-3) / peak_b, 4),
                    'source': f'profiles/{pmc.get("_file")}: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of every dispatch of ONE eager single-stream '
                              'step from separate rocprofv3 --pmc passes (tools/pmc_stages.py), cut at the stage stamps; NOT measured in this '
                              'bench run; backward rows include the weight gradients of the stage (issued inline in that step)'}
    out = {'stages': rows, 'step_pmc_traffic': step_pmc, 'turnaround_ms': round(turn / 1e3, 4),
           'grad_exchange_adamw_ms': round((end - t[('bwd', 'weight_gradients')]) / 1e3, 4) if end is not None else None,
           'step_ms_with_stamps': round((end if end is not None else t[('bwd', 'weight_gradients')]) / 1e3, 4),
           'timer': f'hrf_stamp (100 MHz GPU counter) at the stage boundaries of the captured step, median of {replays} replays; '
                    'fractions = SURVEY App. B-2 forward GFLOP / eager-equivalent MB of the stage x images over the measured time, '
                    'against 157.3 TFLOP/s and 8 TB/s; backward rows price the data-gradient work (1x forward), the deferred '
                    'weight-gradient phase the weight-gradient work of all stages (1x forward)'}
    return out


def merged_launch_counts(trainer, x, mods, cots):
    """One eager step with the multi-problem launches as configured: (launches issued by hrf_group_end, C-ABI calls'
    launches they carried) - the step's launch count is its call count minus (carried - issued)."""
    L = _lib.lib()
    if not hasattr(L, 'hrf_group_count'):
        return 0, 0
    torch.cuda.synchronize()
    c0 = [L.hrf_group_count(k) for k in range(2)]
    trainer.step(x, mods, cots)
    torch.cuda.synchronize()
    c1 = [L.hrf_group_count(k) for k in range(2)]
    return c1[0] - c0[0], c1[1] - c0[1]


def _row(key, t, peak_f, peak_b):
    n, sec, fl, by, steps = t
    bound = 'mfma' if (fl / peak_f) > (by / peak_b) else 'hbm'
    if bound == 'mfma':
        ach, peak, unit = fl / sec / 1e12, peak_f / 1e12, 'TFLOP/s'
    else:
        ach, peak, unit = by / sec / 1e9, peak_b / 1e9, 'GB/s'
    return {'kernel': key, 'bound': bound, 'achieved': round(ach, 3), 'peak': peak, 'unit': unit,
            'frac': round(ach / peak, 4), 'traffic': None, 'launches_per_step': n // steps,
            'avg_launch_us': round(sec / n * 1e6, 2), 'time_per_step_ms': round(sec / steps * 1e3, 4),
            'flops_per_launch': fl / n, 'bytes_per_launch': by / n}


def _newest(profiles_dir, suffix):
    """Newest committed profiles/rNN_<suffix> (by round number) or (None, None)."""
    import glob
    import re
    best = None
    for f in glob.glob(os.path.join(profiles_dir or '', 'r[0-9][0-9]_' + suffix)):
        m = re.match(r'r(\d+)_', os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f)
    return (best[1], os.path.basename(best[1])) if best else (None, None)


def _family_row(table, key, peak_f, peak_b, traffic, tname):
    row = _row(key, table[key], peak_f, peak_b)
    sigs = [r for r in getattr(profile_step, 'last_signatures', []) if r['kernel'] == key]
    if sigs:
        top = sigs[0]
        bound_b = (top['flops_per_launch'] / peak_f) <= (top['bytes_per_launch'] / peak_b)
        ach = top['bytes_per_launch'] / (top['avg_launch_us'] * 1e-6) / 1e9 if bound_b else \
            top['flops_per_launch'] / (top['avg_launch_us'] * 1e-6) / 1e12
        row['dominant_shape'] = {'shape': top['shape'], 'launches_per_step': top['launches_per_step'],
                                 'avg_launch_us': round(top['avg_launch_us'], 2),
                                 'bytes_per_launch': top['bytes_per_launch'], 'flops_per_launch': top['flops_per_launch'],
                                 'achieved': round(ach, 3), 'unit': 'GB/s' if bound_b else 'TFLOP/s',
                                 'frac': round(ach / ((peak_b / 1e9) if bound_b else (peak_f / 1e12)), 4)}
        t, tkey = traffic.get(top['shape']), top['shape']
        if t is None:                                   # (the PMC pass keys the fused attention launches by kernel name)
            t, tkey = traffic.get('kernel:' + key), key
        if t is not None:
            row['traffic'] = t['bytes_per_launch']
            row['traffic_source'] = (f'profiles/{tname}[{tkey}]: 2 x FETCH_SIZE + WRITE_SIZE of that launch (2 x 96 x 160 map) from a '
                                     f'separate rocprofv3 --pmc run (tools/prof_round.sh {tname[:3]}); NOT measured in this bench run')
    return row


def kernel_resources(key):
    """Registers / scratch / LDS / waves per SIMD of kernel family `key` ('attn_block_bwd_kernel<18, 1>': that instantiation;
    'lin_fwd_kernel': every instantiation of the template) from the table the BUILD wrote (hrfuser_amd/kernel_resources.json,
    hipcc -Rpass-analysis=kernel-resource-usage of the shipped objects; empty when it belongs to other sources)."""
    from . import build_ext
    res = build_ext.resources()
    hits = {k: v for k, v in res.items() if k == key or k.startswith(key + '<') or k.replace(' ', '') == key.replace(' ', '')}
    if not hits and key.endswith('>'):              # the work model names the fused kernels by their FIRST template arguments
        pre = key[:-1].replace(' ', '')
        hits = {k: v for k, v in res.items() if k.replace(' ', '').startswith(pre + ',') or k.replace(' ', '').startswith(pre + '>')}
    if not hits:
        return None
    tot = lambda v: v.get('vgpr', 0) + v.get('agpr', 0)
    worst = max(hits.items(), key=lambda kv: (-kv[1].get('waves_per_simd', 8), tot(kv[1])))
    out = {'instantiations': len(hits), 'waves_per_simd_min': min(v.get('waves_per_simd', 8) for v in hits.values()),
           'waves_per_simd_max': max(v.get('waves_per_simd', 8) for v in hits.values()),
           'vgpr_plus_agpr_max': max(tot(v) for v in hits.values()), 'lds_static_max': max(v.get('lds_static', 0) for v in hits.values()),
           'spilling': sorted(k for k, v in hits.items() if v.get('scratch', 0) > 0),
           'lowest_occupancy': {'kernel': worst[0], **{k: worst[1].get(k) for k in ('vgpr', 'agpr', 'scratch', 'lds_static', 'waves_per_simd')}},
           'source': 'hrfuser_amd/kernel_resources.json (written by the build that produced the loaded library; register-limited '
                     'occupancy - dynamic LDS can lower it further)'}
    return out


def roofline_of_dominant(table, peak_f, peak_b, profiles_dir=None, grouped=None):
    """Roofline entry of the DOMINANT kernel family of the training step, selected and measured LIVE in this run (VERDICT r4
    #13: the selection used to come from a committed timeline and named last round's winner after a kernel change).
    Selection: kernel time per step of every family = launches x isolated per-launch duration (HIP events, every distinct
    (entry point, shape) re-issued in a captured graph), instantiations of one template summed; the weight-gradient family is
    priced by its GROUPED launches as the step issues them (in situ, HIP events around every grouped launch of 3 eager steps:
    `grouped`) instead of its 240 problems one at a time.  The family with the largest time is the entry, its instantiation
    with the largest time the kernel: `achieved` = algorithmic bytes (or flops) per launch / average launch duration,
    `dominant_shape` = its heaviest single (entry point, shape), `resources` = registers / scratch / waves per SIMD of the
    build.  `timeline_dominant`: what the newest committed rocprofv3 step timeline names (cross-check only).
    `traffic` = PMC-measured fabric bytes per launch of the dominant shape when profiles/rNN_hbm_traffic.json holds it (a
    SEPARATE `rocprofv3 --pmc` run; the source is stated), else null."""
    import json
    traffic, tname = {}, None
    tpath, tname = _newest(profiles_dir, 'hbm_traffic.json')
    if tpath:
        with open(tpath) as fh:
            tj = json.load(fh)
        traffic = dict(tj.get('shapes', {}))
        traffic.update({'kernel:' + k: v for k, v in tj.get('kernels', {}).items()})
    base = lambda k: k.split('<')[0]
    fam_ms = {}
    for k, t in table.items():
        fam_ms[base(k)] = fam_ms.get(base(k), 0.0) + t[1] / t[4] * 1e3
    iso_ms = dict(fam_ms)
    g_ms = sum(r['time_per_step_ms'] for r in grouped) if grouped else None
    if g_ms and 'wgrad_dense_kernel' in fam_ms:
        fam_ms['wgrad_dense_kernel'] = g_ms
    fam = max(fam_ms, key=fam_ms.get)
    ranking = [{'family': k, 'ms_per_step': round(v, 3)} for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])[:6]]
    sel = ('family with the largest kernel time per step MEASURED IN THIS RUN (launches x isolated launch duration, template '
           'instantiations summed; weight gradients priced by their grouped in-situ launches'
           + (f': {g_ms:.2f} ms grouped vs {iso_ms.get("wgrad_dense_kernel", 0.0):.2f} ms one problem at a time' if g_ms else '') + ')')
    def attach_traffic(r, shape_key, kernel_key):
        """PMC fabric bytes per launch of `r` from the newest committed rNN_hbm_traffic.json (a SEPARATE rocprofv3 --pmc run) +
        its ratio to the algorithmic bytes of that launch; null when the file has no entry for it."""
        t, tkey = traffic.get(shape_key) if shape_key else None, shape_key
        if t is None:
            t, tkey = traffic.get('kernel:' + kernel_key), kernel_key
        if t is None:                                   # the work model names the fused kernels by their first template arguments
            pre = 'kernel:' + kernel_key.rstrip('>')
            hits = sorted(k for k in traffic if k.startswith(pre + ',') or k.startswith(pre + '>'))
            if hits:
                t, tkey = traffic[hits[0]], hits[0][len('kernel:'):]
        r['traffic'] = None
        if t is not None:
            r['traffic'] = t['bytes_per_launch']
            alg = t.get('algorithmic_bytes') or r.get('bytes_per_launch')
            if alg:
                r['traffic_ratio'] = round(t['bytes_per_launch'] / alg, 3)
            r['traffic_source'] = (f'profiles/{tname}[{tkey}]: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of one launch on the 2 x 96 x 160 map, '
                                   f'separate rocprofv3 --pmc passes (tools/prof_round.sh {tname[:3]}); NOT measured in this bench run')

    if fam == 'wgrad_dense_kernel' and grouped:
        hv = roofline_grouped(grouped, peak_f, peak_b)
        row = dict(hv.pop('family'))
        row['kernel'] = 'wgrad_dense_kernel'
        row['timing'] = hv.get('timing')
        hv['resources'] = kernel_resources(hv['kernel']) or kernel_resources('wgrad_dense_kernel')
        attach_traffic(hv, None, hv['kernel'])
    else:
        keys = [k for k in table if base(k) == fam]
        n = sum(table[k][0] / table[k][4] for k in keys)
        sec = sum(table[k][1] / table[k][4] for k in keys)
        fl = sum(table[k][2] / table[k][4] for k in keys)
        by = sum(table[k][3] / table[k][4] for k in keys)
        row = _row(fam, [n, sec, fl, by, 1], peak_f, peak_b)
        row['launches_per_step'] = round(n, 1)
        row['timing'] = ('every distinct (entry point, shape) of the step re-issued 20x in a captured hipGraph, HIP events on the launch '
                         'stream (isolated launch durations), summed over the family')
        key = max(keys, key=lambda k: table[k][1])
        hv = _family_row(table, key, peak_f, peak_b, {}, tname)
        hv['resources'] = kernel_resources(key)
        ds = hv.get('dominant_shape') or {}
        attach_traffic(hv, ds.get('shape'), key)
    # the headline object = the dominant FAMILY (all of its instantiations: what the step spends its time in); the instantiation
    # with the most step time rides along as `heaviest_variant` with its own achieved / frac / traffic (rounds 4
    # and 5 swapped the two between them - 0.086 and 0.27 were not the same quantity)
    row['scope'] = f'kernel family: {len([k for k in table if base(k) == fam])} instantiation(s) of {fam}'
    row['traffic'] = hv.get('traffic')
    if hv.get('traffic') is not None:
        row['traffic_ratio'] = hv.get('traffic_ratio')
        row['traffic_note'] = 'fabric bytes per launch of the heaviest variant (below); ratio to ITS algorithmic bytes'
    row['heaviest_variant'] = hv
    row['resources'] = kernel_resources(fam)
    row['selection'] = sel
    row['family_ranking_ms'] = ranking
    lpath, lname = _newest(profiles_dir, 'step_timeline.json')
    if lpath:
        with open(lpath) as fh:
            tl = json.load(fh).get('families', {})
        if tl:
            top = max(tl.items(), key=lambda kv: kv[1].get('in_step_us', 0.0))
            row['timeline_dominant'] = {'family': top[0], 'in_step_ms': round(top[1].get('in_step_us', 0.0) / 1e3, 3),
                                        'source': f'profiles/{lname} (rocprofv3 --kernel-trace of one replayed step, committed; NOT this run)',
                                        'agrees': base(top[0]) == fam}
            # the same family IN SITU (four lanes sharing the chip, under the tracer): launches are longer than alone - both
            # fractions are stated so that the rocprofv3 summary and this line can be read against each other
            mine = next((v for k, v in tl.items() if base(k) == fam), None)
            if mine and mine.get('in_step_us') and row.get('launches_per_step') and row.get('avg_launch_us'):
                us = mine['in_step_us'] / row['launches_per_step']
                row['in_situ'] = {'avg_launch_us': round(us, 2), 'frac': round(row['frac'] * row['avg_launch_us'] / us, 4),
                                  'source': f'profiles/{lname}: the family\'s kernel time inside one traced graph replay / its launches'}
    # the next families, for context
    others = sorted(table.items(), key=lambda kv: -kv[1][1])[:7]
    row['next_families'] = [{k2: _row(k, t, peak_f, peak_b)[k2] for k2 in ('kernel', 'bound', 'frac', 'launches_per_step', 'time_per_step_ms')}
                            for k, t in others if k != row.get('kernel')][:6]
    return row


def table_json(table, peak_f, peak_b):
    rows = [_row(k, t, peak_f, peak_b) for k, t in table.items()]
    rows.sort(key=lambda r: -r['time_per_step_ms'])
    return rows


def time_eval_forward(net, x, mods, iters=30, use_graph=True):
    """Eval-mode (running-stat BN) forward latency in ms per image."""
    net.eval()
    B = x.shape[0]
    with torch.no_grad():
        for _ in range(3):
            net(x, list(mods))
        torch.cuda.synchronize()
        run = lambda: net(x, list(mods))
        if use_graph:
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    net(x, list(mods))
                torch.cuda.current_stream().wait_stream(side)
                g = torch.cuda.CUDAGraph()
                with _gc_paused(), torch.cuda.graph(g):
                    net(x, list(mods))
                run = g.replay
            except Exception:
                torch.cuda.synchronize()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            run()
        torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / iters / B * 1e3, 4)


def time_neck(in_channels, B, H, W, out_channels=256, iters=20, warm=15, seed=0):
    """HRFPN neck on cuda:0 at the backbone's output shapes, random weights / inputs:
    -> dict(fwd_ms, fwd_bwd_ms, fwd_gflop).  Inputs are channels-last like the backbone's outputs."""
    import time
    from .neck import HRFPN
    dev = torch.device('cuda', torch.cuda.current_device())
    g = torch.Generator().manual_seed(seed)
    xs = [torch.randn(B, c, H >> i, W >> i, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
          for i, c in enumerate(in_channels)]
    net = HRFPN(in_channels=list(in_channels), out_channels=out_channels)
    net.init_weights()
    net.to(dev)

    def timed(fn):
        t_w = time.perf_counter()
        n_w = 0
        while n_w < warm or time.perf_counter() - t_w < 0.3:      # the GPU may come out of an idle (down-clocked) phase
            fn()
            n_w += 1
            if n_w % 8 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    net.eval()
    with torch.no_grad():
        fwd = timed(lambda: net(xs))
    net.train()
    xr = [t.detach().requires_grad_(True) for t in xs]
    ones = None

    def step():
        nonlocal ones
        ys = net(xr)
        if ones is None:
            ones = [torch.ones_like(y) for y in ys]
        torch.autograd.backward(list(ys), ones)
    both = timed(step)
    flop = 2.0 * B * H * W * (sum(in_channels) * out_channels + 9 * out_channels * out_channels * sum(4.0 ** -i for i in range(5)))
    return {'workload': f'HRFPN {list(in_channels)}->{out_channels}, {B}x{H}x{W} finest grid, 5 levels',
            'fwd_ms': round(fwd, 3), 'fwd_bwd_ms': round(both, 3), 'fwd_gflop': round(flop / 1e9, 2),
            'fwd_tflops': round(flop / fwd / 1e9, 1)}


def grouped_wgrad_report(trainer, x, mods, cots, steps=3):
    """In-situ durations of the grouped weight-gradient launches (include/hrfuser_hip.h: hrf_wgrad_group_*): `steps`
    EAGER training steps with HIP events recorded by the library around every grouped launch, on the lane stream it is
    issued on and with the rest of the step running beside it - the same situation rocprofv3 --kernel-trace sees.
    -> rows sorted by time: kernel (the rocprof kernel name), launches_per_step, problems_per_launch, avg_launch_us,
    bytes_per_launch / flops_per_launch (algorithmic, summed over the problems of a launch), heaviest problem."""
    import ctypes
    L = _lib.lib()
    if os.environ.get('HRF_WGRAD_GROUP', '1') == '0' or os.environ.get('HRF_LANES', '1') == '0':
        return []
    trainer.step(x, mods, cots)                          # warm (allocator, lanes)
    torch.cuda.synchronize()
    buf = (ctypes.c_double * (12 * 64))()
    L.hrf_wgrad_group_report(ctypes.addressof(buf), 64)  # drop anything logged earlier
    L.hrf_debug_knob(7, 1)
    try:
        for _ in range(steps):
            trainer.step(x, mods, cots)
        torch.cuda.synchronize()
    finally:
        L.hrf_debug_knob(7, 0)
    n = L.hrf_wgrad_group_report(ctypes.addressof(buf), 64)
    rows = []
    for r in range(n):
        key, launches, problems, us, nbytes, flops, cin, cout, h, w, stride, kh = [buf[12 * r + k] for k in range(12)]
        key = int(key)
        tf = lambda b: 'true' if b else 'false'
        if key >= 4096:                                                   # csrc/wgrad_tiled.hip: 4096 + (((mt * 2 + swap) * 2 + bnb) * 4 + act)
            k = key - 4096
            name = f'wgrad_tiled_kernel<{k >> 4}, {tf((k >> 3) & 1)}, {tf((k >> 2) & 1)}, {k & 3}>'
        else:
            tap, act, bnb, nt, mt = key & 1, (key >> 1) & 3, (key >> 3) & 1, (key >> 4) & 7, key >> 7
            tapm = 2 if (tap and nt == 7) else (1 if tap else 0)          # nt code 7 = tap-blocked (NT 9, TAPM 2)
            name = f'wgrad_dense_kernel<{mt}, {9 if nt == 7 else nt}, {tf(bnb)}, {act}, {tapm}>'
        rows.append({'kernel': name,
                     'launches_per_step': round(launches / steps, 2), 'problems_per_launch': round(problems / launches, 2),
                     'avg_launch_us': round(us / launches, 2), 'time_per_step_ms': round(us / steps / 1e3, 4),
                     'bytes_per_launch': nbytes / launches, 'flops_per_launch': flops / launches,
                     'heaviest_problem': f'conv_bwd_weight[Cin={int(cin)},Cout={int(cout)},H={int(h)},W={int(w)},'
                                         f'KH={int(kh)},stride={int(stride)}]'})
    rows.sort(key=lambda r: -r['time_per_step_ms'])
    return rows


def roofline_grouped(rows, peak_f, peak_b, traffic=None):
    """Roofline object of the dominant kernel when it is the (grouped) weight-gradient family: family-wide aggregate of
    the in-situ launches + the heaviest variant.  A launch = one grouped kernel launch (what rocprofv3 counts)."""
    if not rows:
        return None
    launches = sum(r['launches_per_step'] for r in rows)
    t_ms = sum(r['time_per_step_ms'] for r in rows)
    nbytes = sum(r['bytes_per_launch'] * r['launches_per_step'] for r in rows)
    flops = sum(r['flops_per_launch'] * r['launches_per_step'] for r in rows)

    def price(b, f, us):
        bound_b = (f / peak_f) <= (b / peak_b)
        ach = b / (us * 1e-6) / 1e9 if bound_b else f / (us * 1e-6) / 1e12
        peak = peak_b / 1e9 if bound_b else peak_f / 1e12
        return {'bound': 'hbm' if bound_b else 'mfma', 'achieved': round(ach, 1), 'peak': peak,
                'unit': 'GB/s' if bound_b else 'TFLOP/s', 'frac': round(ach / peak, 4)}
    avg_us = t_ms * 1e3 / launches
    fam = {'kernel': 'wgrad_dense_kernel (all variants)'}
    fam.update(price(nbytes / launches, flops / launches, avg_us))
    fam.update({'launches_per_step': round(launches, 1), 'avg_launch_us': round(avg_us, 2),
                'time_per_step_ms': round(t_ms, 4), 'bytes_per_launch': nbytes / launches,
                'flops_per_launch': flops / launches})
    # the object itself = the heaviest variant of the family (one rocprofv3 kernel name); the family aggregate beside it
    row = dict(rows[0])
    row.update(price(row['bytes_per_launch'], row['flops_per_launch'], row['avg_launch_us']))
    row['traffic'] = (traffic or {}).get(row['kernel'])
    row['timing'] = 'HIP events around every grouped launch of 3 eager steps, on the launching stream (in situ)'
    row['family'] = fam
    if row['traffic'] is not None:
        row['traffic_note'] = ('PMC (FETCH_SIZE x2 + WRITE_SIZE) bytes of ONE representative grouped launch of this variant '
                               '(4 problems, 97.1 MB algorithmic: profiles/r01_hbm_traffic.json "grouped", tools/pmc_grouped.py)')
    return row
