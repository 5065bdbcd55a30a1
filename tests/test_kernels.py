"""Kernel-level parity through the C ABI: every entry point of include/hrfuser_hip.h against plain
PyTorch fp32/fp64 math of the same op (the oracle's building blocks)."""
import pytest
import torch
import torch.nn.functional as F

import hrfuser_oracle as O
from helpers import use_backend
from hrfuser_amd import _lib

TOL = 2e-5
KC = _lib.STAT_COPIES      # cross-block accumulators are replicated (include/hrfuser_hip.h)


def _rep_moments(rows, count, g, dev):
    """[KC][2*C] doubles whose copies sum to `rows` * count (rows: [2, C] per-sample moments)."""
    wts = torch.rand(KC, 1, 1, generator=g).double()
    wts /= wts.sum()
    return (wts * (rows.double() * count)[None]).reshape(-1).contiguous().to(dev)


def make_fin(L, C, count, dev, g, write=1):
    """hrf_bn_fin_t over synthetic replicated moments, its buffers, and the reference results of hrf_bn_finalize."""
    mean = torch.randn(C, generator=g) * 0.3
    var = torch.rand(C, generator=g) * 0.8 + 0.4
    t = dict(stats=_rep_moments(torch.stack([mean, var + mean ** 2]), count, g, dev),
             gamma=(torch.rand(C, generator=g) + 0.5).to(dev), beta=(torch.randn(C, generator=g) * 0.3).to(dev),
             rm=torch.randn(C, generator=g).to(dev), rv=(torch.rand(C, generator=g) + 0.5).to(dev))
    for k in ('scale', 'shift', 'mean', 'invstd'):
        t[k] = torch.zeros(C, device=dev)
        t['ref_' + k] = torch.zeros(C, device=dev)
    t['ref_rm'], t['ref_rv'] = t['rm'].clone(), t['rv'].clone()
    L.hrf_bn_finalize(t['stats'], t['gamma'], t['beta'], t['ref_rm'], t['ref_rv'], float(count), 1e-5, 0.1, 1,
                      t['ref_scale'], t['ref_shift'], t['ref_mean'], t['ref_invstd'], C, _lib.stream_ptr())
    P = _lib._ptr
    fin = _lib.BnFin(P(t['stats']), P(t['gamma']), P(t['beta']), P(t['rm']), P(t['rv']), P(t['scale']), P(t['shift']),
                     P(t['mean']), P(t['invstd']), float(count), 1e-5, 0.1, 1, write, C)
    return fin, t


def check_fin(t):
    """the designated writer block of an on-load finalize publishes exactly what hrf_bn_finalize computes"""
    for k in ('scale', 'shift', 'mean', 'invstd', 'rm', 'rv'):
        assert r(t[k], t['ref_' + k]) < 1e-6, k


def make_bfin(L, C, count, dev, g, train=1):
    """hrf_bn_bfin_t over synthetic replicated (sum du, sum du*y), and the reference results of hrf_bn_bwd_finalize."""
    t = dict(gstats=_rep_moments(torch.randn(2, C, generator=g) * 0.2, count, g, dev),
             gamma=(torch.rand(C, generator=g) + 0.5).to(dev), mean=(torch.randn(C, generator=g) * 0.3).to(dev),
             invstd=(torch.rand(C, generator=g) + 0.7).to(dev),
             dgamma=torch.randn(C, generator=g).to(dev), dbeta=torch.randn(C, generator=g).to(dev))
    for k in ('cA', 'cB', 'cC'):
        t[k] = torch.zeros(C, device=dev)
        t['ref_' + k] = torch.zeros(C, device=dev)
    t['ref_dgamma'], t['ref_dbeta'] = t['dgamma'].clone(), t['dbeta'].clone()
    L.hrf_bn_bwd_finalize(t['gstats'], None, t['gamma'], t['mean'], t['invstd'], float(count), train, t['ref_dgamma'],
                          t['ref_dbeta'], t['ref_cA'], t['ref_cB'], t['ref_cC'], C, _lib.stream_ptr())
    P = _lib._ptr
    bfin = _lib.BnBFin(P(t['gstats']), P(t['gamma']), P(t['mean']), P(t['invstd']), P(t['dgamma']), P(t['dbeta']),
                       P(t['cA']), P(t['cB']), P(t['cC']), float(count), train, 1, C)
    return bfin, t


def check_bfin(t):
    for k in ('cA', 'cB', 'cC', 'dgamma', 'dbeta'):
        assert r(t[k], t['ref_' + k]) < 1e-6, k


def zstat(C, dev):
    return torch.zeros(KC * 2 * C, dtype=torch.float64, device=dev)


def fold(st):
    """sum of the replicated copies -> [2*C]"""
    return st.view(KC, -1).sum(0)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def r(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _tf_apply(x, tf, sc, sh):
    """reference of the transform-on-load; returns (u leaf, transformed, rowstat)"""
    if tf == 4:
        xl = nhwc(x)
        mean = xl.mean(-1, keepdim=True)
        rstd = (xl.var(-1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
        rowstat = torch.cat([mean, rstd], -1).reshape(-1, 2).contiguous()
        u = (((xl - mean) * rstd) * sc + sh).permute(0, 3, 1, 2).detach().requires_grad_(True)
        return u, u, rowstat
    if tf:
        u = (x * sc[None, :, None, None] + sh[None, :, None, None]).requires_grad_(True)
        return u, (u if tf == 1 else (F.relu(u) if tf == 2 else F.gelu(u))), None
    u = x.clone().requires_grad_(True)
    return u, u, None


CONV_CASES = [  # B,H,W,Cin,Cout,KH,stride,tf,bnb,epi
    (2, 9, 11, 18, 72, 1, 1, 4, True, False),
    (2, 9, 11, 72, 18, 1, 1, 3, True, True),
    (2, 10, 13, 64, 64, 3, 1, 2, True, True),
    (2, 10, 13, 20, 36, 3, 2, 0, False, False),
    (2, 11, 13, 20, 36, 3, 2, 2, True, True),
    (1, 6, 5, 256, 18, 3, 1, 0, True, False),
    (3, 40, 50, 16, 40, 1, 1, 1, False, True),
    (2, 16, 24, 64, 256, 1, 1, 2, True, True),
    (2, 7, 9, 144, 80, 1, 1, 0, False, False),     # several 64-wide tile groups in both dims, ragged
    (1, 5, 3, 78, 78, 1, 1, 4, False, True),       # HRFuser-B width, fewer pixels than one batch
    (1, 50, 130, 64, 64, 3, 1, 2, True, True),
    (1, 18, 35, 64, 64, 3, 2, 2, True, True),      # stride-2 backward: parity-class kernel, ragged tiles
    (1, 33, 47, 256, 36, 3, 2, 0, True, False),    # stride-2 forward at transition1's depth (2 304-deep contraction), odd grid
    (2, 21, 40, 18, 72, 3, 2, 1, True, True),      # ... the modality transitions' 18 input channels, two column blocks
    (1, 70, 66, 64, 256, 1, 1, 2, True, True),     # wide 1x1 on the LDS-tiled engine (M >= 4096)
    (1, 66, 70, 256, 64, 1, 1, 1, True, False),      # M >= 128*.. exercises the BM=128 tile on GPU sizes
    (2, 5, 6, 288, 72, 1, 1, 3, True, True),         # few rows, deep contraction (HRFuser-B's coarse branches in miniature)
    (1, 6, 7, 330, 300, 1, 1, 4, True, False),       # deep in both directions, ragged K and N
    (2, 6, 5, 144, 144, 1, 1, 4, True, False),       # 9 channel tiles in ONE wave (whole-row LayerNorm statistics) + split K
    (1, 130, 128, 16, 24, 1, 1, 2, True, True),      # >= 16 384 rows with moments: 16-wave "fat" blocks of the row GEMMs, ragged last block
]


def _pack3x(L, w, dev, direction):
    """tap-major pack of an OIHW weight for csrc/conv3x_engine.hip (hrf_conv3x_pack: one job)"""
    Cout, Cin = w.shape[:2]
    wp = torch.full((L.hrf_conv3x_pack_size(Cout, Cin, direction),), float('nan'), device=dev)
    jobs = (_lib.Conv3xPackJob * 1)()
    jobs[0] = _lib.Conv3xPackJob(_lib._ptr(w), _lib._ptr(wp), Cout, Cin, direction)
    L.hrf_conv3x_pack(jobs, 1, _lib.stream_ptr())
    return wp


def run_conv(case, backend, packed=False):
    """packed: forward / backward-data through hrf_conv_fwd_packed / hrf_conv_bwd_data_packed (csrc/conv3x_engine.hip) where
    hrf_conv3x_supported takes the shape - same arguments, same expected results"""
    dev = use_backend(backend)
    L = _lib.lib()
    B, H, W, Cin, Cout, KH, stride, tf, bnb, epi = case
    conv_fwd, conv_bwd_data = L.hrf_conv_fwd, L.hrf_conv_bwd_data
    if packed:
        assert L.hrf_conv3x_supported(Cin, Cout, KH, stride, 0) or L.hrf_conv3x_supported(Cin, Cout, KH, stride, 1), case
    g = torch.Generator().manual_seed(sum(case[:7]))
    rn = lambda *s: torch.randn(*s, generator=g)
    xraw, w, bias = rn(B, Cin, H, W), rn(Cout, Cin, KH, KH) * 0.2, rn(Cout)
    fin = ft = None
    sc, sh = torch.rand(Cin, generator=g) + 0.5, rn(Cin) * 0.3
    if tf in (1, 2, 3):
        # the input's BatchNorm is finalised ON LOAD from replicated moments (hrf_bn_fin_t): scale / shift are what
        # hrf_bn_finalize derives from the same moments; the kernel gets no scale/shift arrays at all
        fin, ft = make_fin(L, Cin, 977.0, dev, g)
        sc, sh = ft['ref_scale'].cpu(), ft['ref_shift'].cpu()
    u, xt, rowstat = _tf_apply(xraw, tf, sc, sh)
    wq, bq = w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    y = F.conv2d(xt, wq, bq, stride, KH // 2)
    Ho, Wo = y.shape[2:]
    res = rn(B, Ho, Wo, Cout)
    yref = nhwc(y.detach()) + res
    D = lambda t: None if t is None else t.to(dev)
    xr = nhwc(xraw)
    st = (H * W * Cin, W * Cin, Cin, 1)
    yk = torch.zeros(B, Ho, Wo, Cout, device=dev)
    stats = zstat(Cout, dev)
    lnrs = torch.zeros(B * Ho * Wo, 2, device=dev)      # fused LayerNorm row statistics of the output
    if packed and L.hrf_conv3x_supported(Cin, Cout, KH, stride, 0):
        wpf = _pack3x(L, D(w), dev, 0)
        conv_fwd = lambda *a: L.hrf_conv_fwd_packed(*a[:-1], wpf, a[-1])
    if packed and L.hrf_conv3x_supported(Cin, Cout, KH, stride, 1):
        wpb = _pack3x(L, D(w), dev, 1)
        conv_bwd_data = lambda *a: L.hrf_conv_bwd_data_packed(*a[:-1], wpb, a[-1])
    conv_fwd(D(xr), *st, B, H, W, Cin, D(w), D(bias), KH, stride, Cout, yk, Cout, 0, D(res), None, Cout,
             tf, D(sc) if (tf and fin is None) else None, D(sh) if (tf and fin is None) else None, D(rowstat), stats,
             fin, lnrs, 1e-6, _lib.stream_ptr())
    if fin is not None:
        check_fin(ft)
    yr_ = yref.reshape(-1, Cout)
    assert r(lnrs[:, 0], yr_.mean(-1)) < TOL and r(lnrs[:, 1], (yr_.var(-1, unbiased=False) + 1e-6).rsqrt()) < 1e-4
    assert r(yk, yref) < TOL
    s1, s2 = yref.reshape(-1, Cout).double().sum(0), (yref.reshape(-1, Cout).double() ** 2).sum(0)
    assert r(fold(stats)[:Cout], s1) < TOL and r(fold(stats)[Cout:], s2) < TOL
    # the split over K (deep 3x3 contractions with few row blocks): same output, same moments, bit-reproducible
    nsc = L.hrf_conv_fwd_split_scratch(*st, B, H, W, Cin, KH, stride, Cout, Cout, 0)
    if nsc > 0 and not packed:
        assert KH == 3 and 9 * Cin >= 1024
        outs = []
        for _ in range(2):
            yk3, st3 = torch.zeros(B, Ho, Wo, Cout, device=dev), zstat(Cout, dev)
            L.hrf_conv_fwd_split(D(xr), *st, B, H, W, Cin, D(w), D(bias), KH, stride, Cout, yk3, Cout, 0, D(res), None, Cout,
                                 tf, D(sc) if (tf and fin is None) else None, D(sh) if (tf and fin is None) else None, D(rowstat), st3,
                                 fin, None, 0.0, torch.empty(nsc, device=dev), _lib.stream_ptr())
            outs.append(yk3)
            if fin is None:
                assert r(yk3, yref) < TOL
                assert r(fold(st3)[:Cout], s1) < TOL and r(fold(st3)[Cout:], s2) < TOL
        assert torch.equal(outs[0], outs[1])
    # NCHW input through strides (stem path)
    if tf == 0 and not packed:
        yk2 = torch.zeros_like(yk)
        L.hrf_conv_fwd(D(xraw.contiguous()), Cin * H * W, W, 1, H * W, B, H, W, Cin, D(w), D(bias), KH, stride, Cout,
                       yk2, Cout, 0, None, None, 0, 0, None, None, None, None, None, None, 0.0, _lib.stream_ptr())
        assert r(yk2, nhwc(y.detach())) < TOL
    # ---- backward
    du, yraw = rn(B, Ho, Wo, Cout), rn(B, Ho, Wo, Cout)
    cA, cB, cC = rn(Cout), rn(Cout) * 0.3, rn(Cout) * 0.1
    dyeff = cA * du + cB * yraw + cC if bnb else du
    y.backward(dyeff.permute(0, 3, 1, 2))
    gu = nhwc(u.grad)
    co = (D(cA), D(cB), D(cC)) if bnb else (None, None, None)
    act = {0: 0, 1: 0, 2: 1, 3: 2, 4: 0}[tf]
    dx = torch.zeros(B, H, W, Cin, device=dev)
    if epi:
        gst = zstat(Cin, dev)
        conv_bwd_data(D(du), Cout, 0, D(yraw), *co, None, D(w), KH, stride, Cout, B, H, W, Cin, dx, *st, 0, 1,
                            D(xr), Cin, D(sc), D(sh), act, gst, _lib.stream_ptr())
        assert r(dx, gu) < TOL
        assert r(fold(gst)[:Cin], gu.reshape(-1, Cin).double().sum(0)) < TOL
        assert r(fold(gst)[Cin:], (gu.reshape(-1, Cin).double() * xr.reshape(-1, Cin).double()).sum(0)) < TOL
    else:
        base = rn(B, H, W, Cin)
        dx.copy_(base)
        conv_bwd_data(D(du), Cout, 0, D(yraw), *co, None, D(w), KH, stride, Cout, B, H, W, Cin, dx, *st, 1, 0,
                            None, 0, None, None, 0, None, _lib.stream_ptr())
        assert r(dx, gu + base) < TOL
    if bnb:
        # BatchNorm-backward coefficients derived ON LOAD (hrf_bn_bfin_t): same data gradient as with the coefficients
        # hrf_bn_bwd_finalize computes from the same moments; block 0 publishes cA/cB/cC and adds dgamma / dbeta
        bfin, bt = make_bfin(L, Cout, 811.0, dev, g)
        dxa, dxb = torch.zeros(B, H, W, Cin, device=dev), torch.zeros(B, H, W, Cin, device=dev)
        tail = (1, D(xr), Cin, D(sc), D(sh), act, zstat(Cin, dev)) if epi else (0, None, 0, None, None, 0, None)
        conv_bwd_data(D(du), Cout, 0, D(yraw), bt['ref_cA'], bt['ref_cB'], bt['ref_cC'], None, D(w), KH, stride, Cout,
                            B, H, W, Cin, dxa, *st, 0, *tail, _lib.stream_ptr())
        tail = (1, D(xr), Cin, D(sc), D(sh), act, zstat(Cin, dev)) if epi else (0, None, 0, None, None, 0, None)
        conv_bwd_data(D(du), Cout, 0, D(yraw), bt['cA'], bt['cB'], bt['cC'], bfin, D(w), KH, stride, Cout,
                            B, H, W, Cin, dxb, *st, 0, *tail, _lib.stream_ptr())
        check_bfin(bt)
        assert r(dxb, dxa) < 1e-6
    dw, db = torch.zeros_like(w, device=dev), torch.zeros(Cout, device=dev)
    L.hrf_conv_bwd_weight(D(du), Cout, 0, D(yraw), *co, D(xr), *st, B, H, W, Cin, KH, stride, Cout, tf,
                          D(sc) if tf else None, D(sh) if tf else None, D(rowstat), dw, db, _lib.stream_ptr())
    assert r(dw, wq.grad) < TOL and r(db, bq.grad) < TOL
    # the LDS-staged 3x3 weight gradient (csrc/wgrad3x_engine.hip: per-split slabs in scratch + fold, no atomics) where the shape
    # qualifies - same gradient, and bit-reproducible
    L.hrf_debug_knob(33, 64)                     # (product threshold: 16 384 output pixels)
    try:
        nsc = L.hrf_conv_bwd_weight_scratch(*st, B, H, W, Cin, KH, stride, Cout, tf, 0)
        if nsc > 0:
            assert KH == 3 and Cin >= 32
            outs = []
            for _ in range(2):
                dw2 = torch.zeros_like(w, device=dev)
                L.hrf_conv_bwd_weight_s(D(du), Cout, 0, D(yraw), *co, D(xr), *st, B, H, W, Cin, KH, stride, Cout, tf,
                                        D(sc) if tf else None, D(sh) if tf else None, D(rowstat), dw2, None,
                                        torch.full((nsc,), float('nan'), device=dev), _lib.stream_ptr())
                outs.append(dw2)
            assert r(outs[0], wq.grad) < TOL
            assert torch.equal(outs[0], outs[1])
    finally:
        L.hrf_debug_knob(33, 0)


# the LDS-tiled row GEMM (csrc/lin2_engine.hip) serves wide 1x1 problems (min(Cin, Cout) >= 64, >= 1024 rows); the debug knob 28
# forces it on the small shapes a CPU emulator run can afford, knob 29 forces the block width (1..4 = 80 / 160 / 256 / 320 channels)
LIN2_CASES = [  # B,H,W,Cin,Cout,KH,stride,tf,bnb,epi
    (2, 9, 11, 18, 72, 1, 1, 4, True, False),       # LayerNorm on load, ragged K (18) and N (72)
    (2, 9, 11, 72, 18, 1, 1, 3, True, True),        # BatchNorm + GELU finalised on load; act' epilogue + moments
    (1, 13, 11, 78, 312, 1, 1, 4, True, True),      # HRFuser-B fc1 shape: K = 78 (tail of 14), N = 312 (two column blocks)
    (1, 13, 11, 312, 78, 1, 1, 3, True, True),      # HRFuser-B fc3
    (2, 7, 9, 144, 80, 1, 1, 0, False, False),
    (1, 16, 9, 64, 256, 1, 1, 2, True, True),       # Bottleneck conv3 (ReLU input)
    (1, 9, 16, 256, 64, 1, 1, 1, True, False),
    (1, 5, 3, 78, 78, 1, 1, 4, False, True),        # fewer rows than one block
]


def run_lin2(case, wn, backend):
    use_backend(backend)
    L = _lib.lib()
    L.hrf_debug_knob(28, 1)
    L.hrf_debug_knob(29, wn)
    try:
        run_conv(case, backend)
    finally:
        L.hrf_debug_knob(28, 0)
        L.hrf_debug_knob(29, 0)


@pytest.mark.parametrize('wn', [1, 2, 3, 4])
@pytest.mark.parametrize('case', LIN2_CASES[:5])
def test_lin2_emul(case, wn):
    run_lin2(case, wn, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('wn', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('case', LIN2_CASES + [(2, 96, 160, 78, 312, 1, 1, 4, True, True), (2, 96, 160, 312, 78, 1, 1, 3, True, True),
                                               (2, 48, 80, 156, 468, 1, 1, 4, True, False), (2, 96, 160, 64, 256, 1, 1, 2, True, True)])
def test_lin2_gpu(case, wn):
    run_lin2(case, wn, 'hip')


DW_CASES = [(2, 9, 11, 72, 1, 3, True, True, True), (2, 17, 35, 40, 1, 0, False, False, False), (2, 19, 21, 18, 1, 3, False, True, True),
            (2, 10, 13, 18, 2, 0, False, True, False), (2, 11, 15, 36, 2, 2, False, True, True),
            (1, 16, 32, 33, 2, 1, False, False, True),
            (1, 13, 18, 144, 1, 3, True, True, True),      # five channel blocks (two 72-channel slabs of the float4-lane forward: test_dwconv_lane4_modes_*), ragged tile rows / columns
            (1, 9, 33, 52, 1, 2, False, True, True)]       # 52 channels = one 13-lane slab of the float4-lane forward (HRFuser-B widths are 13 x 4 x k)


def run_dw(case, backend):
    dev = use_backend(backend)
    L = _lib.lib()
    B, H, W, C, S, tf, has_bias, bnb, epi = case
    g = torch.Generator().manual_seed(sum(case[:5]))
    rn = lambda *s: torch.randn(*s, generator=g)
    D = lambda t: None if t is None else t.to(dev)
    xraw, w = rn(B, C, H, W), rn(C, 1, 3, 3) * 0.3
    b = rn(C) if has_bias else None
    sc, sh = torch.rand(C, generator=g) + 0.5, rn(C) * 0.3
    fin = ft = None
    if tf in (1, 2, 3):                      # input BatchNorm finalised on load, as in run_conv
        fin, ft = make_fin(L, C, 977.0, dev, g)
        sc, sh = ft['ref_scale'].cpu(), ft['ref_shift'].cpu()
    u, xt, _ = _tf_apply(xraw, tf, sc, sh)
    wq, bq = w.clone().requires_grad_(True), torch.zeros(C, requires_grad=True)
    y = F.conv2d(xt, wq, (b + bq) if has_bias else bq, S, 1, groups=C)
    Ho, Wo = y.shape[2:]
    yk = torch.zeros(B, Ho, Wo, C, device=dev)
    st = zstat(C, dev)
    xr = nhwc(xraw)
    L.hrf_dwconv_fwd(D(xr), B, H, W, C, D(w), D(b), S, tf, D(sc) if (tf and fin is None) else None,
                     D(sh) if (tf and fin is None) else None, yk, st, fin, _lib.stream_ptr())
    if fin is not None:
        check_fin(ft)
    yr = nhwc(y.detach())
    assert r(yk, yr) < TOL
    assert r(fold(st)[:C], yr.reshape(-1, C).double().sum(0)) < TOL and r(fold(st)[C:], (yr.reshape(-1, C).double() ** 2).sum(0)) < TOL
    du, yraw = rn(B, Ho, Wo, C), rn(B, Ho, Wo, C)
    cA, cB, cC = rn(C), rn(C) * 0.3, rn(C) * 0.1
    y.backward((cA * du + cB * yraw + cC if bnb else du).permute(0, 3, 1, 2))
    gu = nhwc(u.grad)
    co = (D(cA), D(cB), D(cC)) if bnb else (None, None, None)
    act = {0: 0, 1: 0, 2: 1, 3: 2}[tf]
    dx = torch.zeros(B, H, W, C, device=dev)
    if epi:
        gst = zstat(C, dev)
        L.hrf_dwconv_bwd_data(D(du), D(yraw), *co, None, D(w), S, B, H, W, C, dx, 0, 1, D(xr), D(sc), D(sh), act, gst,
                              _lib.stream_ptr())
        assert r(dx, gu) < TOL
        assert r(fold(gst)[C:], (gu.reshape(-1, C).double() * xr.reshape(-1, C).double()).sum(0)) < TOL
    else:
        base = rn(B, H, W, C)
        dx.copy_(base)
        L.hrf_dwconv_bwd_data(D(du), D(yraw), *co, None, D(w), S, B, H, W, C, dx, 1, 0, None, None, None, 0, None,
                              _lib.stream_ptr())
        assert r(dx, gu + base) < TOL
    if bnb:                                  # coefficients derived on load (hrf_bn_bfin_t), as in run_conv
        bfin, bt = make_bfin(L, C, 811.0, dev, g)
        dxa, dxb = torch.zeros(B, H, W, C, device=dev), torch.zeros(B, H, W, C, device=dev)
        L.hrf_dwconv_bwd_data(D(du), D(yraw), bt['ref_cA'], bt['ref_cB'], bt['ref_cC'], None, D(w), S, B, H, W, C, dxa, 0, 0,
                              None, None, None, 0, None, _lib.stream_ptr())
        L.hrf_dwconv_bwd_data(D(du), D(yraw), bt['cA'], bt['cB'], bt['cC'], bfin, D(w), S, B, H, W, C, dxb, 0, 0,
                              None, None, None, 0, None, _lib.stream_ptr())
        check_bfin(bt)
        assert r(dxb, dxa) < 1e-6
    dw, db = torch.zeros_like(w, device=dev), torch.zeros(C, device=dev)
    L.hrf_dwconv_bwd_weight(D(du), D(yraw), *co, D(xr), B, H, W, C, S, tf, D(sc) if tf else None,
                            D(sh) if tf else None, dw, db, 0, _lib.stream_ptr())
    assert r(dw, wq.grad) < TOL and r(db, bq.grad) < TOL
    # the same call QUEUED between hrf_wgrad_group_begin / _end (how the weight-gradient phase issues it: DWG problems per
    # launch), twice and next to a half-height problem of the same stride: geometry larger than a member's own grid
    dwg, dbg = torch.zeros(2, *w.shape, device=dev), torch.zeros(2, C, device=dev)
    Hh = max(1, H // 2)
    duh, yrh, xrh = D(du)[:, :((Hh - 1) // S + 1)].contiguous(), D(yraw)[:, :((Hh - 1) // S + 1)].contiguous(), D(xr)[:, :Hh].contiguous()
    dwh, dbh = torch.zeros_like(w, device=dev), torch.zeros(C, device=dev)
    dwr, dbr = torch.zeros_like(w, device=dev), torch.zeros(C, device=dev)
    L.hrf_dwconv_bwd_weight(duh, yrh, *co, xrh, B, Hh, W, C, S, tf, D(sc) if tf else None, D(sh) if tf else None, dwr, dbr, 0,
                            _lib.stream_ptr())
    held = (D(du), D(yraw), D(xr), D(sc) if tf else None, D(sh) if tf else None, dwg[0], dwg[1], dbg[0], dbg[1])
    L.hrf_wgrad_group_begin()                  # (queued launches read their operands at _end: no temporaries here)
    for k in range(2):
        L.hrf_dwconv_bwd_weight(held[0], held[1], *co, held[2], B, H, W, C, S, tf, held[3], held[4], held[5 + k], held[7 + k], 0,
                                _lib.stream_ptr())
    L.hrf_dwconv_bwd_weight(duh, yrh, *co, xrh, B, Hh, W, C, S, tf, held[3], held[4], dwh, dbh, 0, _lib.stream_ptr())
    assert float(dwg.abs().max()) == 0.0                                  # nothing launched yet
    L.hrf_wgrad_group_end(_lib.stream_ptr())
    for k in range(2):
        assert r(dwg[k], wq.grad) < TOL and r(dbg[k], bq.grad) < TOL
    assert r(dwh, dwr) < 1e-6 and r(dbh, dbr) < 1e-6
    # replicated accumulators (copy_stride > 0) + hrf_fold_copies into a "gradient arena"
    n = 10 * C
    scr = torch.zeros(KC * n, device=dev)
    L.hrf_dwconv_bwd_weight(D(du), D(yraw), *co, D(xr), B, H, W, C, S, tf, D(sc) if tf else None,
                            D(sh) if tf else None, scr, scr[9 * C:], n, _lib.stream_ptr())
    arena = torch.ones(n + 5, device=dev)
    amap = (torch.arange(n, dtype=torch.int32) + 5).to(dev)
    L.hrf_fold_copies(scr, n, amap, arena, n, _lib.stream_ptr())
    assert r(arena[5:5 + 9 * C] - 1, wq.grad.reshape(-1)) < TOL and r(arena[5 + 9 * C:] - 1, bq.grad) < TOL
    assert float(arena[:5].sum()) == 5.0
    if epi and S == 1 and tf:
        # data gradient + weight gradient of the same convolution in ONE pass (hrf_dwconv_bwd_data_weight)
        dx2, gst2, scr2 = torch.zeros(B, H, W, C, device=dev), zstat(C, dev), torch.zeros(KC * n, device=dev)
        L.hrf_dwconv_bwd_data_weight(D(du), D(yraw), *co, None, D(w), B, H, W, C, dx2, D(xr), D(sc), D(sh), act, gst2,
                                     scr2, scr2[9 * C:] if has_bias else None, n, _lib.stream_ptr())
        assert r(dx2, gu) < TOL and r(fold(gst2), fold(gst)) < 1e-6
        arena2 = torch.zeros(n, device=dev)
        L.hrf_fold_copies(scr2, n, torch.arange(n, dtype=torch.int32).to(dev), arena2, n, _lib.stream_ptr())
        assert r(arena2[:9 * C], wq.grad.reshape(-1)) < TOL
        if has_bias:
            assert r(arena2[9 * C:], bq.grad) < TOL


ATTN_CASES = [(2, 10, 13, 18, 1), (1, 7, 7, 36, 2), (2, 15, 8, 72, 4), (1, 9, 16, 78, 2), (1, 6, 10, 64, 8),
              (3, 20, 31, 18, 1)]


def run_attn(case, backend):
    dev = use_backend(backend)
    L = _lib.lib()
    B, H, W, C, heads = case
    P = B * H * W
    g = torch.Generator().manual_seed(sum(case))
    rn = lambda *s: torch.randn(*s, generator=g)
    q, kv = rn(P, C).requires_grad_(True), rn(P, 2 * C).requires_grad_(True)
    kb, vb = rn(C).requires_grad_(True), rn(C).requires_grad_(True)
    T = (rn(169, heads) * 0.5).requires_grad_(True)
    idx = O.rel_pos_index()

    def part(t, padval):     # padded tokens carry the projection bias (zero input after LayerNorm)
        return O.window_partition(t.view(B, H * W, -1) - padval, H, W) + padval
    ow = O._window_attention_core(O.window_partition(q.view(B, H * W, C), H, W), part(kv[:, :C], kb),
                                  part(kv[:, C:], vb), heads, T, idx)
    o_ref = O.window_merge(ow, B, H, W).reshape(P, C)
    go = rn(P, C)
    o_ref.backward(go)
    D = lambda t: t.detach().to(dev)
    qd, kvd, kbd, vbd, Td = D(q), D(kv), D(kb), D(vb), D(T)
    o = torch.zeros(P, C, device=dev)
    s = _lib.stream_ptr()
    L.hrf_window_attn_fwd(qd, C, 0, kvd, 2 * C, 0, kvd, 2 * C, C, kbd, vbd, Td, o, C, B, H, W, C, heads, s)
    assert r(o, o_ref) < TOL
    dq, dkv = torch.zeros(P, C, device=dev), torch.zeros(P, 2 * C, device=dev)
    dkb, dvb, dT = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(169, heads, device=dev)
    L.hrf_window_attn_bwd(qd, C, 0, kvd, 2 * C, 0, kvd, 2 * C, C, kbd, vbd, Td, D(go), C, dq, C, 0, dkv, 2 * C, 0,
                          dkv, 2 * C, C, dkb, dvb, dT, 0, B, H, W, C, heads, s)
    assert r(dq, q.grad) < TOL and r(dkv, kv.grad) < TOL and r(dT, T.grad) < TOL
    if (H % 7) or (W % 7):
        assert r(dkb, kb.grad) < 1e-4 and r(dvb, vb.grad) < TOL
    # replicated parameter-gradient accumulators: [copies][dT | dkb | dvb]
    n = 169 * heads + 2 * C
    scr = torch.zeros(KC * n, device=dev)
    dq2, dkv2 = torch.zeros_like(dq), torch.zeros_like(dkv)
    L.hrf_window_attn_bwd(qd, C, 0, kvd, 2 * C, 0, kvd, 2 * C, C, kbd, vbd, Td, D(go), C, dq2, C, 0, dkv2, 2 * C, 0,
                          dkv2, 2 * C, C, scr[169 * heads:], scr[169 * heads + C:], scr, n, B, H, W, C, heads, s)
    tot = scr.view(KC, n).sum(0)
    assert r(dq2, q.grad) < TOL and r(tot[:169 * heads].view(169, heads), T.grad) < TOL
    if (H % 7) or (W % 7):
        assert r(tot[169 * heads:169 * heads + C], kb.grad) < 1e-4 and r(tot[169 * heads + C:], vb.grad) < TOL


def run_pointwise(backend):
    dev = use_backend(backend)
    L = _lib.lib()
    s = _lib.stream_ptr()
    g = torch.Generator().manual_seed(11)
    rn = lambda *sh: torch.randn(*sh, generator=g)
    D = lambda t: None if t is None else t.detach().to(dev)
    # ---- BatchNorm train forward/backward through the finalize kernels
    B, H, W, C = 2, 7, 9, 20
    y = rn(B, H, W, C) * 2 + 1
    yq = y.clone().requires_grad_(True)
    bn = torch.nn.BatchNorm2d(C, momentum=0.1)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5); bn.bias.copy_(rn(C))
        bn.running_mean.copy_(rn(C)); bn.running_var.copy_(torch.rand(C, generator=g) + 0.5)
    rm, rv = D(bn.running_mean), D(bn.running_var)
    out = F.relu(bn(yq.permute(0, 3, 1, 2))).permute(0, 2, 3, 1)
    gg = rn(B, H, W, C)
    out.backward(gg)
    st = zstat(C, dev)
    st[:2 * C] = torch.stack([y.reshape(-1, C).double().sum(0), (y.reshape(-1, C).double() ** 2).sum(0)]).reshape(-1).to(dev) * 0.75
    st[6 * C:8 * C] = st[:2 * C] / 3          # the finalize kernels sum the copies
    sc, sh, mean, inv = (torch.zeros(C, device=dev) for _ in range(4))
    n = float(B * H * W)
    L.hrf_bn_finalize(st, D(bn.weight), D(bn.bias), rm, rv, n, 1e-5, 0.1, 1, sc, sh, mean, inv, C, s)
    o = torch.zeros(B, H, W, C, device=dev)
    L.hrf_affine_act_res(D(y), sc, sh, None, None, None, None, None, 1, 1, 0, o, B * H * W, C, None, 0.0, None, None, s)
    assert r(o, out) < TOL and r(rm, bn.running_mean) < TOL and r(rv, bn.running_var) < TOL
    # the same materialisation with BOTH BatchNorms finalised on load (Bottleneck tail: relu(bn3(y) + bn_ds(y2)))
    f1, t1 = make_fin(L, C, 733.0, dev, g)
    f2, t2 = make_fin(L, C, 733.0, dev, g)
    y2 = rn(B, H, W, C)
    o2 = torch.zeros(B, H, W, C, device=dev)
    L.hrf_affine_act_res(D(y), None, None, D(y2), None, None, None, None, 1, 1, 0, o2, B * H * W, C, None, 0.0, f1, f2, s)
    check_fin(t1)
    check_fin(t2)
    assert r(o2, F.relu(y * t1['ref_scale'].cpu() + t1['ref_shift'].cpu() + y2 * t2['ref_scale'].cpu() + t2['ref_shift'].cpu())) < TOL
    gk = torch.zeros(B, H, W, C, device=dev)
    gst = zstat(C, dev)
    L.hrf_act_bwd(D(gg), o, D(y), None, None, None, 1, 0, gk, None, None, gst, None, None, B * H * W, C, s)
    dg, db, cA, cB, cC = (torch.zeros(C, device=dev) for _ in range(5))
    L.hrf_bn_bwd_finalize(gst, None, D(bn.weight), mean, inv, n, 1, dg, db, cA, cB, cC, C, s)
    assert r(cA * gk + cB * D(y) + cC, yq.grad) < TOL and r(dg, bn.weight.grad) < TOL and r(db, bn.bias.grad) < TOL
    # ---- CrossFFN tail: res + rowscale*gelu(bn(y)) and its adjoint
    res, rs = rn(B, H, W, C), torch.tensor([0.0, 1.25])
    lnr = torch.zeros(B * H * W, 2, device=dev)
    L.hrf_affine_act_res(D(y), sc, sh, None, None, None, D(res), D(rs), H * W, 2, 1, o, B * H * W, C, lnr, 1e-6, None, None, s)
    ref = res + rs.view(B, 1, 1, 1) * F.gelu(y * sc.cpu() + sh.cpu())
    assert r(o, ref) < TOL
    f3, t3 = make_fin(L, C, 733.0, dev, g)           # CrossFFN tail with BN3 finalised on load
    o3_, lnr3 = torch.zeros(B, H, W, C, device=dev), torch.zeros(B * H * W, 2, device=dev)
    L.hrf_affine_act_res(D(y), None, None, None, None, None, D(res), D(rs), H * W, 2, 1, o3_, B * H * W, C, lnr3, 1e-6, f3, None, s)
    check_fin(t3)
    ref3_ = res + rs.view(B, 1, 1, 1) * F.gelu(y * t3['ref_scale'].cpu() + t3['ref_shift'].cpu())
    assert r(o3_, ref3_) < TOL and r(lnr3[:, 0], ref3_.reshape(-1, C).mean(-1)) < TOL
    rr_ = ref.reshape(-1, C)
    assert r(lnr[:, 0], rr_.mean(-1)) < TOL and r(lnr[:, 1], (rr_.var(-1, unbiased=False) + 1e-6).rsqrt()) < 1e-4
    uq = (y * sc.cpu() + sh.cpu()).requires_grad_(True)
    (F.gelu(uq) * rs.view(B, 1, 1, 1) * gg).sum().backward()
    L.hrf_act_bwd(D(gg), None, D(y), sc, sh, D(rs), H * W, 1, gk, None, None, gst.zero_(), None, None, B * H * W, C, s)
    assert r(gk, uq.grad) < TOL
    # ---- dropout / droppath arithmetic
    mask = torch.empty(B, H, W, C).bernoulli_(0.9, generator=g)
    L.hrf_scale_add(D(y), D(mask), 1 / 0.9, D(rs), H * W, D(res), D(gg), o, B * H * W, C, s)
    assert r(o, res + gg + y * mask / 0.9 * rs.view(B, 1, 1, 1)) < TOL
    # ---- LayerNorm (every channel-slot template: C/16 <= 2, 3, 5, 10, 20, 40)
    for rows, Cl in ((37, 18), (33, 36), (21, 78), (19, 144), (9, 312), (5, 624)):
        x = rn(rows, Cl) * 2 + 0.5
        xq = x.clone().requires_grad_(True)
        ln = torch.nn.LayerNorm(Cl, eps=1e-6)
        with torch.no_grad():
            ln.weight.copy_(torch.rand(Cl, generator=g) + 0.5); ln.bias.copy_(rn(Cl))
        da = rn(rows, Cl)
        ln(xq).backward(da)
        rsb = torch.zeros(rows, 2, device=dev)
        L.hrf_ln_stats(D(x), rows, Cl, 1e-6, rsb, s)
        assert r(rsb[:, 0], x.mean(-1)) < TOL and r(rsb[:, 1], (x.var(-1, unbiased=False) + 1e-6).rsqrt()) < TOL
        base = rn(rows, Cl)
        dx, dgl, dbl = D(base).clone(), torch.zeros(Cl, device=dev), torch.zeros(Cl, device=dev)
        L.hrf_ln_bwd(D(da), D(x), rsb, D(ln.weight), rows, Cl, dx, 1, dgl, dbl, 0, s)
        assert r(dx, xq.grad + base) < TOL and r(dgl, ln.weight.grad) < TOL and r(dbl, ln.bias.grad) < TOL
        dx2 = torch.full((rows, Cl), 7.0, device=dev)
        L.hrf_ln_bwd(D(da), D(x), rsb, D(ln.weight), rows, Cl, dx2, 0, dgl, dbl, 0, s)
        assert r(dx2, xq.grad) < TOL and r(dgl, 2 * ln.weight.grad) < TOL
        scr = torch.zeros(KC * 2 * Cl, device=dev)                      # replicated accumulators
        L.hrf_ln_bwd(D(da), D(x), rsb, D(ln.weight), rows, Cl, dx2, 0, scr, scr[Cl:], 2 * Cl, s)
        tot = scr.view(KC, 2 * Cl).sum(0)
        assert r(tot[:Cl], ln.weight.grad) < TOL and r(tot[Cl:], ln.bias.grad) < TOL
    # ---- act_bwd with three moment sets, narrow and > 256-channel rows
    for rows, Ca in ((45, 20), (301, 18), (7, 300), (3, 600)):
        dd, oo, ya, yb, yc = rn(rows, Ca), rn(rows, Ca), rn(rows, Ca), rn(rows, Ca), rn(rows, Ca)
        gk = torch.zeros(rows, Ca, device=dev)
        sa, sb, sc3 = (zstat(Ca, dev) for _ in range(3))
        L.hrf_act_bwd(D(dd), D(oo), D(ya), None, None, None, 1, 0, gk, D(yb), D(yc), sa, sb, sc3, rows, Ca, s)
        gref = (dd * (oo > 0)).double()
        assert r(gk, gref) < TOL
        for st_, y_ in ((sa, ya), (sb, yb), (sc3, yc)):
            assert r(fold(st_)[:Ca], gref.sum(0)) < TOL and r(fold(st_)[Ca:], (gref * y_.double()).sum(0)) < TOL
    # ---- odd channel counts: the dword kernels (vector paths need C % 2 == 0)
    rows, Ca = 33, 15
    dd, oo, ya = rn(rows, Ca), rn(rows, Ca), rn(rows, Ca)
    gk, sa = torch.zeros(rows, Ca, device=dev), zstat(Ca, dev)
    L.hrf_act_bwd(D(dd), D(oo), D(ya), None, None, None, 1, 0, gk, None, None, sa, None, None, rows, Ca, s)
    gref = (dd * (oo > 0)).double()
    assert r(gk, gref) < TOL and r(fold(sa)[Ca:], (gref * ya.double()).sum(0)) < TOL
    sco, sho, oo2 = torch.rand(Ca, generator=g) + 0.5, rn(Ca), torch.zeros(rows, Ca, device=dev)
    L.hrf_affine_act_res(D(dd), D(sco), D(sho), None, None, None, D(oo), None, 1, 1, 0, oo2, rows, Ca, None, 0.0, None, None, s)
    assert r(oo2, F.relu(dd * sco + sho + oo)) < TOL
    # ---- the same two kernels on tensors that are only 4-byte aligned (16-byte accesses on 4-byte aligned addresses)
    def unal(t):
        buf = torch.zeros(t.numel() + 1, device=dev)
        buf[1:] = t.reshape(-1).to(dev)
        return buf[1:].view(t.shape)
    rows, Ca = 45, 20
    dd, oo, ya = rn(rows, Ca), rn(rows, Ca), rn(rows, Ca)
    gk = unal(torch.zeros(rows, Ca))
    sa = zstat(Ca, dev)
    L.hrf_act_bwd(unal(dd), unal(oo), unal(ya), None, None, None, 1, 0, gk, None, None, sa, None, None, rows, Ca, s)
    gref = (dd * (oo > 0)).double()
    assert gk.data_ptr() % 16 != 0 and r(gk, gref) < TOL and r(fold(sa)[Ca:], (gref * ya.double()).sum(0)) < TOL
    scu, shu = torch.rand(Ca, generator=g) + 0.5, rn(Ca)
    ou = unal(torch.zeros(rows, Ca))
    L.hrf_affine_act_res(unal(dd), D(scu), D(shu), None, None, None, unal(oo), None, 1, 1, 0, ou, rows, Ca, None, 0.0, None, None, s)
    assert r(ou, F.relu(dd * scu + shu + oo)) < TOL
    # ---- cross-resolution exchange + bilinear adjoint (x2, x4, non-integer ratio)
    B, H, W, C = 2, 12, 20, 10
    x0 = rn(B, H, W, C).requires_grad_(True)
    ylo, ylo2 = rn(B, 6, 10, C).requires_grad_(True), rn(B, 3, 5, C).requires_grad_(True)
    ysame = rn(B, H, W, C).requires_grad_(True)
    scs = [torch.rand(C, generator=g) + 0.5 for _ in range(3)]
    shs = [rn(C) for _ in range(3)]
    up = lambda t, a, b: F.interpolate((t * a + b).permute(0, 3, 1, 2), size=(H, W), mode='bilinear',
                                       align_corners=False).permute(0, 2, 3, 1)
    ref = F.relu(x0 + up(ylo, scs[0], shs[0]) + up(ylo2, scs[1], shs[1]) + (ysame * scs[2] + shs[2]))
    gg = rn(B, H, W, C)
    ref.backward(gg)
    o = torch.zeros(B, H, W, C, device=dev)
    L.hrf_fuse_sum(1, D(x0), None, None, 0, 0, 3, D(ylo), D(scs[0]), D(shs[0]), 6, 10, 3, D(ylo2), D(scs[1]),
                   D(shs[1]), 3, 5, 2, D(ysame), D(scs[2]), D(shs[2]), 0, 0, o, B, H, W, C, None, s)
    assert r(o, ref) < TOL
    # terms 1 and 3 with their BatchNorms finalised on load (an array of four hrf_bn_fin_t)
    fa, ta = make_fin(L, C, 611.0, dev, g)
    fb, tb = make_fin(L, C, 611.0, dev, g)
    fins = (_lib.BnFin * 4)()
    fins[1], fins[3] = fa, fb
    of = torch.zeros(B, H, W, C, device=dev)
    L.hrf_fuse_sum(1, D(x0), None, None, 0, 0, 3, D(ylo), None, None, 6, 10, 3, D(ylo2), D(scs[1]),
                   D(shs[1]), 3, 5, 2, D(ysame), None, None, 0, 0, of, B, H, W, C, fins, s)
    check_fin(ta)
    check_fin(tb)
    reff = F.relu(x0 + up(ylo, ta['ref_scale'].cpu(), ta['ref_shift'].cpu()) + up(ylo2, scs[1], shs[1])
                  + (ysame * tb['ref_scale'].cpu() + tb['ref_shift'].cpu()))
    assert r(of, reff) < TOL
    gk = torch.zeros(B, H, W, C, device=dev)
    st3 = zstat(C, dev)
    L.hrf_act_bwd(D(gg), o, D(ysame), None, None, None, 1, 0, gk, None, None, st3, None, None, B * H * W, C, s)
    assert r(gk, x0.grad) < TOL
    assert r(fold(st3)[C:], (x0.grad.reshape(-1, C).double() * ysame.detach().reshape(-1, C).double()).sum(0)) < TOL
    du, stl = torch.zeros(B, 6, 10, C, device=dev), zstat(C, dev)
    L.hrf_bilinear_up_bwd(gk, C, 0, B, H, W, C, D(ylo), 6, 10, du, stl, s)
    assert r(du * D(scs[0]), ylo.grad) < TOL and r(fold(stl)[:C], du.reshape(-1, C).double().sum(0)) < TOL
    du2 = torch.zeros(B, 3, 5, C, device=dev)
    L.hrf_bilinear_up_bwd(gk, C, 0, B, H, W, C, D(ylo2), 3, 5, du2, None, s)
    assert r(du2 * D(scs[1]), ylo2.grad) < TOL
    ylo3 = rn(1, 8, 4, C).requires_grad_(True)
    ref3 = F.interpolate(ylo3.permute(0, 3, 1, 2), size=(15, 7), mode='bilinear', align_corners=False).permute(0, 2, 3, 1)
    g3 = rn(1, 15, 7, C)
    ref3.backward(g3)
    du3 = torch.zeros(1, 8, 4, C, device=dev)
    L.hrf_bilinear_up_bwd(D(g3), C, 0, 1, 15, 7, C, None, 8, 4, du3, None, s)
    o3 = torch.zeros(1, 15, 7, C, device=dev)
    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    L.hrf_fuse_sum(3, D(ylo3), one, zero, 8, 4, 0, None, None, None, 0, 0, 0, None, None, None, 0, 0, 0, None, None,
                   None, 0, 0, o3, 1, 15, 7, C, None, s)
    assert r(o3, F.relu(ref3)) < TOL and r(du3, ylo3.grad) < TOL
    # ---- AdamW vs torch.optim.AdamW (3 steps, device-side step counter)
    n = 1000
    p, gr = rn(n), rn(n)
    pp = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([pp], lr=3e-4, weight_decay=0.01)
    pk, gk2, m, v, state = D(p).clone(), D(gr), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(4, device=dev)
    for _ in range(3):
        pp.grad = gr.clone()
        opt.step()
        L.hrf_adamw_tick(state, 0.9, 0.999, s)
        L.hrf_adamw(pk, gk2, m, v, None, n, 3e-4, 0.9, 0.999, 1e-8, 0.01, state, 1.0, s)
    assert r(pk, pp.data) < TOL and float(state[2]) == 3.0


# ------------------------------------------------------------------ wide packed-weight 3x3 convolution
C3W_CASES = [  # B, H, W, Cin, Cout, forced channel groups per block (0 = dispatcher's choice)
    (1, 8, 16, 32, 64, 0), (2, 11, 21, 64, 128, 0), (1, 9, 17, 64, 256, 4), (1, 5, 7, 96, 64, 1), (1, 3, 40, 64, 320, 0),
    (1, 10, 9, 64, 128, 2)]


def run_conv3w(case, backend):
    """hrf_conv3_pack + hrf_conv3_packed (forward, and backward-data on the flipped pack) vs F.conv2d / autograd."""
    B, H, W, Cin, Cout, wn = case
    dev = use_backend(backend)
    try:
        L, s = _lib.lib(), _lib.stream_ptr()
        g = torch.Generator().manual_seed(7)
        x = torch.randn(B, Cin, H, W, generator=g).requires_grad_(True)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).requires_grad_(True)
        bias = torch.randn(Cout, generator=g)
        y = F.conv2d(x, w, bias, padding=1)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        D = lambda t: t.detach().to(dev).contiguous()
        L.hrf_debug_knob(24, wn)
        xk, wk = D(nhwc(x)), D(w)
        wp = torch.empty(9 * Cout * Cin, device=dev)
        yk = torch.full((B, H, W, Cout + 4), 7.0, device=dev)          # ldY > N: the pad columns must stay untouched
        L.hrf_conv3_pack(wk, Cout, Cin, 0, wp, s)
        assert torch.equal(wp.view(9, Cout, Cin).cpu(), w.detach().permute(2, 3, 0, 1).reshape(9, Cout, Cin))
        L.hrf_conv3_packed(xk, Cin, wp, D(bias), yk, Cout + 4, 0, B, H, W, Cin, Cout, s)
        assert r(yk[..., :Cout], nhwc(y)) < TOL
        assert float((yk[..., Cout:] - 7.0).abs().max()) == 0.0
        if Cin % 64 == 0 and Cout % 32 == 0:
            prev = torch.randn(B, H, W, Cin, generator=g)
            dx = D(prev).clone()
            L.hrf_conv3_pack(wk, Cout, Cin, 1, wp, s)
            L.hrf_conv3_packed(D(nhwc(dy)), Cout, wp, None, dx, Cin, 1, B, H, W, Cout, Cin, s)
            assert r(dx - D(prev), nhwc(x.grad)) < TOL
        if Cin % 64 == 0 and Cout % 128 == 0:
            dw0, db0 = torch.randn(w.shape, generator=g), torch.randn(Cout, generator=g)
            dw, db = D(dw0).clone(), D(db0).clone()
            scr = torch.empty(L.hrf_conv3_wgrad_wide_scratch(B, H, W, Cin, Cout), device=dev)
            L.hrf_conv3_wgrad_wide(D(nhwc(dy)), Cout, xk, Cin, B, H, W, Cin, Cout, dw, db, scr, s)
            assert r(dw - D(dw0), w.grad) < TOL
            assert r(db - D(db0), dy.sum((0, 2, 3))) < TOL
        with pytest.raises(_lib.HRFuserHipError):                     # unsupported K: refused, never a silent fallback
            L.hrf_conv3_packed(xk, Cin, wp, None, yk, Cout + 4, 0, B, H, W, Cin - 1, Cout, s)
    finally:
        _lib.lib().hrf_debug_knob(24, 0)
        use_backend('hip')


RG_CASES = [(200, 270, 256, 0), (131, 48, 64, 0), (300, 1170, 256, 4), (128, 256, 270, 2), (77, 64, 40, 0)]   # M, K, N, groups


def run_rowgemm(case, backend):
    """hrf_rowgemm_pack + hrf_rowgemm: forward y = x W^T + b on zero-padded K, backward dx = dy W on padded N."""
    M, K, N, wn = case
    dev = use_backend(backend)
    try:
        L, s = _lib.lib(), _lib.stream_ptr()
        g = torch.Generator().manual_seed(9)
        x = torch.randn(M, K, generator=g)
        w = torch.randn(N, K, generator=g) / K ** 0.5
        bias = torch.randn(N, generator=g)
        Kp, Np = (K + 15) // 16 * 16, (N + 15) // 16 * 16
        L.hrf_debug_knob(24, wn)
        xk = torch.zeros(M, Kp, device=dev); xk[:, :K] = x.to(dev)
        wp = torch.empty(Np * Kp, device=dev)
        bk = torch.zeros(Np, device=dev); bk[:N] = bias.to(dev)
        L.hrf_rowgemm_pack(w.to(dev).contiguous(), N, K, 0, Np, Kp, wp, s)
        y = torch.full((M, Np + 4), 3.0, device=dev)
        L.hrf_rowgemm(xk, Kp, wp, bk, y, Np + 4, 0, M, Kp, Np, s)
        ref = x @ w.t() + bias
        assert r(y[:, :N], ref) < TOL
        assert float(y[:, N:Np].abs().max()) == 0.0 if Np > N else True        # zero weight rows, zero bias
        assert float((y[:, Np:] - 3.0).abs().max()) == 0.0
        # backward-data operand: dx[m][k] = sum_n dy[m][n] w[n][k], accumulated into an existing gradient
        dy = torch.randn(M, N, generator=g)
        dyk = torch.zeros(M, Np, device=dev); dyk[:, :N] = dy.to(dev)
        wq = torch.empty(Kp * Np, device=dev)
        L.hrf_rowgemm_pack(w.to(dev).contiguous(), N, K, 1, Kp, Np, wq, s)
        prev = torch.randn(M, Kp, generator=g)
        dx = prev.to(dev).clone()
        L.hrf_rowgemm(dyk, Np, wq, None, dx, Kp, 1, M, Np, Kp, s)
        assert r(dx[:, :K] - prev[:, :K].to(dev), dy @ w) < TOL
    finally:
        _lib.lib().hrf_debug_knob(24, 0)
        use_backend('hip')


@pytest.mark.parametrize('case', RG_CASES[:2] + RG_CASES[3:], ids=str)
def test_rowgemm_emul(case):
    run_rowgemm(case, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('case', RG_CASES + [(61440, 272, 256, 0)], ids=str)
def test_rowgemm_gpu(case):
    run_rowgemm(case, 'hip')


@pytest.mark.parametrize('case', C3W_CASES[:4] + C3W_CASES[-1:], ids=str)
def test_conv3w_emul(case):
    run_conv3w(case, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('case', C3W_CASES + [(2, 96, 160, 256, 256, 0)], ids=str)
def test_conv3w_gpu(case):
    run_conv3w(case, 'hip')


# ------------------------------------------------------------------ LDS-tiled weight gradient
# csrc/wgrad_tiled.hip serves wide 1x1 problems (>= 32 channels on both sides, >= 256 pixels, >= 60 % tile efficiency); the debug
# knob 8 = 2 forces it on every stride-1 1x1 problem: ragged widths on either side (the 128-wide side is chosen per problem),
# LayerNorm / BatchNorm + GELU / ReLU on load, BatchNorm-backward coefficients, fewer pixels than one chunk
WT_CASES = LIN2_CASES + [(2, 16, 24, 64, 256, 1, 1, 2, True, True), (3, 40, 50, 16, 40, 1, 1, 1, False, True),
                         (1, 6, 7, 330, 300, 1, 1, 4, True, False)]


def run_wgrad_tiled(case, backend):
    use_backend(backend)
    L = _lib.lib()
    L.hrf_debug_knob(8, 2)
    try:
        run_conv(case, backend)
    finally:
        L.hrf_debug_knob(8, 0)


# ------------------------------------------------------------------ grouped weight-gradient launches
def run_wgrad_group(backend):
    """hrf_wgrad_group_begin/_end: queued problems (several shapes / kernel variants, > 16 of one variant so that the
    group is split) must produce what separate launches produce."""
    dev = use_backend(backend)
    try:
        L, s = _lib.lib(), _lib.stream_ptr()
        g = torch.Generator().manual_seed(3)
        probs = []
        shapes = [(1, 6, 10, 18, 72, 1)] * 18 + [(1, 6, 10, 72, 18, 1)] * 3 + [(2, 5, 7, 64, 64, 3)] * 2 + [(1, 4, 9, 36, 36, 1)] + \
            [(1, 16, 17, 64, 256, 1)] * 2 + [(1, 16, 18, 256, 64, 1)]      # (the last three: the LDS-tiled kernel)
        for (B, H, W, Cin, Cout, KH) in shapes:
            x = torch.randn(B, H, W, Cin, generator=g).to(dev)
            dy = torch.randn(B, H, W, Cout, generator=g).to(dev)
            probs.append((B, H, W, Cin, Cout, KH, x, dy))

        def run(grouped):
            outs = []
            if grouped:
                L.hrf_wgrad_group_begin()
            for (B, H, W, Cin, Cout, KH, x, dy) in probs:
                dw = torch.zeros(Cout, Cin, KH, KH, device=dev)
                db = torch.zeros(Cout, device=dev)
                L.hrf_conv_bwd_weight(dy, Cout, 0, None, None, None, None, x, H * W * Cin, W * Cin, Cin, 1, B, H, W, Cin,
                                      KH, 1, Cout, 0, None, None, None, dw, db, s)
                outs.append((dw, db))
            if grouped:
                assert all(float(dw.abs().max()) == 0.0 for dw, _ in outs)      # nothing launched yet
                L.hrf_wgrad_group_end(s)
            return outs
        ref, got = run(False), run(True)
        for (dw0, db0), (dw1, db1) in zip(ref, got):
            assert float(dw0.abs().max()) > 0
            assert r(dw1, dw0) < TOL and r(db1, db0) < TOL
    finally:
        use_backend('hip')


def test_wgrad_group_emul():
    run_wgrad_group('emul')


@pytest.mark.parametrize('case', WT_CASES, ids=str)
def test_wgrad_tiled_emul(case):
    run_wgrad_tiled(case, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('case', WT_CASES + [(1, 70, 66, 64, 256, 1, 1, 2, True, True), (1, 66, 70, 256, 64, 1, 1, 1, True, False),
                                             (2, 48, 80, 312, 78, 1, 1, 3, True, True), (2, 48, 80, 78, 312, 1, 1, 4, True, True)], ids=str)
def test_wgrad_tiled_gpu(case):
    run_wgrad_tiled(case, 'hip')


@pytest.mark.gpu
def test_wgrad_group_gpu():
    run_wgrad_group('hip')


# ------------------------------------------------------------------ emulator (CPU suite)
@pytest.mark.parametrize('case', CONV_CASES[:8] + CONV_CASES[12:13] + CONV_CASES[-4:], ids=str)      # ([12]: the split over K)
def test_conv_emul(case):
    run_conv(case, 'emul')


@pytest.mark.parametrize('case', DW_CASES, ids=str)
def test_dwconv_emul(case):
    run_dw(case, 'emul')


@pytest.mark.parametrize('mode', [0, 2, 3])
def test_dwconv_lane4_modes_emul(mode):
    """hrf_debug_knob(40): the float4-lane depthwise forward (not the default: csrc/dwconv.hip) with automatic / 8-row / 4-row
    tiles - same results as the one-channel-lane kernel the other tests run"""
    use_backend('emul')
    L = _lib.lib()
    L.hrf_debug_knob(40, mode)
    try:
        run_dw((2, 9, 11, 72, 1, 3, True, True, True), 'emul')
        run_dw((1, 9, 33, 52, 1, 2, False, True, True), 'emul')          # one 13-lane slab, ReLU on load
    finally:
        L.hrf_debug_knob(40, 1)


@pytest.mark.parametrize('case', ATTN_CASES[:5], ids=str)
def test_attention_emul(case):
    run_attn(case, 'emul')


def test_pointwise_emul():
    run_pointwise('emul')


# ------------------------------------------------------------------ MI355X (product path)
@pytest.mark.gpu
@pytest.mark.parametrize('case', CONV_CASES, ids=str)
def test_conv_gpu(case):
    run_conv(case, 'hip')


# the packed-weight front-end engine (csrc/conv3x_engine.hip): stems / Bottleneck conv2 (64 -> 64, stride 1 and the stride-2
# backward by parity classes), the transitions' backward (18 / 36 -> 256 channels), ragged grids and channel counts
C3X_CASES = [  # B,H,W,Cin,Cout,KH,stride,tf,bnb,epi
    (2, 10, 13, 64, 64, 3, 1, 2, True, True),
    (1, 9, 35, 64, 64, 3, 2, 2, True, True),       # stride 2: forward over parity planes (two 32-channel slabs), backward over parity classes
    (1, 7, 18, 96, 40, 3, 1, 1, True, False),      # two halo slabs (64 + 32), ragged output channels, accumulate epilogue
    (1, 6, 5, 256, 18, 3, 1, 0, True, False),      # transition backward: 18 -> 256 channels (ragged K, four channel blocks)
    (1, 9, 7, 72, 36, 3, 2, 3, True, True),        # GELU on load / GELU' epilogue, stride-2 backward 36 -> 72
    (1, 7, 9, 40, 96, 3, 1, 2, True, True),        # backward with two halo slabs of dY (96 channels), ragged 40 outputs
    (1, 18, 33, 64, 64, 3, 1, 0, False, False),
    (1, 11, 8, 80, 36, 3, 2, 1, True, False),      # stride-2 forward: three slabs (32 + 32 + 16), ragged 36 outputs, odd source rows
]


def run_im2col(backend):
    """hrf_im2col3x3 (the stem's first convolution as a row GEMM): patches == F.unfold, NCHW and channels-last inputs, both strides;
    conv = rows . w.view(Cout, 9 Cin)^T through hrf_conv_fwd(KH = 1), grad_weight through hrf_conv_bwd_weight(KH = 1)"""
    dev = use_backend(backend)
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    for (B, C, H, W, stride, cl) in [(2, 3, 9, 14, 2, False), (1, 3, 8, 7, 1, True), (2, 1, 5, 6, 2, False)]:
        x = torch.randn(B, C, H, W, generator=g)
        xs = x.contiguous(memory_format=torch.channels_last) if cl else x
        Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
        ref = F.unfold(x, 3, padding=1, stride=stride).transpose(1, 2).reshape(B, Ho, Wo, 9 * C)       # [.., ci * 9 + tap]
        xd = xs.to(dev)
        sB, sC, sY, sX = xd.stride()
        cols = torch.full((B, Ho, Wo, 9 * C), float('nan'), device=dev)
        L.hrf_im2col3x3(xd, sB, sY, sX, sC, B, H, W, C, stride, cols, 9 * C, _lib.stream_ptr())
        assert torch.equal(cols.cpu(), ref)
        Cout = 20
        w = torch.randn(Cout, C, 3, 3, generator=g) * 0.3
        y = torch.zeros(B, Ho, Wo, Cout, device=dev)
        K9 = 9 * C
        L.hrf_conv_fwd(cols, Ho * Wo * K9, Wo * K9, K9, 1, B, Ho, Wo, K9, w.to(dev), None, 1, 1, Cout, y, Cout, 0, None, None, 0, 0, None,
                       None, None, None, None, None, 0.0, _lib.stream_ptr())
        wq = w.clone().requires_grad_(True)
        yr = F.conv2d(x, wq, None, stride, 1)
        assert r(y, nhwc(yr.detach())) < TOL
        du = torch.randn(B, Ho, Wo, Cout, generator=g)
        yr.backward(du.permute(0, 3, 1, 2))
        dw = torch.zeros(Cout, C, 3, 3, device=dev)
        L.hrf_conv_bwd_weight(du.to(dev), Cout, 0, None, None, None, None, cols, Ho * Wo * K9, Wo * K9, K9, 1, B, Ho, Wo, K9, 1, 1, Cout, 0,
                              None, None, None, dw, None, _lib.stream_ptr())
        assert r(dw, wq.grad) < TOL


def test_im2col_emul():
    run_im2col('emul')


@pytest.mark.gpu
def test_im2col_gpu():
    run_im2col('hip')


@pytest.mark.parametrize('case', C3X_CASES[:6] + C3X_CASES[7:], ids=str)
def test_conv3x_emul(case):
    run_conv(case, 'emul', packed=True)


@pytest.mark.gpu
@pytest.mark.parametrize('case', C3X_CASES + [(2, 50, 130, 64, 64, 3, 1, 2, True, True), (1, 33, 47, 256, 36, 3, 2, 0, True, False),
                                              (2, 37, 70, 64, 64, 3, 2, 2, True, True)], ids=str)
def test_conv3x_gpu(case):
    run_conv(case, 'hip', packed=True)


@pytest.mark.gpu
@pytest.mark.parametrize('case', DW_CASES, ids=str)
def test_dwconv_gpu(case):
    run_dw(case, 'hip')


@pytest.mark.gpu
@pytest.mark.parametrize('mode', [0, 2, 3])
def test_dwconv_lane4_modes_gpu(mode):
    use_backend('hip')
    L = _lib.lib()
    L.hrf_debug_knob(40, mode)
    try:
        run_dw((2, 19, 37, 72, 1, 3, True, True, True), 'hip')
        run_dw((1, 13, 18, 144, 1, 3, True, True, True), 'hip')
    finally:
        L.hrf_debug_knob(40, 1)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ATTN_CASES, ids=str)
def test_attention_gpu(case):
    run_attn(case, 'hip')


@pytest.mark.gpu
def test_pointwise_gpu():
    run_pointwise('hip')
