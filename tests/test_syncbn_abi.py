"""Direct C-ABI tests of the packed SyncBN entry points (reached only through whole-net tests before) - incl.
BatchNorms wider than HRF_FIN_MAXC = 576 channels (HRFuser-B's CrossFFN hidden widths 624 / 1248 / 2496, which take the stand-alone
packed finalize instead of the on-load one) - and of hrf_nearest_up_bwd (the adjoint of the HRNet-based exchange's up-sampling)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from hrfuser_amd import _lib
from helpers import use_backend
from test_kernels import KC, _rep_moments, r


def _packed(backend):
    dev = use_backend(backend)
    L = _lib.lib()
    s = _lib.stream_ptr()
    P = _lib._ptr
    g = torch.Generator().manual_seed(11)
    Cs = [18, 624, 72, 1248]                                   # two of them beyond the on-load limit
    rows = [977.0, 480.0, 1920.0, 60.0]
    n = len(Cs)
    stats, gstats, ref = [], [], []
    for C, cnt in zip(Cs, rows):
        mean = torch.randn(C, generator=g) * 0.3
        var = torch.rand(C, generator=g) * 0.8 + 0.4
        stats.append(_rep_moments(torch.stack([mean, var + mean ** 2]), cnt, g, dev))
        gstats.append(_rep_moments(torch.randn(2, C, generator=g) * 0.2, cnt, g, dev))
    # ---- hrf_bn_pack: the KC replicated copies of every layer folded into 2*C doubles each, the row counts behind them
    total = sum(2 * C for C in Cs)
    packed = torch.zeros(total + n, dtype=torch.float64, device=dev)
    ptrs = (ctypes.c_void_p * n)(*[P(t) for t in stats])
    cs = (ctypes.c_int * n)(*Cs)
    rw = (ctypes.c_double * n)(*rows)
    L.hrf_bn_pack(ptrs, cs, n, rw, packed, s)
    off = 0
    for C, t in zip(Cs, stats):
        assert r(packed[off:off + 2 * C], t.view(KC, 2 * C).sum(0)) < 1e-12
        off += 2 * C
    assert packed[total:].tolist() == rows
    # ---- hrf_bn_finalize_packed == hrf_bn_finalize on the folded sums, with the count read from the device (count_ptr)
    fins = (_lib.BnFin * n)()
    bufs = []
    off = 0
    for i, (C, cnt) in enumerate(zip(Cs, rows)):
        t = dict(gamma=(torch.rand(C, generator=g) + 0.5).to(dev), beta=(torch.randn(C, generator=g) * 0.3).to(dev),
                 rm=torch.randn(C, generator=g).to(dev), rv=(torch.rand(C, generator=g) + 0.5).to(dev))
        for k in ('scale', 'shift', 'mean', 'invstd'):
            t[k], t['ref_' + k] = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        t['ref_rm'], t['ref_rv'] = t['rm'].clone(), t['rv'].clone()
        L.hrf_bn_finalize(stats[i], t['gamma'], t['beta'], t['ref_rm'], t['ref_rv'], cnt, 1e-5, 0.1, 1, t['ref_scale'], t['ref_shift'],
                          t['ref_mean'], t['ref_invstd'], C, s)
        fins[i] = _lib.BnFin(None, P(t['gamma']), P(t['beta']), P(t['rm']), P(t['rv']), P(t['scale']), P(t['shift']), P(t['mean']),
                             P(t['invstd']), 1.0, 1e-5, 0.1, 1, 1, C, 1, packed.data_ptr() + 8 * (total + i))
        bufs.append(t)
        off += 2 * C
    L.hrf_bn_finalize_packed(fins, n, packed, s)
    for t in bufs:
        for k in ('scale', 'shift', 'mean', 'invstd', 'rm', 'rv'):
            assert r(t[k], t['ref_' + k]) < 1e-6, k
    # ---- hrf_bn_bwd_finalize_packed == hrf_bn_bwd_finalize; pgrad_scale = 1/world on the all-reduced sums (no rank-local copy)
    gp = torch.zeros(total, dtype=torch.float64, device=dev)
    gptrs = (ctypes.c_void_p * n)(*[P(t) for t in gstats])
    L.hrf_bn_pack(gptrs, cs, n, None, gp, s)
    bfins = (_lib.BnBFin * n)()
    bb = []
    for i, (C, cnt) in enumerate(zip(Cs, rows)):
        t = dict(gamma=bufs[i]['gamma'], mean=bufs[i]['mean'], invstd=bufs[i]['invstd'],
                 dgamma=torch.randn(C, generator=g).to(dev), dbeta=torch.randn(C, generator=g).to(dev))
        for k in ('cA', 'cB', 'cC'):
            t[k], t['ref_' + k] = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        t['dg0'], t['db0'] = t['dgamma'].clone(), t['dbeta'].clone()
        t['ref_dgamma'], t['ref_dbeta'] = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        L.hrf_bn_bwd_finalize(gstats[i], None, t['gamma'], t['mean'], t['invstd'], cnt, 1, t['ref_dgamma'], t['ref_dbeta'], t['ref_cA'],
                              t['ref_cB'], t['ref_cC'], C, s)
        bfins[i] = _lib.BnBFin(None, P(t['gamma']), P(t['mean']), P(t['invstd']), P(t['dgamma']), P(t['dbeta']), P(t['cA']), P(t['cB']),
                               P(t['cC']), cnt, 1, 1, C, 1, None, 0.5)
        bb.append(t)
    L.hrf_bn_bwd_finalize_packed(bfins, n, gp, None, s)
    for t in bb:
        for k in ('cA', 'cB', 'cC'):
            assert r(t[k], t['ref_' + k]) < 1e-6, k
        assert r(t['dgamma'] - t['dg0'], 0.5 * t['ref_dgamma']) < 1e-5 and r(t['dbeta'] - t['db0'], 0.5 * t['ref_dbeta']) < 1e-5


def test_packed_syncbn_emul():
    _packed('emul')


@pytest.mark.gpu
def test_packed_syncbn_gpu():
    _packed('hip')


def _nearest(backend):
    dev = use_backend(backend)
    L = _lib.lib()
    s = _lib.stream_ptr()
    g = torch.Generator().manual_seed(4)
    for B, Hs, Ws, f, C in ((2, 3, 5, 2, 18), (1, 2, 3, 4, 36), (2, 1, 2, 8, 20)):
        H, W = Hs * f, Ws * f
        ylow = torch.randn(B, Hs, Ws, C, generator=g)
        gout = torch.randn(B, H, W, C, generator=g)
        yq = ylow.permute(0, 3, 1, 2).clone().requires_grad_(True)
        F.interpolate(yq, scale_factor=f, mode='nearest').backward(gout.permute(0, 3, 1, 2))
        ref = yq.grad.permute(0, 2, 3, 1)
        du = torch.zeros(B, Hs, Ws, C, device=dev)
        st = torch.zeros(KC * 2 * C, dtype=torch.float64, device=dev)
        L.hrf_nearest_up_bwd(gout.to(dev), C, 0, B, H, W, C, ylow.to(dev), Hs, Ws, du, st, s)
        assert r(du, ref) < 1e-6
        tot = st.view(KC, 2 * C).sum(0)
        assert r(tot[:C], ref.double().sum((0, 1, 2))) < 1e-6 and r(tot[C:], (ref.double() * ylow.double()).sum((0, 1, 2))) < 1e-6


def test_nearest_up_bwd_emul():
    _nearest('emul')


@pytest.mark.gpu
def test_nearest_up_bwd_gpu():
    _nearest('hip')
