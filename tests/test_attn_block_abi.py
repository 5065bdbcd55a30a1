"""Direct C-ABI tests of the fused window-attention block and its helpers (VERDICT r2 "What's weak" #2): hrf_attn_block_fwd /
hrf_attn_block_bwd, hrf_fold_slots, hrf_rpb_grad on EDGE grids (one window, H or W < 7, B = 3, grids whose window count is
not a multiple of anything), hrf_bn_pack / hrf_bn_finalize_packed / hrf_bn_bwd_finalize_packed with C > HRF_FIN_MAXC, and
hrf_nearest_up_bwd - every entry point called through ctypes with raw buffers, against plain fp64 torch math of the
reference's formulas (hrformer.py:96-131,184-236,365-373; hrfuser_hrformer_based.py:106-151,189-248,305-317)."""
import ctypes

import pytest
import torch

import hrfuser_oracle as O
from helpers import LN, NORM, use_backend
from hrfuser_amd import _lib

KC = _lib.STAT_COPIES


def r(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _reference(blk, x, xkv, res2, cross, with_ffn):
    """out = res (+ res2) + out_proj(attn(LN_q(x), LN_kv(xkv))) and h1 = conv1x1(LN_2(out)), NLC rows (fp64)."""
    B, C, H, W = x.shape
    t = O.nchw_to_nlc(x)
    if cross:
        z = O.nchw_to_nlc(xkv)
        out = t + (z if res2 else 0) + blk.attn[0](blk.norm1[0](t), blk.norm2[0](z), H, W)
        ln2 = blk.norm3
    else:
        out = t + blk.attn(blk.norm1(t), H, W)
        ln2 = blk.norm2
    h1 = None
    if with_ffn:
        h1 = blk.ffn.layers[0](O.nlc_to_nchw(ln2(out), H, W)).permute(0, 2, 3, 1).reshape(B, H * W, -1)
    return out, h1


def _run(C, heads, B, H, W, cross, with_ffn, backward, backend, tail=False):
    """tail: the block input is formed on load, x = tail_res + rowscale[b] * GELU(scale * tail_raw + shift) (the CrossFFN tail
    of the preceding block, hrformer.py:371-372); the backward also emits tail_du and the tail BatchNorm's moments."""
    dev = use_backend(backend)
    L = _lib.lib()
    if cross:
        blk = O.HRFuserFusionBlock(C, heads, 4, NORM, LN, 0.0, 1, 0.0)
        msa, lnq, lnkv, ln2 = blk.attn[0].attn, blk.norm1[0], blk.norm2[0], blk.norm3
    else:
        blk = O.HRFormerBlock(C, heads, 4, NORM, LN)
        msa, lnq, lnkv, ln2 = blk.attn.attn, blk.norm1, blk.norm1, blk.norm2
    O.seeded_fill_(blk, 5)
    blk = blk.double()
    g = torch.Generator().manual_seed(3)
    if tail:
        assert not cross
        t_res = torch.randn(B, C, H, W, generator=g).double().requires_grad_(True)
        t_raw = torch.randn(B, C, H, W, generator=g).double()
        t_sc, t_sh = (torch.rand(C, generator=g) + 0.5).double(), torch.randn(C, generator=g).double()
        t_rs = torch.tensor([1.25, 0.0, 1.25][:B] + [1.25] * max(0, B - 3)).double()
        t_u = (t_raw * t_sc.view(1, C, 1, 1) + t_sh.view(1, C, 1, 1)).requires_grad_(True)
        x = t_res + t_rs.view(B, 1, 1, 1) * torch.nn.functional.gelu(t_u)
    else:
        x = torch.randn(B, C, H, W, generator=g).double().requires_grad_(True)
    xkv = torch.randn(B, C, H, W, generator=g).double().requires_grad_(True) if cross else x
    out_ref, h1_ref = _reference(blk, x, xkv, True, cross, with_ffn)

    f32 = lambda t: t.detach().float().contiguous().to(dev)
    rows = lambda t: f32(t.permute(0, 2, 3, 1).reshape(B * H * W, C))
    xq_d = rows(x) if not tail else torch.full((B * H * W, C), float('nan'), device=dev)     # tail: written by the launch
    xkv_d = rows(xkv) if cross else xq_d
    P = _lib._ptr
    keep = []

    def dp(t):
        d = f32(t)
        keep.append(d)
        return d
    a = _lib.AttnBlock()
    a.B, a.H, a.W, a.C, a.heads = B, H, W, C, heads
    a.xq, a.xkv = P(xq_d), P(xkv_d)
    a.lnq_g, a.lnq_b, a.lnkv_g, a.lnkv_b, a.ln_eps = P(dp(lnq.weight)), P(dp(lnq.bias)), P(dp(lnkv.weight)), P(dp(lnkv.bias)), 1e-6
    if cross:
        wq, bq, wk, bk, wv, bv = (dp(msa.q_proj.weight), dp(msa.q_proj.bias), dp(msa.k_proj.weight), dp(msa.k_proj.bias),
                                  dp(msa.v_proj.weight), dp(msa.v_proj.bias))
        a.wq, a.bq, a.wk, a.bk, a.wv, a.bv = P(wq), P(bq), P(wk), P(bk), P(wv), P(bv)
    else:
        wqkv, bqkv = dp(msa.qkv.weight), dp(msa.qkv.bias)
        a.wq, a.bq = wqkv.data_ptr(), bqkv.data_ptr()
        a.wk, a.bk = wqkv.data_ptr() + 4 * C * C, bqkv.data_ptr() + 4 * C
        a.wv, a.bv = wqkv.data_ptr() + 8 * C * C, bqkv.data_ptr() + 8 * C
    rpb, wo, bo = dp(msa.relative_position_bias_table), dp(msa.out_proj.weight), dp(msa.out_proj.bias)
    a.rpb, a.wo, a.bo = P(rpb), P(wo), P(bo)
    a.res, a.res2 = P(xq_d), (P(xkv_d) if cross else None)
    a.mask, a.mscale, a.rowscale, a.rows_per_sample = None, 1.0, None, H * W
    if tail:
        tres_d, traw_d, tsc_d, tsh_d, trs_d = rows(t_res), rows(t_raw), dp(t_sc), dp(t_sh), dp(t_rs)
        a.tail_res, a.tail_raw, a.tail_scale, a.tail_shift, a.tail_rowscale, a.x_out = P(tres_d), P(traw_d), P(tsc_d), P(tsh_d), P(trs_d), P(xq_d)
    out = torch.full((B * H * W, C), float('nan'), device=dev)
    a.out = P(out)
    N1 = 4 * C
    h1 = stats = None
    if with_ffn:
        conv1 = blk.ffn.layers[0]
        w1, b1 = dp(conv1.weight.reshape(N1, C)), dp(conv1.bias)
        h1 = torch.full((B * H * W, N1), float('nan'), device=dev)
        stats = torch.zeros(KC * 2 * N1, dtype=torch.float64, device=dev)
        a.ln2_g, a.ln2_b, a.out_eps = P(dp(ln2.weight)), P(dp(ln2.bias)), 1e-6
        a.w1, a.b1, a.h1, a.stats1, a.hidden = P(w1), P(b1), P(h1), P(stats), N1
    L.hrf_attn_block_fwd(a, _lib.stream_ptr())
    assert r(out.reshape(B, H * W, C), out_ref) < 2e-5
    if tail:
        assert r(xq_d.reshape(B, H, W, C).permute(0, 3, 1, 2), x) < 1e-5        # the formed rows, written for the backward
    if with_ffn:
        assert r(h1.reshape(B, H * W, N1), h1_ref) < 2e-5
        st = stats.view(KC, 2, N1).sum(0)
        assert r(st[0], h1_ref.sum((0, 1))) < 1e-4 and r(st[1], (h1_ref ** 2).sum((0, 1))) < 1e-4
    if not backward:
        return

    # ---- backward: loss = <out, gout> + <h1, du1>
    gout = torch.randn(B, H * W, C, generator=g).double()
    du1 = torch.randn(B, H * W, N1, generator=g).double() if with_ffn else None
    loss = (out_ref * gout).sum() + ((h1_ref * du1).sum() if with_ffn else 0.0)
    loss.backward()
    nwin = B * ((H + 6) // 7) * ((W + 6) // 7)
    names = (['w1', 'b1', 'g2', 'bt2'] if with_ffn else []) + ['wo', 'bo', 'wq', 'bq', 'wk', 'bk', 'wv', 'bv', 'gq', 'btq'] + \
        (['gkv', 'btkv'] if cross else [])
    size = dict(w1=N1 * C, b1=N1, g2=C, bt2=C, wo=C * C, bo=C, wq=C * C, bq=C, wk=C * C, bk=C, wv=C * C, bv=C, gq=C, btq=C, gkv=C, btkv=C)
    offs, o = {}, 0
    for n in names:
        offs[n] = o
        o += size[n]
    slot = o
    pslot = torch.full((nwin * slot,), float('nan'), device=dev)
    dsp = torch.full((nwin * heads * 49 * 49,), float('nan'), device=dev)
    gout_d = f32(gout.reshape(-1, C))
    a.gout = P(gout_d)
    if with_ffn:
        du1_d = f32(du1.reshape(-1, N1))
        cA, cB, cC = torch.ones(N1, device=dev), torch.zeros(N1, device=dev), torch.zeros(N1, device=dev)
        a.du1, a.cA1, a.cB1, a.cC1 = P(du1_d), P(cA), P(cB), P(cC)
    dq = torch.full((B * H * W, C), float('nan'), device=dev)
    a.dq, a.dq_acc = P(dq), 0
    if tail:
        tdu = torch.full((B * H * W, C), float('nan'), device=dev)
        tgs = torch.zeros(KC * 2 * C, dtype=torch.float64, device=dev)
        a.tail_du, a.tail_gstats = P(tdu), P(tgs)
    if cross:
        dkv = torch.full((B * H * W, C), float('nan'), device=dev)
        a.dkv, a.dkv_acc, a.dkv_add_res = P(dkv), 0, 1
        a.dq_add_res = 1                                 # res is xq: the residual gradient rides on the dq write
    else:
        a.dq_add_res = 1
    a.pslot, a.slot_stride, a.ds_plane = P(pslot), slot, P(dsp)
    park = torch.full((nwin * 64 * 32,), float('nan'), device=dev)       # scratch of the 8-wave 18-channel backward
    a.gx_park = P(park)
    for n in ('w1', 'b1', 'g2', 'bt2', 'wo', 'bo', 'wq', 'bq', 'wk', 'bk', 'wv', 'bv', 'gq', 'btq', 'gkv', 'btkv', 'rpb'):
        setattr(a, 'off_' + n, offs.get(n, -1))
    L.hrf_attn_block_bwd(a, _lib.stream_ptr())
    if tail:
        assert r(dq.reshape(B, H, W, C).permute(0, 3, 1, 2), t_res.grad) < 5e-5
        du_ref = t_u.grad.permute(0, 2, 3, 1).reshape(-1, C)
        assert r(tdu, du_ref) < 5e-5
        gs = tgs.view(KC, 2, C).sum(0)
        raw_rows = t_raw.permute(0, 2, 3, 1).reshape(-1, C)
        assert r(gs[0], du_ref.sum(0)) < 1e-4 and r(gs[1], (du_ref * raw_rows).sum(0)) < 1e-4
    else:
        assert r(dq.reshape(B, H, W, C).permute(0, 3, 1, 2), x.grad) < 5e-5
    if cross:
        assert r(dkv.reshape(B, H, W, C).permute(0, 3, 1, 2), xkv.grad) < 5e-5

    # ---- hrf_fold_slots: the per-window slots into a flat "gradient arena" (here: one segment, identity map + an offset)
    base = 7
    dst = torch.zeros(base + slot, device=dev)
    seg = torch.tensor([[0, nwin, slot, slot, 0]], dtype=torch.long, device=dev)
    mp = torch.arange(base, base + slot, dtype=torch.int32, device=dev)
    mp[offs['bk']:offs['bk'] + C] = -1                  # a frozen parameter: its entries are skipped
    L.hrf_fold_slots(pslot, seg, 1, mp, dst, slot, _lib.stream_ptr())
    got = lambda n: dst[base + offs[n]:base + offs[n] + size[n]]
    if cross:
        ref = dict(wo=msa.out_proj.weight, bo=msa.out_proj.bias, wq=msa.q_proj.weight, bq=msa.q_proj.bias, wk=msa.k_proj.weight,
                   wv=msa.v_proj.weight, bv=msa.v_proj.bias, gq=lnq.weight, btq=lnq.bias, gkv=lnkv.weight, btkv=lnkv.bias)
        ref = {k: v.grad.reshape(-1) for k, v in ref.items()}
    else:
        gw, gb = msa.qkv.weight.grad, msa.qkv.bias.grad
        ref = dict(wo=msa.out_proj.weight.grad.reshape(-1), bo=msa.out_proj.bias.grad, wq=gw[:C].reshape(-1), bq=gb[:C],
                   wk=gw[C:2 * C].reshape(-1), wv=gw[2 * C:].reshape(-1), bv=gb[2 * C:], gq=lnq.weight.grad, btq=lnq.bias.grad)
    if with_ffn:
        ref.update(w1=blk.ffn.layers[0].weight.grad.reshape(-1), b1=blk.ffn.layers[0].bias.grad, g2=ln2.weight.grad, bt2=ln2.bias.grad)
    gmax = max(float(v.abs().max()) for v in ref.values())
    for n, q in ref.items():
        err = float((got(n).double().cpu() - q).abs().max()) / max(float(q.abs().max()), 1e-3 * gmax)
        assert err < 1e-4, (n, err)
    assert float(got('bk').abs().max()) == 0.0          # skipped by the map (and analytically zero anyway: softmax shift)
    assert float(dst[:base].abs().max()) == 0.0

    # ---- hrf_rpb_grad: dRPB gathered from the dS planes into the replicated accumulator
    drpb = torch.zeros(KC * 169 * heads, device=dev)
    L.hrf_rpb_grad(dsp, nwin, heads, drpb, 169 * heads, _lib.stream_ptr())
    assert r(drpb.view(KC, 169, heads).sum(0), msa.relative_position_bias_table.grad) < 1e-4
    # ---- hrf_rpb_grad_all: the same gather for several layers in one launch (here: this layer at an offset of a common arena,
    # a second row that is the first half of its windows with its own accumulator, geometry larger than either row needs)
    arena = torch.zeros(8 + dsp.numel(), device=dev)
    arena[8:] = dsp
    acc2 = torch.zeros(2, KC * 169 * heads, device=dev)
    h1 = max(1, nwin // 2)
    seg = torch.tensor([[8, nwin, heads, acc2[0].data_ptr(), 169 * heads],
                        [8, h1, heads, acc2[1].data_ptr(), 169 * heads]], dtype=torch.long).to(dev)
    L.hrf_rpb_grad_all(arena, seg, 2, nwin + 3, heads + 1, _lib.stream_ptr())
    want1 = torch.zeros(KC * 169 * heads, device=dev)
    L.hrf_rpb_grad(dsp, h1, heads, want1, 169 * heads, _lib.stream_ptr())
    assert r(acc2[0].view(KC, 169, heads).sum(0), msa.relative_position_bias_table.grad) < 1e-4
    assert r(acc2[1].view(KC, 169, heads).sum(0), want1.view(KC, 169, heads).sum(0)) < 1e-5


EDGE = [  # C, heads, B, H, W
    (18, 1, 1, 7, 7),      # exactly one window, no padding
    (18, 1, 3, 5, 6),      # H and W < 7: one padded window per sample, B = 3
    (18, 1, 2, 10, 13),    # 2 x 2 windows, asymmetric centre pad
    (36, 2, 1, 3, 20),     # H < 7, three windows in a row
    (36, 2, 3, 8, 15),
]


@pytest.mark.parametrize('cross', [False, True])
@pytest.mark.parametrize('case', EDGE[:3])
def test_attn_block_abi_emul(case, cross):
    _run(*case, cross, True, True, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('with_ffn', [False, True])
@pytest.mark.parametrize('cross', [False, True])
@pytest.mark.parametrize('case', EDGE)
def test_attn_block_abi_gpu(case, cross, with_ffn):
    _run(*case, cross, with_ffn, True, 'hip')


@pytest.mark.parametrize('case', [EDGE[1], EDGE[2]])
def test_attn_block_tail_abi_emul(case):
    _run(*case, False, True, True, 'emul', tail=True)


@pytest.mark.gpu
@pytest.mark.parametrize('with_ffn', [False, True])
@pytest.mark.parametrize('case', EDGE)
def test_attn_block_tail_abi_gpu(case, with_ffn):
    """the preceding block's CrossFFN tail formed on load (forward) / its du + BatchNorm moments emitted (backward)"""
    _run(*case, False, with_ffn, True, 'hip', tail=True)


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(72, 4, 2, 8, 9), (144, 8, 3, 5, 10)])
def test_attn_block_fwd_wide_abi_gpu(case):
    """forward-only widths of the fused block (the backward of 72 / 144 runs on the per-op kernels)"""
    _run(*case, False, True, False, 'hip')


# ------------------------------------------------------------------------------------------- packed SyncBN forms, wide C
def _packed(C_list, backend):
    dev = use_backend(backend)
    L = _lib.lib()
    g = torch.Generator().manual_seed(9)
    n = len(C_list)
    count = 240.0
    stats, ref = [], []
    for C in C_list:
        s = torch.rand(KC, 2 * C, generator=g).double()
        s[:, C:] += 2.0                                   # sum of squares > (sum)^2 / count
        stats.append((s * count / KC).reshape(-1).contiguous().to(dev))
        ref.append(stats[-1].view(KC, 2 * C).sum(0))
    tail = sum(2 * C for C in C_list)
    packed = torch.full((tail + n,), float('nan'), dtype=torch.float64, device=dev)
    ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in stats])
    cs = (ctypes.c_int * n)(*C_list)
    rows = (ctypes.c_double * n)(*[count + 7.0 * i for i in range(n)])         # this rank's sample counts, behind the sums
    L.hrf_bn_pack(ptrs, cs, n, rows, packed, _lib.stream_ptr())
    assert r(packed[:tail], torch.cat(ref)) < 1e-12
    assert packed[tail:].tolist() == [count + 7.0 * i for i in range(n)]
    packed[tail:] = count                                     # (as if all-reduced: the layers below are finalised with `count`)
    # finalize every layer from the packed sums and compare with the stand-alone finalize on the replicated moments
    P = _lib._ptr
    off = 0
    for C, st in zip(C_list, stats):
        bufs = {k: torch.zeros(C, device=dev) for k in ('scale', 'shift', 'mean', 'invstd', 'rscale', 'rshift', 'rmean', 'rinvstd')}
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
        rm, rv = torch.randn(C, generator=g).to(dev), (torch.rand(C, generator=g) + 0.5).to(dev)
        rm2, rv2 = rm.clone(), rv.clone()
        L.hrf_bn_finalize(st, gamma, beta, rm2, rv2, count, 1e-5, 0.1, 1, bufs['rscale'], bufs['rshift'], bufs['rmean'],
                          bufs['rinvstd'], C, _lib.stream_ptr())
        # the host-side count is deliberately WRONG: the kernels must take the device-side count behind the sums (count_ptr)
        li = C_list.index(C)
        fin = _lib.BnFin(None, P(gamma), P(beta), P(rm), P(rv), P(bufs['scale']), P(bufs['shift']), P(bufs['mean']),
                         P(bufs['invstd']), 3.0, 1e-5, 0.1, 1, 1, C, 1, packed.data_ptr() + 8 * (tail + li))
        L.hrf_bn_finalize_packed(fin, 1, packed.data_ptr() + 8 * off, _lib.stream_ptr())
        for k in ('scale', 'shift', 'mean', 'invstd'):
            assert r(bufs[k], bufs['r' + k]) < 1e-6, (C, k)
        assert r(rm, rm2) < 1e-6 and r(rv, rv2) < 1e-6
        # backward form on the same packed slice (as gstats = (sum du, sum du*y))
        mean, invstd = bufs['rmean'], bufs['rinvstd']
        c = {k: torch.zeros(C, device=dev) for k in ('cA', 'cB', 'cC', 'rA', 'rB', 'rC')}
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        dg2, db2 = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        L.hrf_bn_bwd_finalize(st, None, gamma, mean, invstd, count, 1, dg2, db2, c['rA'], c['rB'], c['rC'], C, _lib.stream_ptr())
        bf = _lib.BnBFin(None, P(gamma), P(mean), P(invstd), P(dg), P(db), P(c['cA']), P(c['cB']), P(c['cC']), 3.0, 1, 1, C, 1, None, 0.5,
                         packed.data_ptr() + 8 * (tail + li))
        L.hrf_bn_bwd_finalize_packed(bf, 1, packed.data_ptr() + 8 * off, None, _lib.stream_ptr())
        for k in 'ABC':
            assert r(c['c' + k], c['r' + k]) < 1e-6, (C, k)
        assert r(dg, 0.5 * dg2) < 1e-6 and r(db, 0.5 * db2) < 1e-6        # pgrad_scale = 1 / world
        off += 2 * C


def test_bn_packed_wide_emul():
    _packed([18, 624, 72], 'emul')


@pytest.mark.gpu
def test_bn_packed_wide_gpu():
    """BatchNorms wider than HRF_FIN_MAXC = 576 (HRFuser-B CrossFFN hidden 624 / 1248 / 2496) take the stand-alone packed
    finalize forms under SyncBN (runtime._finalize_now / bn_backward_coef)."""
    assert max([624, 1248, 2496]) > _lib.FIN_MAXC
    _packed([18, 624, 1248, 2496, 72], 'hip')


def _nearest(B, Hs, Ws, f, C, backend):
    dev = use_backend(backend)
    L = _lib.lib()
    g = torch.Generator().manual_seed(4)
    H, W = Hs * f, Ws * f
    gr = torch.randn(B, H, W, C, generator=g)
    ylow = torch.randn(B, Hs, Ws, C, generator=g)
    du = torch.full((B, Hs, Ws, C), float('nan'), device=dev)
    st = torch.zeros(KC * 2 * C, dtype=torch.float64, device=dev)
    L.hrf_nearest_up_bwd(gr.to(dev), C, 0, B, H, W, C, ylow.to(dev), Hs, Ws, du, st, _lib.stream_ptr())
    ref = gr.double().view(B, Hs, f, Ws, f, C).sum((2, 4))
    assert r(du, ref) < 1e-5
    s = st.view(KC, 2, C).sum(0)
    assert r(s[0], ref.sum((0, 1, 2))) < 1e-5 and r(s[1], (ref * ylow.double()).sum((0, 1, 2))) < 1e-5


def test_nearest_up_bwd_emul():
    _nearest(2, 3, 5, 2, 18, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('case', [(2, 3, 5, 2, 18), (1, 2, 3, 8, 36), (3, 5, 4, 4, 300)])
def test_nearest_up_bwd_gpu(case):
    _nearest(*case, 'hip')


def _rpb_many_windows(backend, nwin=300, heads=2):
    """hrf_rpb_grad with more windows than window chunks (HRF_RPB_CHUNKS = 128): a block walks several planes, the next one in flight
    in registers while this one is gathered (round 6) - against the definition drpb[(yi-yj+6)*13 + (xi-xj+6)][h] = sum dS[w][h][j][i]"""
    dev = use_backend(backend)
    L = _lib.lib()
    g = torch.Generator().manual_seed(11)
    ds = torch.randn(nwin, heads, 49, 49, generator=g)
    idx = torch.arange(49)
    yi, xi = idx // 7, idx % 7
    bins = (yi[None, :] - yi[:, None] + 6) * 13 + (xi[None, :] - xi[:, None] + 6)            # [j][i]
    want = torch.zeros(169, heads, dtype=torch.float64)
    want.index_add_(0, bins.reshape(-1), ds.double().sum(0).permute(1, 2, 0).reshape(49 * 49, heads))
    drpb = torch.zeros(KC * 169 * heads, device=dev)
    L.hrf_rpb_grad(ds.to(dev), nwin, heads, drpb, 169 * heads, _lib.stream_ptr())
    assert r(drpb.view(KC, 169, heads).sum(0), want) < 1e-5


def test_rpb_grad_many_windows_emul():
    _rpb_many_windows('emul')


@pytest.mark.gpu
def test_rpb_grad_many_windows_gpu():
    _rpb_many_windows('hip', nwin=644)
