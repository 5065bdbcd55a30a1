"""the stochastic layers of the fusion block - nn.Dropout(proj_drop_rate) on the cross-attention projection
(hrfuser_hrformer_based.py:97,147-150) and mmcv DropPath on every residual path (:301-303,311-317) - with PINNED random
draws on both sides: a Dropout mask with real zeros and DropPath draws that drop one of the two samples on each
residual path.  This pins WHERE the mask / scale enters (before the window merge, after out_proj, on the attention
output only - not on the modality residual) and the 1/keep scaling.  Plus the statistics of the product's draws."""
import pytest
import torch

import hrfuser_oracle as O
from helpers import LN, NORM, relmax, rel_l2, use_backend

import hrfuser_amd.backbone as B
from hrfuser_amd.testing import BlockHarness


def _run(C, heads, M, H, W, backend):
    dev = use_backend(backend)
    orc = O.HRFuserFusionBlock(C, heads, 4, NORM, LN, 0.2, M, 0.1)
    O.seeded_fill_(orc, 3)
    blk = B.HRFuserFusionBlock(C, C, heads, norm_cfg=NORM, transformer_norm_cfg=LN, num_fused_modalities=M,
                               drop_path=0.2, proj_drop_rate=0.1)
    h = BlockHarness(blk, lambda k, b, x: b.run(k, x[0], x[1:]))
    h.block.load_state_dict(orc.state_dict(), strict=True)
    h.to(dev)
    orc = orc.double()
    h.train()
    orc.train()
    Bn = 2
    g = torch.Generator().manual_seed(11)
    # the pinned draws: Dropout keep-masks (NHWC, ~10 % zeros) per modality; DropPath per-sample scales (mmcv: floor(keep+U)/keep)
    # for the M attention paths and the FFN path - sample 0 / 1 dropped alternately, so every path loses a sample once
    masks = [(torch.rand(Bn, H, W, C, generator=g) >= 0.1).float() for _ in range(M)]
    assert all(float(m.min()) == 0.0 for m in masks)
    keep = 0.8
    scales = [torch.tensor([0.0, 1.0 / keep]) if k % 2 == 0 else torch.tensor([1.0 / keep, 0.0]) for k in range(M + 1)]

    # ---- product side: the engine's random pools return the pinned tensors, in draw order
    eng = h._engine()
    mq, sq = [m.to(dev) for m in masks], [s.to(dev) for s in scales]
    eng.dropout_mask = lambda shape, p: mq.pop(0).reshape(shape)
    eng.droppath_scale = lambda Bb, p: sq.pop(0)

    # ---- oracle side: the same tensors at the reference's call sites
    oq = [s.double() for s in scales]

    def pinned_droppath(x):
        s = oq.pop(0)
        return x * s.view(-1, *([1] * (x.ndim - 1)))
    orc.drop_path.forward = pinned_droppath
    for k in range(M):
        # the reference drops on the WINDOWED tensor (B*nW, 49, C), padded tokens included: partition the mask the same way
        mw = O.window_partition(masks[k].reshape(Bn, H * W, C).double(), H, W)
        orc.attn[k].attn.proj_drop.forward = (lambda mk: (lambda x: x * mk / 0.9))(mw)

    ins = [torch.randn(Bn, C, H, W, generator=torch.Generator().manual_seed(40 + i)) for i in range(M + 1)]
    a = [t.clone().to(dev).requires_grad_(True) for t in ins]
    b = [t.clone().double().requires_grad_(True) for t in ins]
    ya = h(*a)[0]
    yb = orc(b[0], list(b[1:]))
    assert not mq and not sq and not oq                     # every pinned draw was consumed, on both sides
    assert relmax(ya, yb) < 1e-4
    cot = torch.randn(yb.shape, generator=torch.Generator().manual_seed(7))
    (ya * cot.to(dev)).sum().backward()
    (yb * cot.double()).sum().backward()
    for p, q in zip(a, b):
        assert rel_l2(p.grad, q.grad) < 1e-4
    pa, pb = dict(h.block.named_parameters()), dict(orc.named_parameters())
    gmax = max(float(q.grad.abs().max()) for q in pb.values() if q.grad is not None)
    for k, q in pb.items():
        if q.grad is None:
            continue
        scale = max(float(q.grad.abs().max()), 1e-3 * gmax)   # floor: analytically-zero grads (k bias, biases in front of a train-mode BN)
        assert float((pa[k].grad.double().cpu() - q.grad).abs().max()) / scale < 1e-3, k
    # a dropped sample really is the identity of its residual path: with the FFN path of sample 1 dropped (M = 2: scales[2]
    # = [0, 1.25] drops sample 0) nothing here; checked through the oracle equality above


@pytest.mark.parametrize('cfg', [(18, 1, 2, 10, 13), (36, 2, 3, 8, 15)])
def test_fusion_block_pinned_stochastic_emul(cfg):
    _run(*cfg, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [(18, 1, 2, 10, 13), (36, 2, 3, 8, 15), (78, 2, 2, 9, 16)])
def test_fusion_block_pinned_stochastic_gpu(cfg):
    """M = 2 and M = 3 on the fused attention block, and an HRFuser-B width on the unfused kernels."""
    _run(*cfg, 'hip')


def _draw_stats(dev):
    blk = B.HRFuserFusionBlock(18, 18, 1, norm_cfg=NORM, transformer_norm_cfg=LN, num_fused_modalities=2)
    h = BlockHarness(blk, lambda k, b, x: b.run(k, x[0], x[1:])).to(dev)
    eng = h._engine()
    eng.ready(dev)
    torch.manual_seed(0)
    for rnd in range(2):                                   # round 0: fresh draws; round 1: served from the per-step pools
        eng.begin_forward(True)
        m = eng.dropout_mask((4, 24, 40, 18), 0.1)
        s = torch.cat([eng.droppath_scale(64, 0.2) for _ in range(16)])
        vals = set(m.unique().tolist())
        assert vals == {0.0, 1.0}
        assert abs(float(m.mean()) - 0.9) < 0.01           # keep rate 1 - p (69 k draws: sigma = 1.1e-3)
        sv = set(round(v, 5) for v in s.unique().tolist())
        assert sv == {0.0, 1.25}                           # mmcv DropPath: floor(keep + U) / keep
        assert abs(float((s > 0).float().mean()) - 0.8) < 0.05
        assert abs(float(s.mean()) - 1.0) < 0.06           # unbiased: E[scale] = 1


def test_draw_statistics_emul():
    _draw_stats(use_backend('emul'))


@pytest.mark.gpu
def test_draw_statistics_gpu():
    _draw_stats(use_backend('hip'))


def test_draws_belong_to_the_call_site_not_to_program_order():
    """The lock-step scheduler and the serial one (HRF_LOCKSTEP=0 / HRF_SYNC_LANE_COMMS=1) reach the layers in different
    orders; with one seed every layer must get the same draws under either (bench.py's sync_ab compares gradients)."""
    dev = use_backend('emul')
    blk = B.HRFuserFusionBlock(18, 18, 1, norm_cfg=NORM, transformer_norm_cfg=LN, num_fused_modalities=2)
    eng = BlockHarness(blk, lambda k, b, x: b.run(k, x[0], x[1:])).to(dev)._engine()
    eng.ready(dev)

    def step(order, seed):
        eng.begin_forward(True)                            # the pools were refilled from torch's generator just now
        out = {}
        for site in order:
            eng.rng_site = site
            out[site] = (eng.dropout_mask((2, 8, 8, 18), 0.1).clone(), eng.droppath_scale(2, 0.2).clone(), eng.droppath_scale(2, 0.2).clone())
        return out

    torch.manual_seed(1)
    step(['a', 'b', 'c'], 1)                               # first step: fresh draws, slots assigned
    torch.manual_seed(2)
    r1 = step(['a', 'b', 'c'], 2)
    torch.manual_seed(2)
    r2 = step(['c', 'a', 'b'], 2)
    for site in 'abc':
        for u, v in zip(r1[site], r2[site]):
            assert torch.equal(u, v)
    assert not torch.equal(r1['a'][0], r1['b'][0])         # distinct slices of the pool
