"""Shared parity-test machinery.

Two backends run the SAME comparisons against the oracle:
  * 'hip'  (tests marked gpu): the product path - libhrfuser_hip.so on cuda:0 through the C ABI;
  * 'emul' (CPU suite): the same kernel sources compiled for the CPU fiber emulator
    (tests/emul) - checks kernel logic/index math without a GPU.  Not a product path.
"""
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'emul'))

import hrfuser_oracle as O                      # noqa: E402  (oracle/ is on sys.path via conftest)
from hrfuser_amd import _lib                    # noqa: E402

NORM = dict(type='BN', requires_grad=True, momentum=0.1)
LN = dict(type='LN', eps=1e-6)
_EMUL = None
_REAL_LIB_FN = _lib.lib


def use_backend(name):
    """Select the library the product code launches on; returns the torch device to use."""
    global _EMUL
    if name == 'emul':
        if _EMUL is None:
            import build_emul
            _EMUL = _lib.Lib(build_emul.build(), require_cuda=False)
        _lib.lib = lambda: _EMUL
        return torch.device('cpu')
    _lib.lib = _REAL_LIB_FN
    return torch.device('cuda:0')


def relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def rel_l2(a, b, floor=0.0):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + floor))


def disable_stochastic(*nets):
    for net in nets:
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if isinstance(m, O.DropPath):
                m.p = 0.0
            if hasattr(m, 'drop_path_prob'):
                m.drop_path_prob = 0.0


def load_cfgs():
    with open(os.path.join(ROOT, 'tests', 'golden', 'backbone_cfgs.json')) as fh:
        return json.load(fh)


def build_pair(tag, device, seed=0, edit=None, stochastic=False):
    """(product backbone on `device`, oracle on CPU) with identical seeded parameters.  edit(cfg): optional in-place
    change of the config dict before both are built (e.g. fewer modules per stage for a quick data-parallel test).
    stochastic: keep Dropout / DropPath live (the caller pins their draws)."""
    from hrfuser_amd import build_backbone
    cfg = copy.deepcopy(load_cfgs()[tag])
    if edit is not None:
        edit(cfg)
    c2 = copy.deepcopy(cfg)
    c2.pop('type')
    orc = O.HRFuserOracle(**c2)
    O.seeded_fill_(orc, seed)
    net = build_backbone(copy.deepcopy(cfg))
    net.load_state_dict(orc.state_dict())
    net.to(device)
    if not stochastic:
        disable_stochastic(net, orc)
    return net, orc, cfg


def pin_fusion_stochastic(net, oracles, B, H, W, seed=21, p_drop=0.1, keep=0.8):
    """Pin the random draws of every fusion block of a WHOLE backbone on both sides: per block and modality
    a Dropout keep-mask with real zeros, per residual path an mmcv DropPath scale floor(keep + U) / keep per sample.
    Product: the block's own `_droppath_scale` returns its scales in order and the engine serves the masks from one
    FIFO per tensor shape (the fusion stages run one after the other, their branches differ in shape - so the order
    inside a queue is the reference's order even though sibling branches execute in lock-step).  Oracles: the same
    tensors at the reference's call sites (hrfuser_hrformer_based.py:147-150,311-317), re-usable across runs."""
    g = torch.Generator().manual_seed(seed)
    dev = next(net.parameters()).device
    fifo = {}
    for tag in ('fusion_a', 'fusion_b', 'fusion_c'):
        for i, pb in enumerate(getattr(net, tag)):
            M = pb.num_fused_modalities
            C = pb.norm3.weight.numel()
            Hi, Wi = (H // 4) >> i, (W // 4) >> i
            masks = [(torch.rand(B, Hi, Wi, C, generator=g) >= p_drop).float() for _ in range(M)]
            scales = [torch.floor(keep + torch.rand(B, generator=g)) / keep for _ in range(M + 1)]
            scales[i % (M + 1)][0] = 0.0                                 # every block really drops a sample on one path
            for m in masks:
                fifo.setdefault((B, Hi, Wi, C), []).append(m.to(dev))
            pq = [s.to(dev) for s in scales]
            pb._droppath_scale = (lambda q: (lambda ctx, Bn, d: q.pop(0)))(pq)
            for orc in oracles:
                ob = getattr(orc, tag)[i]
                dt = next(orc.parameters()).dtype
                oq, cnt = [s.to(dt) for s in scales], [0]

                def pinned_droppath(x, oq=oq, cnt=cnt):
                    s = oq[cnt[0] % len(oq)]
                    cnt[0] += 1
                    return x * s.view(-1, *([1] * (x.ndim - 1)))
                ob.drop_path.forward = pinned_droppath
                for k in range(M):
                    mw = O.window_partition(masks[k].reshape(B, Hi * Wi, C).to(dt), Hi, Wi)
                    ob.attn[k].attn.proj_drop.forward = (lambda mk: (lambda x: x * mk / (1.0 - p_drop)))(mw)
    eng = net._engine()
    eng.dropout_mask = lambda shape, p: fifo[tuple(shape)].pop(0).reshape(shape)
    return fifo


class PinnedReLU:
    """Context manager that makes every ReLU of the ORACLE follow the product's sign decisions.

    An fp32 pre-activation within rounding noise of 0 lands on either side of the ReLU in two correct fp32
    implementations; each such "mask flip" changes the gradient inside one receptive field, which is why an un-pinned
    fp32-vs-fp64 gradient comparison has outliers.  Instead of excusing outliers, the comparison is made
    flip-free: the product records the sign mask of every ReLU it applied (runtime.collect_relu_masks), and while this
    context is active `torch.nn.functional.relu` (reached by F.relu and nn.ReLU alike) returns `u * mask_product`.
    The mask of a call is found by content: the recorded mask of the same shape that agrees best with (u > 0); the
    agreement must be essentially total (<= max(4, 1e-4 * numel) differing elements per site), so a wrong pairing or a
    genuinely different activation pattern fails the test instead of being pinned over.  `flips` counts the imposed
    differences; check() asserts what is measured on MI355X (0-11 flips per run): the TOTAL over all sites stays below
    max(3, 1e-5 * total elements) - a kernel that mis-signs 0.1 % of a map cannot be pinned over."""

    def __init__(self, masks):
        self.by_shape = {}
        for m in masks:
            self.by_shape.setdefault(tuple(m.shape), []).append(m.detach().cpu())
        self.flips = 0
        self.sites = 0
        self.numel = 0

    def check(self, tag=''):
        lim = max(3, 1e-5 * self.numel)
        assert self.flips <= lim, (tag, f'{self.flips} pinned ReLU decisions over {self.sites} sites / {self.numel} elements (limit {lim:.1f})')
        return self.flips

    def _relu(self, u, inplace=False):
        cands = self.by_shape.get(tuple(u.shape))
        assert cands, f'no product ReLU site of shape {tuple(u.shape)}'
        tgt = u.detach() > 0
        best, bad = None, None
        for m in cands:
            d = int((m != tgt).sum())
            if bad is None or d < bad:
                best, bad = m, d
        assert bad <= max(4, 1e-4 * tgt.numel()), f'ReLU site {tuple(u.shape)}: best product mask differs in {bad} elements'
        self.flips += bad
        self.sites += 1
        self.numel += tgt.numel()
        return u * best.to(u.dtype)

    def __enter__(self):
        import torch.nn.functional as F
        self._orig = F.relu
        F.relu = self._relu
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.relu = self._orig
        return False


def enable_relu_probe(net):
    net.__dict__['_relu_probe'] = True


def relu_masks(net):
    return net.__dict__['_relu_masks']


def tight_grad_gate(named_prod, named_ref64, named_ref32, tol=1e-3, tag=''):
    """SURVEY 8c gate, per tensor: rel-L2 err(build, fp64) <= max(tol, 3 * e_ref[k]) with e_ref[k] = the oracle's own
    fp32-vs-fp64 error on the same tensor (same pinned ReLU masks, so e_ref is rounding noise only).  No global
    relaxation and no denominator padding.  Analytically-zero gradients (biases in front of a train-mode BatchNorm,
    the key bias under the softmax: SURVEY App. E) are recognised by their fp64 norm (< 1e-9 of the largest gradient
    norm) and gated absolutely: max|g| <= 1e-4 * max|g| over the whole net.  Returns (worst, n_above_tol)."""
    pa, pb, pc = dict(named_prod), dict(named_ref64), dict(named_ref32)
    ref = {k: v.grad.double() for k, v in pb.items() if v.grad is not None}
    gmax = max(float(g.abs().max()) for g in ref.values())
    nmax = max(float(g.norm()) for g in ref.values())
    worst, above, zeros = (0.0, ''), 0, 0
    for k, q in ref.items():
        g = pa[k].grad
        assert g is not None, k
        g = g.detach().double().cpu()
        if float(q.norm()) < 1e-9 * nmax:
            zeros += 1
            assert float(g.abs().max()) <= 1e-4 * gmax, (tag, k, 'analytically zero', float(g.abs().max()), gmax)
            continue
        den = float(q.norm())
        e = float((g - q).norm()) / den
        e_ref = float((pc[k].grad.double() - q).norm()) / den
        above += e > tol
        assert e <= max(tol, 3 * e_ref), (tag, k, e, e_ref)
        worst = max(worst, (e, k))
    print(f'[grad gate {tag}] {len(ref)} tensors, {zeros} analytically zero, worst rel-L2 {worst[0]:.2e} ({worst[1]}), '
          f'{above} above {tol:g} (all within 3x the oracle\'s own fp32 error)')
    return worst, above
