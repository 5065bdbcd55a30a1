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


def grad_close(a, b, tol=1e-3, max_flip_frac=0.02):
    """Gradient comparison robust to fp32 ReLU-mask flips: a pre-activation within fp32 noise of 0
    may land on either side in two correct fp32 implementations, which changes the gradient inside
    ONE receptive field.  Pass if max-rel error <= tol, or if the elements beyond tol are a small
    localized fraction of the tensor (and the tensor-wide rel-L2 stays small)."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = float(b.abs().max()) + 1e-30
    bad = ((a - b).abs() > tol * scale).double().mean().item()
    if bad == 0.0:
        return True
    return bad <= max_flip_frac and float((a - b).norm() / (b.norm() + 1e-30)) < 0.1


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


def build_pair(tag, device, seed=0, edit=None):
    """(product backbone on `device`, oracle on CPU) with identical seeded parameters.  edit(cfg): optional in-place
    change of the config dict before both are built (e.g. fewer modules per stage for a quick data-parallel test)."""
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
    disable_stochastic(net, orc)
    return net, orc, cfg


class PinnedReLU:
    """Context manager that makes every ReLU of the ORACLE follow the product's sign decisions.

    An fp32 pre-activation within rounding noise of 0 lands on either side of the ReLU in two correct fp32
    implementations; each such "mask flip" changes the gradient inside one receptive field, which is why an un-pinned
    fp32-vs-fp64 gradient comparison has outliers.  Instead of excusing outliers, the comparison is made
    flip-free: the product records the sign mask of every ReLU it applied (runtime.collect_relu_masks), and while this
    context is active `torch.nn.functional.relu` (reached by F.relu and nn.ReLU alike) returns `u * mask_product`.
    The mask of a call is found by content: the recorded mask of the same shape that agrees best with (u > 0); the
    agreement must be essentially total (<= max(4, 2e-3 * numel) differing elements), so a wrong pairing or a
    genuinely different activation pattern fails the test instead of being pinned over.  `flips` counts the imposed
    differences."""

    def __init__(self, masks):
        self.by_shape = {}
        for m in masks:
            self.by_shape.setdefault(tuple(m.shape), []).append(m.detach().cpu())
        self.flips = 0
        self.sites = 0

    def _relu(self, u, inplace=False):
        cands = self.by_shape.get(tuple(u.shape))
        assert cands, f'no product ReLU site of shape {tuple(u.shape)}'
        tgt = u.detach() > 0
        best, bad = None, None
        for m in cands:
            d = int((m != tgt).sum())
            if bad is None or d < bad:
                best, bad = m, d
        assert bad <= max(4, 2e-3 * tgt.numel()), f'ReLU site {tuple(u.shape)}: best product mask differs in {bad} elements'
        self.flips += bad
        self.sites += 1
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


def grad_check(prod_named, orc_named, tol=1e-3, abs_frac=2e-3):
    """Per-tensor gradient gate: rel-L2 <= tol, with an absolute floor for analytically-zero
    gradients (k-bias, biases in front of a train-mode BN: SURVEY App. E)."""
    pb = {k: v for k, v in orc_named if v.grad is not None}
    pa = dict(prod_named)
    gscale = max(float(v.grad.abs().max()) for v in pb.values())
    worst = (0.0, '')
    for k, q in pb.items():
        g = pa[k].grad
        assert g is not None, k
        err = float((g.detach().double().cpu() - q.grad.double()).norm())
        den = float(q.grad.double().norm()) + abs_frac * gscale * (q.numel() ** 0.5)
        e = err / den
        if e > worst[0]:
            worst = (e, k)
    assert worst[0] <= tol, worst
    return worst
