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


def build_pair(tag, device, seed=0):
    """(product backbone on `device`, oracle on CPU) with identical seeded parameters."""
    from hrfuser_amd import build_backbone
    cfg = load_cfgs()[tag]
    c2 = copy.deepcopy(cfg)
    c2.pop('type')
    orc = O.HRFuserOracle(**c2)
    O.seeded_fill_(orc, seed)
    net = build_backbone(copy.deepcopy(cfg))
    net.load_state_dict(orc.state_dict())
    net.to(device)
    disable_stochastic(net, orc)
    return net, orc, cfg


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
