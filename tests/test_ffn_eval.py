"""Eval-mode CrossFFN in one launch (csrc/ffn_eval.hip, hrf_ffn_eval): hrformer.py:351 (norm2) + :267-295 (CrossFFN.forward with
frozen BatchNorm statistics) + :371-372 (residual) against the same modules in torch fp64 - through the C ABI on the CPU
emulator and on the GPU, ragged grids (H, W not multiples of the 4 x 16 tile, fewer pixels than one tile), every supported
width; and the whole backbone in eval mode (the route HRFormerBlock / HRFuserFusionBlock take without a tape) against the
oracle."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from helpers import use_backend
from hrfuser_amd import _lib

CASES = [(2, 9, 21, 18), (1, 5, 16, 36), (1, 4, 7, 18), (1, 6, 17, 36), (1, 3, 5, 36), (2, 17, 33, 18)]


def _run(case, backend):
    dev = use_backend(backend)
    L = _lib.lib()
    B, H, W, C = case
    Hd = 4 * C
    assert L.hrf_ffn_eval_supported(C, Hd) == 1 and L.hrf_ffn_eval_supported(C, 3 * C) == 0 and L.hrf_ffn_eval_supported(20, 80) == 0 and \
        L.hrf_ffn_eval_supported(72, 288) == 0
    g = torch.Generator().manual_seed(B * 1000 + H * 100 + W * 10 + C)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(B, H, W, C)
    ln = nn.LayerNorm(C, eps=1e-6)
    c1, d, c3 = nn.Conv2d(C, Hd, 1), nn.Conv2d(Hd, Hd, 3, 1, 1, groups=Hd), nn.Conv2d(Hd, C, 1)
    bns = [nn.BatchNorm2d(Hd), nn.BatchNorm2d(Hd), nn.BatchNorm2d(C)]
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * rn(C)); ln.bias.copy_(0.2 * rn(C))
        for m in (c1, d, c3):
            m.weight.copy_(rn(*m.weight.shape) * (0.4 if m is not d else 0.3)); m.bias.copy_(0.2 * rn(*m.bias.shape))
        for bn in bns:
            n = bn.num_features
            bn.weight.copy_(1 + 0.3 * rn(n)); bn.bias.copy_(0.2 * rn(n))
            bn.running_mean.copy_(0.3 * rn(n)); bn.running_var.copy_(0.5 + torch.rand(n, generator=g))
    mods = nn.ModuleList([ln, c1, d, c3] + bns).double().eval()
    with torch.no_grad():
        xd = x.double()
        h = ln(xd).permute(0, 3, 1, 2)
        h = F.gelu(bns[0](c1(h)))
        h = F.gelu(bns[1](d(h)))
        h = F.gelu(bns[2](c3(h)))
        want = xd + h.permute(0, 2, 3, 1)
    mods.float()
    a = _lib.FfnEval()
    keep = []

    def P(t):
        t = t.detach().float().contiguous().to(dev)
        keep.append(t)
        return t.data_ptr()
    a.B, a.H, a.W, a.C, a.hidden = B, H, W, C, Hd
    a.x = P(x)
    a.ln_g, a.ln_b, a.ln_eps = P(ln.weight), P(ln.bias), 1e-6
    aff = []
    for bn in bns:
        sc = bn.weight.double() / torch.sqrt(bn.running_var.double() + bn.eps)
        aff.append((sc.float(), (bn.bias.double() - bn.running_mean.double() * sc).float()))
    a.w1, a.b1, a.s1, a.t1 = P(c1.weight), P(c1.bias), P(aff[0][0]), P(aff[0][1])
    a.wd, a.bd, a.s2, a.t2 = P(d.weight), P(d.bias), P(aff[1][0]), P(aff[1][1])
    a.w3, a.b3, a.s3, a.t3 = P(c3.weight), P(c3.bias), P(aff[2][0]), P(aff[2][1])
    out = torch.full((B, H, W, C), float('nan'), device=dev)
    a.out = out.data_ptr()
    L.hrf_ffn_eval(a, _lib.stream_ptr() if backend == 'hip' else 0)
    if backend == 'hip':
        torch.cuda.synchronize()
    err = float((out.double().cpu() - want).abs().max() / want.abs().max())
    assert err < 2e-5, (case, err)


@pytest.mark.parametrize('case', CASES[:5], ids=str)
def test_ffn_eval_emul(case):
    _run(case, 'emul')


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES + [(2, 96, 160, 18), (2, 48, 80, 36)], ids=str)
def test_ffn_eval_gpu(case):
    _run(case, 'hip')
