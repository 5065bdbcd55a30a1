"""Isolated launch times of hrf_ffn_eval at the models' shapes (GPU-side, graph-batched)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hrfuser_amd import _lib
from hrfuser_amd.profiling import _graph_time
L = _lib.lib()
dev = torch.device('cuda:0')
for (B, H, W, C) in [(2, 96, 160, 18), (2, 48, 80, 36), (2, 24, 40, 72), (2, 12, 20, 144), (2, 96, 160, 78), (2, 48, 80, 156), (2, 96, 312, 18)]:
    Hd = 4 * C
    t = lambda *s: torch.randn(*s, device=dev)
    x = t(B, H, W, C); out = torch.empty_like(x)
    a = _lib.FfnEval(); a.B, a.H, a.W, a.C, a.hidden = B, H, W, C, Hd
    bufs = [t(C), t(C), t(Hd, C) * 0.3, t(Hd), t(Hd), t(Hd), t(Hd, 9) * 0.3, t(Hd), t(Hd), t(Hd), t(C, Hd) * 0.2, t(C), t(C), t(C)]
    a.x = x.data_ptr(); a.ln_g, a.ln_b, a.ln_eps = bufs[0].data_ptr(), bufs[1].data_ptr(), 1e-6
    a.w1, a.b1, a.s1, a.t1 = [b.data_ptr() for b in bufs[2:6]]
    a.wd, a.bd, a.s2, a.t2 = [b.data_ptr() for b in bufs[6:10]]
    a.w3, a.b3, a.s3, a.t3 = [b.data_ptr() for b in bufs[10:14]]
    a.out = out.data_ptr()
    dt = _graph_time(lambda: L.hrf_ffn_eval(a, _lib.stream_ptr()))
    fl = 2.0 * B * H * W * C * Hd * 2 + 2.0 * B * H * W * Hd * 9
    print(f'ffn_eval B{B} {H}x{W} C{C}: {dt * 1e6:7.1f} us  {fl / dt / 1e12:6.2f} TFLOP/s  {(2 * B * H * W * C * 4) / dt / 1e9:7.1f} GB/s (x in + out)')
