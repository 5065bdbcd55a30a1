"""Is the captured training step paced by the HOST submission of its nodes?  A multi-stream hipGraph is re-enqueued node by node
at every launch (tools/microbench/graph_nodes.hip: ~3 us of host time per node, against 20 us for a whole single-stream graph),
and the rocprofv3 timeline of a step shows the sibling lanes of a stage starting 100-500 us apart.  Measures
  (1) the host time of consecutive graph.replay() calls and the step time of back-to-back replays,
  (2) the same with TWO captured instances of the step replayed alternately,
  (3) the GPU duration of ONE replay that starts behind a long filler kernel (every node is submitted before the first one may run)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                   # noqa: E402


def main():
    args = bench.parse(['--no-cpu-baseline', '--no-neck', '--no-eager', '--no-roofline'] + sys.argv[1:])
    rank, world, dev, group, force_coll = bench.init_ranks(args)
    tag, cfg, stf, H, W, mc, net, B, x, mods, cots, trainer = bench.build_workload(args, rank, world, dev, group, force_coll)
    trainer.step(x, mods, cots)
    torch.cuda.synchronize()
    g1 = trainer.capture(x, mods, cots)
    for _ in range(5):
        g1.replay()
    torch.cuda.synchronize()
    # (1) host time per replay() and step time
    N = 20
    host = []
    t0 = time.perf_counter()
    for _ in range(N):
        a = time.perf_counter()
        g1.replay()
        host.append(time.perf_counter() - a)
    t_sub = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f'(1) one graph: {t_all / N * 1e3:.3f} ms/step; host time of replay(): first {host[0] * 1e3:.2f} median {sorted(host)[N // 2] * 1e3:.2f} '
          f'max {max(host) * 1e3:.2f} ms; all {N} submitted after {t_sub * 1e3:.1f} ms of {t_all * 1e3:.1f} ms')
    # (3) one replay behind a filler: everything is submitted before the GPU may start
    filler = torch.empty(1 << 28, device=dev)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    res = []
    for rep in range(5):
        torch.cuda.synchronize()
        for _ in range(12):
            filler.add_(1.0)                                       # ~1 GB read+write each: >= 6 ms of GPU work in total
        e0.record()
        g1.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    print('(3) GPU time of one replay queued behind ~6 ms of filler kernels:', ' '.join(f'{v:.3f}' for v in res), 'ms')
    res = []
    for rep in range(5):
        torch.cuda.synchronize()
        e0.record()
        g1.replay()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    print('    GPU time of one replay on an idle GPU:                       ', ' '.join(f'{v:.3f}' for v in res), 'ms')
    # (2) two instances alternately
    trainer2 = trainer
    g2 = trainer2.capture(x, mods, cots)
    for _ in range(4):
        g1.replay(); g2.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N // 2):
        g1.replay(); g2.replay()
    torch.cuda.synchronize()
    print(f'(2) two graphs alternately: {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step')


if __name__ == '__main__':
    main()
