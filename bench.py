#!/usr/bin/env python3
"""bench.py — HRFuser-T backbone training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = forward (train-mode BN) + backward + gradient exchange + fused AdamW of the
HRFuser-T nuScenes backbone on a synthetic batch of 2 images per GPU (2x3x384x640 camera + lidar +
radar; BASELINE.json configs[1]; SyncBN + RCCL all-reduce when N > 1; weak scaling).  Everything in
the timed region runs on the hand-written HIP kernels (no oracle, no CPU fallback).

The JSON line also carries
  roofline      dominant kernel (largest share of GPU time per step): algorithmic FLOPs (or bytes)
                per launch / average launch duration.  Durations are measured in-process with HIP
                events on the launch stream: every distinct (entry point, shape) of one training step
                is re-issued 20x back-to-back inside a captured hipGraph right after the timed region
                (events cannot bracket single kernels inside the step's own graph, and an event pair
                around one eager launch measures the ~10 us host gap, not the kernel);
                cross-checked by profiles/*.csv (rocprofv3 --kernel-trace --stats)
  cpu_baseline  the PyTorch-CPU oracle (kind "port": bit-exact restatement of the reference
                backbone) timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA = 157.3e12     # MI355X_MICROARCH.md: dense fp32 MFMA peak (= fp32 vector peak)
PEAK_HBM = 8.0e12            # HBM3E spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--model', default='t_nus_bn', help='t_nus[_bn] | b_nus[_bn] | t_stf[_bn]')
    ap.add_argument('--batch', type=int, default=2, help='images per GPU')
    ap.add_argument('--height', type=int, default=0)
    ap.add_argument('--width', type=int, default=0)
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-neck', action='store_true', help='skip the HRFPN neck timing (the consumer of the 4 maps)')
    ap.add_argument('--profile-steps', type=int, default=1)
    ap.add_argument('--dump-kernels', default='', help='write the per-kernel table (JSON) to this path')
    return ap.parse_args()


def load_cfg(tag):
    with open(os.path.join(ROOT, 'tests', 'golden', 'backbone_cfgs.json')) as fh:
        return json.load(fh)[tag]


def cpu_baseline(tag, B, H, W, mc, iters=2):
    """Oracle (PyTorch-CPU restatement of the reference backbone, oracle/hrfuser_oracle.py) timed on
    the host cores: train-mode forward+backward of the same workload.  Bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import hrfuser_oracle as O
    cfg = copy.deepcopy(load_cfg(tag))
    cfg.pop('type')
    orc = O.HRFuserOracle(**cfg)
    O.seeded_fill_(orc, 0)
    orc.train()
    cores = torch.get_num_threads()
    x, mods = O.seeded_inputs(B, H, W, mc, seed=1)

    def step():
        orc.zero_grad(set_to_none=True)
        ys = orc(x, [m.clone() for m in mods])
        sum(y.mean() for y in ys).backward()
    step()                                                        # warm-up (primitive creation)
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    dt = (time.perf_counter() - t0) / iters
    return {'value': round(B / dt, 4), 'unit': 'images/sec', 'cores': cores, 'kind': 'port',
            'sample': f'{iters} timed train fwd+bwd iterations (after 1 warm-up) of the same {B}x3x{H}x{W} '
                      f'+ {len(mc)} modality batch, torch CPU fp32, {cores} threads, no optimizer step',
            'ms_per_step': round(dt * 1e3, 1)}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the HIP path has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    group = None
    force_coll = os.environ.get('HRF_FORCE_COLLECTIVES', '0') == '1'
    if world > 1 or force_coll:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=dev)          # 'nccl' is RCCL on ROCm
        group = dist.group.WORLD
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    from hrfuser_amd import build_backbone, _lib
    from hrfuser_amd.trainer import Trainer, make_cotangents
    from hrfuser_amd import profiling

    tag = args.model if (world == 1 and not force_coll) else args.model.replace('_bn', '')
    cfg = load_cfg(tag)
    stf = tag.startswith('t_stf')
    H = args.height or 384
    W = args.width or (1248 if stf else 640)
    mc = cfg.get('mod_in_channels', [3, 3])
    torch.manual_seed(1234)                                       # same init on every rank (DP)
    net = build_backbone(copy.deepcopy(cfg)).to(dev)
    with torch.no_grad():                                         # non-trivial RPB so that path is exercised
        for n, p in net.named_parameters():
            if n.endswith('relative_position_bias_table'):
                p.normal_(0, 0.02)
    net.train()
    g = torch.Generator().manual_seed(100 + rank)                 # different data per rank
    B = args.batch
    x = torch.randn(B, 3, H, W, generator=g).to(dev)
    mods = [torch.randn(B, c, H, W, generator=g).to(dev) for c in mc]
    if os.environ.get('HRF_BENCH_CHANNELS_LAST', '0') == '1':     # inputs as hrfuser_amd.pipeline delivers them (experiment)
        x = x.contiguous(memory_format=torch.channels_last)
        mods = [m.contiguous(memory_format=torch.channels_last) for m in mods]
    cots = make_cotangents(net, x, mods)
    trainer = Trainer(net, lr=1e-3 if stf else 3e-4, group=group, world_size=world)

    use_graph = not args.no_graph
    if use_graph:
        try:
            trainer.capture(x, mods, cots)
        except Exception as e:                                    # e.g. collective not capturable
            if rank == 0:
                print(f'[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager', file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()
    run = trainer.replay if use_graph else (lambda: trainer.step(x, mods, cots))

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    # secondary metric: eval forward ms/img (BN running stats), hipGraph replay
    fwd_ms = None
    if rank == 0:
        try:
            fwd_ms = profiling.time_eval_forward(net, x, mods, use_graph=not args.no_graph)
        except Exception as e:
            print(f'[bench] eval-forward timing failed: {e}', file=sys.stderr)
        net.train()

    roof = None
    if not args.no_roofline:          # every rank runs it (the step contains collectives when N > 1)
        table = profiling.profile_step(trainer, x, mods, cots, steps=args.profile_steps)
        traffic = {}
        tpath = os.path.join(ROOT, 'profiles', 'r01_hbm_traffic.json')     # PMC-measured bytes per launch by shape
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = {k: v['bytes_per_launch'] for k, v in json.load(fh).get('shapes', {}).items()}
        roof = profiling.roofline_of_dominant(table, PEAK_F32_MFMA, PEAK_HBM, traffic)
        if roof is not None and roof.get('kernel') == 'wgrad_dense_kernel' and world == 1:
            # the product issues this family as grouped launches (several problems per launch): price the launches
            # that actually run - in situ, HIP events on the launching stream - and keep the isolated per-problem
            # figures (graph-batched single launches) beside them
            try:
                grows = profiling.grouped_wgrad_report(trainer, x, mods, cots)
                gtraffic = {}
                if os.path.exists(tpath):
                    with open(tpath) as fh:
                        gtraffic = {k: v['bytes_per_launch'] for k, v in json.load(fh).get('grouped', {}).items()}
                groof = profiling.roofline_grouped(grows, PEAK_F32_MFMA, PEAK_HBM, gtraffic)
                if groof is not None:
                    groof['isolated_per_problem'] = {k: roof[k] for k in ('achieved', 'frac', 'launches_per_step', 'avg_launch_us',
                                                                          'time_per_step_ms', 'dominant_shape') if k in roof}
                    roof = groof
                    if args.dump_kernels and rank == 0:
                        with open(args.dump_kernels.replace('.json', '_grouped_wgrad.json'), 'w') as fh:
                            json.dump(grows, fh, indent=1)
            except Exception as e:
                print(f'[bench] grouped weight-gradient timing failed: {e}', file=sys.stderr)
        if args.dump_kernels and rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump_kernels)), exist_ok=True)
            with open(args.dump_kernels, 'w') as fh:
                json.dump({'kernels': profiling.table_json(table, PEAK_F32_MFMA, PEAK_HBM),
                           'signatures': profiling.profile_step.last_signatures[:60]}, fh, indent=1)

    neck = None
    if rank == 0 and world == 1 and not args.no_neck:
        # the HRFPN neck that consumes the four maps, timed on its own (not part of `value`)
        try:
            widths = cfg['extra']['stage4']['num_channels']
            neck = profiling.time_neck(list(widths), B, H // 4, W // 4)
        except Exception as e:
            neck = {'error': str(e)[:200]}

    feat = None
    if rank == 0 and world == 1 and not args.no_neck and use_graph:
        # extract_feat training step = backbone + neck on both tapes in ONE hipGraph (not part of `value`)
        try:
            from hrfuser_amd.detector import FeatureExtractor, ExtractTrainer, make_pyramid_cotangents
            from hrfuser_amd import HRFPN
            widths = list(cfg['extra']['stage4']['num_channels'])
            nk = HRFPN(in_channels=widths, out_channels=256)
            nk.init_weights()
            fx = FeatureExtractor(net, nk.to(dev))
            fx.train()
            pc = make_pyramid_cotangents(fx, x, mods)
            et = ExtractTrainer(fx)
            et.capture(x, mods, pc)
            for _ in range(5):
                et.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                et.replay()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            feat = {'workload': f'extract_feat train step: {tag} backbone + HRFPN {widths}->256, {B} img, one hipGraph',
                    'ms_per_step': round(ms, 3), 'images_per_sec': round(B / ms * 1e3, 2)}
        except Exception as e:
            feat = {'error': str(e)[:200]}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(tag, B, H, W, mc)

    if rank == 0:
        line = {
            'metric': 'train images/sec HRFuser-T r640 3-modal @1/2/4/8 MI355X; fwd ms/img',
            'value': round(value, 3), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'{tag} backbone train step (fwd+bwd+grad exchange+AdamW), {B} img/GPU, '
                                   f'{B}x3x{H}x{W} camera + {len(mc)} modalities, '
                                   + ('SyncBN+RCCL all-reduce' if world > 1 else 'BN, single GPU'),
                       'global_batch': B * world, 'parallelism': f'dp{world}',
                       'launch': 'hipGraph replay' if use_graph else 'eager'},
            'fwd_ms_per_img': fwd_ms,
            'roofline': roof, 'cpu_baseline': cpu, 'neck': neck, 'extract_feat': feat,
        }
        print(json.dumps(line), flush=True)
    sys.stdout.flush()
    sys.stderr.flush()
    if world > 1 or force_coll:
        import torch.distributed as dist
        try:
            dist.barrier()
        except Exception:
            pass
        # RCCL/HIP teardown of a process that holds captured graphs with collectives can abort at
        # interpreter exit on ROCm 7.0; the result line is already out, so leave without destructors.
        os._exit(0)


if __name__ == '__main__':
    main()
