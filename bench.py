#!/usr/bin/env python3
"""bench.py — HRFuser backbone training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = forward (train-mode BN) + backward + gradient exchange + fused AdamW of the HRFuser-T nuScenes backbone
on a synthetic batch of 2 images per GPU (2x3x384x640 camera + lidar + radar; BASELINE.json configs[1]; SyncBN + RCCL
all-reduce when N > 1; weak scaling).  Everything in the timed region runs on the hand-written HIP kernels (no oracle,
no CPU fallback).  `value` = images of all ranks / wall time of exactly K steps between two barriers (max over ranks).

The JSON line also carries
  step_ms         per-step GPU durations of the same K steps from HIP events on the launch stream: median / p10 / p90
  finite          outputs and parameters checked for NaN / Inf after the timed steps (a non-finite run is not a result)
  eager_autograd  the route an mmdet user gets: `backbone(img, mods)` + `loss.backward()` through torch.autograd + the
                  fused AdamW, eager (no hipGraph), ms per step
  roofline        the kernel family with the LARGEST time per step (isolated per-launch durations: every distinct
                  (entry point, shape) of a step re-issued 20x back-to-back in a captured hipGraph, HIP events on the launch
                  stream): algorithmic bytes (or flops) per launch / average launch duration, against 8 TB/s resp. the
                  dense fp32 MFMA peak; `traffic` = PMC-measured fabric bytes per launch with its source stated (a
                  separate `rocprofv3 --pmc` run kept under profiles/), or null
  step_roofline   the whole step against its bound: eager-equivalent fp32 bytes and flops of the reference graph
                  (SURVEY 8d / BASELINE.md section 4) over the measured step time
  cpu_baseline    the PyTorch-CPU oracle (kind "port": bit-exact restatement of the reference backbone) timed on the host
                  cores, same workload, train forward+backward, at 8 threads and at all physical cores
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
# algorithmic work per image of the reference graph (train = 3 x forward): GFLOP, eager-equivalent fp32 GB
WORK = {'t_nus': (106.2, 3 * 5.58), 'b_nus': (750.2, 3 * 20.04), 't_stf': (265.4, 3 * 13.58)}


def parse(argv=None):
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
    ap.add_argument('--no-neck', action='store_true', help='skip the HRFPN neck / extract_feat timings (consumers of the 4 maps)')
    ap.add_argument('--no-eager', action='store_true', help='skip the eager torch.autograd route timing')
    ap.add_argument('--profile-steps', type=int, default=1)
    ap.add_argument('--dump-kernels', default='', help='write the per-kernel table (JSON) to this path')
    ap.add_argument('--no-sync-ab', action='store_true', help='N > 1: skip the A/B of the per-lane-communicator SyncBN schedule')
    ap.add_argument('--sync-ab-child', action='store_true', help=argparse.SUPPRESS)     # internal: the A/B child job
    ap.add_argument('--sync-ab-timeout', type=float, default=240.0)
    ap.add_argument('--backend', default=os.environ.get('HRF_BENCH_BACKEND', 'nccl'),
                    help="torch.distributed backend: 'nccl' (= RCCL, the product path) | 'gloo' (flow tests of the N > 1 path on a "
                         "box with fewer GPUs than ranks: the ranks share GPU 0, eager launches)")
    return ap.parse_args(argv)


def load_cfg(tag):
    from hrfuser_amd.configs import backbone_cfg          # the resolved configs ship with the package (not with the test tree)
    return backbone_cfg(tag)


def cpu_baseline(tag, B, H, W, mc, iters=3):
    """Oracle (PyTorch-CPU restatement of the reference backbone, oracle/hrfuser_oracle.py) timed on the host cores:
    train-mode forward+backward of the same workload, at 8 threads (the setting of SURVEY section 6) and at all physical
    cores.  Bounded sample: 1 warm-up + `iters` timed iterations per setting."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import hrfuser_oracle as O
    cfg = copy.deepcopy(load_cfg(tag))
    cfg.pop('type')
    orc = O.HRFuserOracle(**cfg)
    O.seeded_fill_(orc, 0)
    orc.train()
    x, mods = O.seeded_inputs(B, H, W, mc, seed=1)

    def step():
        orc.zero_grad(set_to_none=True)
        ys = orc(x, [m.clone() for m in mods])
        sum(y.mean() for y in ys).backward()

    def eval_fwd():
        with torch.no_grad():
            orc(x, [m.clone() for m in mods])

    phys = max(1, (os.cpu_count() or 2) // 2)
    runs, eruns = [], []
    prev = torch.get_num_threads()
    for n in sorted({min(8, phys), phys}):
        torch.set_num_threads(n)
        orc.train()
        step()                                                        # warm-up (primitive creation)
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        runs.append({'value': round(B / med, 4), 'cores': n, 'ms_per_step': round(med * 1e3, 1)})
        # eval forward (BN running statistics) beside the GPU path's fwd_ms_per_img: BASELINE.md section 3 (2 warm-ups, 5 timed, median)
        orc.eval()
        eval_fwd()
        eval_fwd()
        es = []
        for _ in range(5):
            t0 = time.perf_counter()
            eval_fwd()
            es.append(time.perf_counter() - t0)
        es.sort()
        eruns.append({'cores': n, 'fwd_ms_per_img': round(es[len(es) // 2] / B * 1e3, 1)})
    torch.set_num_threads(prev)
    best = max(runs, key=lambda r: r['value'])
    ebest = min(eruns, key=lambda r: r['fwd_ms_per_img'])
    return {'value': best['value'], 'unit': 'images/sec', 'cores': best['cores'], 'kind': 'port',
            'sample': f'median of {iters} timed train fwd+bwd iterations (after 1 warm-up) of the same {B}x3x{H}x{W} + '
                      f'{len(mc)} modality batch, torch CPU fp32, no optimizer step; run at 8 threads and at all '
                      f'{phys} physical cores, the faster one is `value`',
            'ms_per_step': best['ms_per_step'], 'runs': runs,
            'eval_fwd_ms_per_img': ebest['fwd_ms_per_img'], 'eval_fwd_cores': ebest['cores'], 'eval_fwd_runs': eruns,
            'eval_fwd_sample': f'median of 5 eval-mode forwards (after 2 warm-ups) of the same {B}-image batch per thread setting'}


# --------------------------------------------------------------------------------------------------------------------
# Multi-rank launch.  The reference starts its ranks with tools/dist_train.sh:22-24 (torch.distributed.launch
# --nproc_per_node=$GPUS); here `python bench.py --gpus N` does the same for itself when no launcher did: it starts
# `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process group before anything in this
# process touches the GPU (a process that has initialised HIP must never exec another program), relays rank 0's JSON
# line and exits with the children's status.
def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_command(argv, gpus, port):
    """The command line of the N-rank job (one process per GPU over RCCL): what tools/dist_train.sh forms for the reference."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}',
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def run_ranks(argv, gpus, env=None, timeout=None):
    """Start the N-rank job in its own process group, wait for it (or kill exactly that group on timeout).
    -> (return code | None on timeout, the last JSON object line of its stdout | None, tail of its other output)"""
    cmd = rank_command(argv, gpus, free_port())
    e = dict(os.environ if env is None else env)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'ROLE_RANK', 'MASTER_ADDR', 'MASTER_PORT',
              'TORCHELASTIC_RUN_ID', 'TORCHELASTIC_RESTART_COUNT', 'TORCHELASTIC_MAX_RESTARTS'):
        e.pop(k, None)                                     # a job started from inside a rank must not inherit its rendezvous
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC (RCCL between processes on this driver)
    e.setdefault('OMP_NUM_THREADS', '4')
    return run_child(cmd, e, timeout)


_AB_ARMS = {
    'lane_comms': ('one communicator per lane (HRF_SYNC_LANE_COMMS=1)', {'HRF_SYNC_LANE_COMMS': '1', 'HRF_SYNC_P2P': '0'}),
    'p2p': ('peer-to-peer exchange through IPC inboxes, no communicator (HRF_SYNC_P2P=1)', {'HRF_SYNC_AB_MODE': 'p2p'}),
}


def _run_ab_arm(arm, argv, gpus, timeout):
    name, extra = _AB_ARMS[arm]
    env = dict(os.environ)
    env.update(extra)
    env.pop('HRF_FORCE_COLLECTIVES', None)
    t0 = time.time()
    rc, line, rest = run_ranks(list(argv) + ['--sync-ab-child'], gpus, env=env, timeout=timeout)
    if rc is None:
        return {'schedule': name, 'error': f'no result within {timeout:.0f} s (child process group killed)'}
    if line is None or 'sync_ab' not in line:
        return {'schedule': name, 'error': f'child exited with {rc} and no result: {rest[-400:]}'}
    res = line['sync_ab']
    res['wall_s'] = round(time.time() - t0, 1)
    return res


def run_sync_ab(argv, gpus, timeout):
    """A/B of the SyncBN schedules that need no hop to the main lane, each in a FRESH child job with a timeout:
    one RCCL communicator per lane (HRF_SYNC_LANE_COMMS=1), and the peer-to-peer exchange through IPC inboxes
    (HRF_SYNC_P2P=1).  A child first checks that one step's gradient arena equals the main-lane schedule's (rel-L2 < 1e-6),
    then times the captured step.  Reported beside the headline, never as the headline; a hang or failure is text."""
    res = _run_ab_arm('lane_comms', argv, gpus, timeout)
    res['p2p'] = _run_ab_arm('p2p', argv, gpus, timeout)
    return res


def strip_flag(argv, flag, has_value=False):
    out, skip = [], False
    for a in argv:
        if skip:
            skip = False
            continue
        if a == flag:
            skip = has_value
            continue
        if has_value and a.startswith(flag + '='):
            continue
        out.append(a)
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher.  Runs before any GPU call of this process."""
    n = args.gpus
    have = torch.cuda.device_count()                       # counting devices does not initialise HIP on this image
    share = args.backend == 'gloo'
    if have < n and not share:
        print(f'[bench] --gpus {n} needs {n} GPUs on this node, {have} visible: nothing was run '
              f'(one process per GPU over RCCL; --backend gloo runs a flow test with the ranks sharing GPU 0)', file=sys.stderr)
        return 3
    env = dict(os.environ)
    env['HRF_BENCH_LAUNCHED_BY_BENCH'] = '1'                # the ranks leave the schedule A/B to this parent
    rc, line, rest = run_ranks(argv, n, env=env)
    if rest:
        print(rest, file=sys.stderr)
    if rc != 0 or line is None:
        print(f'[bench] the {n}-rank job exited with {rc}' + ('' if line is not None else ' and printed no result line'), file=sys.stderr)
        return rc if rc else 1
    if not args.no_sync_ab and line.get('sync_ab') is None:
        line['sync_ab'] = run_sync_ab(strip_flag(argv, '--dump-kernels', True), n, args.sync_ab_timeout)
    print(json.dumps(line), flush=True)
    return 0


# --------------------------------------------------------------------------------------------------------------------
# First contact with N > 1 GPUs must not be able to hang or to lie.  A rank process started by ANY launcher
# (the driver's torch.distributed.run, or launch_ranks above) is a SUPERVISOR: it never touches the GPU, it starts the real
# rank (HRF_BENCH_WORKER=1: the same script, same RANK / LOCAL_RANK / WORLD_SIZE, a rendezvous port of its own) as a child
# process group under a wall-clock limit, and - when that attempt fails or runs out of time on the default SyncBN schedule
# (peer-to-peer exchange kernels that spin on IPC inboxes, never run across xGMI before) - kills exactly that group and starts
# a FRESH child on the RCCL packed schedule (HRF_SYNC_P2P=0).  Every supervisor takes the same decisions from its own child's
# exit status: a collective job fails on all ranks or on none (the survivors of a lost peer sit in a barrier / exchange until
# their limit).  The device-side spin limit of the exchange is 60 s here (HRF_P2P_TIMEOUT_S), the job limits add up to less
# than the driver's.  Never an exec of a process that has initialised HIP.
ATTEMPTS = (('auto', {}), ('rccl_packed_fallback', {'HRF_SYNC_P2P': '0'}))


def worker_command(argv):
    over = os.environ.get('HRF_BENCH_WORKER_CMD')           # tests/test_bench_launch.py: a stand-in worker (no GPU needed)
    if over:
        import shlex
        return shlex.split(over)
    return [sys.executable, os.path.abspath(__file__)] + list(argv)


def run_child(cmd, env, timeout):
    """-> (rc | None on timeout, last JSON object line of stdout | None, tail of the other stdout lines)"""
    import signal
    import subprocess
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True, start_new_session=True)

    def kill_group():
        try:
            os.killpg(proc.pid, signal.SIGKILL)            # the process GROUP we started, nothing else
        except ProcessLookupError:
            pass

    def on_signal(sig, _frame):                            # the launcher ends this supervisor (another rank failed, ^C): the worker
        kill_group()                                       # sits in a session of its own and would otherwise keep its GPU
        os._exit(128 + sig)

    saved = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        out, _ = proc.communicate(timeout=timeout)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        kill_group()
        out, _ = proc.communicate()
        rc = None
    finally:
        for sg, h in saved.items():
            signal.signal(sg, h)
    line, rest = None, []
    for ln in (out or '').splitlines():
        if ln.startswith('{') and ln.rstrip().endswith('}'):
            try:
                line = json.loads(ln)
                continue
            except ValueError:
                pass
        rest.append(ln)
    return rc, line, '\n'.join(rest[-15:])


def supervise_rank(args, argv):
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ['WORLD_SIZE'])
    base_port = int(os.environ.get('MASTER_PORT', '29500'))
    t0 = time.time()
    budget = float(os.environ.get('HRF_BENCH_BUDGET_S', '1380'))          # main job + fallback + schedule A/B, all of it
    per = float(os.environ.get('HRF_BENCH_ATTEMPT_TIMEOUT_S', '480'))
    attempts = ATTEMPTS[1:] if os.environ.get('HRF_SYNC_P2P') == '0' else ATTEMPTS
    last_rc, tried = 1, []
    for k, (name, extra) in enumerate(attempts):
        left = budget - (time.time() - t0)
        if left < 30:
            break
        env = dict(os.environ)
        env.update(extra)
        env.update(HRF_BENCH_WORKER='1', HRF_BENCH_ATTEMPT=name, HRF_BENCH_LAUNCHED_BY_BENCH='1',
                   MASTER_PORT=str(base_port + 17 + k), TORCHELASTIC_USE_AGENT_STORE='False')
        env.setdefault('MASTER_ADDR', '127.0.0.1')
        env.setdefault('HRF_P2P_TIMEOUT_S', '60')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        rc, line, rest = run_child(worker_command(argv), env, min(per, left))
        tried.append({'attempt': name, 'rc': rc, 'wall_s': round(time.time() - t0, 1)})
        if rc == 0 and (rank != 0 or line is not None):
            if rank == 0:
                line.setdefault('config', {})['attempts'] = tried
                ab_left = budget - (time.time() - t0)
                if not args.no_sync_ab and os.environ.get('HRF_BENCH_LAUNCHED_BY_BENCH') != '1' and line.get('sync_ab') is None:
                    if ab_left > 2 * 60:
                        line['sync_ab'] = run_sync_ab(strip_flag(argv, '--dump-kernels', True), world,
                                                      min(args.sync_ab_timeout, (ab_left - 20) / 2))
                    else:
                        line['sync_ab'] = {'skipped': f'{ab_left:.0f} s of the wall budget left'}
                print(json.dumps(line), flush=True)
            return 0
        why = 'no result within its limit (process group killed)' if rc is None else f'exit status {rc}' + ('' if line is not None or rank else ', no result line')
        print(f'[bench] rank {rank}: attempt "{name}" failed: {why}' + (f'\n{rest}' if rest else ''), file=sys.stderr, flush=True)
        last_rc = rc if rc else 1
    return last_rc


def init_ranks(args):
    """-> rank, world, device, group (None on one rank without forced collectives), force_coll"""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a ROCm GPU: the HIP path has no CPU fallback')
    if world != args.gpus:
        raise SystemExit(f'[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks')
    share = args.backend == 'gloo'
    ndev = torch.cuda.device_count()
    if world > 1 and not share and local >= ndev:
        raise SystemExit(f'[bench] rank {rank}: LOCAL_RANK {local} but only {ndev} GPUs are visible (one process per GPU)')
    dev = torch.device('cuda', 0 if share else local)
    torch.cuda.set_device(dev)
    group = None
    force_coll = os.environ.get('HRF_FORCE_COLLECTIVES', '0') == '1'
    if world > 1 or force_coll:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if share:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)      # 'nccl' is RCCL on ROCm
        group = dist.group.WORLD
    return rank, world, dev, group, force_coll


def build_workload(args, rank, world, dev, group, force_coll):
    from hrfuser_amd import build_backbone
    from hrfuser_amd.trainer import Trainer, make_cotangents
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
    return tag, cfg, stf, H, W, mc, net, B, x, mods, cots, trainer


def time_steps(args, run, world, dev):
    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        run()
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(args.steps):
        run()
        evs[k + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    per = sorted(evs[k].elapsed_time(evs[k + 1]) for k in range(args.steps))
    return dt, per


def leave(world, force_coll):
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
        if os.environ.get('HRF_BENCH_SOFT_EXIT', '0') != '1':     # (profilers need the normal exit path to flush)
            os._exit(0)


def sync_ab_child(args):
    """One arm of run_sync_ab (N ranks): gradient equivalence of the arm's SyncBN schedule with the main-lane one on one eager
    step each, then the captured step of the arm, timed like the headline.  Rank 0 prints {"sync_ab": {...}}."""
    from hrfuser_amd import runtime as R
    p2p_arm = os.environ.get('HRF_SYNC_AB_MODE') == 'p2p'
    rank, world, dev, group, force_coll = init_ranks(args)
    tag, cfg, stf, H, W, mc, net, B, x, mods, cots, trainer = build_workload(args, rank, world, dev, group, force_coll)
    res = {'schedule': _AB_ARMS['p2p' if p2p_arm else 'lane_comms'][0]}
    try:
        eng = net._engine()

        def select(on):
            if p2p_arm:
                os.environ['HRF_SYNC_P2P'] = '1' if on else '0'      # read when a forward starts (hrfuser_amd/p2p.py)
            else:
                R.set_lane_comms(on)
        select(False)
        trainer.step(x, mods, cots)                               # sizes of the random pools are known after one step
        torch.cuda.synchronize()

        def grads(on):
            select(on)
            torch.manual_seed(777)                                # the same Dropout / DropPath draws for both schedules
            trainer.step(x, mods, cots, grads_only=True)
            trainer.check()
            torch.cuda.synchronize()
            return eng.flat_g.clone(), trainer.collectives_per_step, getattr(trainer, 'p2p_exchanges_per_step', 0)
        g_main, n_main, _ = grads(False)
        g_arm, n_arm, n_px = grads(True)
        err = float((g_arm.double() - g_main.double()).norm() / g_main.double().norm().clamp_min(1e-300))
        res.update({'grad_rel_l2_vs_main_lane': err, 'collectives_per_step': n_arm, 'main_lane_collectives_per_step': n_main})
        if p2p_arm:
            res['p2p_exchanges_per_step'] = n_px
        if not (err < 1e-6):
            raise RuntimeError(f'gradient arena differs from the main-lane schedule: rel-L2 {err:.3e}')
        use_graph = not args.no_graph and args.backend == 'nccl'
        if use_graph:
            trainer.capture(x, mods, cots)
        run = trainer.replay if use_graph else (lambda: trainer.step(x, mods, cots))
        dt, per = time_steps(args, run, world, dev)
        trainer.check()
        res.update({'ms_per_step': round(dt / args.steps * 1e3, 4), 'images_per_sec': round(B * world * args.steps / dt, 3),
                    'launch': 'hipGraph replay' if use_graph else 'eager', 'finite': bool(torch.isfinite(eng.flat_p).all())})
        if p2p_arm:
            # ... and the collective main-lane schedule timed the same way in the same job: the baseline of the comparison
            # when the headline itself ran on the peer-to-peer exchange
            try:
                select(False)
                trainer.step(x, mods, cots)
                torch.cuda.synchronize()
                if use_graph:
                    trainer.capture(x, mods, cots)
                run = trainer.replay if use_graph else (lambda: trainer.step(x, mods, cots))
                dt, per = time_steps(args, run, world, dev)
                res['main_lane_ms_per_step'] = round(dt / args.steps * 1e3, 4)
            except Exception as e:
                res['main_lane_error'] = f'{type(e).__name__}: {str(e)[:200]}'
    except Exception as e:
        res['error'] = f'{type(e).__name__}: {str(e)[:300]}'
    if rank == 0:
        print(json.dumps({'sync_ab': res}), flush=True)
    leave(world, True)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args, argv))                         # nothing above touched the GPU
    if args.gpus > 1 and os.environ.get('HRF_BENCH_WORKER') != '1' and not args.sync_ab_child:
        if int(os.environ.get('WORLD_SIZE', '1')) != args.gpus:
            raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE')} ranks")
        sys.exit(supervise_rank(args, argv))                       # this process stays off the GPU
    if args.sync_ab_child:
        return sync_ab_child(args)
    rank, world, dev, group, force_coll = init_ranks(args)
    from hrfuser_amd import profiling
    tag, cfg, stf, H, W, mc, net, B, x, mods, cots, trainer = build_workload(args, rank, world, dev, group, force_coll)

    ranks_in_group = None
    if group is not None:
        import torch.distributed as dist
        ranks_in_group = dist.get_world_size(group)               # what the communicator itself reports
        if ranks_in_group != world:
            raise SystemExit(f'[bench] the process group has {ranks_in_group} ranks, WORLD_SIZE says {world}')
    use_graph = not args.no_graph and args.backend == 'nccl'
    capture_note = None
    if world > 1:
        # first contact: two EAGER steps before anything is captured - a protocol hang of the peer-to-peer exchange ends in its
        # 60 s device-side limit with (source, slot) in the error, every rank exits non-zero, the supervisors start the fallback
        for _ in range(2):
            trainer.step(x, mods, cots)
            trainer.check()
        torch.cuda.synchronize()
    if use_graph:
        try:
            trainer.capture(x, mods, cots)
        except Exception as e:                                    # e.g. collective not capturable
            capture_note = f'hipGraph capture failed ({type(e).__name__}: {str(e)[:120]}); timed EAGER'
            if rank == 0:
                print(f'[bench] {capture_note}', file=sys.stderr)
            use_graph = False
            torch.cuda.synchronize()
    run = trainer.replay if use_graph else (lambda: trainer.step(x, mods, cots))
    dt, per = time_steps(args, run, world, dev)
    trainer.check()                                               # a timed-out peer-to-peer SyncBN exchange is not a result
    ms_per_step = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    pick = lambda f: round(per[min(len(per) - 1, int(f * len(per)))], 4)
    step_ms = {'median': pick(0.5), 'p10': pick(0.1), 'p90': pick(0.9),
               'timer': 'HIP events on the launch stream around each of the timed steps'}

    # a run that went non-finite is not a result
    eng = net._engine()
    outs = trainer._graph_outs if use_graph else trainer.step(x, mods, cots)
    finite = bool(torch.isfinite(eng.flat_p).all()) and all(bool(torch.isfinite(o.t).all()) for o in outs)
    if not finite:
        raise SystemExit('[bench] non-finite parameters / outputs after the timed steps: no result')

    # secondary metric: eval forward ms/img (BN running stats), hipGraph replay
    fwd_ms = None
    if rank == 0:
        try:
            fwd_ms = profiling.time_eval_forward(net, x, mods, use_graph=not args.no_graph)
        except Exception as e:
            print(f'[bench] eval-forward timing failed: {e}', file=sys.stderr)
        net.train()

    eager = None
    module_graph = None
    if rank == 0 and world == 1 and not force_coll and not args.no_eager:
        # what `backbone(img, mods)` + `loss.backward()` costs through torch.autograd (the drop-in route an mmdet user gets):
        # with eager launches, and with the module-boundary hipGraphs (captured from the third call of a signature on)
        def autograd_route(graphs, n_warm, n_timed):
            saved = os.environ.get('HRF_MODULE_GRAPH')
            os.environ['HRF_MODULE_GRAPH'] = '1' if graphs else '0'
            try:
                for it in range(n_warm + n_timed):
                    if it == n_warm:
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                    ys = net(x, list(mods))
                    loss = sum((y * c.permute(0, 3, 1, 2)).sum() for y, c in zip(ys, cots))
                    eng.flat_g.zero_()
                    loss.backward()
                    trainer.optimizer_step()
                torch.cuda.synchronize()
                return round((time.perf_counter() - t1) / n_timed * 1e3, 3)
            finally:
                if saved is None:
                    os.environ.pop('HRF_MODULE_GRAPH', None)
                else:
                    os.environ['HRF_MODULE_GRAPH'] = saved
        try:
            eager = {'ms_per_step': autograd_route(False, 3, 10),
                     'route': 'backbone(img, mods) -> synthetic loss -> loss.backward() (torch.autograd.Function bridge) -> '
                              'fused AdamW; eager launches, no hipGraph (HRF_MODULE_GRAPH=0)'}
        except Exception as e:
            eager = {'error': f'{type(e).__name__}: {str(e)[:160]}'}
            torch.cuda.synchronize()
        try:
            import warnings
            with warnings.catch_warnings(record=True) as wl:
                warnings.simplefilter('always')
                ms = autograd_route(True, 5, 20)
            ents = [e for k, e in net.__dict__.get('_hrf_graphs', {}).items() if k != '_setup']
            module_graph = {'ms_per_step': ms, 'captured': any(e.fwd is not None and e.bwd is not None for e in ents),
                            'route': 'the same calls; the module replays one forward and one backward hipGraph per input signature '
                                     '(copy-in / copy-out at the module boundary), captured at the third call',
                            'warnings': [str(w.message)[:160] for w in wl][:3]}
        except Exception as e:
            module_graph = {'error': f'{type(e).__name__}: {str(e)[:160]}'}
            torch.cuda.synchronize()

    roof = None
    if not args.no_roofline:          # every rank runs it (the step contains collectives when N > 1)
        table = profiling.profile_step(trainer, x, mods, cots, steps=args.profile_steps)
        try:
            grouped = profiling.grouped_wgrad_report(trainer, x, mods, cots) if use_graph else None
        except Exception:
            grouped = None
            torch.cuda.synchronize()
        roof = profiling.roofline_of_dominant(table, PEAK_F32_MFMA, PEAK_HBM, os.path.join(ROOT, 'profiles'), grouped=grouped)
        nl = sum(t[0] // t[4] for t in table.values())
        if args.dump_kernels and rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(args.dump_kernels)), exist_ok=True)
            with open(args.dump_kernels, 'w') as fh:
                json.dump({'kernels': profiling.table_json(table, PEAK_F32_MFMA, PEAK_HBM),
                           'signatures': profiling.profile_step.last_signatures[:80], 'launches_per_step': nl}, fh, indent=1)
        issued, carried = profiling.merged_launch_counts(trainer, x, mods, cots)
        if roof is not None:
            # C-ABI calls of a step, and what reaches the GPU after the equal-shape calls of the sensor streams were merged
            roof['library_calls_per_step'] = nl
            roof['library_launches_per_step'] = nl - (carried - issued)
            roof['merged_launches'] = {'launches': issued, 'calls_carried': carried}

    stages = None
    if rank == 0 and world == 1 and use_graph and not args.no_roofline and (H, W) == (384, 1248 if stf else 640):
        try:
            stages = profiling.stage_table(trainer, x, mods, cots, tag, PEAK_F32_MFMA, PEAK_HBM)
        except Exception as e:
            stages = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
            torch.cuda.synchronize()

    gfl, gby = WORK.get(tag.replace('_bn', ''), (None, None))
    step_roof = None
    if gfl is not None and (H, W) == (384, 1248 if stf else 640):
        sec = ms_per_step * 1e-3
        step_roof = {'eager_bytes_per_step_GB': round(gby * B, 2), 'flops_per_step_G': round(gfl * B, 1),
                     'bytes_frac': round(gby * 1e9 * B / sec / PEAK_HBM, 4), 'flops_frac': round(gfl * 1e9 * B / sec / PEAK_F32_MFMA, 4),
                     'bound_img_s': round(1.0 / max(gby * 1e9 / PEAK_HBM, gfl * 1e9 / PEAK_F32_MFMA), 1),
                     'note': 'eager-equivalent fp32 bytes / flops of the reference graph per step (SURVEY 8d, train = 3 x forward) over '
                             'the measured step time, against 8 TB/s and 157.3 TFLOP/s; bound_img_s = per-GPU ceiling of that model'}

    neck = None
    feat = None
    if rank == 0 and world == 1 and not args.no_neck:
        # the HRFPN neck that consumes the four maps, timed on its own (not part of `value`)
        try:
            widths = cfg['extra']['stage4']['num_channels']
            neck = profiling.time_neck(list(widths), B, H // 4, W // 4)
        except Exception as e:
            neck = {'error': str(e)[:200]}
        if use_graph:
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
                t1 = time.perf_counter()
                for _ in range(20):
                    et.replay()
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / 20 * 1e3
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
                       'launch': 'hipGraph replay' if use_graph else 'eager',
                       'collectives_per_step': trainer.collectives_per_step,
                       'p2p_exchanges_per_step': getattr(trainer, 'p2p_exchanges_per_step', 0),
                       'sync_schedule': getattr(trainer, 'sync_schedule', None) if (world > 1 or force_coll) else None,
                       'backend': ('RCCL (torch.distributed nccl)' if args.backend == 'nccl' else args.backend + ' (flow test, ranks share GPU 0)') if (world > 1 or force_coll) else None,
                       'ranks_in_group': ranks_in_group, 'attempt': os.environ.get('HRF_BENCH_ATTEMPT'),
                       'exchange_lanes_hist': {f'{k[0]}{"m" if k[1] else ""}': v for k, v in sorted(getattr(trainer, 'exchange_hist', {}).items())}},
            'step_ms': step_ms, 'finite': finite, 'fwd_ms_per_img': fwd_ms, 'eager_autograd': eager, 'module_graph': module_graph,
            'roofline': roof, 'step_roofline': step_roof, 'stage_roofline': stages, 'cpu_baseline': cpu, 'neck': neck,
            'extract_feat': feat,
        }
        if capture_note:
            line['config']['capture_note'] = capture_note
    if world > 1 and not args.no_sync_ab and os.environ.get('HRF_BENCH_LAUNCHED_BY_BENCH') != '1':
        # started by an external launcher (the driver's torch.distributed.run): rank 0 runs the schedule A/B itself, as a
        # child job, once every rank is done with the GPU-heavy part (a child process, never an exec)
        import torch.distributed as dist
        dist.barrier()
        torch.cuda.synchronize()
        if rank != 0:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)                                        # no barrier kernel left spinning beside the child job
        line['sync_ab'] = run_sync_ab(strip_flag(strip_flag(argv, '--dump-kernels', True), '--sync-ab-child'), world, args.sync_ab_timeout)
        print(json.dumps(line), flush=True)
        sys.stdout.flush()
        os._exit(0)
    if rank == 0:
        print(json.dumps(line), flush=True)
    leave(world, force_coll)


if __name__ == '__main__':
    main()
