"""Peer-to-peer SyncBN exchange (csrc/p2p_exchange.hip, hrfuser_amd/p2p.py) - what torch.nn.SyncBatchNorm's exchange of
the per-layer moments does in the reference (norm_cfg type SyncBN, configs/_base_/models/cascade_rcnn_hrfuser_fpn_nus_clr_
fusion.py:2), without a communicator.

  * the C-ABI entry points directly: two ranks' contexts over two inboxes, the protocol in halves on the CPU emulator (one
    launch at a time) and as two CONCURRENT launches that wait for each other on the GPU; generations, parity, time-out word;
  * the whole backbone with HRF_SYNC_P2P=1 on a forced one-rank group (every exchange a real launch through the inbox) against
    the same step through the collective schedule;
  * two PROCESSES sharing one GPU (IPC-mapped inboxes, gloo for the control plane and the gradient buckets): gradients equal to
    the collective schedule's at rel-L2 < 1e-6, and a peer that never arrives makes the step fail loudly."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

from helpers import use_backend
from hrfuser_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KC = _lib.STAT_COPIES


def _contexts(L, dev, world, slot_doubles, nslots, timeout_ticks=0):
    """`world` contexts in ONE process: every rank's inbox is a plain buffer all of them can address."""
    data = world * 2 * slot_doubles
    bufs = [torch.zeros(data + world * 2 * nslots, dtype=torch.float64, device=dev) for _ in range(world)]
    gens = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    errs = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    ctxs = []
    for r in range(world):
        c = _lib.P2p()
        c.world, c.rank = world, r
        for p in range(world):
            c.inbox[p] = bufs[p].data_ptr()
            c.flags[p] = bufs[p].data_ptr() + 8 * data
        c.slot_doubles, c.nslots = slot_doubles, nslots
        c.gen, c.err = gens[r].data_ptr(), errs[r].data_ptr()
        c.timeout_ticks = timeout_ticks
        ctxs.append(c)
    return ctxs, bufs, gens, errs


def _layers(dev, world, Cs, seed):
    g = torch.Generator().manual_seed(seed)
    stats = [[torch.randn(KC * 2 * C, generator=g, dtype=torch.float64).to(dev) for C in Cs] for _ in range(world)]
    rows = [[float(100 + 7 * r + k) for k in range(len(Cs))] for r in range(world)]
    return stats, rows


def _call(L, ctx, stats, Cs, rows, offs, ids, packed, phase, stream=0):
    n = len(Cs)
    L.hrf_p2p_exchange(ctx, (ctypes.c_void_p * n)(*[s.data_ptr() for s in stats]), (ctypes.c_int * n)(*Cs), n,
                       (ctypes.c_double * n)(*rows) if rows is not None else None, (ctypes.c_long * n)(*offs),
                       (ctypes.c_int * n)(*ids), packed, phase, stream)


def _expect(stats, rows, Cs, world):
    sums = [sum(stats[r][k].view(KC, 2 * C).sum(0) for r in range(world)) for k, C in enumerate(Cs)]
    cnt = [sum(rows[r][k] for r in range(world)) for k in range(len(Cs))]
    return torch.cat(sums + [torch.tensor(cnt, dtype=torch.float64, device=sums[0].device)])


def test_p2p_protocol_in_halves_emul():
    """Two ranks, three layers (more than one workgroup, odd widths), two generations (both parities), on the emulator: push
    of both ranks, then wait + reduce of both; a wait without the peer's push sets the error word."""
    dev = use_backend('emul')
    L = _lib.lib()
    Cs = [18, 624, 5]
    offs, o = [], 0
    for C in Cs:
        offs.append(o)
        o += 2 * C + 1
    ids = [4, 0, 2]
    ctxs, bufs, gens, errs = _contexts(L, dev, 2, o + 3, 6)
    for gen in (1, 2):
        stats, rows = _layers(dev, 2, Cs, 10 + gen)
        for r in range(2):
            L.hrf_p2p_tick(gens[r], 0)
            assert int(gens[r]) == gen
        packed = [torch.full((sum(2 * C for C in Cs) + len(Cs),), float('nan'), dtype=torch.float64, device=dev) for _ in range(2)]
        for r in range(2):
            _call(L, ctxs[r], stats[r], Cs, rows[r], offs, ids, packed[r], 1)
        for r in range(2):
            _call(L, ctxs[r], stats[r], Cs, rows[r], offs, ids, packed[r], 2)
        want = _expect(stats, rows, Cs, 2)
        assert torch.equal(packed[0], packed[1])                    # bit-identical on both ranks: sums in rank order
        for r in range(2):
            assert float((packed[r] - want).abs().max()) <= 1e-12 * float(want.abs().max()), r
            assert int(errs[r]) == 0
    # generation 3: rank 1 never pushes -> rank 0's wait reports source 1, slot id 4 (first layer)
    L.hrf_p2p_tick(gens[0], 0)
    stats, rows = _layers(dev, 2, Cs, 99)
    junk = torch.zeros(sum(2 * C for C in Cs) + len(Cs), dtype=torch.float64, device=dev)
    _call(L, ctxs[0], stats[0], Cs, rows[0], offs, ids, junk, 3)
    e = int(errs[0])
    assert e != 0 and (e >> 32) - 1 == 1 and (e & 0xffffffff) - 1 in ids
    assert bool(torch.isnan(junk).all())                # a lost exchange never yields statistics: NaN, all of it
    # ... and every later exchange of the process is poisoned as well, without waiting (the host raises at its next step boundary)
    L.hrf_p2p_tick(gens[0], 0)
    junk2 = torch.zeros_like(junk)
    _call(L, ctxs[0], stats[0], Cs, rows[0], offs, ids, junk2, 3)
    assert bool(torch.isnan(junk2).all())


def test_p2p_argument_checks_emul():
    dev = use_backend('emul')
    L = _lib.lib()
    ctxs, bufs, gens, errs = _contexts(L, dev, 1, 40, 2)
    st = torch.zeros(KC * 2 * 18, dtype=torch.float64)
    out = torch.zeros(37, dtype=torch.float64)
    with pytest.raises(_lib.HRFuserHipError):                      # slot beyond the region
        _call(L, ctxs[0], [st], [18], [1.0], [10], [0], out, 3)
    with pytest.raises(_lib.HRFuserHipError):                      # flag index beyond the table
        _call(L, ctxs[0], [st], [18], [1.0], [0], [2], out, 3)
    _call(L, ctxs[0], [st], [18], [1.0], [0], [1], out, 3)          # world = 1: self-delivery


@pytest.mark.gpu
def test_p2p_two_ranks_concurrent_launches_gpu():
    """The product form (phase 3) on the GPU: rank 0's and rank 1's launches run CONCURRENTLY on two streams of one process and
    wait for each other's flags; 20 generations back to back (slot reuse, both parities), wide and narrow layers."""
    dev = use_backend('hip')
    L = _lib.lib()
    Cs = [18, 72, 624, 2496, 36, 144, 288, 64, 256]                # 9 layers: two launches of <= 8 workgroups
    offs, o = [], 0
    for C in Cs:
        offs.append(o)
        o += 2 * C + 1
    ids = list(range(len(Cs)))
    ctxs, bufs, gens, errs = _contexts(L, dev, 2, o, len(Cs), timeout_ticks=int(20e8))
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    n = sum(2 * C for C in Cs) + len(Cs)
    for gen in range(1, 21):
        stats, rows = _layers(dev, 2, Cs, gen)
        packed = [torch.full((n,), float('nan'), dtype=torch.float64, device=dev) for _ in range(2)]
        torch.cuda.synchronize()
        for r in (1, 0) if gen % 2 else (0, 1):
            with torch.cuda.stream(s[r]):
                L.hrf_p2p_tick(gens[r], s[r].cuda_stream)
                _call(L, ctxs[r], stats[r], Cs, rows[r], offs, ids, packed[r], 0, s[r].cuda_stream)
        torch.cuda.synchronize()
        want = _expect(stats, rows, Cs, 2)
        assert torch.equal(packed[0], packed[1]), gen
        for r in range(2):
            assert int(errs[r]) == 0
            assert float((packed[r] - want).abs().max()) <= 1e-12 * float(want.abs().max()), (gen, r)


@pytest.mark.gpu
def test_p2p_timeout_sets_error_word_gpu():
    """A peer that never arrives: the launch gives up after the time-out (here 5 ms), reports (source, slot) and the GPU
    stays usable."""
    dev = use_backend('hip')
    L = _lib.lib()
    ctxs, bufs, gens, errs = _contexts(L, dev, 2, 64, 2, timeout_ticks=int(5e5))
    st = torch.ones(KC * 2 * 18, dtype=torch.float64, device=dev)
    out = torch.zeros(37, dtype=torch.float64, device=dev)
    L.hrf_p2p_tick(gens[0], _lib.stream_ptr())
    _call(L, ctxs[0], [st], [18], [3.0], [0], [1], out, 0, _lib.stream_ptr())
    torch.cuda.synchronize()
    e = int(errs[0])
    assert (e >> 32) - 1 == 1 and (e & 0xffffffff) - 1 == 1
    assert bool(torch.isnan(out).all())                 # never a sum over an inbox whose flag did not arrive
    assert float(torch.ones(4, device=dev).sum()) == 4.0


WORKER = r'''
import copy, json, os, sys, torch
ROOT = %r
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch.distributed as dist
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
mode = sys.argv[1]                                        # 'ab' | 'dead_peer' | 'skew'
dev = torch.device('cuda:0')                               # the ranks SHARE GPU 0 (RCCL refuses that; IPC inboxes do not)
torch.cuda.set_device(dev)
dist.init_process_group('gloo')
import hrfuser_oracle as O
from helpers import build_pair, use_backend
from hrfuser_amd.trainer import Trainer
use_backend('hip')


def edit(cfg):                                             # a short HRFuser-T: every kind of stage once
    for k in ('stage2', 'stage3', 'stage4', 'LidarStageB', 'LidarStageC'):
        cfg['extra'][k]['num_modules'] = 1


def grads(p2p):
    os.environ['HRF_SYNC_P2P'] = '1' if p2p else '0'
    net, _, cfg = build_pair('t_nus', dev, edit=edit)
    net.train()
    B = 1 + rank                                           # unequal per-rank batches: the global counts ride in the exchange
    x, mods = O.seeded_inputs(3, 64, 96, [3, 3], seed=1)
    sl = slice(0, 1) if rank == 0 else slice(1, 3)
    xd, md = x[sl].to(dev), [m[sl].to(dev) for m in mods]
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn((3, 64 // 4 >> i, 96 // 4 >> i, c), generator=g)[sl].to(dev) for i, c in enumerate(cfg['extra']['stage4']['num_channels'])]
    tr = Trainer(net, lr=0.0, weight_decay=0.0, group=dist.group.WORLD, world_size=world)
    for _ in range(2):
        tr.step(xd, md, cots, grads_only=True)
    tr.check()
    torch.cuda.synchronize()
    return net._engine().flat_g.clone(), tr, net


if mode == 'ab':
    g_coll, tr0, _ = grads(False)
    assert tr0.p2p_exchanges_per_step == 0 and tr0.collectives_per_step > 20
    g_p2p, tr1, _ = grads(True)
    assert tr1.p2p_exchanges_per_step > 40, tr1.p2p_exchanges_per_step
    assert 'peer-to-peer' in tr1.sync_schedule
    err = float((g_p2p.double() - g_coll.double()).norm() / g_coll.double().norm())
    print(json.dumps({'rank': rank, 'rel_l2': err, 'p2p_exchanges': tr1.p2p_exchanges_per_step, 'collectives_left': tr1.collectives_per_step}))
    assert err < 1e-6, err
    dist.barrier()
    print('P2P_AB_OK', rank)
elif mode == 'skew':
    # two-process race check: the lane timing of ONE rank is perturbed (idle launches in front of every 18 / 36-channel block,
    # one weight-gradient lane instead of four) - the exchange launches of that rank reach their spin-waits late and in another
    # interleaving across lanes.  Both ranks must still finish (no dead-lock: the enqueue order per lane is the program's, not
    # the clock's) with the unperturbed gradients.
    g_ref, tr0, _ = grads(True)
    if rank == 1:
        os.environ.update(HRF_DEBUG_PAD='18:40,36:40', HRF_WGRAD_LANES='1')
    g_pad, tr1, _ = grads(True)
    assert tr1.p2p_exchanges_per_step > 40
    err = float((g_pad.double() - g_ref.double()).norm() / g_ref.double().norm())
    print(json.dumps({'rank': rank, 'rel_l2_vs_unperturbed': err}))
    assert err < 1e-3, err
    dist.barrier()
    print('P2P_SKEW_OK', rank)
else:
    os.environ['HRF_P2P_TIMEOUT_S'] = '2'
    g, tr, net = grads(True)                                # builds the inboxes with both ranks present
    dist.barrier()
    if rank == 1:
        print('P2P_PEER_LEAVES')
        sys.stdout.flush()
        os._exit(0)                                        # this rank never runs the next step
    try:
        x, mods = O.seeded_inputs(1, 64, 96, [3, 3], seed=1)
        net._execute((x.to(dev),) + tuple(m.to(dev) for m in mods), True)
        tr.check()
        print('P2P_NO_ERROR')
    except Exception as e:
        print('P2P_TIMEOUT_RAISED', type(e).__name__, str(e)[:160])
    sys.stdout.flush()
    os._exit(0)
sys.stdout.flush()
os._exit(0)
'''


def _run_two(mode, port, timeout=900):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'HRF_FORCE_COLLECTIVES', 'HRF_SYNC_P2P'):
        env.pop(k, None)
    env.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, '-c', WORKER % ROOT, mode], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, e = p.communicate()
            o += '\n[killed after timeout]'
        outs.append((p.returncode, o, e))
    return outs


@pytest.mark.gpu
def test_p2p_two_processes_share_one_gpu():
    """Two ranks = two processes on ONE GPU, inboxes mapped through IPC handles: the SyncBN step through the peer-to-peer
    exchange equals the same step through the (gloo) collective schedule, 1 + 2 images per rank."""
    outs = _run_two('ab', 29671)
    for rc, o, e in outs:
        sys.stdout.write(o[-1500:])
        assert 'P2P_AB_OK' in o, (rc, o[-1500:], e[-3000:])


@pytest.mark.gpu
def test_p2p_two_processes_one_rank_perturbed_gpu():
    """lane-timing perturbation on ONE rank only - the other must still finish, gradients equal."""
    outs = _run_two('skew', 29679, timeout=600)
    for rc, o, e in outs:
        sys.stdout.write(o[-600:])
        assert 'P2P_SKEW_OK' in o, (rc, o[-1500:], e[-3000:])


@pytest.mark.gpu
def test_p2p_dead_peer_fails_loudly():
    """The peer leaves between two steps: the surviving rank's next step must end in an HRFuserHipError that names the missing
    source (time-out 2 s) - never a hang, never a silent wrong result."""
    outs = _run_two('dead_peer', 29673, timeout=300)
    rc0, o0, e0 = outs[0]
    sys.stdout.write(o0[-800:])
    assert 'P2P_TIMEOUT_RAISED' in o0 and 'HRFuserHipError' in o0 and 'rank 1 never delivered' in o0, (rc0, o0[-1500:], e0[-3000:])
    assert 'P2P_NO_ERROR' not in o0


FORCED = r'''
import copy, os, sys, torch
ROOT = %r
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], RANK='0', WORLD_SIZE='1', HRF_FORCE_COLLECTIVES='1')
import torch.distributed as dist
backend = sys.argv[1]
dist.init_process_group('gloo')
import hrfuser_oracle as O
from helpers import build_pair, use_backend
from hrfuser_amd.trainer import Trainer
dev = use_backend(backend)


def edit(cfg):
    for k in ('stage2', 'stage3', 'stage4', 'LidarStageB', 'LidarStageC'):
        cfg['extra'][k]['num_modules'] = 1


def grads(p2p):
    os.environ['HRF_SYNC_P2P'] = '1' if p2p else '0'
    net, _, cfg = build_pair('t_nus', dev, edit=edit)
    net.train()
    H, W = (32, 32) if backend == 'emul' else (64, 96)
    x, mods = O.seeded_inputs(2, H, W, [3, 3], seed=1)
    g = torch.Generator().manual_seed(5)
    cots = [torch.randn((2, H // 4 >> i, W // 4 >> i, c), generator=g).to(dev) for i, c in enumerate(cfg['extra']['stage4']['num_channels'])]
    tr = Trainer(net, lr=0.0, weight_decay=0.0, group=dist.group.WORLD, world_size=1)
    tr.step(x.to(dev), [m.to(dev) for m in mods], cots, grads_only=True)
    if backend == 'hip':
        tr.check()
    return net._engine().flat_g.clone(), tr
g0, t0 = grads(False)
g1, t1 = grads(True)
assert t0.p2p_exchanges_per_step == 0 and t1.p2p_exchanges_per_step > 40, (t0.p2p_exchanges_per_step, t1.p2p_exchanges_per_step)
err = float((g1.double() - g0.double()).norm() / g0.double().norm())
print('rel_l2', err, 'exchanges', t1.p2p_exchanges_per_step, 'collectives', t0.collectives_per_step, '->', t1.collectives_per_step)
assert err < 1e-6, err
print('P2P_FORCED_OK')
sys.stdout.flush()
os._exit(0)
'''


def _forced(backend, port):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', FORCED % ROOT, backend, str(port)], capture_output=True, text=True, timeout=1500, env=env)
    sys.stdout.write(r.stdout[-1000:])
    assert 'P2P_FORCED_OK' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_p2p_forced_one_rank_emul():
    """The whole (shortened) backbone on the emulator with every SyncBN exchange going through the inbox of a one-rank group:
    same gradients as the collective schedule."""
    _forced('emul', 29675)


@pytest.mark.gpu
def test_p2p_forced_one_rank_gpu():
    _forced('hip', 29677)
