"""Data-parallel logic on CPU: 2 ranks (gloo) each run the product tape/SyncBN/grad-exchange code
on the kernel emulator with a part of a batch - UNEQUAL parts: one image on rank 0, two on rank 1, so the BatchNorm
sample counts must travel with the moments (torch.nn.SyncBatchNorm all-gathers them) - and the result must equal ONE
process on the whole batch (= the oracle on the concatenated batch): SyncBN statistics, summed/averaged gradients, AdamW."""
import copy
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HW = (64, 64)            # 2 images of 64x64: the stride-32 BatchNorms see 8 samples (32x64 left 4: fp32 noise x 1e4)


def _short(cfg):
    """HRFuser-T with ONE module per stage (every layer type, branch count and exchange still present): the subject here
    is the data-parallel logic, and the CPU kernel emulator is slow."""
    for k in ('stage3', 'stage4', 'LidarStageC'):
        cfg['extra'][k]['num_modules'] = 1


def _worker(rank, world, port, ret, schedule):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HRF_EMUL_THREADS='2')
    os.environ.pop('HRF_SYNC_P2P', None)                 # auto: peer-to-peer after the handshake, or the agreed fallback
    os.environ['HRF_P2P_TIMEOUT_S'] = '120'
    # the gradient exchange inside the weight-gradient phase: four bucket groups (auto would use one for 16 MB of gradients),
    # leaves issued group by group, a group's all-reduce behind its own leaves and the folds - same gradients as one exchange
    os.environ['HRF_GRAD_OVERLAP'] = '4' if schedule == 'p2p' else 'auto'
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import hrfuser_oracle as O
    from helpers import build_pair, enable_relu_probe, relu_masks, use_backend
    from hrfuser_amd.trainer import Trainer
    dev = use_backend('emul')
    if schedule == 'fallback':
        # the auto mode's fallback (more than 8 ranks, a peer on another node, no IPC): mapping a peer's inbox fails on THIS
        # rank only - every rank must still end up on the collective schedule, with a warning, and Trainer.check() must work
        from hrfuser_amd import _lib
        L = _lib.lib()
        real = L._fns['hrf_p2p_open']

        def refuse(*a):
            if rank == 1:
                raise _lib.HRFuserHipError('hrf_p2p_open: refused by the test')
            return real(*a)
        L._fns['hrf_p2p_open'] = refuse
    net, orc, cfg = build_pair('t_nus', dev, edit=_short)           # SyncBN config
    net.train()
    enable_relu_probe(net)
    B, (H, W) = 3, (HW if schedule == 'p2p' else (32, 32))
    x, mods = O.seeded_inputs(B, H, W, [3, 3], seed=1)
    sl = slice(0, 1) if rank == 0 else slice(1, 3)      # rank 0: one image, rank 1: two
    g = torch.Generator().manual_seed(5)
    shapes = [(B, H // 4 >> i, W // 4 >> i, c) for i, c in enumerate((18, 36, 72, 144))]
    cots = [torch.randn(s, generator=g) for s in shapes]
    tr = Trainer(net, lr=1e-3, group=dist.group.WORLD, world_size=world)
    assert len(tr.buckets(4_000_000)) == 4 and tr.buckets(10)[0] == (0, 10)
    assert tr.overlap_rounds(4_000_000) == (4 if schedule == 'p2p' else 1) and Trainer(net, group=None).overlap_rounds(61_000_000) in (4, 1)
    import warnings
    with warnings.catch_warnings(record=True) as wlog:
        warnings.simplefilter('always')
        outs = tr.step(x[sl], [m[sl] for m in mods], [c[sl] * world for c in cots])
    tr.check()                                             # (crashed on the fallback's (group, world, None, mode) entry)
    fell_back = any('peer-to-peer exchange is not available' in str(w.message) for w in wlog)
    if schedule == 'p2p':
        # two emulator PROCESSES, inboxes in POSIX shared memory: the product's exchange protocol itself (push, flags, spin,
        # rank-order sums, generation parity) across address spaces - not the fallback
        assert not fell_back and tr.p2p_exchanges_per_step > 40 and 'peer-to-peer' in tr.sync_schedule, (tr.p2p_exchanges_per_step, tr.sync_schedule)
        assert tr.grad_collectives_per_step == len(tr.buckets(net._engine().flat_g.numel()))
    else:
        assert fell_back and tr.p2p_exchanges_per_step == 0 and 'peer-to-peer' not in tr.sync_schedule
        assert 20 < tr.collectives_per_step <= 150, tr.collectives_per_step   # packed exchanges of the lock-step schedule
        assert bool(torch.isfinite(net._engine().flat_g).all())
        ret[f'fallback{rank}'] = float(net._engine().flat_g.double().norm())
        dist.barrier()
        dist.destroy_process_group()
        return
    eng = net._engine()
    N = lambda t: t.detach().float().cpu().numpy().copy()
    res = dict(out=[N(o.t) for o in outs], grad=N(eng.flat_g), param=N(eng.flat_p),
               rm=N(net.bn1.running_mean), rv=N(net.bn1.running_var), ncoll=tr.collectives_per_step)
    ret[f'masks{rank}'] = [m.cpu().numpy().copy() for m in relu_masks(net)]
    # ---- the neck's arena goes through the same exchange: rank-sum of local gradients
    import hrfpn_oracle as NO
    from hrfuser_amd import HRFPN
    norc = NO.HRFPNOracle(in_channels=[18, 36], out_channels=32, num_outs=3)
    O.seeded_fill_(norc, 11)
    neck = HRFPN(in_channels=[18, 36], out_channels=32, num_outs=3)
    neck.load_state_dict(norc.state_dict())
    neck.train()
    gm = torch.Generator().manual_seed(9)
    maps = [torch.randn(B, 18, 8, 16, generator=gm), torch.randn(B, 36, 4, 8, generator=gm)]
    assert net.bn1.running_mean is not None
    ncots = [torch.randn(B, 8 >> i, 16 >> i, 32, generator=gm) for i in range(3)]
    ntr = Trainer(neck, lr=0.0, weight_decay=0.0, group=dist.group.WORLD, world_size=world)
    ntr.step(maps[0][sl], [maps[1][sl]], [c[sl] for c in ncots])
    ntr.check()
    res['neck_grad'] = N(neck._engine().flat_g)
    if rank == 0:
        ys = norc(maps)
        sum((y.permute(0, 2, 3, 1) * c).sum() for y, c in zip(ys, ncots)).backward()
        ret['neck_ref'] = N(torch.cat([p.grad.reshape(-1) for p in norc.parameters()]))
    if rank == 0:
        ret['names'] = [(n, tuple(p.shape)) for n, p in net.named_parameters()]
        ret['spans'] = list(eng._spans)
        ret['res'] = res
        ret['wd_mask_sum'] = float(tr.wd_mask.sum())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('schedule', ['p2p', 'fallback'])
def test_two_rank_syncbn_and_grad_exchange_equal_single_process(schedule):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + (0 if schedule == 'p2p' else 1)
    mp.spawn(_worker, args=(2, port, ret, schedule), nprocs=2, join=True)
    if schedule == 'fallback':
        # the agreed fallback of the auto mode (one rank cannot map its peer's inbox): both ranks on the collective schedule,
        # identical all-reduced gradient arenas, Trainer.check() usable.  (The schedule's parity against the oracle is the
        # subject of tests/test_syncbn_gpu.py and of the peer-to-peer == collective comparisons of tests/test_p2p_exchange.py.)
        assert ret['fallback0'] > 0 and ret['fallback0'] == ret['fallback1']
        return
    res = ret['res']
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hrfuser_oracle as O
    from helpers import PinnedReLU, build_pair
    T = torch.as_tensor
    rel = lambda a, b: float((T(a).double() - T(b).double()).abs().max() / (T(b).double().abs().max() + 1e-30))
    # single-process reference on the WHOLE batch: the fp64 (and fp32) oracle with the ReLU decisions of the two ranks
    # pinned (each rank saw one image: their masks concatenate along the batch), so the gradient gate is the tight one
    _, orc, _ = build_pair('t_nus', torch.device('cpu'), edit=_short)
    B, (H, W) = 3, HW
    x, mods = O.seeded_inputs(B, H, W, [3, 3], seed=1)
    g = torch.Generator().manual_seed(5)
    shapes = [(B, H // 4 >> i, W // 4 >> i, c) for i, c in enumerate((18, 36, 72, 144))]
    cots = [torch.randn(s, generator=g) for s in shapes]
    masks = [torch.cat([T(a), T(b)], 0) for a, b in zip(ret['masks0'], ret['masks1'])]
    refs = []
    for dt in (torch.float64, torch.float32):
        o = copy.deepcopy(orc).to(dt).train()
        with PinnedReLU(masks):
            ys = o(x.to(dt), [m.to(dt) for m in mods])
        sum((y.permute(0, 2, 3, 1) * c.to(dt)).sum() for y, c in zip(ys, cots)).backward()
        refs.append((o, ys))
    o64, ys = refs[0]
    o32 = refs[1][0]
    for o, r in zip(res['out'], ys):
        assert rel(o, r.permute(0, 2, 3, 1)[0:1].detach()) < 1e-3      # SyncBN: rank-0 slice of the global-batch forward
    assert rel(res['rm'], o64.bn1.running_mean) < 1e-4 and rel(res['rv'], o64.bn1.running_var) < 1e-4
    # flat gradient arena after the all-reduce = SUM over ranks of local grads (cotangents were pre-scaled by world) =
    # world x gradient of the global-batch loss; AdamW divides by world again.  Per tensor: rel-L2 <= max(1e-3, 3 e_ref).
    ref64 = {k: v.grad for k, v in o64.named_parameters() if v.grad is not None}
    ref32 = {k: v.grad for k, v in o32.named_parameters() if v.grad is not None}
    flat = T(res['grad'])
    nmax = max(float(v.norm()) for v in ref64.values())
    gmax = max(float(v.abs().max()) for v in ref64.values())
    worst = (0.0, 0.0, '')
    for (name, shape), (off, n) in zip(ret['names'], ret['spans']):
        gk = flat[off:off + n].view(shape).double() / 2.0            # world * mean-convention
        if name not in ref64:
            assert float(gk.abs().max()) == 0.0                        # unused params stay zero
            continue
        q = ref64[name]
        if float(q.norm()) < 1e-9 * nmax:                              # analytically zero
            assert float(gk.abs().max()) <= 1e-4 * gmax, name
            continue
        e = float((gk - q).norm() / q.norm())
        e_ref = float((ref32[name].double() - q).norm() / q.norm())
        assert e <= max(1e-3, 3 * e_ref), (name, e, e_ref)
        worst = max(worst, (e / max(e_ref, 1e-30) if e > 1e-3 else 0.0, e, name))
    print(f'[2-rank gloo] largest e / e_ref among tensors above 1e-3: {worst[0]:.2f} (e = {worst[1]:.2e}, {worst[2]}); '
          f'{res["ncoll"]} collectives in the step')
    assert ret['wd_mask_sum'] > 0
    # batching: 330 BatchNorms x 2 directions would be 660 exchanges one by one; the lock-step schedule packs the
    # independent ones (sensor streams, HRModule branches, exchange chains)
    # peer-to-peer schedule: only the gradient buckets are collectives
    assert res['ncoll'] <= 8, res['ncoll']
    # neck: all-reduced arena = gradient of the whole-batch loss (no normalisation layers in HRFPN)
    assert rel(res['neck_grad'], ret['neck_ref']) < 1e-4
