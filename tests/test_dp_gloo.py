"""Data-parallel logic on CPU: 2 ranks (gloo) each run the product tape/SyncBN/grad-exchange code
on the kernel emulator with half of a batch; the result must equal ONE process on the whole batch
(= the oracle on the concatenated batch): SyncBN statistics, summed/averaged gradients, AdamW."""
import copy
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HRF_EMUL_THREADS='2')
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import hrfuser_oracle as O
    from helpers import build_pair, use_backend
    from hrfuser_amd.trainer import Trainer
    dev = use_backend('emul')
    net, orc, cfg = build_pair('t_nus', dev)           # SyncBN config
    net.train()
    B, H, W = 2, 32, 64
    x, mods = O.seeded_inputs(B, H, W, [3, 3], seed=1)
    sl = slice(rank, rank + 1)                          # one image per rank
    g = torch.Generator().manual_seed(5)
    shapes = [(B, H // 4 >> i, W // 4 >> i, c) for i, c in enumerate((18, 36, 72, 144))]
    cots = [torch.randn(s, generator=g) for s in shapes]
    tr = Trainer(net, lr=1e-3, group=dist.group.WORLD, world_size=world)
    assert len(tr.buckets(4_000_000)) == 4 and tr.buckets(10)[0] == (0, 10)
    outs = tr.step(x[sl], [m[sl] for m in mods], [c[sl] * world for c in cots])
    eng = net._engine()
    N = lambda t: t.detach().float().cpu().numpy().copy()
    res = dict(out=[N(o.t) for o in outs], grad=N(eng.flat_g), param=N(eng.flat_p),
               rm=N(net.bn1.running_mean), rv=N(net.bn1.running_var))
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
    ncots = [torch.randn(B, 8 >> i, 16 >> i, 32, generator=gm) for i in range(3)]
    ntr = Trainer(neck, lr=0.0, weight_decay=0.0, group=dist.group.WORLD, world_size=world)
    ntr.step(maps[0][sl], [maps[1][sl]], [c[sl] for c in ncots])
    res['neck_grad'] = N(neck._engine().flat_g)
    if rank == 0:
        ys = norc(maps)
        sum((y.permute(0, 2, 3, 1) * c).sum() for y, c in zip(ys, ncots)).backward()
        ret['neck_ref'] = N(torch.cat([p.grad.reshape(-1) for p in norc.parameters()]))
    if rank == 0:
        # single-process reference on the WHOLE batch: oracle fp64 + torch AdamW with the same masks
        o64 = copy.deepcopy(orc).double().train()
        ys = o64(x.double(), [m.double() for m in mods])
        sum((y.permute(0, 2, 3, 1) * c.double()).sum() for y, c in zip(ys, cots)).backward()
        ret['ref_out'] = [N(y.permute(0, 2, 3, 1)[sl]) for y in ys]
        ret['ref_grad'] = {n: N(p.grad) for n, p in o64.named_parameters() if p.grad is not None}
        ret['ref_rm'] = N(o64.bn1.running_mean)
        ret['ref_rv'] = N(o64.bn1.running_var)
        ret['names'] = [(n, tuple(p.shape)) for n, p in net.named_parameters()]
        ret['spans'] = list(eng._spans)
        ret['res'] = res
        ret['wd_mask_sum'] = float(tr.wd_mask.sum())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_syncbn_and_grad_exchange_equal_single_process():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    res = ret['res']
    T = torch.as_tensor
    rel = lambda a, b: float((T(a).double() - T(b).double()).abs().max() / (T(b).double().abs().max() + 1e-30))
    for o, r in zip(res['out'], ret['ref_out']):
        assert rel(o, r) < 1e-3                      # SyncBN: rank-0 slice of the global-batch forward
    assert rel(res['rm'], ret['ref_rm']) < 1e-4 and rel(res['rv'], ret['ref_rv']) < 1e-4
    # flat gradient arena after the all-reduce = SUM over ranks of local grads (cotangents were
    # pre-scaled by world) = gradient of the global-batch loss; AdamW divides by world again.
    ref_grad = {k: T(v) for k, v in ret['ref_grad'].items()}
    flat = T(res['grad'])
    gscale = max(float(v.abs().max()) for v in ref_grad.values())
    worst = 0.0
    for (name, shape), (off, n) in zip(ret['names'], ret['spans']):
        if name not in ref_grad:
            assert float(flat[off:off + n].abs().max()) == 0.0       # unused params stay zero
            continue
        g = flat[off:off + n].view(shape) / 2.0                      # world * mean-convention
        ref = ref_grad[name]
        err = float((g - ref).norm()) / (float(ref.norm()) + 2e-3 * gscale * ref.numel() ** 0.5)
        worst = max(worst, err)
    assert worst < 2e-2, worst
    assert ret['wd_mask_sum'] > 0
    # neck: all-reduced arena = gradient of the whole-batch loss (no normalisation layers in HRFPN)
    assert rel(res['neck_grad'], ret['neck_ref']) < 1e-4
