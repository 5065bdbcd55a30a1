"""Data-parallel training step for the HIP backbone (one process per GPU, RCCL over xGMI).

Mirrors what the reference gets from mmdet/apis/train.py:113-121 (MMDistributedDataParallel,
find_unused_parameters=True), nn.SyncBatchNorm and the AdamW `paramwise_cfg` of
configs/hrfuser/*_fusion.py:37-48 - restricted to the backbone hot path:

  * forward + backward run on the explicit tape (no torch.autograd graph);
  * every parameter gradient already lives in ONE flat fp32 arena, so the gradient exchange is a
    handful of large RCCL all-reduces on slices of that arena (no per-parameter bucketing copies,
    unused parameters such as transition1.0.1.* are simply zeros in the arena);
  * SyncBN = all-reduce of the per-channel fp64 (sum, sumsq) slots between the producing conv and
    the BN-finalize kernel (runtime.bn_forward / bn_backward_coef);
  * the optimizer is one fused AdamW launch over the flat arena with a per-element decay mask
    (decay_mult=0 for `relative_position_bias_table` and `norm` keys) and a device-side step count;
  * the whole step is captured into a hipGraph when possible (launch-bound regime at batch 2).
"""
import os

import torch

from . import _lib
from . import runtime as R

NO_DECAY_KEYS = ('absolute_pos_embed', 'relative_position_bias_table', 'norm')


class Trainer:
    def __init__(self, net, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, group=None,
                 world_size=1, n_buckets=4):
        self.net = net
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.group, self.world = group, world_size
        self.n_buckets = n_buckets
        self.force = R.force_collectives() and group is not None
        if world_size > 1 or self.force:
            net.set_sync_group(group, world_size)
        self._ready = False
        self.graph = None
        self.collectives_per_step = 0

    # -------------------------------------------------------------------------------------------
    def _setup(self, device):
        eng = self.net._engine()
        eng.ready(device)
        n = eng.flat_p.numel()
        self.m = torch.zeros(n, device=device)
        self.v = torch.zeros(n, device=device)
        self.state = torch.zeros(4, device=device)
        mask = torch.ones(n, device=device)
        unused = set(getattr(self.net, 'unused_parameter_names', lambda: ())())
        for (name, p), (off, cnt) in zip(self.net.named_parameters(), eng._spans):
            # mmcv DefaultOptimizerConstructor custom_keys: substring match on the parameter name
            if any(k in name for k in NO_DECAY_KEYS):
                mask[off:off + cnt] = 0.0
            if name in unused or not p.requires_grad:
                mask[off:off + cnt] = -1.0            # no gradient ever: torch.optim leaves such parameters untouched
        self.wd_mask = mask
        self.state[3] = self.lr
        self._ready = True

    def set_lr(self, lr):
        """Learning rate of the NEXT steps, also for an already captured hipGraph (the kernel reads it from the device)."""
        self.lr = float(lr)
        if self._ready:
            self.state[3:4].fill_(self.lr)

    def optimizer_step(self):
        """Fused AdamW over the flat arenas (device-side step count and learning rate)."""
        net = self.net
        eng = net._engine()
        L = net._lib_handle()
        s = _lib.stream_ptr()
        L.hrf_adamw_tick(self.state, self.betas[0], self.betas[1], s)
        L.hrf_adamw(eng.flat_p, eng.flat_g, self.m, self.v, self.wd_mask, eng.flat_p.numel(), -1.0,
                    self.betas[0], self.betas[1], self.eps, self.wd, self.state, 1.0 / self.world, s)

    def buckets(self, n):
        """Contiguous slices of the flat gradient arena for the RCCL all-reduce (>= 1 MiB each)."""
        nb = max(1, min(self.n_buckets, n // (256 * 1024)))
        step = (n + nb - 1) // nb
        return [(i, min(n, i + step)) for i in range(0, n, step)]

    def overlap_rounds(self, n):
        """Groups of buckets whose all-reduce overlaps the weight-gradient leaves of the next group.  HRF_GRAD_OVERLAP: a number,
        or auto = one group per bucket from 64 MB of gradients (HRFuser-B: 243 MB, ~5 ms on xGMI beside a > 10 ms
        weight-gradient phase), a single group below (HRFuser-T: 16 MB = 0.3 ms - every extra fork / join of the leaf lanes
        costs more than it hides)."""
        v = os.environ.get('HRF_GRAD_OVERLAP', 'auto').strip().lower()
        if v not in ('', 'auto'):
            return max(1, int(v))
        return self.n_buckets if 4 * n >= (64 << 20) else 1

    def _step_impl(self, x, mods, cots, grads_only=False):
        net = self.net
        eng = net._engine()
        L = net._lib_handle()
        R.gpu_zero_(eng.flat_g)
        ctx, outs, _ = net._execute((x,) + tuple(mods), True)
        for o, c in zip(outs, cots):
            o.grad = R.gpu_clone(c)         # synthetic loss  L = sum_i <out_i, cot_i>   
        if self.world > 1 or self.force:
            import torch.distributed as dist
            # the gradient exchange is part of the backward pass: the weight-gradient leaves are issued bucket group by bucket
            # group and a group's all-reduce runs on a communication lane beside the next group's leaves (runtime.Ctx.
            # _exchange_rounds; the reference: DDP's bucketed overlap, mmdet/apis/train.py:113-121)
            ctx.exchange = (self.buckets(eng.flat_g.numel()),
                            lambda a, b: dist.all_reduce(eng.flat_g[a:b], group=self.group), self.overlap_rounds(eng.flat_g.numel()))
        ctx.run_backward()
        ncoll = ctx.n_collectives + ctx.n_grad_collectives
        self.p2p_exchanges_per_step = ctx.n_p2p
        self.grad_collectives_per_step = ctx.n_grad_collectives
        self.collectives_per_step = ncoll
        self.exchange_hist = dict(ctx.xhist)
        self.sync_schedule = ctx.schedule_desc()
        if not grads_only:
            self.optimizer_step()
        st = net.__dict__.get('_stage_stamps')
        if st is not None:
            st.take(ctx, 'step', 'step_end')
        return outs

    def step(self, x, mods, cots, grads_only=False):
        """One eager training step (forward, backward, gradient exchange, AdamW; grads_only: no optimizer step - the
        all-reduced gradient arena of the step stays in net._engine().flat_g)."""
        if not self._ready:
            self._setup(x.device)
        outs = self._step_impl(x, mods, cots, grads_only)
        self._poll_exchange()
        return outs

    def _exchange(self):
        """The peer-to-peer SyncBN exchange context of the engine, or None (plain BatchNorm, the collective schedules, or the
        agreed fallback of the auto mode: Engine.p2p_context stores (group, world, None, mode) then)."""
        px = self.net._engine().__dict__.get('_p2p')
        return px[2] if (px is not None and px[2] is not None) else None

    def _poll_exchange(self):
        """Step boundary: a lost SyncBN exchange (a peer that did not arrive within HRF_P2P_TIMEOUT_S) raises here, one step
        late at the latest, without a host synchronisation (P2PExchange.poll)."""
        px = self._exchange()
        if px is not None:
            px.poll()

    # -------------------------------------------------------------------------------------------
    def capture(self, x, mods, cots, warmup=2):
        """Capture the full step into a hipGraph (static input buffers x/mods/cots)."""
        if not self._ready:
            self._setup(x.device)
        side = torch.cuda.Stream(priority=int(os.environ.get('HRF_LANE_PRIORITY', '0')))
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._step_impl(x, mods, cots)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        old = self.__dict__.pop('_graph_px', None)
        if old is not None:
            old.unpin()                  # the graph this capture replaces goes away with its pin
        px = self._exchange()
        if px is not None:
            # the per-lane exchange-order digest of the first eager step is compared across ranks HERE, outside the capture: with
            # warmup = 0 after one eager step the second forward would be the captured one and skip it
            px.verify_order()
        if self.world > 1 or self.force:
            # The RCCL watchdog thread is still polling the end events of the warm-up collectives (one
            # sweep per 100 ms); an event query that lands on the communicator stream after it joined the
            # capture invalidates the capture ("capturing stream has unjoined work": 3 of 6 runs).  Give
            # the watchdog time to retire the eager work first (6 of 6 runs pass with the pause).
            import time
            time.sleep(float(os.environ.get('HRF_CAPTURE_SETTLE', '1.0')))
        g = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread polls events of earlier (eager) collectives while we
        # capture; in the default "global" mode such a call from another thread invalidates the capture
        # ("capturing stream has unjoined work", seen in ~3 of 4 runs with collectives in the graph)
        with R.gc_paused(), torch.cuda.graph(g, capture_error_mode='thread_local'):
            self._graph_outs = self._step_impl(x, mods, cots)
        self.graph = g
        px = self._exchange()
        if px is not None:
            px.pin()                     # the graph's exchange launches carry the context's inbox / flag / counter pointers
            self._graph_px = px          # (Engine.p2p_context never closes a pinned context)
        return g

    def replay(self):
        self.graph.replay()
        self._poll_exchange()

    def check(self):
        """Raise if a peer-to-peer SyncBN exchange timed out (synchronises: call where the caller synchronises anyway - the
        end of a timed loop, an evaluation interval)."""
        px = self._exchange()
        if px is not None:
            if px.err.device.type == 'cuda':
                torch.cuda.synchronize()
            px.check()


def make_cotangents(net, x, mods, seed=5):
    """Fixed random output cotangents (NHWC) for the synthetic loss, shaped by a dry eval forward."""
    was = net.training
    net.eval()
    with torch.no_grad():
        ys = net(x, list(mods))
    net.train(was)
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(tuple(y.permute(0, 2, 3, 1).shape), generator=g).to(x.device) / y.numel() for y in ys]
