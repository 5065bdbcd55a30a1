"""MI355X-native HRFuser backbone: host-side mirror of the reference's module tree.

Same class names, constructor keywords, error behaviour and state-dict keys as
  /root/reference/mmdet/models/backbones/hrfuser_hrformer_based.py  (HRFuserHRFormerBased :330-628,
      HRFuserFusionBlock :250-326, MultiWindowCrossAttention :153-248, WindowMCA :21-151)
  /root/reference/mmdet/models/backbones/hrformer.py  (WindowMSA :18-131, LocalWindowSelfAttention
      :134-236, CrossFFN :239-295, HRFormerBlock :298-386, HRFomerModule :389-561)
  /root/reference/mmdet/models/backbones/hrnet.py     (stem/_make_layer/_make_transition_layer)
  /root/reference/mmdet/models/backbones/resnet.py    (Bottleneck :100-302)
but the modules only OWN parameters: all arithmetic is issued through `runtime.py` onto the
hand-written gfx950 kernels behind `include/hrfuser_hip.h`.  There is no eager/CPU fallback.
"""
import os
import warnings

import torch
import torch.nn as nn

from . import _lib
from . import runtime as R
from .registry import BACKBONES

WIN = 7


# ----------------------------------------------------------------------------- norm factories
def build_bn(norm_cfg, c):
    """mmcv.build_norm_layer subset for norm_cfg type BN / SyncBN (SyncBN = BN + RCCL stat exchange, decided at run time by
    the engine's process group, so the module class is the same) and GN (nn.GroupNorm(num_groups, c): per-sample statistics,
    no exchange; runtime.gn_forward)."""
    cfg = dict(norm_cfg or dict(type='BN'))
    kind = cfg.pop('type')
    if kind not in ('BN', 'SyncBN', 'GN'):
        raise KeyError(f'unsupported norm_cfg type {kind}')
    requires_grad = cfg.pop('requires_grad', True)
    if kind == 'GN':
        if 'num_groups' not in cfg:
            raise AssertionError('norm_cfg of type GN needs num_groups')          # mmcv: assert 'num_groups' in cfg_
        bn = nn.GroupNorm(cfg['num_groups'], c, eps=cfg.get('eps', 1e-5))
    else:
        bn = nn.BatchNorm2d(c, eps=cfg.get('eps', 1e-5), momentum=cfg.get('momentum', 0.1))
    for p in bn.parameters():
        p.requires_grad_(requires_grad)
    return bn


def norm_name(norm_cfg, postfix):
    """The attribute name mmcv.build_norm_layer gives a postfixed norm layer (abbreviation + postfix): state-dict keys of
    the stems and the residual blocks follow the norm type (hrnet.py:338-360, resnet.py:34-49,161-206)."""
    kind = dict(norm_cfg or dict(type='BN')).get('type', 'BN')
    return ('gn' if kind == 'GN' else 'bn') + str(postfix)


def build_ln(cfg, c):
    cfg = dict(cfg or dict(type='LN', eps=1e-6))
    if cfg.pop('type') != 'LN':
        raise KeyError('transformer_norm_cfg must be LN')
    return nn.LayerNorm(c, eps=cfg.get('eps', 1e-5))


def _conv_bn_seq(cin, cout, k, stride, norm_cfg, relu, groups=1):
    mods = [nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False), build_bn(norm_cfg, cout)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return nn.Sequential(*mods)


def _rel_index():
    ys, xs = torch.meshgrid(torch.arange(WIN), torch.arange(WIN), indexing='ij')
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    return (ys[:, None] - ys[None, :] + WIN - 1) * (2 * WIN - 1) + (xs[:, None] - xs[None, :] + WIN - 1)


# ----------------------------------------------------------------------------- engine
class Engine:
    """Device-side state owned by a root module: flat parameter / gradient arenas, per-BatchNorm
    scratch slots, eval-mode BN affine cache, SyncBN group."""

    def __init__(self, root):
        self.root = root
        self.device = None
        self.flat_p = self.flat_g = None
        self.slots = {}
        self._bns = []
        self.keep = []                 # step-lifetime buffers of THIS engine (runtime.use_keep_list)
        self.fs_layers, self.fs_step, self.fs_used, self.fs_sig, self.fs_arena = {}, [], [], None, None

    def ready(self, device):
        params = [p for p in self.root.parameters()]
        if self.device != device or self.flat_p is None or (params and params[0].data_ptr() != self._p0) or \
                (self._bns and self._bns[0].running_mean is not None and self._bns[0].running_mean.data_ptr() != self._rs0):
            self._setup(device, params)
        # re-bind gradients dropped by zero_grad(set_to_none=True)
        dirty = False
        for p, (off, n) in zip(params, self._spans):
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off:off + n].view_as(p)
                dirty = True
        if dirty:
            self.flat_g.zero_()

    def _setup(self, device, params):
        total = sum(p.numel() for p in params)
        self.flat_p = torch.empty(total, device=device, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=device, dtype=torch.float32)
        self._spans = []
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.data.reshape(-1).to(device))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                self._spans.append((off, n))
                off += n
        self._p0 = params[0].data_ptr() if params else 0
        bns = [m for m in self.root.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
        self._bns = bns
        csum = sum(m.num_features for m in bns)
        # running statistics become views of ONE flat buffer (like the parameters), so that the eval-mode affine of every
        # BatchNorm is recomputed on the device by a handful of flat torch ops at the start of each forward that has
        # frozen BatchNorms - never cached across steps (the kernels update these buffers through raw pointers)
        self.rstat = torch.zeros(2 * max(1, csum), device=device, dtype=torch.float32)
        self.eval_f = torch.zeros(4 * max(1, csum), device=device, dtype=torch.float32)    # scale | shift | mean | invstd
        gidx, bidx, epsv = [], [], []
        offs0 = {id(p): o for p, (o, _) in zip(params, self._spans)}
        K = _lib.STAT_COPIES                       # replicated accumulators: see include/hrfuser_hip.h
        self.arena_d = torch.zeros(4 * csum * K, device=device, dtype=torch.float64)
        self.arena_f = torch.zeros(7 * csum, device=device, dtype=torch.float32)
        self.slots = {}
        od = of = oc = 0
        for bi, m in enumerate(bns):
            C = m.num_features
            d = self.arena_d
            f = self.arena_f
            with torch.no_grad():
                if m.running_mean is not None:
                    self.rstat[oc:oc + C].copy_(m.running_mean.to(device))
                    self.rstat[csum + oc:csum + oc + C].copy_(m.running_var.to(device))
                    m.running_mean = self.rstat[oc:oc + C]
                    m.running_var = self.rstat[csum + oc:csum + oc + C]
            ar = torch.arange(C, dtype=torch.long)
            gidx.append(ar + offs0[id(m.weight)] if m.weight is not None else torch.full((C,), -1, dtype=torch.long))
            bidx.append(ar + offs0[id(m.bias)] if m.bias is not None else torch.full((C,), -1, dtype=torch.long))
            epsv.append(torch.full((C,), float(m.eps)))
            e = self.eval_f
            self.slots[id(m)] = dict(
                stats=d[od:od + 2 * C * K], gstats=d[od + 2 * C * K:od + 4 * C * K],
                scale=f[of:of + C], shift=f[of + C:of + 2 * C], mean=f[of + 2 * C:of + 3 * C],
                invstd=f[of + 3 * C:of + 4 * C], cA=f[of + 4 * C:of + 5 * C], cB=f[of + 5 * C:of + 6 * C],
                cC=f[of + 6 * C:of + 7 * C],
                eval=(e[oc:oc + C], e[csum + oc:csum + oc + C], e[2 * csum + oc:2 * csum + oc + C],
                      e[3 * csum + oc:3 * csum + oc + C]))
            od += 4 * C * K
            of += 7 * C
            oc += C
        self._csum = csum
        self._rs0 = bns[0].running_mean.data_ptr() if (bns and bns[0].running_mean is not None) else 0
        if bns:
            self._gidx, self._bidx = torch.cat(gidx).to(device), torch.cat(bidx).to(device)
            self._epsv = torch.cat(epsv).to(device)
        # parameter gradients that MANY blocks add into (LayerNorm gamma/beta, depthwise weights/bias)
        # accumulate in K replicated fp32 copies and are folded into the arena once per backward
        offs = {id(p): o for p, (o, _) in zip(params, self._spans)}
        self._poffs = offs
        self.fs_layers, self.fs_sig = {}, None
        self.pslot, cols, ps = {}, [], 0
        for m in self.root.modules():
            hit = isinstance(m, nn.LayerNorm) or (isinstance(m, nn.Conv2d) and m.groups == m.in_channels and m.groups > 1)
            cand = (m.weight, m.bias) if hit else ()
            if hasattr(m, 'relative_position_bias_table'):        # window attention: dRPB + pad-key/value bias grads
                cand = (m.relative_position_bias_table,) + tuple(
                    getattr(m, n).bias for n in ('qkv', 'k_proj', 'v_proj') if hasattr(m, n))
            for q in cand:
                if q is None or id(q) not in offs or id(q) in self.pslot:
                    continue
                self.pslot[id(q)] = ps
                cols.append(torch.arange(offs[id(q)], offs[id(q)] + q.numel(), dtype=torch.int32))
                ps += q.numel()
        self.ps_n = ps
        self.ps_map = (torch.cat(cols) if cols else torch.zeros(0, dtype=torch.int32)).to(device)
        self.ps_scratch = torch.zeros(max(1, K * ps), device=device, dtype=torch.float32)
        self.ps_dirty = False
        nbt = [m.num_batches_tracked for m in bns if m.num_batches_tracked is not None]
        self.nbt_flat = torch.zeros(len(nbt), device=device, dtype=torch.long)
        with torch.no_grad():
            for i, m in enumerate(b for b in bns if b.num_batches_tracked is not None):
                self.nbt_flat[i] = m.num_batches_tracked.to(device)
                m.num_batches_tracked = self.nbt_flat[i]          # 0-dim view: one add_ per step updates all
        self.device = device
        self._build_packs()

    # ---- tap-major weight packs of the front-end 3x3 convolutions (csrc/conv3x_engine.hip): one persistent buffer per
    # convolution and direction, refreshed by ONE hrf_conv3x_pack launch at the start of every forward (the optimizer and
    # captured replays write the weights through raw pointers: nothing derived from them is cached across steps)
    def _build_packs(self):
        L = self.root._lib_handle()
        self.packs, jobs = {}, []
        if os.environ.get('HRF_CONV3X', '1') == '0' or not hasattr(L, 'hrf_conv3x_pack'):
            self._pack_jobs = None
            return
        for m in self.root.modules():
            if not (isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.groups == 1 and m.padding == (1, 1)):
                continue
            Cout, Cin = m.weight.shape[:2]
            ent = [None, None]
            for d in (0, 1):
                if L.hrf_conv3x_supported(Cin, Cout, 3, m.stride[0], d):
                    ent[d] = torch.zeros(L.hrf_conv3x_pack_size(Cout, Cin, d), device=self.device, dtype=torch.float32)
                    jobs.append((m.weight, ent[d], Cout, Cin, d))
            if ent[0] is not None or ent[1] is not None:
                self.packs[id(m.weight)] = tuple(ent)
        arr = (_lib.Conv3xPackJob * max(1, len(jobs)))()
        for i, (w, wp, Cout, Cin, d) in enumerate(jobs):
            arr[i] = _lib.Conv3xPackJob(w.data_ptr(), wp.data_ptr(), Cout, Cin, d)
        self._pack_jobs = (arr, len(jobs)) if jobs else None

    def pack_weights(self):
        if self.__dict__.get('_pack_jobs') is not None:
            arr, n = self._pack_jobs
            self.root._lib_handle().hrf_conv3x_pack(arr, n, _lib.stream_ptr())

    def packed(self, weight, direction):
        """-> the tap-major pack of `weight` for hrf_conv_fwd_packed (0) / hrf_conv_bwd_data_packed (1), or None"""
        ent = self.packs.get(id(weight))
        return None if ent is None else ent[direction]

    # ---- peer-to-peer SyncBN exchange (hrfuser_amd/p2p.py; HRF_SYNC_P2P=1 on every rank)
    def p2p_context(self, group, world):
        """The exchange context of this engine for (group, world), built at the first training forward that needs it - a
        COLLECTIVE operation (IPC handles are all-gathered over `group`, a handshake runs): every rank gets here at the same
        point of its program.  None: collectives go through the communicator (HRF_SYNC_P2P=0, a one-rank group in auto mode,
        or a failed handshake in auto mode - decided by ALL ranks together, with a warning)."""
        from . import p2p
        if group is None or not p2p.wanted(world):
            return None
        cur = self.__dict__.get('_p2p')
        if cur is not None and cur[0] is group and cur[1] == world and cur[3] == p2p.mode():
            return cur[2]
        if cur is not None and cur[2] is not None:
            # captured graphs (Trainer.capture, the module-boundary entries) carry the old context BY VALUE in the kernel
            # arguments of their exchange launches: inbox / flag pointers of every peer, the generation counter, the error
            # word.  A context any capture has pinned is retired, never closed - a replay after set_sync_group or a change
            # of HRF_SYNC_P2P must not write into freed or unmapped (remote!) memory
            if cur[2].pins:
                cur[2].retired = True                            # closed by the unpin of its last capture (P2PExchange.unpin)
                self.__dict__.setdefault('_p2p_retired', []).append(cur[2])
                self._p2p_retired[:] = [c for c in self._p2p_retired if c.base]      # (closed ones drop out)
            else:
                cur[2].close()
        import torch.distributed as dist
        rank = dist.get_rank(group) if world > 1 else 0
        ctx = p2p.P2PExchange.create(self.root._lib_handle(), self._bns, group, world, rank, self.device, strict=p2p.mode() == 'on')
        self.__dict__['_p2p'] = (group, world, ctx, p2p.mode())
        return ctx

    def grad_acc(self, p):
        """-> (accumulator tensor, copy_stride) for a parameter gradient written by many blocks."""
        o = self.pslot.get(id(p))
        if o is None:
            return p.grad, 0
        self.ps_dirty = True
        return self.ps_scratch[o:o + p.numel()], self.ps_n

    def fold_grads(self, L, stream):
        """Sum the replicated accumulators and the per-window slots of the fused attention blocks into the gradient
        arena (one launch each per backward)."""
        if self.ps_dirty and self.ps_n:
            L.hrf_fold_copies(self.ps_scratch, self.ps_n, self.ps_map, self.flat_g, self.ps_n, stream)
            R.gpu_zero_(self.ps_scratch)
        self.ps_dirty = False
        self.fold_slots_now(L, stream)

    def fs_bytes(self):
        """Bytes of slot data the pending hrf_fold_slots launch reads (cost model of the leaf balancer)."""
        return 4.0 * sum(self.fs_layers[k]['nslots'] * self.fs_layers[k]['n'] for k in self.fs_used)

    def fold_slots_now(self, L, stream):
        """hrf_fold_slots over the slots of this backward pass (once: as a leaf of the weight-gradient phase, or by fold_grads)."""
        if self.fs_used:
            L.hrf_fold_slots(self.fs_arena, self.fs_seg, len(self.fs_used), self.fs_map, self.flat_g, self.fs_maxn, stream)
            self.fs_used = []

    # ---- parameter-gradient slots of the fused attention blocks (runtime.attn_block): every window's workgroup writes
    # the complete partial sums of its window with plain stores (no atomics, deterministic), hrf_fold_slots adds them up
    def fs_register(self, key, nslots, entries, rpb=None, heads=0):
        """Called by the forward of a fused layer: slot layout {name: offset, '_n': slot size} of layer `key`;
        entries = [(name, parameter, first element, count)].  `rpb` (a trainable relative-position-bias table): the layer's dS
        planes [nslots][heads][49][49] live in the same arena, behind its slots, and hrf_rpb_grad_all gathers every layer's
        table gradient in one launch (rpb_grad_now)."""
        lay = self.fs_layers.get(key)
        rg = tuple(bool(p.requires_grad) for _, p, _, _ in entries) + (rpb is not None, heads)   # freezing / unfreezing re-maps
        if lay is None or lay['nslots'] != nslots or lay['rg'] != rg:
            offs, idx, o = {}, [], 0
            for name, p, first, cnt in entries:
                offs[name] = o
                base = self._poffs.get(id(p))
                idx.append(torch.arange(base + first, base + first + cnt, dtype=torch.int32) if (base is not None and p.requires_grad)
                           else torch.full((cnt,), -1, dtype=torch.int32))
                o += cnt
            offs['_n'] = o
            lay = self.fs_layers[key] = dict(nslots=nslots, offs=offs, map=torch.cat(idx), n=o, rg=rg, rpb=rpb, heads=heads)
            self.fs_sig = None
        self.fs_step.append(key)
        return lay['offs']

    def fs_prepare(self):
        """Start of a backward pass: place the slots of every fused layer of this step in one arena and build the
        segment table of hrf_fold_slots (cached while the set of layers does not change)."""
        keys = tuple(self.fs_step)
        self.fs_step = []
        self.fs_used = list(keys)
        self._rpb_todo = bool(keys)
        self.fs_tables(keys)

    def fs_tables(self, keys):
        """Arena + segment table for the fused layers `keys` (host-to-device copies: a module-boundary graph capture calls
        this BEFORE it starts capturing the backward pass)."""
        if not keys or keys == self.fs_sig:
            return
        seg, maps, off, moff = [], [], 0, 0
        rseg, self.fs_rpb_geo, self.fs_rpb_bytes = [], (0, 0), 0.0
        self.fs_off, self.fs_plane_off = {}, {}
        for k in keys:
            lay = self.fs_layers[k]
            self.fs_off[k] = off
            seg.append([off, lay['nslots'], lay['n'], lay['n'], moff])
            maps.append(lay['map'])
            off += lay['nslots'] * lay['n']
            moff += lay['n']
            if lay['rpb'] is not None:                         # dS planes of the layer + its row of hrf_rpb_grad_all's table
                off = (off + 3) & ~3
                self.fs_plane_off[k] = off
                acc, cs = self.grad_acc(lay['rpb'])
                rseg.append([off, lay['nslots'], lay['heads'], acc.data_ptr(), cs])
                off += lay['nslots'] * lay['heads'] * 49 * 49
                self.fs_rpb_geo = (max(self.fs_rpb_geo[0], lay['nslots']), max(self.fs_rpb_geo[1], lay['heads']))
                self.fs_rpb_bytes += 4.0 * lay['nslots'] * lay['heads'] * 49 * 49
        if self.fs_arena is None or self.fs_arena.numel() < off:
            self.fs_arena = torch.empty(off, device=self.device, dtype=torch.float32)
        self.fs_seg = torch.tensor(seg, dtype=torch.long).to(self.device)
        self.fs_map = torch.cat(maps).to(self.device)
        self.fs_rpb = torch.tensor(rseg, dtype=torch.long).to(self.device) if rseg else None
        self.fs_maxn = max(self.fs_layers[k]['n'] for k in keys)
        self.fs_sig = keys

    def fs_buffer(self, key):
        return self.fs_arena.data_ptr() + 4 * self.fs_off[key]

    def fs_plane(self, key):
        """dS planes of fused layer `key` (written by hrf_attn_block_bwd, read by rpb_grad_now)."""
        return self.fs_arena.data_ptr() + 4 * self.fs_plane_off[key]

    def rpb_pending(self):
        """True between the start of a backward pass with fused layers (fs_prepare) and its rpb_grad_now."""
        return self.__dict__.get('_rpb_todo', False) and getattr(self, 'fs_rpb', None) is not None

    def rpb_grad_now(self, L, stream):
        """relative_position_bias_table.grad of every fused layer of this backward pass, gathered from the dS planes in ONE
        launch (a leaf of the weight-gradient phase; it adds into the replicated accumulators, so it precedes fold_grads)."""
        if self.rpb_pending():
            L.hrf_rpb_grad_all(self.fs_arena, self.fs_rpb, int(self.fs_rpb.shape[0]), self.fs_rpb_geo[0], self.fs_rpb_geo[1], stream)
            self.ps_dirty = True
            self._rpb_todo = False

    # ---- per-step random pools: ONE Bernoulli launch (per drop probability) and one DropPath draw per step
    # instead of one torch RNG kernel chain per fusion block (every graph node costs ~5 us of host time)
    def _rng_begin(self):
        plan = self.__dict__.setdefault('_rng_plan', {})
        pools = self.__dict__.setdefault('_rng_pool', {})
        for key, need in plan.items():
            kind, p = key
            ent = pools.get(key)
            if ent is None or ent.numel() != need:             # persistent buffers: refilled in place
                ent = pools[key] = torch.empty(need, device=self.device, dtype=torch.float32)
            if kind == 'mask':
                ent.bernoulli_(1.0 - p)
            else:
                keep = 1.0 - p
                ent.uniform_().add_(keep).floor_().div_(keep)
        self._rng_calls = {}

    def _rng_take(self, kind, p, n, fresh):
        """A slice of the step's pool that belongs to the CALL SITE (`rng_site`, set by the block right before it draws, and
        the how-manieth draw of that site this step) - not to the position of the call in program order: the lock-step
        scheduler interleaves sibling strands differently from the serial one (HRF_LOCKSTEP=0, HRF_SYNC_LANE_COMMS=1), and
        the same seed must give every layer the same draws under either (bench.py sync_ab compares their gradients)."""
        key = (kind, float(p))
        calls = self.__dict__.setdefault('_rng_calls', {})
        site = self.__dict__.get('rng_site')
        k = calls.get((key, site), 0)
        calls[(key, site)] = k + 1
        slots = self.__dict__.setdefault('_rng_slots', {}).setdefault(key, {})
        slot = slots.get((site, k, n))                   # keyed by the size too: a shape that comes back re-uses its span
        if slot is None:                                 # (alternating input shapes used to grow the pools without bound)
            off = self._rng_plan.get(key, 0)
            slot = slots[(site, k, n)] = (off, n)
            self._rng_plan[key] = off + n
        ent = self._rng_pool.get(key)
        if ent is not None and slot[0] + n <= ent.numel():
            return ent[slot[0]:slot[0] + n]
        return fresh()                          # first training step (sizes unknown yet)

    def dropout_mask(self, shape, p):
        n = 1
        for d in shape:
            n *= d
        return self._rng_take('mask', p, n, lambda: R._new((n,), self.device).bernoulli_(1.0 - p)).view(shape)

    def droppath_scale(self, B, p):
        keep = 1.0 - p
        return self._rng_take('dp', p, B, lambda: (torch.rand(B, device=self.device) + keep).floor_().div_(keep))

    def pre_step(self, training):
        """The torch-side work of a step (random pools, num_batches_tracked): everything that is NOT a
        library launch, so that a recorded replay program can be preceded by exactly this call."""
        if training:
            self._rng_begin()
            # num_batches_tracked moves only for BatchNorms that are IN training mode (norm_eval keeps them frozen:
            # hrnet.py:588-596; F.batch_norm(training=False) leaves the buffer alone)
            flags = tuple(m.training for m in self._bns if m.num_batches_tracked is not None)
            if all(flags):
                self.nbt_flat.add_(1)
            elif any(flags):
                inc = self.__dict__.get('_nbt_inc')
                if inc is None or inc[0] != flags:
                    inc = self.__dict__['_nbt_inc'] = (flags, torch.tensor([int(f) for f in flags], dtype=torch.long).to(self.device))
                self.nbt_flat.add_(inc[1])

    def begin_forward(self, training, pre=True):
        # every forward re-uses the per-BatchNorm slots and the step's buffers: a tape recorded by an EARLIER forward of this
        # module must not be back-propagated afterwards (Ctx.run_backward checks the generation and raises)
        self.gen = getattr(self, 'gen', 0) + 1
        R.use_keep_list(self.keep)
        R.release_step_buffers()
        self.fs_step = []
        if pre:
            self.pre_step(training)
        else:
            self._rng_calls = {}
        if self.arena_d.numel():
            R.gpu_zero_(self.arena_d)
        self.pack_weights()
        if training and self.root.sync_group is not None and (self.root.sync_world > 1 or R.force_collectives()):
            px = self.p2p_context(self.root.sync_group, self.root.sync_world)
            if px is not None:
                if self.device.type != 'cuda' or not torch.cuda.is_current_stream_capturing():
                    px.verify_order()                 # (second training step: one object all-gather, then never again)
                px.tick(_lib.stream_ptr())            # the step generation the exchanges of this forward / backward carry
        if self._bns and not (training and all(m.training for m in self._bns)):
            self.refresh_eval_affine()

    def refresh_eval_affine(self):
        """Frozen-statistics affine of EVERY BatchNorm (eval mode / norm_eval): scale = gamma * rsqrt(running_var + eps),
        shift = beta - running_mean * scale, as F.batch_norm(training=False) computes it.  Six flat torch launches on the
        current stream, re-run at the start of every forward that uses them: weights and running statistics are written
        by the kernels through raw pointers (optimizer, train-mode forwards, hipGraph replays), so nothing here may be
        cached across steps."""
        n = self._csum
        e = self.eval_f
        with torch.no_grad():
            torch.add(self.rstat[n:2 * n], self._epsv, out=e[3 * n:4 * n])
            e[3 * n:4 * n].rsqrt_()
            torch.mul(self.flat_p[self._gidx.clamp_min(0)], e[3 * n:4 * n], out=e[0:n])
            torch.where(self._gidx >= 0, e[0:n], e[3 * n:4 * n], out=e[0:n])            # affine=False: gamma == 1
            e[2 * n:3 * n].copy_(self.rstat[0:n])
            beta = torch.where(self._bidx >= 0, self.flat_p[self._bidx.clamp_min(0)], torch.zeros((), device=e.device))
            torch.addcmul(beta, e[2 * n:3 * n], e[0:n], value=-1.0, out=e[n:2 * n])

    def bn_eval_affine(self, bn):
        return self.slots[id(bn)]['eval']


class EngineOwner:
    """Mixin for root modules that execute on the HIP engine."""
    sync_group = None
    sync_world = 1

    def _engine(self):
        eng = self.__dict__.get('_hrf_engine')
        if eng is None:
            eng = Engine(self)
            self.__dict__['_hrf_engine'] = eng
        return eng

    def _lib_handle(self):
        return _lib.lib()

    def _bn_slot(self, bn):
        return self._engine().slots[id(bn)]

    def _bn_eval_affine(self, bn):
        return self._engine().bn_eval_affine(bn)

    use_lanes = True          # multi-stream execution of independent branches / modality streams

    def _lane_pool(self, grow=False):
        pool = self.__dict__.setdefault('_hrf_lanes', [])
        if grow:
            lane = R.Lane(torch.cuda.Stream(priority=int(os.environ.get('HRF_LANE_PRIORITY', '0'))))
            pool.append(lane)
            return lane
        return pool

    def _lane_group(self, lane, base_group):
        """HRF_SYNC_LANE_COMMS=1: the communicator of a lane (same ranks as `base_group`), created on first use - every
        rank runs the same program, so the collective `new_group` calls happen in the same order everywhere."""
        groups = self.__dict__.setdefault('_hrf_lane_groups', {})
        g = groups.get(id(lane))
        if g is None:
            import torch.distributed as dist
            g = groups[id(lane)] = dist.new_group(ranks=dist.get_process_group_ranks(base_group))
        return g

    def _side_pool(self):
        pool = self.__dict__.get('_hrf_side')
        if pool is None:
            n = int(os.environ.get('HRF_SIDE_LANES', '6') or 6)
            # leaf work (weight gradients) must not delay the latency-bound data-gradient chain: lowest stream priority
            try:
                low = max(torch.cuda.Stream.priority_range())
            except Exception:
                low = 0
            if os.environ.get('HRF_SIDE_PRIORITY', '1') == '0':
                low = 0
            pool = [R.Lane(torch.cuda.Stream(priority=low)) for _ in range(max(1, n))]
            self.__dict__['_hrf_side'] = pool
        return pool

    def set_sync_group(self, group, world):
        """Enable SyncBN semantics: BN statistics are all-reduced over `group` (RCCL)."""
        self.sync_group, self.sync_world = group, world
        if group is not None:
            R.check_sync_schedule(group, world)

    def params_updated(self):
        """Kept for callers that signal a weight update; nothing is cached across steps any more."""


# ----------------------------------------------------------------------------- blocks
class Bottleneck(nn.Module):
    """resnet.py:263-302 - 1x1 -> 3x3 -> 1x1 (+downsample) with BN/ReLU, residual add, ReLU."""
    expansion = 4

    def __init__(self, inplanes, planes, norm_cfg, downsample=None):
        super().__init__()
        self._nn = [norm_name(norm_cfg, k) for k in (1, 2, 3)]
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.add_module(self._nn[0], build_bn(norm_cfg, planes))
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.add_module(self._nn[1], build_bn(norm_cfg, planes))
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.add_module(self._nn[2], build_bn(norm_cfg, planes * 4))
        self.downsample = downsample

    def run(self, ctx, x):
        n1, n2, n3 = (getattr(self, k) for k in self._nn)
        y = R.conv_bn(ctx, x, self.conv1, n1, R.TF_RELU)
        y = R.conv_bn(ctx, y, self.conv2, n2, R.TF_RELU)
        y = R.conv_bn(ctx, y, self.conv3, n3, R.TF_AFFINE)
        if self.downsample is not None:
            idt = R.conv_bn(ctx, x, self.downsample[0], self.downsample[1], R.TF_AFFINE)
            return R.materialize(ctx, y, R.ACT_RELU, lazy2=idt)
        return R.materialize(ctx, y, R.ACT_RELU, res=x)


class BasicBlock(nn.Module):
    """resnet.py:14-97 as the convolutional HRModule uses it (stride 1, no downsample): conv3x3-BN-ReLU-conv3x3-BN, + x, ReLU."""
    expansion = 1

    def __init__(self, inplanes, planes, norm_cfg, downsample=None):
        super().__init__()
        if downsample is not None or inplanes != planes:
            raise NotImplementedError('BasicBlock with a downsample path is unused by the HRNet branches')
        self._nn = [norm_name(norm_cfg, k) for k in (1, 2)]
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1, bias=False)
        self.add_module(self._nn[0], build_bn(norm_cfg, planes))
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.add_module(self._nn[1], build_bn(norm_cfg, planes))
        self.downsample = None

    def run(self, ctx, x):
        y = R.conv_bn(ctx, x, self.conv1, getattr(self, self._nn[0]), R.TF_RELU)
        y = R.conv_bn(ctx, y, self.conv2, getattr(self, self._nn[1]), R.TF_AFFINE)
        return R.materialize(ctx, y, R.ACT_RELU, res=x)


class CrossFFN(nn.Module):
    """hrformer.py:267-295.  Returns the LAZY tail BN(h3) (GELU applied by the caller's residual add)."""

    def __init__(self, in_channels, hidden_channels=None, out_channels=None, norm_cfg=dict(type='SyncBN'), **kw):
        super().__init__()
        out_channels = out_channels or in_channels
        hidden_channels = hidden_channels or in_channels
        self.layers = nn.Sequential(
            nn.Conv2d(in_channels, hidden_channels, 1), build_bn(norm_cfg, hidden_channels), nn.GELU(),
            nn.Conv2d(hidden_channels, hidden_channels, 3, 1, 1, groups=hidden_channels),
            build_bn(norm_cfg, hidden_channels), nn.GELU(),
            nn.Conv2d(hidden_channels, out_channels, 1), build_bn(norm_cfg, out_channels), nn.GELU())

    def run(self, ctx, ln_in):
        l = self.layers
        return self.run_tail(ctx, R.conv_bn(ctx, ln_in, l[0], l[1], R.TF_GELU))

    def run_tail(self, ctx, h1):
        """Everything after the 1x1 expansion (+BN+GELU, `h1`): the fused attention block produces h1 itself."""
        l = self.layers
        h = R.dwconv_bn(ctx, h1, l[3], l[4], R.TF_GELU)
        return R.conv_bn(ctx, h, l[6], l[7], R.TF_GELU)


class WindowMSA(nn.Module):
    def __init__(self, embed_dims, num_heads, window_size=(WIN, WIN), **kw):
        super().__init__()
        if tuple(window_size) != (WIN, WIN):
            raise NotImplementedError('the HIP attention core is specialised for 7x7 windows')
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WIN - 1) ** 2, num_heads))
        self.register_buffer('relative_position_index', _rel_index())
        self.qkv = nn.Linear(embed_dims, embed_dims * 3)
        self.out_proj = nn.Linear(embed_dims, embed_dims)


class LocalWindowSelfAttention(nn.Module):
    """hrformer.py:184-236 + WindowMSA.forward :96-131; returns x + out_proj(attn(LN(x)))."""

    def __init__(self, embed_dims, num_heads, window_size=WIN, with_pad_mask=False, **kw):
        super().__init__()
        if with_pad_mask:
            raise NotImplementedError('with_pad_mask=True is unused by every reference config')
        ws = (window_size, window_size) if isinstance(window_size, int) else tuple(window_size)
        self.attn = WindowMSA(embed_dims, num_heads, ws)

    def run(self, ctx, x, ln, cache=None, drop=None):
        a = self.attn
        B, H, W, C = x.t.shape
        lin = R.ln_input(ctx, x, ln, cache)
        qkv = R.Plain(R._new((B * H * W, 3 * C), x.t.device))
        R.linear_into(ctx, lin, a.qkv, qkv, 0)
        bq = a.qkv.bias
        o = R.window_attention(ctx, qkv, 0, qkv, C, qkv, 2 * C, bq[C:2 * C], bq[2 * C:], bq, C, bq, 2 * C,
                               a.relative_position_bias_table, a.num_heads, (B, H, W, C))
        return R.linear_residual(ctx, o, a.out_proj, x, drop=drop)


_DEBUG_PAD_CACHE = {}


def _debug_pad():
    """HRF_DEBUG_PAD="18:10,72:10" -> {width: microseconds} (read per call: tools/race_check.py changes it between steps)."""
    env = os.environ.get('HRF_DEBUG_PAD', '')
    if env not in _DEBUG_PAD_CACHE:
        _DEBUG_PAD_CACHE[env] = {int(k): float(v) for k, v in (kv.split(':') for kv in env.split(',') if ':' in kv)}
    return _DEBUG_PAD_CACHE[env]


class HRFormerBlock(nn.Module):
    """hrformer.py:365-373: x += DropPath(LSA(LN1(x))); x += DropPath(CrossFFN(LN2(x))).  DropPath is Identity on
    the HRFuser path (App. D-2: the rate never reaches the stages); the plain HRFormer applies its linspace schedule."""
    expansion = 1

    def __init__(self, in_channels, out_channels, num_heads, window_size=WIN, mlp_ratio=4, drop_path=0.0,
                 norm_cfg=dict(type='SyncBN'), transformer_norm_cfg=dict(type='LN', eps=1e-6), **kw):
        super().__init__()
        self.norm1 = build_ln(transformer_norm_cfg, in_channels)
        self.attn = LocalWindowSelfAttention(in_channels, num_heads, window_size,
                                             with_pad_mask=kw.get('with_pad_mask', False))
        self.norm2 = build_ln(transformer_norm_cfg, out_channels)
        self.ffn = CrossFFN(in_channels, int(in_channels * mlp_ratio), out_channels, norm_cfg)
        self.drop_path_prob = float(drop_path)

    def run(self, ctx, x, defer=False):
        """defer: the caller hands the result to another HRFormerBlock (the next block of the branch, or of the next
        single-branch module): the CrossFFN tail may then stay lazy (R.LazyTail) - the next block's fused attention launch
        forms the rows on load, no hrf_affine_act_res / hrf_act_bwd launch for this block."""
        p = self.drop_path_prob
        s1 = s2 = None
        if p > 0.0 and ctx.training and self.training:
            # mmcv DropPath: per-sample floor(keep + U[0,1)) / keep, drawn independently for the two residual paths
            eng = ctx.owner._engine()
            eng.rng_site = id(self)
            s1, s2 = eng.droppath_scale(x.shape[0], p), eng.droppath_scale(x.shape[0], p)
        msa = self.attn.attn
        C = x.shape[-1]
        if _debug_pad().get(C):
            # critical-lane probe (HRF_DEBUG_PAD="18:10,72:10": 10 us of idle time per block of that width, forward and
            # backward): a lane whose padding shows up in the step time is on the critical path
            ticks = int(_debug_pad()[C] * 100)
            ctx.L.hrf_debug_spin(ticks, ctx.stream)
            ctx.push(lambda: ctx.L.hrf_debug_spin(ticks, ctx.stream))
        if R.ffn_eval_ok(ctx, C, self.ffn):
            # eval forward (frozen BatchNorms, no tape): the attention half, then the WHOLE CrossFFN half in one launch with the
            # hidden tensor in LDS (csrc/ffn_eval.hip; hrformer.py:284-295) - 2 launches per block instead of 3 - 10
            if R.attn_block_ok(ctx, C, msa.num_heads):
                x1, _ = R.attn_block(ctx, id(self), msa.num_heads, x, x, self.norm1, self.norm1, (msa.qkv, 0), (msa.qkv, C),
                                     (msa.qkv, 2 * C), msa.relative_position_bias_table, msa.out_proj, x)
            else:
                x1 = self.attn.run(ctx, R.force(ctx, x), self.norm1)
            return R.ffn_eval(ctx, x1, self.norm2, self.ffn)
        if R.attn_block_ok(ctx, C, msa.num_heads) and self.ffn.layers[0].weight.shape[0] == 4 * C and not R.is_gn(self.ffn.layers[1]):
            # one launch: norm1 -> qkv -> window attention -> out_proj -> residual -> norm2 -> CrossFFN 1x1 expansion
            # (x may be the previous block's lazy tail: the launch forms it on load)
            x, h1 = R.attn_block(ctx, id(self), msa.num_heads, x, x, self.norm1, self.norm1, (msa.qkv, 0), (msa.qkv, C),
                                 (msa.qkv, 2 * C), msa.relative_position_bias_table, msa.out_proj, x,
                                 drop=None if s1 is None else (None, 1.0, s1),
                                 ffn=(self.norm2, self.ffn.layers[0], self.ffn.layers[1]))
            tail = self.ffn.run_tail(ctx, h1)
            if defer and R.tail_onload():
                return R.LazyTail(x, tail, s2)
            return R.materialize(ctx, tail, R.ACT_GELU, res=x, act_first=True, rowscale=s2)
        x = R.force(ctx, x)
        x = self.attn.run(ctx, x, self.norm1, drop=None if s1 is None else (None, 1.0, s1))
        tail = self.ffn.run(ctx, R.ln_input(ctx, x, self.norm2))
        return R.materialize(ctx, tail, R.ACT_GELU, res=x, act_first=True, rowscale=s2)


class WindowMCA(nn.Module):
    def __init__(self, embed_dim, num_heads, window_size=(WIN, WIN), proj_drop_rate=0., **kw):
        super().__init__()
        if tuple(window_size) != (WIN, WIN):
            raise NotImplementedError('the HIP attention core is specialised for 7x7 windows')
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * WIN - 1) ** 2, num_heads))
        self.register_buffer('relative_position_index', _rel_index())
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.proj_drop = nn.Dropout(proj_drop_rate)


class MultiWindowCrossAttention(nn.Module):
    """hrfuser_hrformer_based.py:189-248 + WindowMCA.forward :106-151."""

    def __init__(self, window_size=WIN, with_pad_mask=False, **kwargs):
        super().__init__()
        if with_pad_mask:
            raise NotImplementedError('with_pad_mask=True is unused by every reference config')
        ws = (window_size, window_size) if isinstance(window_size, int) else tuple(window_size)
        self.attn = WindowMCA(window_size=ws, **kwargs)

    def run(self, ctx, q_in, kv_in, acc, z, drop_path_scale):
        """acc + z + DropPath(Dropout(out_proj(attn(q_in, kv_in))))"""
        a = self.attn
        B, H, W, C = acc.t.shape
        dev = acc.t.device
        q = R.Plain(R._new((B * H * W, C), dev))
        kv = R.Plain(R._new((B * H * W, 2 * C), dev))
        R.linear_into(ctx, q_in, a.q_proj, q, 0)
        R.linear_into(ctx, kv_in, a.k_proj, kv, 0)
        R.linear_into(ctx, kv_in, a.v_proj, kv, C)
        o = R.window_attention(ctx, q, 0, kv, 0, kv, C, a.k_proj.bias, a.v_proj.bias, a.k_proj.bias, 0,
                               a.v_proj.bias, 0, a.relative_position_bias_table, a.num_heads, (B, H, W, C))
        drop = None
        p = a.proj_drop.p
        if ctx.training and a.proj_drop.training and (p > 0 or drop_path_scale is not None):
            mask = None
            if p > 0:
                ctx.owner._engine().rng_site = id(self)
                mask = ctx.owner._engine().dropout_mask((B, H, W, C), p)
            drop = (mask, 1.0 / (1.0 - p) if p > 0 else 1.0, drop_path_scale)
        return R.linear_residual(ctx, o, a.out_proj, acc, res2=z, drop=drop)


class HRFuserFusionBlock(nn.Module):
    """hrfuser_hrformer_based.py:305-317."""
    expansion = 1

    def __init__(self, in_channels, out_channels, num_heads, window_size=WIN, mlp_ratio=4, drop_path=0.0,
                 norm_cfg=dict(type='SyncBN'), transformer_norm_cfg=dict(type='LN', eps=1e-6), with_cp=False,
                 num_fused_modalities=2, **kwargs):
        super().__init__()
        self.with_cp = with_cp
        self.num_fused_modalities = M = num_fused_modalities
        self.norm1 = nn.ModuleList(build_ln(transformer_norm_cfg, in_channels) for _ in range(M))
        self.norm2 = nn.ModuleList(build_ln(transformer_norm_cfg, out_channels) for _ in range(M))
        self.attn = nn.ModuleList(MultiWindowCrossAttention(embed_dim=in_channels, num_heads=num_heads,
                                                            window_size=window_size, **kwargs) for _ in range(M))
        self.norm3 = build_ln(transformer_norm_cfg, out_channels)
        self.ffn = CrossFFN(in_channels, int(in_channels * mlp_ratio), out_channels, norm_cfg)
        self.drop_path_prob = float(drop_path)

    def _droppath_scale(self, ctx, B, dev):
        """mmcv DropPath: per-sample floor(keep + U[0,1)) / keep (train only)."""
        p = self.drop_path_prob
        if not (ctx.training and self.training) or p <= 0.0:
            return None
        ctx.owner._engine().rng_site = id(self)
        return ctx.owner._engine().droppath_scale(B, p)

    def run(self, ctx, x, mods):
        if self.with_cp and ctx.record:
            raise Exception('with_cp is currently not possible with CA Fusion module')
        B = x.t.shape[0]
        dev = x.t.device
        cache = {}
        acc = x
        M = self.num_fused_modalities
        C = x.t.shape[-1]
        heads = self.attn[0].attn.num_heads
        if R.attn_block_ok(ctx, C, heads) and self.ffn.layers[0].weight.shape[0] == 4 * C and not R.is_gn(self.ffn.layers[1]):
            # one launch per modality: norm1[k] / norm2[k] -> q / k / v -> window cross-attention -> out_proj -> Dropout ->
            # DropPath -> + z_k + running sum; the last one also runs norm3 and the CrossFFN 1x1 expansion
            h1 = None
            for k in range(M):
                a = self.attn[k].attn
                z = mods[k]
                dps = self._droppath_scale(ctx, B, dev)
                p = a.proj_drop.p
                drop = None
                if ctx.training and a.proj_drop.training and (p > 0 or dps is not None):
                    ctx.owner._engine().rng_site = (id(self), k)
                    mask = ctx.owner._engine().dropout_mask(tuple(x.t.shape), p) if p > 0 else None
                    drop = (mask, 1.0 / (1.0 - p) if p > 0 else 1.0, dps)
                fe = R.ffn_eval_ok(ctx, C, self.ffn)
                acc, h1 = R.attn_block(ctx, (id(self), k), heads, x, z, self.norm1[k], self.norm2[k], (a.q_proj, 0),
                                       (a.k_proj, 0), (a.v_proj, 0), a.relative_position_bias_table, a.out_proj, acc,
                                       res2=z, drop=drop,
                                       ffn=(self.norm3, self.ffn.layers[0], self.ffn.layers[1]) if (k == M - 1 and not fe) else None)
            if fe:                                  # eval: norm3 + the whole CrossFFN + residual in one launch
                return R.ffn_eval(ctx, acc, self.norm3, self.ffn)
            tail = self.ffn.run_tail(ctx, h1)
            return R.materialize(ctx, tail, R.ACT_GELU, res=acc, act_first=True,
                                 rowscale=self._droppath_scale(ctx, B, dev))
        for k in range(M):
            z = mods[k]
            q_in = R.ln_input(ctx, x, self.norm1[k], cache)      # every modality queries the PRE-fusion camera
            kv_in = R.ln_input(ctx, z, self.norm2[k])
            acc = self.attn[k].run(ctx, q_in, kv_in, acc, z, self._droppath_scale(ctx, B, dev))
        if R.ffn_eval_ok(ctx, C, self.ffn):
            return R.ffn_eval(ctx, acc, self.norm3, self.ffn)
        tail = self.ffn.run(ctx, R.ln_input(ctx, acc, self.norm3))
        return R.materialize(ctx, tail, R.ACT_GELU, res=acc, act_first=True,
                             rowscale=self._droppath_scale(ctx, B, dev))


_STAGE_ANCHOR = os.environ.get('HRF_STAGE_ANCHOR', '0') == '1'
_CAM_LANES = int(os.environ.get('HRF_CAM_LANES', '0') or 0)      # 0: one stream per camera branch (A/B knob)
_FORK_EXCHANGE = os.environ.get('HRF_FORK_EXCHANGE', '1') != '0'   # exchange chains on sibling lanes (0: serial)
_BRANCH_ORDER = os.environ.get('HRF_BRANCH_ORDER', '')
_PENDING_FUSION = os.environ.get('HRF_PENDING_FUSION', '1') != '0'
_ONE_FORK = os.environ.get('HRF_MODULE_ONE_FORK', '1') != '0'      # one fork / join per HRFomerModule (0: the round-3 structure)


class HRFomerModule(nn.Module):
    """hrnet.py:184-207 (HRModule.forward) with HRFormer fuse layers hrformer.py:498-561."""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels, num_heads,
                 num_window_sizes, num_mlp_ratios, multiscale_output=True, drop_paths=(0.0,), with_rpe=True,
                 with_pad_mask=False, conv_cfg=None, norm_cfg=dict(type='SyncBN', requires_grad=True),
                 transformer_norm_cfg=dict(type='LN', eps=1e-6), with_cp=False):
        super().__init__()
        if num_branches != len(num_blocks):
            raise ValueError(f'NUM_BRANCHES({num_branches}) != NUM_BLOCKS({len(num_blocks)})')
        if num_branches != len(num_channels):
            raise ValueError(f'NUM_BRANCHES({num_branches}) != NUM_CHANNELS({len(num_channels)})')
        if num_branches != len(num_inchannels):
            raise ValueError(f'NUM_BRANCHES({num_branches}) != NUM_INCHANNELS({len(num_inchannels)})')
        if not with_rpe:
            raise NotImplementedError('with_rpe=False is unused by every reference config')
        self.num_branches = nb = num_branches
        self.in_channels = list(num_inchannels)
        self.multiscale_output = multiscale_output
        ch = self.in_channels
        self.branches = nn.ModuleList(
            nn.Sequential(*[block(ch[i], num_channels[i], num_heads=num_heads[i], window_size=num_window_sizes[i],
                                  mlp_ratio=num_mlp_ratios[i],
                                  drop_path=drop_paths[min(b, len(drop_paths) - 1)] if drop_paths else 0.0,   # hrformer.py:453-497: block b <- drop_paths[b]
                                  norm_cfg=norm_cfg, transformer_norm_cfg=transformer_norm_cfg,
                                  with_pad_mask=with_pad_mask)
                            for b in range(num_blocks[i])]) for i in range(nb))
        self.fuse_layers = None
        if nb > 1:
            rows = []
            for i in range(nb if multiscale_output else 1):
                row = []
                for j in range(nb):
                    if j > i:
                        row.append(_conv_bn_seq(ch[j], ch[i], 1, 1, norm_cfg, relu=False))
                    elif j == i:
                        row.append(None)
                    else:
                        steps = []
                        for s in range(i - j):
                            last = s == i - j - 1
                            cout = ch[i] if last else ch[j]
                            mods = [nn.Conv2d(ch[j], ch[j], 3, 2, 1, groups=ch[j], bias=False), build_bn(norm_cfg, ch[j]),
                                    nn.Conv2d(ch[j], cout, 1, bias=False), build_bn(norm_cfg, cout)]
                            if not last:
                                mods.append(nn.ReLU(False))
                            steps.append(nn.Sequential(*mods))
                        row.append(nn.Sequential(*steps))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)

    def run(self, ctx, xs, defer_out=False):
        """defer_out (single-branch modules only): the output feeds the first block of the NEXT module of the stage and may
        stay a lazy CrossFFN tail."""
        nb = self.num_branches
        xs = list(xs)
        # parallel branches (hrnet.py:189-190); with launch merging the finest branch stays on the current lane, where the
        # modality stages run as well: equal calls of the three sensor streams become one multi-problem launch
        cap = ctx.__dict__.get('branch_lane_cap', 0)
        if cap and nb > cap:
            # beside the modality stages the runtime's 4 hardware queues are oversubscribed (camera branches + M streams) and
            # two streams aliased onto one queue run in order: the coarse ("thin") branches share ONE stream on purpose
            # instead - branch i runs on lane min(i, cap - 1) - so that no thin chain queues behind a fat one
            base = ctx.fork(cap, keep_first=ctx.keep_first)
            lanes = [base[min(i, cap - 1)] for i in range(nb)]
        else:
            lanes = ctx.fork(nb, keep_first=ctx.keep_first)

        def branch(i):
            if isinstance(xs[i], R.Pending):              # the previous module's exchange sum for this branch, on this lane
                xs[i] = R.force(ctx, xs[i])               # (a lazy CrossFFN tail of the previous single-branch module stays
            blocks = list(self.branches[i])               # lazy: the first block's fused attention launch forms it on load)
            for b, blk in enumerate(blocks):
                last = b == len(blocks) - 1
                if isinstance(blk, HRFormerBlock):
                    xs[i] = blk.run(ctx, xs[i], defer=(not last) or (nb == 1 and defer_out))
                else:
                    xs[i] = blk.run(ctx, R.force(ctx, xs[i]))
        nrows = len(self.fuse_layers) if nb > 1 else 0
        terms = [[None] * nb for _ in range(nrows)]

        def source(j):
            # every conv chain that reads xs[j] (the backward of this lane accumulates into xs[j].grad only: no cross-lane
            # gradient races)
            for i, row in enumerate(self.fuse_layers):
                if j == i:
                    terms[i][j] = ('id', xs[j])
                elif j > i:
                    terms[i][j] = ('up', R.conv_bn(ctx, xs[j], row[j][0], row[j][1], R.TF_AFFINE))
                else:
                    cur = xs[j]
                    for step in row[j]:
                        cur = R.dwconv_bn(ctx, cur, step[0], step[1], R.TF_AFFINE)
                        cur = R.conv_bn(ctx, cur, step[2], step[3], R.TF_RELU if len(step) == 5 else R.TF_AFFINE)
                    terms[i][j] = ('same', cur)
        order = list(range(nb))
        if _BRANCH_ORDER == 'rev':
            order.reverse()                              # (experiment: which lane of a fork starts late - the last one, or the coarse one?)
        if _ONE_FORK and nb > 1:
            # ONE fork / join per module: lane j = [pending exchange sum of branch j] -> its blocks -> the exchange chains that
            # read branch j.  (Rounds 1-3: branches | join | fork | exchange chains | join | all fuse_sums on the main lane:
            # ~25 us of cross-queue latency and 70-90 us of serial fuse_sum launches per module and direction.)
            def lane_body(i):
                branch(i)
                source(i)
            ctx.parallel([lanes[i] for i in order], [lambda i=i: lane_body(i) for i in order])
            ctx.join(lanes)
            return [R.LazyFuse(tuple(xs[i].t.shape), terms[i]) for i in range(nrows)]
        ctx.parallel([lanes[i] for i in order], [lambda i=i: branch(i) for i in order])
        ctx.join(lanes)
        if nb == 1:
            return [xs[0]]
        # exchange on sibling lanes of its own (HRF_MODULE_ONE_FORK=0), the sums on the main lane
        lanes = ctx.fork(nb, keep_first=ctx.keep_first) if _FORK_EXCHANGE else [ctx.cur] * nb
        ctx.parallel(lanes, [lambda j=j: source(j) for j in range(nb)])
        ctx.join(lanes)
        return [R.fuse_sum(ctx, tuple(xs[i].t.shape), terms[i]) for i in range(nrows)]


class HRModule(nn.Module):
    """The convolutional HRModule (hrnet.py:14-207): branches of BasicBlocks; exchange = 1x1 conv + BN + nn.Upsample
    (nearest) upwards (the F.interpolate(bilinear) that follows in hrnet.py:199-203 is the identity when the grids nest,
    which Pad(size_divisor=32) guarantees; other sizes raise) and chains of DENSE 3x3 stride-2 conv + BN (+ReLU) downwards."""

    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels, multiscale_output=True,
                 norm_cfg=dict(type='BN'), **kw):
        super().__init__()
        if num_branches != len(num_blocks):
            raise ValueError(f'NUM_BRANCHES({num_branches}) != NUM_BLOCKS({len(num_blocks)})')
        if num_branches != len(num_channels):
            raise ValueError(f'NUM_BRANCHES({num_branches}) != NUM_CHANNELS({len(num_channels)})')
        if num_branches != len(num_inchannels):
            raise ValueError(f'NUM_BRANCHES({num_branches}) != NUM_INCHANNELS({len(num_inchannels)})')
        self.num_branches = nb = num_branches
        self.in_channels = ch = list(num_inchannels)
        self.multiscale_output = multiscale_output
        self.branches = nn.ModuleList(
            nn.Sequential(*[block(ch[i], num_channels[i] * block.expansion, norm_cfg) for _ in range(num_blocks[i])])
            for i in range(nb))
        self.fuse_layers = None
        if nb > 1:
            rows = []
            for i in range(nb if multiscale_output else 1):
                row = []
                for j in range(nb):
                    if j > i:
                        row.append(nn.Sequential(nn.Conv2d(ch[j], ch[i], 1, bias=False), build_bn(norm_cfg, ch[i]),
                                                 nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                    elif j == i:
                        row.append(None)
                    else:
                        steps = []
                        for k in range(i - j):
                            last = k == i - j - 1
                            cout = ch[i] if last else ch[j]
                            mods = [nn.Conv2d(ch[j], cout, 3, 2, 1, bias=False), build_bn(norm_cfg, cout)]
                            if not last:
                                mods.append(nn.ReLU(inplace=False))
                            steps.append(nn.Sequential(*mods))
                        row.append(nn.Sequential(*steps))
                rows.append(nn.ModuleList(row))
            self.fuse_layers = nn.ModuleList(rows)

    def run(self, ctx, xs):
        nb = self.num_branches
        xs = list(xs)
        lanes = ctx.fork(nb, keep_first=ctx.keep_first)

        def branch(i):
            xs[i] = R.force(ctx, xs[i])                   # (a pending fusion block in front of the stage runs on this lane)
            for blk in self.branches[i]:
                xs[i] = blk.run(ctx, xs[i])
        ctx.parallel(lanes, [lambda i=i: branch(i) for i in range(nb)])
        ctx.join(lanes)
        if nb == 1:
            return [xs[0]]
        nrows = len(self.fuse_layers)
        terms = [[None] * nb for _ in range(nrows)]
        lanes = ctx.fork(nb, keep_first=ctx.keep_first)

        def source(j):                             # one lane per SOURCE branch (its backward accumulates into xs[j] only)
            for i, row in enumerate(self.fuse_layers):
                if j == i:
                    terms[i][j] = ('id', xs[j])
                elif j > i:
                    Hi, Wi, Hj, Wj = xs[i].t.shape[1], xs[i].t.shape[2], xs[j].t.shape[1], xs[j].t.shape[2]
                    if Hi != Hj << (j - i) or Wi != Wj << (j - i):
                        raise NotImplementedError('HRModule exchange on grids that do not nest (input not a multiple of 32)')
                    terms[i][j] = ('near', R.conv_bn(ctx, xs[j], row[j][0], row[j][1], R.TF_AFFINE))
                else:
                    cur = xs[j]
                    for step in row[j]:
                        cur = R.conv_bn(ctx, cur, step[0], step[1], R.TF_RELU if len(step) == 3 else R.TF_AFFINE)
                    terms[i][j] = ('same', cur)
        ctx.parallel(lanes, [lambda j=j: source(j) for j in range(nb)])
        ctx.join(lanes)
        return [R.fuse_sum(ctx, tuple(xs[i].t.shape), terms[i]) for i in range(nrows)]


def _make_transition(pre, cur, norm_cfg):
    """hrnet.py:419-463."""
    layers = []
    for i, c in enumerate(cur):
        if i < len(pre):
            layers.append(_conv_bn_seq(pre[i], c, 3, 1, norm_cfg, relu=True) if c != pre[i] else None)
        else:
            steps = []
            for j in range(i + 1 - len(pre)):
                cout = c if j == i - len(pre) else pre[-1]
                steps.append(_conv_bn_seq(pre[-1], cout, 3, 2, norm_cfg, relu=True))
            layers.append(nn.Sequential(*steps))
    return nn.ModuleList(layers)


def _run_conv_chain(ctx, x, seqs):
    """[Sequential(conv, bn, relu)]* -> materialised ReLU(BN(conv(...)))"""
    cur = x
    for seq in seqs:
        cur = R.conv_bn(ctx, cur, seq[0], seq[1], R.TF_RELU)
    return R.materialize(ctx, cur, R.ACT_RELU)


def _make_res_layer(inplanes, planes, blocks, norm_cfg):
    ds = None
    if inplanes != planes * 4:
        ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, bias=False), build_bn(norm_cfg, planes * 4))
    layers = [Bottleneck(inplanes, planes, norm_cfg, ds)]
    layers += [Bottleneck(planes * 4, planes, norm_cfg) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


# ----------------------------------------------------------------------------- weight initialisation
def resolve_init_cfg(pretrained, init_cfg):
    """hrnet.py:301-318: `pretrained` (deprecated str) becomes an init_cfg of type 'Pretrained'; both at once assert."""
    assert not (init_cfg and pretrained), 'init_cfg and pretrained cannot be specified at the same time'
    if isinstance(pretrained, str):
        warnings.warn('DeprecationWarning: pretrained is deprecated, please use "init_cfg" instead')
        return dict(type='Pretrained', checkpoint=pretrained)
    if pretrained is not None:
        raise TypeError('pretrained must be a str or None')
    return init_cfg


def load_pretrained(module, init_cfg):
    """mmcv `PretrainedInit` for a local checkpoint file: takes `state_dict` when present, strips `module.` and the
    optional `prefix` (e.g. 'backbone.'), loads non-strictly and reports missing / unexpected keys as a warning."""
    path = init_cfg['checkpoint']
    if not os.path.isfile(path):
        raise FileNotFoundError(f'init_cfg Pretrained: {path} is not a local file (there is no downloader here)')
    ckpt = torch.load(path, map_location='cpu')
    sd = ckpt.get('state_dict', ckpt) if isinstance(ckpt, dict) else ckpt
    prefix = init_cfg.get('prefix')
    out = {}
    for k, v in sd.items():
        if k.startswith('module.'):
            k = k[7:]
        if prefix:
            p = prefix if prefix.endswith('.') else prefix + '.'
            if not k.startswith(p):
                continue
            k = k[len(p):]
        out[k] = v
    res = module.load_state_dict(out, strict=False)
    if res.missing_keys or res.unexpected_keys:
        warnings.warn(f'Pretrained init from {path}: {len(res.missing_keys)} missing, '
                      f'{len(res.unexpected_keys)} unexpected keys')
    return res


def default_or_pretrained_init(module):
    """Kaiming(conv) / constant-1 (BN) as the reference's default init_cfg (hrnet.py:309-316) - RPB tables stay zero
    (App. D-4) - or the checkpoint of an init_cfg of type 'Pretrained'.  Any other init_cfg raises."""
    cfg = getattr(module, 'init_cfg', None)
    if cfg:
        if isinstance(cfg, dict) and cfg.get('type') == 'Pretrained':
            return load_pretrained(module, cfg)
        raise NotImplementedError(f'init_cfg {cfg!r}: only None (Kaiming / Constant default) and type=Pretrained are built')
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm)):      # hrnet.py:311-316
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


# ----------------------------------------------------------------------------- autograd bridge
class _BackboneFn(torch.autograd.Function):
    """Bridges the explicit tape to torch.autograd at the backbone boundary only."""

    @staticmethod
    def forward(fctx, module, anchor, *inputs):
        record = anchor is not None
        ctx, outs, srcs = module._execute(inputs, record)
        fctx.hrf = (ctx, outs, srcs)
        fctx.set_materialize_grads(False)
        return tuple(o.t.permute(0, 3, 1, 2) for o in outs)

    @staticmethod
    def backward(fctx, *gouts):
        ctx, outs, srcs = fctx.hrf
        for o, g in zip(outs, gouts):
            if g is None:
                o.grad = torch.zeros_like(o.t)
            else:
                o.grad = g.permute(0, 2, 3, 1).contiguous()
                if o.grad.data_ptr() == g.data_ptr():
                    o.grad = o.grad.clone()
        ctx.run_backward()
        grads = tuple(fctx_module_input_grad(s) for s in srcs)
        fctx.hrf = None
        _poll_exchange(ctx.owner)
        return (None, None) + grads


def _poll_exchange(module):
    """Step boundary of the autograd route (what mmdet runs: backbone() ... loss.backward()): a SyncBN exchange that lost a
    peer raises HERE, one step late at the latest, instead of training on NaN / unsynchronised statistics for ever
    (P2PExchange.poll: a non-blocking read of the error word)."""
    px = module._engine().__dict__.get('_p2p')
    if px is not None and px[2] is not None:
        px[2].poll()


def fctx_module_input_grad(src):
    if not src.needs_grad or src.grad is None:
        return None
    if isinstance(src, R.RawInput):
        return src.grad                        # already NCHW
    return src.grad.permute(0, 3, 1, 2)        # NHWC Act of a block harness -> logical NCHW


class _GraphEntry:
    """One captured (shape, mode) instance of a module: static input / cotangent buffers, the forward hipGraph, the tape of the
    capture-time forward and - from the first backward on - the backward hipGraph replaying that tape's launches."""
    __slots__ = ('fwd', 'bwd', 'ins', 'ctx', 'outs', 'srcs', 'gouts', 'keep', 'gen', 'calls', 'needs', 'failed', 'ctx_record',
                 'refs')          # engine-owned buffers whose raw pointers the captured launches carry (see _pin_engine_buffers)


class _GraphedFn(torch.autograd.Function):
    """torch.autograd bridge of a captured module instance: copy-in, replay, copy-out (forward and backward)."""

    @staticmethod
    def forward(fctx, module, ent, anchor, *inputs):
        fctx.set_materialize_grads(False)
        outs = module._graph_forward(ent, inputs)
        fctx.hrf = (module, ent, ent.gen)                  # the engine generation of THIS replay
        return outs

    @staticmethod
    def backward(fctx, *gouts):
        module, ent, gen = fctx.hrf
        fctx.hrf = None
        return (None, None, None) + module._graph_backward(ent, gouts, gen)


class HipModule(nn.Module, EngineOwner):
    """Root of a module tree executed on the HIP engine (the backbone, or a sub-block under test)."""

    def _execute(self, inputs, record, needs=None):
        dev = inputs[0].device
        if dev.type != 'cuda' and _lib.lib().require_cuda:
            raise _lib.HRFuserHipError(
                f'HRFuser HIP path invoked with tensors on {dev}: there is no CPU fallback '
                '(the CPU oracle lives in oracle/ and is test infrastructure only)')
        eng = self._engine()
        eng.ready(dev)
        eng.begin_forward(self.training)
        ctx = R.Ctx(self, self.training, record)
        ctx.gen = eng.gen
        st = self.__dict__.get('_stage_stamps')
        if st is not None:
            st.reset()
        ls = self.__dict__.get('_lane_stamps')
        if ls is not None:
            ls.reset()
            ls.nfork = 0
            ls.lane_fork = {}
        with torch.no_grad():
            srcs = self._wrap_inputs(inputs)
            if needs is not None:                       # static graph inputs: the flags of the caller's tensors
                for src, n in zip(srcs, needs):
                    src.needs_grad = bool(n)
            ctx.mark('start')
            outs = R.force_all(ctx, self._run(ctx, srcs))       # (a sub-module harness may hand back pending exchange sums)
            if ctx.probe is not None:
                self.__dict__['_relu_masks'] = R.collect_relu_masks(ctx)
        return ctx, outs, srcs

    def enable_stage_stamps(self, on=True):
        """Measurement aid (bench.py per-stage report): GPU timestamps at the stage boundaries of every following pass
        (runtime.StageStamps); returns the stamp object, or None when switched off."""
        if not on:
            self.__dict__.pop('_stage_stamps', None)
            return None
        dev = next(self.parameters()).device
        st = self.__dict__['_stage_stamps'] = R.StageStamps(dev)
        return st

    def enable_lane_stamps(self, on=True, cap=8192):
        """Measurement aid (tools/lane_stamps.py): GPU timestamps at every fork / join of the lanes (runtime.Ctx._lane_stamp)."""
        if not on:
            self.__dict__.pop('_lane_stamps', None)
            return None
        st = self.__dict__['_lane_stamps'] = R.StageStamps(next(self.parameters()).device, cap)
        return st

    # ---- captured graphs at the module boundary -----------------------------------------------------------------------
    # What mmdet calls is `backbone(img, mods)` + `loss.backward()` (two_stage.py:76-84): ~1 000 host-paced launches per
    # direction when issued eagerly.  A module instance that sees the SAME (shapes, mode, requires-grad pattern) again is
    # captured - forward into one hipGraph, the backward of that forward into a second one - and replayed from then on with
    # copy-in / copy-out of the inputs, outputs, cotangents and input gradients.  HRF_MODULE_GRAPH=0 switches it off; a
    # failed capture falls back to eager launches with a warning.
    _graphable = False                 # the backbones opt in (sub-block harnesses and the neck launch eagerly)
    _graph_warmup = 2                  # eager calls of a key before it is captured (allocator, random-pool sizes, slot tables)

    def reset_graphs(self):
        """Drop every captured instance (call after structural changes: freezing parameters, loading another device)."""
        cache = self.__dict__.pop('_hrf_graphs', None) or {}
        from . import p2p as _p2p
        for k, ent in cache.items():
            if k == '_setup':
                continue
            for r in getattr(ent, 'refs', None) or []:         # the entry's captures pinned their exchange context
                if isinstance(r, _p2p.P2PExchange):
                    r.unpin()

    def _graph_key(self, inputs, record):
        flags = tuple(m.training for m in self._engine()._bns) if self._engine()._bns else ()
        return (self.training, record, tuple((tuple(t.shape), tuple(t.stride()), bool(t.requires_grad)) for t in inputs),
                hash(flags), R.force_collectives(), self.sync_group is not None)

    def _graph_entry(self, inputs, record):
        """-> the _GraphEntry to run this call through, or None for an eager call."""
        if not self._graphable or not inputs[0].is_cuda or os.environ.get('HRF_MODULE_GRAPH', '1') == '0' or \
                self.__dict__.get('_relu_probe') or self.__dict__.get('_stage_stamps') is not None or \
                torch.cuda.is_current_stream_capturing():
            return None
        cache = self.__dict__.setdefault('_hrf_graphs', {})
        eng = self._engine()
        if cache.get('_setup') is not eng.flat_p:           # parameters re-homed (device move, new arena): start over
            cache.clear()
            cache['_setup'] = eng.flat_p
        key = self._graph_key(inputs, record)
        ent = cache.get(key)
        if ent is None:
            ent = cache[key] = _GraphEntry()
            ent.fwd = ent.bwd = ent.ctx = None
            ent.calls, ent.failed = 0, False
        ent.calls += 1
        if ent.failed or ent.calls <= self._graph_warmup:
            return None
        return ent

    @staticmethod
    def _pin_engine_buffers(ent, eng):
        """The captured launches carry RAW pointers into buffers the engine allocates outside the capture and REBINDS when
        another input signature comes along: the slot arena / segment table / index map of hrf_fold_slots and the fused
        attention backward (Engine.fs_tables), and the Dropout / DropPath pools (Engine._rng_begin).  The entry keeps every
        such tensor alive for as long as it exists, so a later eager step or a second captured signature can rebind the
        engine's attributes (new tensors) without the replays of THIS entry reading freed or reused memory."""
        ent.refs = getattr(ent, 'refs', None) or []
        for t in (eng.fs_arena, getattr(eng, 'fs_seg', None), getattr(eng, 'fs_map', None), getattr(eng, 'fs_rpb', None),
                  eng.ps_scratch, eng.ps_map):
            if t is not None:
                ent.refs.append(t)
        ent.refs.extend(eng.__dict__.get('_rng_pool', {}).values())
        px = eng.__dict__.get('_p2p')
        if px is not None and px[2] is not None and not any(r is px[2] for r in ent.refs):
            px[2].pin()                                      # exchange launches of this capture point into the context's inboxes
            ent.refs.append(px[2])

    def _graph_forward(self, ent, inputs):
        eng = self._engine()
        if ent.fwd is None:
            ent.ins = [torch.empty_like(t) for t in inputs]
            ent.needs = [bool(t.requires_grad) for t in inputs]
            ent.keep = []
            ent.refs = []
        for d, t in zip(ent.ins, inputs):
            d.copy_(t)
        if ent.fwd is None:
            torch.cuda.synchronize()
            cur = eng.__dict__.get('_p2p')
            if cur is not None and cur[2] is not None:
                cur[2].verify_order()                        # the liveness assertion runs OUTSIDE the capture, before any replay
            g = torch.cuda.CUDAGraph()
            saved = eng.keep
            eng.keep = ent.keep                              # the capture's step buffers live as long as the entry
            try:
                with R.gc_paused(), torch.cuda.graph(g, capture_error_mode='thread_local'):
                    ent.ctx, ent.outs, ent.srcs = self._execute(tuple(ent.ins), ent.ctx_record, ent.needs)
            finally:
                eng.keep = saved
            ent.fwd = g
            ent.gouts = None
            self._pin_engine_buffers(ent, eng)
        ent.fwd.replay()
        eng.gen = getattr(eng, 'gen', 0) + 1                 # a replay overwrites the BatchNorm slots like any forward
        ent.gen = eng.gen
        return tuple(o.t.clone().permute(0, 3, 1, 2) for o in ent.outs)

    def _graph_backward(self, ent, gouts, gen):
        eng = self._engine()
        if gen != getattr(eng, 'gen', gen):
            raise _lib.HRFuserHipError(
                'backward of a forward pass that is no longer the latest one of this module (the BatchNorm statistics slots '
                'were overwritten by a later forward): run backward before the next forward of the same module')
        if ent.gouts is None:
            ent.gouts = [torch.zeros_like(o.t) for o in ent.outs]
        for d, g in zip(ent.gouts, gouts):
            if g is None:
                d.zero_()
            else:
                d.copy_(g.permute(0, 2, 3, 1))
        if ent.bwd is None:
            # this entry gets tables (and an arena) of its OWN: the engine's current ones may be shared with another
            # signature's replays, and fs_tables re-uses an arena that is large enough
            eng.fs_sig, eng.fs_arena = None, None
            eng.fs_tables(tuple(eng.fs_step))                # (host-to-device table copies: not inside the capture)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            saved = eng.keep
            eng.keep = ent.keep
            ent.ctx.gen = eng.gen
            try:
                with R.gc_paused(), torch.cuda.graph(g, pool=ent.fwd.pool(), capture_error_mode='thread_local'):
                    R.use_keep_list(ent.keep)
                    for o, d in zip(ent.outs, ent.gouts):
                        o.grad = R.gpu_clone(d)
                    ent.ctx.run_backward()
            finally:
                eng.keep = saved
            ent.bwd = g
            self._pin_engine_buffers(ent, eng)
            eng.fs_sig = None                                # the next eager backward builds its own tables
        ent.bwd.replay()
        _poll_exchange(self)
        return tuple((fctx_module_input_grad(s).clone() if (s.needs_grad and s.grad is not None) else None) for s in ent.srcs)

    def _call_engine(self, inputs):
        # NCHW-contiguous (the reference's layout) or channels-last storage (what hrfuser_amd.pipeline / the HIP modules
        # produce) are both read in place through element strides; anything else is made contiguous first
        inputs = tuple((t if (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)) else t.contiguous()).float()
                       for t in inputs)
        anchor = None
        if torch.is_grad_enabled():
            # parameters are not autograd inputs (their grads are written by the kernels straight
            # into param.grad); a dummy leaf makes autograd call our tape when outputs get grads
            anchor = self.__dict__.get('_hrf_anchor')
            if anchor is None or anchor.device != inputs[0].device:
                anchor = torch.zeros(1, device=inputs[0].device, requires_grad=True)
                self.__dict__['_hrf_anchor'] = anchor
        record = anchor is not None
        ent = None
        if self._graphable and inputs[0].is_cuda:
            self._engine().ready(inputs[0].device)
            ent = self._graph_entry(inputs, record)
        if ent is not None:
            ent.ctx_record = record
            try:
                return list(_GraphedFn.apply(self, ent, anchor, *inputs))
            except _lib.HRFuserHipError:
                raise
            except Exception as e:                              # capture refused: this key launches eagerly from now on
                if ent.fwd is not None and ent.bwd is not None:
                    raise
                ent.failed = True
                ent.fwd = ent.bwd = ent.ctx = None
                torch.cuda.synchronize()
                warnings.warn(f'{type(self).__name__}: hipGraph capture at the module boundary failed '
                              f'({type(e).__name__}: {str(e)[:160]}); this input signature launches eagerly')
        res = _BackboneFn.apply(self, anchor, *inputs)
        return list(res)


@BACKBONES.register_module()
class HRFuserHRFormerBased(HipModule):
    """HRFuser backbone (camera HRFormer stream + M modality streams + multi-window cross-attention
    fusion before stages 2/3/4) on hand-written gfx950 kernels.

    Drop-in for mmdet's `HRFuserHRFormerBased` (hrfuser_hrformer_based.py:330-628): same registry
    name, constructor keywords, `forward(x, x_mod) -> list[4]` contract, exceptions and state-dict.
    """
    blocks_dict = {'BOTTLENECK': Bottleneck, 'HRFORMER': HRFormerBlock, 'CA': HRFuserFusionBlock,
                   'MWCA': HRFuserFusionBlock}
    _hrformer_trunk = True
    _graphable = True

    def __init__(self, extra, in_channels=3, conv_cfg=None, norm_cfg=dict(type='SyncBN', requires_grad=True),
                 transformer_norm_cfg=dict(type='LN', eps=1e-6), norm_eval=False, with_cp=False,
                 drop_path_rate=0., zero_init_residual=False, multiscale_output=True, pretrained=None,
                 init_cfg=None, num_fused_modalities=2, mod_in_channels=[3, 3]):
        super().__init__()
        assert 'stage1' in extra and 'stage2' in extra and 'stage3' in extra and 'stage4' in extra
        for i in range(4):
            cfg = extra[f'stage{i + 1}']
            assert len(cfg['num_blocks']) == cfg['num_branches'] and \
                len(cfg['num_channels']) == cfg['num_branches']
        if conv_cfg is not None:
            raise NotImplementedError('conv_cfg must be None (plain Conv2d), as in every reference config')
        if with_cp:
            # the reference raises inside the fusion blocks as soon as gradients are needed (:321-322)
            pass
        self.extra = extra
        self.pretrained = pretrained
        self.init_cfg = resolve_init_cfg(pretrained, init_cfg)
        # zero_init_residual is dead code in the reference: hrnet.py:480-481,521-522 require `not hasattr(self, 'init_cfg')`,
        # and BaseModule.__init__ always sets that attribute - norm3 is never zero-initialised.  Kept as an attribute only.
        self.zero_init_residual = zero_init_residual
        self.norm_cfg, self.transformer_norm_cfg = norm_cfg, transformer_norm_cfg
        self.norm_eval, self.with_cp = norm_eval, with_cp
        self.num_fused_modalities = M = num_fused_modalities
        self.pre_neck_fusion = True if extra.get('LidarStageD') else False          # :364-366 (off in every config)
        ncfg, lcfg = norm_cfg, transformer_norm_cfg
        if self._hrformer_trunk:
            # HRFormer.__init__ :666-678 - drop_path_rate is swallowed (always 0 here); `extra` is mutated
            for s in ('stage2', 'stage3', 'stage4'):
                n = extra[s]['num_blocks'][0] * extra[s]['num_modules']
                extra[s]['drop_path_rates'] = [0.0] * n
            extra['LidarStageB']['drop_path_rates'] = extra['stage2']['drop_path_rates']
            extra['LidarStageC']['drop_path_rates'] = extra['stage3']['drop_path_rates']
            if self.pre_neck_fusion:
                extra['LidarStageD']['drop_path_rates'] = extra['stage4']['drop_path_rates']

        # camera stem + stage 1 (hrnet.py:337-371)
        self.conv1 = nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False)
        self._stem_nn = [norm_name(ncfg, 1), norm_name(ncfg, 2)]      # 'bn1' / 'bn2', or 'gn1' / 'gn2' (hrnet.py:338-360)
        self.add_module(self._stem_nn[0], build_bn(ncfg, 64))
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.add_module(self._stem_nn[1], build_bn(ncfg, 64))
        self.stage1_cfg = extra['stage1']
        blk = self.blocks_dict[self.stage1_cfg['block']]
        if blk is not Bottleneck:
            raise NotImplementedError('stage1 block must be BOTTLENECK')
        c1 = self.stage1_cfg['num_channels'][0]
        self.layer1 = _make_res_layer(64, c1, self.stage1_cfg['num_blocks'][0], ncfg)
        pre = [c1 * 4]
        for si in (2, 3, 4):
            cfg = extra[f'stage{si}']
            setattr(self, f'stage{si}_cfg', cfg)
            ch = [c * self.blocks_dict[cfg['block']].expansion for c in cfg['num_channels']]
            setattr(self, f'transition{si - 1}', _make_transition(pre, ch, ncfg))
            stage, pre = self._make_stage(cfg, ch, multiscale_output if si == 4 else True)
            setattr(self, f'stage{si}', stage)

        # modality stems + stage A (hrfuser_hrformer_based.py:375-412)
        self.conv_a = nn.ModuleList(nn.Conv2d(mod_in_channels[k], 64, 3, 2, 1, bias=False) for k in range(M))
        self.norm_a = nn.ModuleList(build_bn(ncfg, 64) for _ in range(M))
        self.conv_b = nn.ModuleList(nn.Conv2d(64, 64, 3, 2, 1, bias=False) for _ in range(M))
        self.norm_b = nn.ModuleList(build_bn(ncfg, 64) for _ in range(M))
        sa = extra['LidarStageA']
        self.stage_a_cfg = sa
        self.layer_a = nn.ModuleList(_make_res_layer(64, sa['num_channels'][0], sa['num_blocks'][0], ncfg)
                                     for _ in range(M))
        pre_m = [[sa['num_channels'][0] * 4] for _ in range(M)]
        for tag, nxt in (('a', 'b'), ('b', 'c'), ('c', None)):
            fcfg = extra[f'ModFusion{tag.upper()}']
            setattr(self, f'fusion_{tag}_cfg', fcfg)
            ch = list(fcfg['num_channels'])
            setattr(self, f'transition_{tag}',
                    nn.ModuleList(_make_transition(pre_m[k], ch, ncfg) for k in range(M)))
            setattr(self, f'fusion_{tag}', self._make_multimodal_fusion(fcfg, ch))
            if nxt is not None:
                scfg = extra[f'LidarStage{nxt.upper()}']
                setattr(self, f'stage_{nxt}_cfg', scfg)
                sch = list(scfg['num_channels'])
                stages = [self._make_stage(scfg, sch)[0] for _ in range(M)]
                setattr(self, f'stage_{nxt}', nn.ModuleList(stages))
                pre_m = [sch for _ in range(M)]
        if self.pre_neck_fusion:
            # modality stage D + a fourth fusion AFTER camera stage 4, then ReLU (hrfuser_hrformer_based.py:454-468,609-625)
            scfg = extra['LidarStageD']
            self.stage_d_cfg = scfg
            sch = list(scfg['num_channels'])
            self.stage_d = nn.ModuleList([self._make_stage(scfg, sch)[0] for _ in range(M)])
            fcfg = extra['ModFusionD']
            self.fusion_d_cfg = fcfg
            ch = list(fcfg['num_channels'])
            self.transition_d = nn.ModuleList(_make_transition(sch, ch, ncfg) for _ in range(M))
            self.fusion_d = self._make_multimodal_fusion(fcfg, ch)
        self.init_weights()

    # -- construction helpers -------------------------------------------------------------------
    def _make_stage(self, cfg, in_channels, multiscale_output=True):
        block = self.blocks_dict[cfg['block']]
        if block is not HRFormerBlock:
            raise NotImplementedError('stages 2-4 / LidarStage B-C must use block HRFORMER')
        n = cfg['num_modules']
        nb0 = cfg['num_blocks'][0]
        dpr = cfg.get('drop_path_rates', [0.0] * (n * nb0))
        mods = []
        for i in range(n):
            ms = multiscale_output or i != n - 1
            mods.append(HRFomerModule(cfg['num_branches'], block, cfg['num_blocks'], in_channels,
                                      cfg['num_channels'], cfg['num_heads'], cfg['window_sizes'], cfg['mlp_ratios'],
                                      ms, drop_paths=dpr[nb0 * i:nb0 * (i + 1)],
                                      with_rpe=self.extra.get('with_rpe', True),
                                      with_pad_mask=self.extra.get('with_pad_mask', False),
                                      norm_cfg=self.norm_cfg, transformer_norm_cfg=self.transformer_norm_cfg,
                                      with_cp=self.with_cp))
            in_channels = mods[-1].in_channels
        return nn.Sequential(*mods), in_channels

    def _make_multimodal_fusion(self, cfg, num_inchannels):
        if cfg['block'] not in ('CA', 'MWCA'):
            raise Exception('Not valid fusion block')
        block = self.blocks_dict[cfg['block']]
        return nn.ModuleList(
            block(num_inchannels[i], cfg['num_channels'][i], num_heads=cfg['num_heads'][i],
                  window_size=cfg['window_sizes'][i], mlp_ratio=cfg['mlp_ratios'][i], drop_path=cfg['drop_path'],
                  norm_cfg=self.norm_cfg, transformer_norm_cfg=self.transformer_norm_cfg,
                  num_fused_modalities=self.num_fused_modalities, proj_drop_rate=cfg['proj_drop_rate'])
            for i in range(cfg['num_branches']))

    def init_weights(self):
        """BaseModule.init_weights with the init_cfg resolved in the constructor (hrnet.py:301-318)."""
        default_or_pretrained_init(self)

    def train(self, mode=True):
        """hrnet.py:588-596 (norm_eval keeps BN frozen).  Unlike the reference this returns self."""
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def unused_parameter_names(self):
        """Parameters that never receive a gradient: the camera stream takes only the FIRST child of
        transition1[0] (hrfuser_hrformer_based.py:550-551), so its BatchNorm is dead; the reference trains with
        find_unused_parameters=True (mmdet/apis/train.py:114) and torch.optim skips parameters without a gradient."""
        return ('transition1.0.1.weight', 'transition1.0.1.bias')

    # -- execution ------------------------------------------------------------------------------
    def forward(self, x, x_mod):
        if not self.num_fused_modalities == len(x_mod):
            raise Exception('num_fused_modalities does not fit the given input length')
        for m in x_mod:
            if m.shape[0] != x.shape[0] or m.shape[2:] != x.shape[2:]:
                raise AssertionError('camera and modality inputs must share batch and spatial size')
        return self._call_engine((x,) + tuple(x_mod))

    def refresh_inputs(self, inputs):
        """Channels-last copies of the NCHW network inputs for the stem weight gradient (torch copies:
        part of the per-step torch-side work, see Engine.pre_step)."""
        bufs = self.__dict__.setdefault('_nhwc_in', {})
        for i, t in enumerate(inputs):
            Bn, C, H, W = t.shape
            if t.is_contiguous(memory_format=torch.channels_last) and not t.is_contiguous():
                bufs[i] = t.permute(0, 2, 3, 1)             # already channels-last storage: a view, no copy
                continue
            b = bufs.get(i)
            if b is None or b.shape != (Bn, H, W, C) or b.device != t.device or not b.is_contiguous():
                b = bufs[i] = torch.empty(Bn, H, W, C, device=t.device, dtype=torch.float32)
            b.copy_(t.permute(0, 2, 3, 1))
        return bufs

    def _wrap_inputs(self, inputs):
        bufs = {}
        if self.training and inputs[0].is_cuda:
            bufs = self.refresh_inputs(inputs)
        return [R.RawInput(t, bool(t.requires_grad), bufs.get(i)) for i, t in enumerate(inputs)]

    def _stem(self, ctx, src, c1, b1, c2, b2, layer):
        y = R.conv_bn(ctx, src, c1, b1, R.TF_RELU)
        y = R.conv_bn(ctx, y, c2, b2, R.TF_RELU)
        x = R.materialize(ctx, y, R.ACT_RELU)
        for blk in layer:
            x = blk.run(ctx, x)
        return x

    def _fuse_stage(self, ctx, cam_in, trans_cam, trans_mod, fusion, nb, mods, first, tag=''):
        M = self.num_fused_modalities
        cams = [None] * nb
        ms = [[None] * M for _ in range(nb)]
        # phase 1 - transitions: lane 0 = camera, lane 1+k = modality k (each lane only ever
        # back-propagates into its own source tensor); equal-shape strands share the lane and merge their launches
        lanes = ctx.bundle_lanes(1 + M, 'trans')

        def cam_trans():
            for i in range(nb):
                if first:
                    # reference quirk (:550-551): transition1[i][0] takes only the FIRST child:
                    #   branch 0 -> the bare conv (no BN / ReLU); branch 1 -> conv + BN + ReLU.
                    t0 = trans_cam[i][0]
                    cams[i] = self._bare_conv(ctx, cam_in, t0) if isinstance(t0, nn.Conv2d) \
                        else _run_conv_chain(ctx, cam_in, [t0])
                elif trans_cam[i] is not None:
                    tr = trans_cam[i]
                    same = isinstance(tr[0], nn.Conv2d)      # same-index channel change (3x3 s1)
                    cams[i] = _run_conv_chain(ctx, cam_in[i] if same else cam_in[-1], [tr] if same else list(tr))
                else:
                    cams[i] = cam_in[i]

        def mod_trans(k):
            for i in range(nb):
                tr = trans_mod[k][i]
                ms[i][k] = mods[k] if tr is None else \
                    _run_conv_chain(ctx, mods[k], [tr] if isinstance(tr[0], nn.Conv2d) else list(tr))
        ctx.parallel(lanes, [cam_trans] + [lambda k=k: mod_trans(k) for k in range(M)])
        ctx.join(lanes)
        ctx.mark('transitions_' + tag)
        # phase 2 - one fusion block per branch: PENDING - the stage's first module runs fusion block i at the head of branch
        # lane i (no fork / join of its own), and the modality stages, which need the transitions only, start beside them
        # instead of behind them.  The modality stage k and fusion block 0 both read ms[0][k] and back-propagate on different
        # lanes: the stage gets a handle with a gradient slot of its own (split_grad).
        xs = [R.Pending(tuple(cams[i].t.shape), lambda c, i=i: fusion[i].run(c, cams[i], ms[i])) for i in range(nb)]
        m0 = [R.split_grad(ctx, ms[0][k]) for k in range(M)]
        if not _PENDING_FUSION:                              # (HRF_PENDING_FUSION=0: the round-3 structure, A/B)
            xs = R.force_all(ctx, xs)
            ctx.mark('fusion_' + tag)
        return xs, m0

    def _bare_conv(self, ctx, x, conv):
        """3x3 conv WITHOUT BatchNorm (transition1[0][0] quirk)."""
        B, H, W, C = x.t.shape
        out = R.Plain(R._new((B * H * W, conv.weight.shape[0]), x.t.device))
        L, s = ctx.L, ctx.stream
        w = conv.weight
        Cout = w.shape[0]
        strides = (H * W * C, W * C, C, 1)
        L.hrf_conv_fwd(x.t, *strides, B, H, W, C, w, None, 3, 1, Cout, out.t, Cout, 0, None, None, 0,
                       R.TF_NONE, None, None, None, None, None, None, 0.0, s)
        act = R.Act(out.t.view(B, H, W, Cout))

        def bwd():
            R._conv_backward(ctx, x, w, None, 3, 1, Cout, act.grad, Cout, 0, None, None)
        ctx.push(bwd)
        return act

    @staticmethod
    def _run_stage(ctx, stage, xs):
        mods = list(stage)
        for k, mod in enumerate(mods):
            if isinstance(mod, HRFomerModule) and mod.num_branches == 1 and k + 1 < len(mods) and \
                    isinstance(mods[k + 1], HRFomerModule) and mods[k + 1].num_branches == 1:
                xs = mod.run(ctx, xs, defer_out=True)       # single-branch stage (LidarStageB / C): the chain continues
            else:
                xs = mod.run(ctx, xs)
        return R.force_all(ctx, xs)                     # the last module's exchange sums (pending: LazyFuse), on sibling lanes

    def _run(self, ctx, srcs):
        M = self.num_fused_modalities
        lanes = ctx.bundle_lanes(1 + M, 'stems')     # camera stem + modality stems: equal shapes, one lane, merged launches
        mods = [None] * M
        cam = [None]

        def cam_stem():
            cam[0] = self._stem(ctx, srcs[0], self.conv1, getattr(self, self._stem_nn[0]), self.conv2, getattr(self, self._stem_nn[1]), self.layer1)

        def mod_stem(k):
            mods[k] = self._stem(ctx, srcs[1 + k], self.conv_a[k], self.norm_a[k], self.conv_b[k], self.norm_b[k],
                                 self.layer_a[k])
        ctx.parallel(lanes, [cam_stem] + [lambda k=k: mod_stem(k) for k in range(M)])
        ctx.join(lanes)
        ctx.mark('stems')                                # stem_cam + layer1 + stem_mod + layer_a
        x = cam[0]
        xs, m0 = self._fuse_stage(ctx, x, self.transition1, self.transition_a, self.fusion_a,
                                  self.stage2_cfg['num_branches'], mods, True, 'a')
        ys, mods = self._stages(ctx, self.stage2, xs, self.stage_b, m0)
        ctx.mark('stage2+stage_b')
        xs, m0 = self._fuse_stage(ctx, ys, self.transition2, self.transition_b, self.fusion_b,
                                  self.stage3_cfg['num_branches'], mods, False, 'b')
        ys, mods = self._stages(ctx, self.stage3, xs, self.stage_c, m0)
        ctx.mark('stage3+stage_c')
        xs, m0 = self._fuse_stage(ctx, ys, self.transition3, self.transition_c, self.fusion_c,
                                  self.stage4_cfg['num_branches'], mods, False, 'c')
        if not self.pre_neck_fusion:
            outs = self._run_stage(ctx, self.stage4, xs)
            ctx.mark('stage4')
            return outs
        ys, mods = self._stages(ctx, self.stage4, xs, self.stage_d, m0)
        nb = self.stage4_cfg['num_branches']
        xs, _ = self._fuse_stage(ctx, ys, [None] * nb, self.transition_d, self.fusion_d, nb, mods, False, 'd')
        lanes = ctx.fork(nb)
        outs = [None] * nb
        for i in range(nb):
            with ctx.on(lanes[i]):
                outs[i] = self._relu(ctx, R.force(ctx, xs[i]))                                  # :622-623 y_list[i] = self.relu(x_list[i])
        ctx.join(lanes)
        return outs

    def _relu(self, ctx, x):
        """Stand-alone ReLU of a materialised map (only the pre-neck fusion needs one: everywhere else the
        activation is applied by the consumer on load)."""
        L, s = ctx.L, ctx.stream
        Bn, H, W, C = x.t.shape
        rows = Bn * H * W
        unit = self.__dict__.setdefault('_unit_vec', {})
        key = (C, x.t.device)
        if key not in unit:
            unit[key] = (torch.ones(C, device=x.t.device), torch.zeros(C, device=x.t.device))
        one, zero = unit[key]
        out = R.Act(R._new_like(x.t))
        L.hrf_affine_act_res(x.t, one, zero, None, None, None, None, None, H * W, R.ACT_RELU, 0, out.t, rows, C, None, 0.0, None, None, s)
        if ctx.probe is not None:
            ctx.probe.append(('out', out.t))

        def bwd():
            if out.grad is None:
                return
            g = R._new_like(out.t)
            ctx.L.hrf_act_bwd(out.grad, out.t, x.t, None, None, None, 1, 0, g, None, None, None, None, None, rows, C,
                              ctx.stream)
            x.add_grad(g)
        ctx.push(bwd)
        return out

    def _stages(self, ctx, cam_stage, xs, mod_stages, m0):
        """The camera stage and the M single-branch modality stages are independent.  The modality
        stages are enqueued on their own lanes first; the camera stage then runs from the main lane,
        each of its modules forking branch lanes from main (flat, never nested)."""
        M = self.num_fused_modalities
        cap = max(1, 4 - len(xs)) if ctx.mod_lanes == 'auto' else int(ctx.mod_lanes or 0)
        lanes = ctx.bundle_lanes(M, 'stages', cap=cap, anchor=_STAGE_ANCHOR)
        mods = [None] * M
        ys = [None]

        def pad(us):
            # critical-lane probe (HRF_DEBUG_PAD="-1:300" pads every modality stage, "-2:300" every camera stage by 300 us of
            # idle time, forward and backward): a strand whose padding shows up in the step time is not hidden behind the other
            if us:
                ticks = int(us * 100)
                ctx.L.hrf_debug_spin(ticks, ctx.stream)
                ctx.push(lambda: ctx.L.hrf_debug_spin(ticks, ctx.stream))

        def mod_stage(k):
            pad(_debug_pad().get(-1))
            mods[k] = self._run_stage(ctx, mod_stages[k], [m0[k]])[0]

        def camera():
            pad(_debug_pad().get(-2))
            ctx.branch_lane_cap = _CAM_LANES          # streams for the camera stage's branches while M modality stages run beside it
            try:
                ys[0] = self._run_stage(ctx, cam_stage, xs)
            finally:
                ctx.branch_lane_cap = 0
        if ctx.cam_first:          # (the camera stage is the long chain: its nodes first in issue / graph order)
            ctx.parallel([ctx.cur] + list(lanes), [camera] + [lambda k=k: mod_stage(k) for k in range(M)])
        else:
            ctx.parallel(list(lanes) + [ctx.cur], [lambda k=k: mod_stage(k) for k in range(M)] + [camera])
        ctx.join(lanes, anchor=_STAGE_ANCHOR)
        return ys[0], mods


@BACKBONES.register_module()
class HRFuserHRNetBased(HRFuserHRFormerBased):
    """Drop-in for mmdet's `HRFuserHRNetBased` (hrfuser_hrnet_based.py:23-315): the same fusion dataflow (`forward` :214-315
    is the HRFormer-based one line for line, including the transition1[i][0] quirk) over a purely CONVOLUTIONAL HRNet trunk:
    stages 2-4 and the modality stages B-D are HRModules of `BASIC` blocks (hrnet.py:512-550).  Same registry name,
    constructor keywords, state-dict keys; no reference config uses it (the golden config is tests/golden/
    hrfuser_hrnet_cfg.json, oracle bit-exact vs the reference class: oracle/tools/make_golden_hrnet_based.py)."""
    blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck, 'CA': HRFuserFusionBlock, 'MWCA': HRFuserFusionBlock}
    _hrformer_trunk = False

    def _make_stage(self, cfg, in_channels, multiscale_output=True):
        block = self.blocks_dict[cfg['block']]
        if block is not BasicBlock:
            raise NotImplementedError('stages 2-4 / LidarStage B-D of HRFuserHRNetBased must use block BASIC')
        n = cfg['num_modules']
        mods = []
        for i in range(n):
            ms = multiscale_output or i != n - 1
            mods.append(HRModule(cfg['num_branches'], block, cfg['num_blocks'], in_channels, cfg['num_channels'], ms,
                                 norm_cfg=self.norm_cfg))
            in_channels = mods[-1].in_channels
        return nn.Sequential(*mods), in_channels


@BACKBONES.register_module()
class HRFormer(HipModule):
    """Plain camera-only HRFormer on the same kernels: drop-in for mmdet's `HRFormer`
    (hrformer.py:565-740 over HRNet, hrnet.py:211-596): registry name, constructor keywords, `forward(x) -> list`,
    state-dict keys.  It is the camera stream of HRFuserHRFormerBased without the fusion blocks, with three
    differences taken from the reference: block key 'HRFORMERBLOCK', the stochastic-depth schedule IS applied
    (hrformer.py:666-678) and `transition1[i]` runs whole (hrnet.py:563-566)."""

    blocks_dict = {'BOTTLENECK': Bottleneck, 'HRFORMERBLOCK': HRFormerBlock}
    _graphable = True
    _make_stage = HRFuserHRFormerBased._make_stage
    init_weights = HRFuserHRFormerBased.init_weights
    refresh_inputs = HRFuserHRFormerBased.refresh_inputs
    _wrap_inputs = HRFuserHRFormerBased._wrap_inputs
    _stem = HRFuserHRFormerBased._stem
    _run_stage = staticmethod(HRFuserHRFormerBased._run_stage)

    def __init__(self, extra, in_channels=3, conv_cfg=None, norm_cfg=dict(type='BN', requires_grad=True),
                 transformer_norm_cfg=dict(type='LN', eps=1e-6), norm_eval=False, with_cp=False,
                 multiscale_output=True, drop_path_rate=0., zero_init_residual=False, pretrained=None, init_cfg=None):
        super().__init__()
        assert 'stage1' in extra and 'stage2' in extra and 'stage3' in extra and 'stage4' in extra     # hrnet.py:296
        for i in range(4):
            cfg = extra[f'stage{i + 1}']
            assert len(cfg['num_blocks']) == cfg['num_branches'] and len(cfg['num_channels']) == cfg['num_branches']
        if conv_cfg is not None:
            raise NotImplementedError('conv_cfg must be None (plain Conv2d), as in every reference config')
        # stochastic depth (hrformer.py:666-678): linspace over the blocks of stages 2-4; mutates `extra` like the reference
        depths = [extra[s]['num_blocks'][0] * extra[s]['num_modules'] for s in ('stage2', 'stage3', 'stage4')]
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        extra['stage2']['drop_path_rates'] = dpr[0:depths[0]]
        extra['stage3']['drop_path_rates'] = dpr[depths[0]:depths[0] + depths[1]]
        extra['stage4']['drop_path_rates'] = dpr[depths[0] + depths[1]:]
        self.extra = extra
        self.pretrained = pretrained
        self.init_cfg = resolve_init_cfg(pretrained, init_cfg)
        self.zero_init_residual = zero_init_residual          # dead in the reference (see HRFuserHRFormerBased)
        self.norm_cfg, self.transformer_norm_cfg = norm_cfg, transformer_norm_cfg
        self.norm_eval, self.with_cp = norm_eval, with_cp
        ncfg = norm_cfg
        self.conv1 = nn.Conv2d(in_channels, 64, 3, 2, 1, bias=False)                                   # hrnet.py:337-371
        self._stem_nn = [norm_name(ncfg, 1), norm_name(ncfg, 2)]      # 'bn1' / 'bn2', or 'gn1' / 'gn2' (hrnet.py:338-360)
        self.add_module(self._stem_nn[0], build_bn(ncfg, 64))
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.add_module(self._stem_nn[1], build_bn(ncfg, 64))
        self.stage1_cfg = extra['stage1']
        if self.blocks_dict[self.stage1_cfg['block']] is not Bottleneck:
            raise NotImplementedError('stage1 block must be BOTTLENECK')
        c1 = self.stage1_cfg['num_channels'][0]
        self.layer1 = _make_res_layer(64, c1, self.stage1_cfg['num_blocks'][0], ncfg)
        pre = [c1 * 4]
        for si in (2, 3, 4):
            cfg = extra[f'stage{si}']
            setattr(self, f'stage{si}_cfg', cfg)
            ch = [c * self.blocks_dict[cfg['block']].expansion for c in cfg['num_channels']]
            setattr(self, f'transition{si - 1}', _make_transition(pre, ch, ncfg))
            stage, pre = self._make_stage(cfg, ch, multiscale_output if si == 4 else True)
            setattr(self, f'stage{si}', stage)
        self.init_weights()

    def train(self, mode=True):
        """hrnet.py:588-596 (norm_eval keeps BN frozen).  Unlike the reference this returns self."""
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self

    def forward(self, x):
        return self._call_engine((x,))

    @staticmethod
    def _transition(ctx, trans, prev, first):
        """hrnet.py:562-582: a non-None transition takes the single stage-1 map (first) / the LAST branch of the
        previous stage; None keeps branch i."""
        nb = len(trans)
        xs = [None] * nb
        # Several transition convolutions may read the SAME map (transition1: every new branch reads the stage-1 output) and
        # back-propagate on different lanes: each consumer after the first gets a handle with a gradient slot of its own
        # (R.split_grad; the slots are added on this lane, behind the join) - two lanes writing one gradient buffer, one with
        # "overwrite" and one with "accumulate", is a race (seen as a 12 % error of every stage-1 gradient when the second
        # lane's launch happened to run first).
        srcs, seen = [None] * nb, set()
        for i in range(nb):
            if trans[i] is not None:
                src = prev if first else prev[-1]
                srcs[i] = src if id(src) not in seen else R.split_grad(ctx, src)
                seen.add(id(src))
        lanes = ctx.fork(nb)
        for i in range(nb):
            tr = trans[i]
            if tr is None:
                xs[i] = prev[i]
                continue
            with ctx.on(lanes[i]):
                xs[i] = _run_conv_chain(ctx, srcs[i], [tr] if isinstance(tr[0], nn.Conv2d) else list(tr))
        ctx.join(lanes)
        return xs

    def _run(self, ctx, srcs):
        x = self._stem(ctx, srcs[0], self.conv1, getattr(self, self._stem_nn[0]), self.conv2, getattr(self, self._stem_nn[1]), self.layer1)
        ys = self._run_stage(ctx, self.stage2, self._transition(ctx, self.transition1, x, True))
        ys = self._run_stage(ctx, self.stage3, self._transition(ctx, self.transition2, ys, False))
        return self._run_stage(ctx, self.stage4, self._transition(ctx, self.transition3, ys, False))
