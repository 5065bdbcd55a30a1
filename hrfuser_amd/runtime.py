"""Execution runtime of the HRFuser HIP path: NHWC activations, a reverse-mode tape, and the op
layer that turns the reference's ATen graph into launches of the C-ABI kernels.

Design (MI355X-first):
  * activations are NHWC fp32 end to end (= the reference's NLC), so every nchw<->nlc copy of the
    reference (~1000 `copy_` per forward) disappears;
  * BatchNorm / LayerNorm / activations are never materialised on their own: a conv writes its RAW
    output plus per-channel sums, consumers apply `act(scale*raw+shift)` while loading;
  * backward is an explicit tape of closures (no torch.autograd graph inside the backbone): every
    gradient buffer is written exactly once or accumulated in place by the kernels themselves;
  * parameter gradients are accumulated by the kernels straight into `param.grad` (views of one
    flat arena owned by the backbone) so the optimizer and the RCCL all-reduce see ONE buffer.

Nothing in this file computes on the CPU: every op is a launch on torch's current HIP stream.
"""
import ctypes
import os

import torch

try:                                     # coroutines for the lock-step SyncBN schedule (Ctx.parallel); optional
    import greenlet as _greenlet
except Exception:                        # pragma: no cover
    _greenlet = None

from . import _lib

# BatchNorm finalize ON LOAD: the consumer of a train-mode BatchNorm derives scale/shift in its own prologue from the
# producer's replicated moments (hrf_bn_fin_t), the data-gradient kernel of the producing convolution does the same
# for the backward coefficients (hrf_bn_bfin_t): no hrf_bn_finalize / hrf_bn_bwd_finalize launches on the chain.
# HRF_FIN_ONLOAD=0 restores the separate launches (A/B measurements; SyncBN always uses them: the moments are
# exchanged between producer and finalize).
_DW_FUSED_WG = os.environ.get('HRF_DW_FUSED_WG', '1') != '0'   # depthwise dW from the data-gradient pass
_FIN_ONLOAD = os.environ.get('HRF_FIN_ONLOAD', '1') != '0'
# SyncBN exchange of a one-lane batch on that lane instead of the main lane (no cross-stream edge).  OFF by default: collectives
# of one communicator must execute in the same order on every rank, and two lanes' collectives are unordered on the GPU -
# only the main lane serialises them.  (Safe, and ~3 % faster, in the forced one-rank measurement mode.)
_XHUB = os.environ.get('HRF_XHUB', '0') != '0'
# EXPERIMENT (opt-in, validated on one GPU only): one RCCL communicator PER LANE.  Collectives of different communicators
# may run concurrently, and every lane's collectives are ordered on its own stream on every rank - so a BatchNorm exchange
# needs no hop to the main lane at all (no batching either: the lanes' exchanges overlap instead of being merged).
_LANE_COMMS = os.environ.get('HRF_SYNC_LANE_COMMS', '0') == '1'


def _p2p_mode_id():
    """The rank's HRF_SYNC_P2P setting (hrfuser_amd/p2p.py): 0 off (collectives through the communicator), 1 on, 2 auto."""
    from . import p2p
    return {'off': 0, 'on': 1, 'auto': 2}[p2p.mode()]


def set_lane_comms(on):
    """Switch the SyncBN schedule of the NEXT forward passes of this process (bench.py's sync_ab child compares the two
    schedules in one process).  Every rank must make the same choice."""
    global _LANE_COMMS
    _LANE_COMMS = bool(on)


def lane_comms():
    return _LANE_COMMS


def sync_schedule_fingerprint():
    """What decides the ORDER and NUMBER of SyncBN collectives a rank issues: ranks that disagree on any of these would
    hang or reduce mismatched buffers."""
    lockstep = os.environ.get('HRF_LOCKSTEP', '1') != '0'
    batch = os.environ.get('HRF_SYNC_BATCH', '1') != '0'
    # the EFFECTIVE batching decision of Ctx.__init__ (sync_batch) rides along: HRF_LOCKSTEP=0 alone turns it off
    eff = 1 if (_greenlet is not None and lockstep and batch and not _LANE_COMMS) else 0
    return [1 if _greenlet is not None else 0, 1 if batch else 0, 1 if _XHUB else 0, 1 if _LANE_COMMS else 0,
            1 if _FIN_ONLOAD else 0, _lib.STAT_COPIES, 1 if lockstep else 0, eff, _p2p_mode_id()]


def check_sync_schedule(group, world):
    """Called when SyncBN is switched on (EngineOwner.set_sync_group): the batched exchange schedule needs `greenlet`
    (a missing module used to degrade silently to one collective per BatchNorm), and every rank must run the same
    schedule - the fingerprint is all-reduced (MIN and MAX) and a disagreement raises before the first exchange."""
    if _greenlet is None and os.environ.get('HRF_SYNC_BATCH', '1') != '0' and not _LANE_COMMS and _p2p_mode_id() != 1:
        raise _lib.HRFuserHipError(
            'SyncBN: the batched exchange schedule needs the `greenlet` module, which is not importable here.  Install it, '
            'or set HRF_SYNC_BATCH=0 on EVERY rank to run one collective per BatchNorm (about 3x the collectives).')
    if group is None or world <= 1:
        return
    import torch.distributed as dist
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    fp = torch.tensor(sync_schedule_fingerprint(), dtype=torch.int64, device=dev)
    lo, hi = fp.clone(), fp.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not bool((lo == hi).all()):
        raise _lib.HRFuserHipError(
            f'SyncBN: the ranks disagree on the exchange schedule (greenlet, HRF_SYNC_BATCH, HRF_XHUB, HRF_SYNC_LANE_COMMS, '
            f'HRF_FIN_ONLOAD, HRF_STAT_COPIES, HRF_LOCKSTEP, effective batching, HRF_SYNC_P2P): min {lo.tolist()} max {hi.tolist()}, this rank {fp.tolist()}')
FIN_MAXC = _lib.FIN_MAXC
LN_EPS = 1e-6            # eps of every transformer LayerNorm of the reference configs (transformer_norm_cfg)
_MAX_LANES = int(os.environ.get('HRF_MAX_LANES', '0'))
_FORK_ANCHOR = os.environ.get('HRF_FORK_ANCHOR', '0') == '1'


def force_collectives():
    """HRF_FORCE_COLLECTIVES=1: issue the collectives even in a 1-rank group (lets a single-GPU box exercise RCCL inside
    the lanes / hipGraph capture exactly as an 8-GPU run would).  Read when a forward starts, not at import."""
    return os.environ.get('HRF_FORCE_COLLECTIVES', '0') == '1'


TF_NONE, TF_AFFINE, TF_RELU, TF_GELU, TF_LN = 0, 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
_TF2ACT = {TF_NONE: ACT_NONE, TF_AFFINE: ACT_NONE, TF_RELU: ACT_RELU, TF_GELU: ACT_GELU}


def gpu_add_(dst, g):
    """dst += g through the library (one launch on the current lane)."""
    _lib.lib().hrf_scale_add(g, None, 1.0, None, 1, dst, None, dst, g.numel(), 1, _lib.stream_ptr())
    return dst


def gpu_clone(g):
    out = _new_like(g)
    _lib.lib().hrf_scale_add(g, None, 1.0, None, 1, None, None, out, g.numel(), 1, _lib.stream_ptr())
    return out


def gpu_zero_(t):
    _lib.lib().hrf_memset(t, 0, t.numel() * t.element_size(), _lib.stream_ptr())
    return t


class Act:
    """A materialised NHWC activation (B,H,W,C) with an explicit gradient slot."""
    __slots__ = ('t', 'grad', 'needs_grad', 'rowstat')

    def __init__(self, t, needs_grad=True):
        self.t = t
        self.grad = None
        self.needs_grad = needs_grad
        self.rowstat = None          # (eps, [rows, 2] (mean, rstd)) when the producer emitted LayerNorm statistics

    @property
    def shape(self):
        return self.t.shape

    def grad_target(self):
        """-> (tensor, accumulate flag) for a kernel that writes this activation's gradient."""
        if self.grad is None:
            self.grad = _new_like(self.t)
            return self.grad, 0
        return self.grad, 1

    def add_grad(self, g):
        """Accumulate a finished gradient tensor (aliasing it when it is the first one)."""
        if self.grad is None:
            self.grad = g
        else:
            gpu_add_(self.grad, g)


class BNState:
    """One BatchNorm application: raw conv output + the per-channel vectors around it."""
    __slots__ = ('bn', 'C', 'raw', 'count', 'scale', 'shift', 'mean', 'invstd', 'stats', 'gstats',
                 'coef', 'du', 'train', 'pending', 'lane', 'bx_done', 'packed', 'bpacked', 'count_ptr', 'fin_lane',
                 'gn_raw', 'gn_stat')          # GroupNorm (gn_forward): the raw convolution output and its (mean, rstd) table


class Lazy:
    """act(scale*raw + shift) of a conv output, applied by the consumer on load (never stored)."""
    __slots__ = ('st', 'mode')

    def __init__(self, st, mode):
        self.st = st
        self.mode = mode

    @property
    def raw(self):
        return self.st.raw

    @property
    def shape(self):
        return self.st.raw.shape


class LNIn:
    """LayerNorm(act) applied by the consumer on load: (x - mean[row]) * rstd[row] * gamma + beta."""
    __slots__ = ('act', 'rowstat', 'ln')

    def __init__(self, act, rowstat, ln):
        self.act, self.rowstat, self.ln = act, rowstat, ln

    @property
    def shape(self):
        return self.act.t.shape


class LazyTail:
    """res + rowscale * GELU(BN(raw)) - the CrossFFN tail of a transformer block (hrformer.py:371-372) - NOT materialised:
    the fused attention launch of the NEXT block of the chain forms the rows while it stages its window (attn_block(xq =
    LazyTail)), and its backward launch emits the tail's du and BatchNorm moments.  force() materialises it for any other
    consumer."""
    __slots__ = ('res', 'lazy', 'rowscale', 'out')

    def __init__(self, res, lazy, rowscale=None):
        self.res, self.lazy, self.rowscale, self.out = res, lazy, rowscale, None

    @property
    def shape(self):
        return self.res.t.shape

    def force(self, ctx):
        if self.out is None:                              # memoised like Pending: a second force must not push a second backward
            self.out = materialize(ctx, self.lazy, ACT_GELU, res=self.res, act_first=True, rowscale=self.rowscale)
        return self.out


class Pending:
    """An Act that is not launched yet: `fn(ctx) -> Act` runs when the CONSUMER forces it, on the consumer's lane.  The output
    branches of an HRModule exchange (LazyFuse) and the fusion blocks in front of a stage are handed over this way: the next
    module's branch i starts with them instead of the main lane running them between a join and the next fork."""
    __slots__ = ('dims', 'fn', 'out')

    def __init__(self, dims, fn):
        self.dims, self.fn, self.out = tuple(dims), fn, None

    @property
    def shape(self):
        return self.dims

    def force(self, ctx):
        if self.out is None:
            self.out = self.fn(ctx)
            self.fn = None
        return self.out


def LazyFuse(dims, terms):
    """One output branch of an HRModule exchange, ReLU(sum of terms) (hrnet.py:192-206), pending."""
    return Pending(dims, lambda ctx: fuse_sum(ctx, tuple(dims), terms))


def split_grad(ctx, act):
    """A second handle on `act` with a gradient slot of its own, for a consumer that back-propagates on ANOTHER lane while
    `act` itself collects gradients elsewhere (the modality stage and the fusion block of branch 0 read the same map and run
    concurrently): the two slots are added where this call sits in the tape - on the current lane, after both consumers."""
    alias = Act(act.t, act.needs_grad)
    alias.rowstat = act.rowstat

    def bwd():
        if alias.grad is not None and act.needs_grad:
            act.add_grad(alias.grad)
    ctx.push(bwd)
    return alias


def force(ctx, x):
    """-> an Act: materialises a LazyTail / launches a pending LazyFuse on the current lane, passes everything else through."""
    return x.force(ctx) if isinstance(x, (LazyTail, Pending)) else x


def force_all(ctx, xs):
    """Acts for a list of stage outputs; pending exchange sums run on sibling lanes (they are independent of each other)."""
    pend = [i for i, x in enumerate(xs) if isinstance(x, Pending) and x.out is None]
    xs = list(xs)
    if len(pend) > 1:
        lanes = ctx.fork(len(pend))

        def one(i):
            xs[i] = force(ctx, xs[i])
        ctx.parallel(lanes, [lambda i=i: one(i) for i in pend])
        ctx.join(lanes)
    return [force(ctx, x) for x in xs]


# the CrossFFN tail of a block formed on load by the next block's fused attention launch (HRF_TAIL_ONLOAD=0: materialised)
_TAIL_ONLOAD = os.environ.get('HRF_TAIL_ONLOAD', '1') != '0'


def tail_onload():
    return _TAIL_ONLOAD


class RawInput:
    """A network input in its native NCHW layout (read through element strides by the stem conv)."""
    __slots__ = ('t', 'grad', 'needs_grad', 'nhwc', 'cols')

    def __init__(self, t, needs_grad, nhwc=None):
        self.t, self.grad, self.needs_grad = t, None, needs_grad
        self.nhwc = nhwc             # optional channels-last copy made by the owner's pre_step()
        self.cols = None             # (B, Ho, Wo, 9 C) 3x3 patches as rows, formed by the stem's first convolution (conv_bn)

    @property
    def shape(self):
        B, C, H, W = self.t.shape
        return (B, H, W, C)


class StageStamps:
    """GPU timestamps at the stage boundaries of a pass (measurement aid; bench.py per-stage roofline).  `take` launches
    hrf_stamp on the current stream: inside a captured hipGraph every replay rewrites the same slots, so the durations of the
    REAL captured step can be read back after a replay - HIP events cannot be timed there.  100 MHz counter."""

    def __init__(self, device, cap=256):
        self.buf = torch.zeros(cap, dtype=torch.int64, device=device)
        self.marks = []

    def reset(self):
        self.marks = []

    def take(self, ctx, direction, name, stream=None):
        i = len(self.marks)
        if i >= self.buf.numel():
            return
        self.marks.append((direction, name))
        ctx.lib.hrf_stamp(self.buf.data_ptr() + 8 * i, ctx.stream if stream is None else stream)

    def read(self):
        """-> [(direction, name, microseconds since the first stamp)]"""
        t = self.buf[:len(self.marks)].cpu().tolist()
        return [(d, n, (v - t[0]) / 100.0) for (d, n), v in zip(self.marks, t)]


class Lane:
    """An execution lane = one HIP stream.  Independent parts of the network (camera branches, per-source exchange
    chains, the deferred weight-gradient phase) run on sibling lanes so that the many small, latency-bound launches of
    HRFuser-T overlap on the 256 CUs instead of queueing serially.  Parts of EQUAL shape (the camera stream's finest
    branch and the modality streams) share a lane instead and merge their launches (Ctx.parallel / Strand)."""
    __slots__ = ('stream', 'ptr')

    def __init__(self, stream):
        self.stream = stream
        self.ptr = stream.cuda_stream if stream is not None else 0


class Strand:
    """One independent chain of the dataflow graph: a body handed to Ctx.parallel, its lane, and its own reverse tape.
    Sibling strands run as coroutines in LOCK-STEP at the granularity of library calls: every strand runs up to its next
    C-ABI call and parks; the sweep then issues the parked calls - those of equal entry point on the same lane between
    hrf_group_begin / hrf_group_end, which merges them into ONE multi-problem launch (the three sensor streams at equal
    depth) - and resumes everybody.  BatchNorm exchanges (SyncBN) park the same way and are flushed as one packed
    collective when no strand can run.  The backward pass runs the recorded strand tree in reverse through the same
    machinery, so the data-gradient launches of sibling strands merge too."""
    __slots__ = ('lane', 'tape', 'glet', 'pending', 'wait', 'parent', 'kids_alive', 'cur')

    def __init__(self, lane, parent=None):
        self.lane, self.parent = lane, parent
        self.tape = []
        self.glet = None
        self.pending = None           # (name, fn, args) of the parked library call
        self.wait = None              # None: runnable | 'call' | 'sync' | 'kids'
        self.kids_alive = 0
        self.cur = lane               # lane in effect when the strand parked (restored on resume)


# entry points whose launches may be merged with an equal launch of a sibling strand (csrc/hrf_group.h: the kernels take
# their arguments as an array of up to HRF_GROUP_MAX problems, blockIdx.z selects the problem)
_GROUPABLE = frozenset((
    'hrf_conv_fwd', 'hrf_conv_bwd_data', 'hrf_conv_fwd_packed', 'hrf_conv_bwd_data_packed', 'hrf_dwconv_fwd', 'hrf_dwconv_bwd_data',
    'hrf_dwconv_bwd_data_weight',
    'hrf_attn_block_fwd', 'hrf_attn_block_bwd', 'hrf_affine_act_res', 'hrf_act_bwd', 'hrf_scale_add', 'hrf_ln_stats',
    'hrf_ln_bwd', 'hrf_window_attn_fwd', 'hrf_window_attn_bwd', 'hrf_fuse_sum', 'hrf_bilinear_up_bwd'))
_NO_PARK = frozenset(_lib._RAW_RETURN) | frozenset((
    'hrf_wgrad_group_begin', 'hrf_wgrad_group_end', 'hrf_group_begin', 'hrf_group_end', 'hrf_debug_knob'))


class _LibProxy:
    """What the ops see as `ctx.L`: inside a lock-step sweep a library call parks its strand (the sweep issues it, possibly
    merged with its siblings' calls); everywhere else it goes straight to the library."""

    def __init__(self, ctx, lib):
        self.__dict__['_ctx'] = ctx
        self.__dict__['_lib'] = lib

    def __getattr__(self, name):
        fn = getattr(self.__dict__['_lib'], name)
        if name in _NO_PARK:
            self.__dict__[name] = fn
            return fn
        ctx = self.__dict__['_ctx']

        def call(*a):
            s = ctx._parkable()
            if s is None:
                return fn(*a)
            s.pending = (name, fn, a)
            s.wait = 'call'
            ctx._yield(s)
        self.__dict__[name] = call
        return call


class gc_paused:
    """Pause Python's cyclic garbage collector for the duration of a hipGraph capture.  The collector may run at any allocation
    (an Event, a list) and destroy whatever unreachable cycle it finds - e.g. the captured graphs of a module that went out of
    scope a moment ago: hipGraphExecDestroy inside ANOTHER stream capture fails, the destructor throws, the process aborts
    (seen as `Fatal Python error: Aborted ... Garbage-collecting` in a wait_stream of a capturing forward).  torch.cuda.graph
    collects once BEFORE the capture starts; this keeps the collector out until it has ended."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()
        return self

    def __exit__(self, *exc):
        if self.was:
            import gc
            gc.enable()
        return False


class Ctx:
    """Per-forward execution context: library handle, current lane/stream, mode, strand tree with the reverse tapes."""

    def __init__(self, owner, training, record):
        self.lib = owner._lib_handle()
        self.L = _LibProxy(self, self.lib)
        self.training = training
        self.record = record            # build the backward tape?
        self.owner = owner
        self.group = owner.sync_group if training else None
        self.world = owner.sync_world if (training and owner.sync_group is not None) else 1
        self.multi = bool(torch.cuda.is_available() and getattr(owner, 'use_lanes', True)
                          and os.environ.get('HRF_LANES', '1') != '0')
        self.main = Lane(torch.cuda.current_stream() if torch.cuda.is_available() else None)
        self.cur = self.main
        self.stream = self.main.ptr
        self.root = Strand(self.main)
        self.strand = self.root
        self._free = list(owner._lane_pool()) if self.multi else []
        self._side_i = 0
        self._side_used = {}
        self._deferred = []
        # test instrumentation (tests/helpers.py: ReLU-mask pinning): a list that receives every ReLU site of the forward
        self.probe = [] if owner.__dict__.get('_relu_probe') else None
        self.n_collectives = 0
        self.xhist = {}                 # (lanes in an exchange, hub is the main lane) -> count
        self.coll = self.group is not None and (self.world > 1 or force_collectives())
        # peer-to-peer SyncBN exchange (HRF_SYNC_P2P=1, hrfuser_amd/p2p.py): no communicator, hence no order to keep between the
        # lanes - every BatchNorm exchanges on its own lane, at once, with ONE launch
        self.p2p = owner._engine().p2p_context(self.group, self.world) if self.coll else None
        self.n_p2p = 0
        # gradient exchange of the step (hrfuser_amd.trainer): (buckets [(a, b) slices of the flat gradient arena], fn(a, b) =
        # all-reduce of one slice, rounds) - set by the trainer before run_backward, which then issues the weight-gradient leaves
        # bucket by bucket and every bucket's all-reduce as soon as its slice of the arena is final
        self.exchange = None
        self.n_grad_collectives = 0
        self.pending = []               # forward BatchNorm exchanges parked by the strands (SyncBN)
        self.bpending = []              # backward ones: (BNState, lane)
        # lock-step strands need coroutines; HRF_LOCKSTEP=0 (or no greenlet) runs the bodies one after the other: no merged
        # launches, and one SyncBN collective per BatchNorm
        glet = _greenlet if os.environ.get('HRF_LOCKSTEP', '1') != '0' else None
        self.sync_batch = self.coll and glet is not None and os.environ.get('HRF_SYNC_BATCH', '1') != '0' and not _LANE_COMMS \
            and self.p2p is None
        # multi-problem launches need a library compiled for them (hrf_group_count(3) = problems per launch; the shipped
        # build has 1: merged launches measured slower than a stream per sensor on MI355X)
        self.merge = glet is not None and os.environ.get('HRF_GROUP', '1') != '0' and hasattr(self.lib, 'hrf_group_begin') \
            and self.lib.hrf_group_count(3) > 1
        self._glet = glet if (self.merge or self.sync_batch or os.environ.get('HRF_LOCKSTEP') == '1') else None
        # equal-shape siblings share the current lane (their launches merge) instead of getting a stream each
        self.bundle = self.merge and os.environ.get('HRF_BUNDLE', '1') != '0'
        # which sibling sets are bundled (A/B knob): stems, trans(itions), stages; keep_first: the finest camera branch
        # stays on the current lane so that it merges with the modality stages bundled there
        self.bundle_what = set(os.environ.get('HRF_BUNDLE_WHAT', 'stems,trans,stages').split(',')) if self.bundle else set()
        # keep_first: the first sibling of a fork stays on the current lane (needed for merging with the bundled modality
        # stages; on its own a lane-count measure: main + nb - 1 streams per HRModule instead of nb, see mod_lanes)
        kf = os.environ.get('HRF_KEEP_FIRST')
        self.keep_first = (self.bundle and 'stages' in self.bundle_what) if kf is None else kf != '0'
        self.cam_first = os.environ.get('HRF_CAM_FIRST', '0') != '0'     # issue the camera stage before the modality stages
        # The hipGraph executor of ROCm 7.2 runs a captured graph on 4 streams (DEBUG_HIP_FORCE_GRAPH_QUEUES; its node -> stream
        # assignment is printed by DEBUG_HIP_GRAPH_DOT_PRINT: first child inherits, siblings round-robin): more than 4 concurrent
        # chains alias two of them onto one in-order stream.  Beside a camera stage of nb branches there is room for 4 - nb
        # modality chains: mod_lanes = lanes for the M modality stages ('auto': max(1, 4 - nb); HRFuser-T 14.35 -> 13.84 ms, STF
        # 28.1 -> 27.8 ms same-box A/B; N: forced, 0: one per modality).
        self.mod_lanes = os.environ.get('HRF_MOD_LANES', 'auto').strip().lower()
        self._sweeper = None            # greenlet running the sweep (None: not inside parallel())
        self._strands = []              # live strands of the running sweep, all levels
        self._xlane = None
        self.n_merged = [0, 0]          # (merged launches issued, calls they carried): reported by bench.py

    @property
    def tape(self):
        return self.root.tape

    def schedule_desc(self):
        """What this pass actually did (bench.py `config.sync_schedule`): not the environment's wish."""
        if not self.coll:
            return None
        if self.p2p is not None:
            return 'peer-to-peer exchange (IPC inbox per rank, one launch per BatchNorm on its own lane; HRF_SYNC_P2P=1)'
        if _LANE_COMMS:
            return 'one communicator per lane, unbatched exchanges (HRF_SYNC_LANE_COMMS=1)'
        if self.sync_batch:
            return f'packed exchanges on the main lane, lock-step strands (greenlet {getattr(_greenlet, "__version__", "?")})'
        return 'one exchange per BatchNorm on the main lane (no lock-step: HRF_SYNC_BATCH=0 / HRF_LOCKSTEP=0)'

    def mark(self, name):
        """End of stage `name` in the forward pass (root level, main lane).  With stage stamps enabled on the owner
        (HipModule.enable_stage_stamps) a GPU timestamp is taken here and again when the backward pass comes back to this
        point (= the start of that stage's backward)."""
        self.owner.__dict__['_stage_tag'] = name
        st = self.owner.__dict__.get('_stage_stamps')
        if st is None or self.strand is not self.root:
            return
        st.take(self, 'fwd', name)
        if self.record:
            self.root.tape.append((lambda: st.take(self, 'bwd', name), self.cur, None))

    # ---- tape ----------------------------------------------------------------------------------
    def push(self, fn, sync=None):
        """sync: BNState whose backward exchange (SyncBN) `fn` needs before it can run - run_backward batches those."""
        if self.record:
            self.strand.tape.append((fn, self.cur, sync))

    # ---- lock-step execution of sibling strands ---------------------------------------------------------------------
    def _resume(self, lane):
        self.cur, self.stream = lane, lane.ptr
        if self.multi and lane.stream is not None:
            torch.cuda.set_stream(lane.stream)

    def _parkable(self):
        """The strand of the running coroutine when its library calls are to be parked, else None."""
        if self._sweeper is None:
            return None
        g = self._glet.getcurrent()
        return None if g is self._sweeper else getattr(g, 'hrf_strand', None)

    def _yield(self, s):
        """Park strand `s` (the running coroutine) and hand control to the sweep; returns when the sweep resumes it."""
        s.cur = self.cur
        self._sweeper.switch()
        self.strand = s
        self._resume(s.cur)

    def parallel(self, lanes, bodies, kids=None):
        """Run bodies[i]() as a strand on lanes[i] (`kids`: the recorded strands whose tapes a backward pass re-runs).
        Without coroutines, or for a single body, one after the other."""
        me = self.strand
        home = self.cur
        fresh = kids is None
        if fresh:
            kids = [Strand(l, me) for l in lanes]
            if self.record:
                me.tape.append(('P', home, kids))
        G = self._glet
        if G is None or len(bodies) <= 1:
            for k, body in zip(kids, bodies):
                self.strand = k
                with _LaneScope(self, k.lane):
                    body()
            self.strand = me
            return kids

        def wrap(k, body):
            def run():
                self.strand = k
                with _LaneScope(self, k.lane):
                    body()
            g = G.greenlet(run, parent=self._sweeper if self._sweeper is not None else G.getcurrent())
            g.hrf_strand = k
            k.glet, k.parent, k.wait, k.pending = g, me, None, None
            return k
        top = self._sweeper is None
        if top:
            self._sweeper = G.getcurrent()
        for k, body in zip(kids, bodies):
            self._strands.append(wrap(k, body))
        me.kids_alive += len(kids)
        if top:
            try:
                self._sweep()
            finally:
                self._sweeper = None
                self._strands = []
            self.strand = me
            self._resume(home)
        else:
            me.wait = 'kids'
            while me.kids_alive > 0:
                self._yield(me)
            me.wait = None
        return kids

    def _sweep(self):
        """Round-robin over the live strands of all levels: each runs to its next library call / exchange / end; then the
        parked calls are issued (merged where siblings on one lane make the same call), and - when nothing can run - the
        parked BatchNorm exchanges are flushed as ONE collective."""
        strands = self._strands
        while strands:
            ran = False
            for s in list(strands):
                if s.wait is not None and not (s.wait == 'kids' and s.kids_alive == 0):
                    continue
                ran = True
                self.strand = s
                self._resume(s.cur)
                s.glet.switch()
                if s.glet.dead:
                    strands.remove(s)
                    if s.parent is not None:
                        s.parent.kids_alive -= 1
            issued = self._issue_round()
            if not ran and not issued:
                if self.pending:
                    self.flush_sync()
                elif self.bpending:
                    self._flush_bpending()
                else:
                    raise _lib.HRFuserHipError('lock-step scheduler: no strand can run and nothing is pending')
                for s in strands:
                    if s.wait == 'sync':
                        s.wait = None

    def _issue_round(self):
        pend = [s for s in self._strands if s.wait == 'call']
        if not pend:
            return False
        buckets = {}
        for s in pend:
            buckets.setdefault((s.cur.ptr, id(s.cur) if s.cur.stream is not None else 0, s.pending[0]), []).append(s)
        lib = self.lib
        for (ptr, _, name), ss in buckets.items():
            if len(ss) > 1 and self.merge and name in _GROUPABLE:
                if self.multi and ss[0].cur.stream is not None:
                    torch.cuda.set_stream(ss[0].cur.stream)
                lib.hrf_group_begin()
                try:
                    for s in ss:
                        s.pending[1](*s.pending[2])
                finally:
                    lib.hrf_group_end(ptr)
                self.n_merged[0] += 1
                self.n_merged[1] += len(ss)
            else:
                for s in ss:
                    if self.multi and s.cur.stream is not None:
                        torch.cuda.set_stream(s.cur.stream)
                    s.pending[1](*s.pending[2])
            for s in ss:
                s.pending, s.wait = None, None
        return True

    def sync_wait(self, st):
        """Forward SyncBN exchange of `st`: parked until the lock-step sweep flushes, or flushed at once outside one."""
        st.lane = self.cur
        self.pending.append(st)
        s = self._parkable() if self.sync_batch else None
        if s is not None:
            s.wait = 'sync'
            self._yield(s)
        else:
            self.flush_sync()

    def sync_wait_bwd(self, st, lane):
        """Backward counterpart: the exchange of (sum du, sum du*y) of `st`, needed by the tape entry about to run."""
        if _LANE_COMMS or self.p2p is not None:          # exchanged at once on the entry's own lane (communicator / inbox)
            self.flush_bwd([st], [lane])
            return
        self.bpending.append((st, lane))
        s = self._parkable() if self.sync_batch else None
        if s is not None:
            s.wait = 'sync'
            self._yield(s)
        else:
            self._flush_bpending()

    def _flush_bpending(self):
        items, self.bpending = self.bpending, []
        todo = [(st, lane) for st, lane in items if not st.bx_done]
        if todo:
            self.flush_bwd([st for st, _ in todo], [lane for _, lane in todo])

    def _exchange(self, sts, lanes, pack_ptrs, finalize, rows=False):
        """One packed collective for the BatchNorms `sts`, issued on the main lane between the lanes involved (every
        cross-stream edge costs ~10 us inside the captured graph, but the main lane is what keeps the collectives of the
        communicator in ONE order on every rank).  HRF_XHUB=1: a batch that lives on one lane is issued on that lane.
        Sibling-to-sibling waits (a non-main hub for several lanes) crash hipStreamEndCapture on ROCm 7.x like nested
        forks do."""
        lanes = [l for l in dict.fromkeys(lanes) if l.stream is not None or l is self.main]
        hub = self.main
        if (_XHUB or _LANE_COMMS or self.p2p is not None) and self.multi and len(lanes) == 1:
            hub = lanes[0]
        others = [l for l in lanes if l is not hub and l.stream is not None]
        if self.multi:
            for l in others:
                hub.stream.wait_stream(l.stream)
        with _LaneScope(self, hub):
            n = len(sts)
            total = sum(2 * st.C for st in sts)
            # behind the sums: this rank's sample count of every layer - the same all-reduce yields the GLOBAL counts, so
            # ranks with unequal batches normalise like torch.nn.SyncBatchNorm (which all-gathers the counts)
            packed = _keep(torch.empty(total + (n if rows else 0), device=sts[0].raw.device, dtype=torch.float64))
            if self.p2p is not None:
                # fold + push to every peer's inbox + wait + reduce in rank order: one launch, no collective
                self.p2p.exchange(sts, not rows, rows, packed, self.stream)
                self.n_p2p += 1
                finalize(packed, True)
            else:
                ptrs = (ctypes.c_void_p * n)(*pack_ptrs)
                cs = (ctypes.c_int * n)(*[st.C for st in sts])
                rw = (ctypes.c_double * n)(*[float(st.raw.numel() // st.C) for st in sts]) if rows else None
                self.lib.hrf_bn_pack(ptrs, cs, n, rw, packed, self.stream)
                self._xlane = hub
                finalize(packed, False)
                self._xlane = None
        if self.multi:
            for l in others:
                l.stream.wait_stream(hub.stream)
        self.xhist[(len(lanes), hub is self.main)] = self.xhist.get((len(lanes), hub is self.main), 0) + 1

    def flush_sync(self):
        """ONE collective for the forward moments of every parked BatchNorm; their consumers then finalise on load
        from the packed, all-reduced sums (hrf_bn_fin_t.copies = 1) - no finalize launch."""
        sts, self.pending = self.pending, []
        if not sts:
            return
        P = _lib._ptr
        box = []
        home, hstrand = self.cur, self.strand

        def fin(packed, exchanged):
            if not exchanged:
                self.all_reduce(packed)
            box.append(packed)
        self._exchange(sts, [st.lane for st in sts], [P(st.stats) for st in sts], fin, rows=True)
        off = 0
        tail = sum(2 * st.C for st in sts)
        for i, st in enumerate(sts):
            st.packed = (box[0], off)
            st.count_ptr = box[0].data_ptr() + 8 * (tail + i)          # the all-reduced sample count of this layer
            off += 2 * st.C
            st.pending = 'fwd' if _FIN_ONLOAD else None
            if not _FIN_ONLOAD:
                with _LaneScope(self, st.lane):
                    _finalize_now(self, st)
        self.strand = hstrand
        self._resume(home)

    def flush_bwd(self, sts, lanes):
        """Backward SyncBN exchange of (sum du, sum du*y) for all of `sts` in one collective; the data-gradient kernels
        of the producing convolutions derive their coefficients on load from the packed sums (bn_backward_coef)."""
        P = _lib._ptr
        box = []
        home, hstrand = self.cur, self.strand

        def fin(packed, exchanged):
            # no rank-local copy: dgamma / dbeta come from the all-reduced sums, scaled by 1/world (hrf_bn_bfin_t.pgrad_scale);
            # the gradient all-reduce that follows restores the sum over ranks
            if not exchanged:
                self.all_reduce(packed)
            box.append((packed, None))
        self._exchange(sts, lanes, [P(st.gstats) for st in sts], fin)
        off = 0
        for st in sts:
            st.bpacked = (box[0][0], box[0][1], off)
            off += 2 * st.C
            st.bx_done = True
        self.strand = hstrand
        self._resume(home)

    def all_reduce(self, t):
        if self.coll:
            import torch.distributed as dist
            group = self.group
            if _LANE_COMMS and self._xlane is not None and self._xlane is not self.main and self._xlane.stream is not None:
                group = self.owner._lane_group(self._xlane, self.group)
            dist.all_reduce(t, group=group)
            self.n_collectives += 1

    # ---- lanes ---------------------------------------------------------------------------------
    def fork(self, n, keep_first=False, cap=0, anchor=False):
        """n sibling lanes that start after everything enqueued so far on the current lane.  keep_first: the first
        sibling stays on the current lane (its launches can then merge with equal-shape strands bundled on that lane)."""
        # lanes are always direct children of the main lane: nested stream forks crash hipGraph
        # capture on ROCm 7.x, and a flat fork/join schedule expresses all the parallelism we need
        if n > 1 and not self.multi and self._glet is not None and self.cur is self.main:
            # no streams (CPU emulator runs): LOGICAL lanes, so that the SyncBN batching sees which tape entries are
            # independent of each other
            uniq = [Lane(None) for _ in range(n)]
            for l in uniq:
                l.ptr = self.cur.ptr        # (HRF_LANES=0 on a GPU: logical lanes launch on the parent's stream)
            if keep_first:
                uniq[0] = self.cur
            if self.record:
                self.strand.tape.append(('F', self.cur, [l for l in uniq if l is not self.cur]))
            return uniq
        if not self.multi or n <= 1 or self.cur is not self.main:
            return [self.cur] * n
        first = [self.cur] if keep_first else []
        n -= len(first)
        m = n if _MAX_LANES <= 0 else min(n, _MAX_LANES)      # HRF_MAX_LANES: fewer streams than siblings
        if cap > 0:
            m = min(m, cap)                                   # siblings share streams on purpose (run one after the other)
        while len(self._free) < m:
            self._free.append(self.owner._lane_pool(grow=True))
        uniq = [self._free.pop() for _ in range(m)]
        for k in uniq:
            k.stream.wait_stream(self.cur.stream)
        self._fork_anchor(self.cur, force=anchor)
        self._lane_stamp('fork', self.cur, uniq)
        if self.record and uniq:
            self.strand.tape.append(('F', self.cur, uniq))
        return first + [uniq[i % m] for i in range(n)]

    def _lane_stamp(self, what, parent, lanes):
        """Measurement aid (HipModule.enable_lane_stamps; tools/lane_stamps.py): a GPU timestamp on the parent lane and on
        every sibling lane at each fork (= when the lane really starts) and join (= when it is done) - the un-profiled truth
        about lane start offsets (rocprofv3's packet interception paces the queues itself)."""
        st = self.owner.__dict__.get('_lane_stamps')
        if st is None or not self.multi:
            return
        ids = st.__dict__.setdefault('lane_fork', {})      # lane -> (fork number, index): a join closes the fork that opened the lane
        if what == 'fork':
            st.nfork = k = st.__dict__.get('nfork', 0) + 1
            for i, l in enumerate(lanes):
                ids[id(l)] = (k, i)
        tag = self.owner.__dict__.get('_stage_tag', '')
        seen = set()
        for i, l in enumerate(lanes):
            if l.stream is None:
                continue
            k, idx = ids.get(id(l), (0, i))
            if k not in seen:
                seen.add(k)
                st.take(self, what, (k, 'parent', tag), parent.ptr)
            st.take(self, what, (k, idx, tag), l.ptr)

    def _fork_anchor(self, parent, force=False):
        """EXPERIMENT (HRF_FORK_ANCHOR=1): one trivial kernel on the PARENT lane right after a fork.  The ROCm graph executor
        gives the first child of a node the node's own stream and resolves a cross-stream dependency through a marker at the
        TAIL of the source stream at enqueue time - so without it the first sibling's whole chain sits between the fork
        point and the markers the other siblings wait for (they started 100-200 us late in the rocprofv3 timeline)."""
        if not (_FORK_ANCHOR or force) or not self.multi or parent.stream is None:
            return
        buf = self.owner.__dict__.get('_fork_scratch')
        if buf is None:
            buf = self.owner.__dict__['_fork_scratch'] = torch.zeros(8, dtype=torch.int64, device=torch.device('cuda', torch.cuda.current_device()))
        self.lib.hrf_stamp(buf.data_ptr(), parent.ptr)

    def bundle_lanes(self, n, what='stems', cap=0, anchor=False):
        """Lanes for n strands of EQUAL shape (camera stem + modality stems, the modality stages beside the camera stage):
        all on the current lane when launch merging is on - their equal calls become one multi-problem launch, in order on
        one queue, no cross-queue edges - otherwise a stream each."""
        if self.bundle and what in self.bundle_what:
            return [self.cur] * n
        return self.fork(n, cap=cap, anchor=anchor)

    def join(self, kids, anchor=False):
        """The current lane continues after all sibling lanes have finished.  `anchor`: the backward pass re-opens these lanes
        with a parent-lane anchor (see _fork_anchor)."""
        kids = [k for k in dict.fromkeys(kids) if k is not self.cur]     # lanes may repeat (HRF_MAX_LANES, bundles)
        if not kids:
            return
        if kids[0].stream is None:                                       # logical lanes
            if self.record:
                self.strand.tape.append(('J', self.cur, kids))
            return
        if not self.multi:
            return
        self._lane_stamp('join', self.cur, kids)
        for k in kids:
            self.cur.stream.wait_stream(k.stream)
        if self.record:
            self.strand.tape.append(('J', self.cur, kids, anchor))
        self._free.extend(kids)

    def on(self, lane):
        return _LaneScope(self, lane)

    def side_launch(self, fn, cost=1.0, key=None, target=None):
        """Run `fn` (one off-critical-path launch) on a side lane that starts after the current lane's
        work so far and is joined into the main lane at the end of the backward pass.  `target`: the parameter whose
        gradient the launch writes STRAIGHT into the flat arena (None: through the replicated accumulators, or unknown) -
        with a gradient exchange pending (Ctx.exchange) the leaves are issued bucket by bucket, see run_backward."""
        mode = os.environ.get('HRF_WGRAD', 'defer')
        bucketed = self.exchange is not None and mode == 'defer'
        if (not self.multi and not bucketed) or mode == 'inline':
            fn()
            return
        if mode in ('defer', 'flush') or bucketed:
            # weight-gradient launches are leaves of the backward graph: collect them and issue them
            # as one wide, fully parallel phase after the (serial, latency-bound) data-gradient chain
            self._deferred.append((float(cost), fn, key, target))
            return
        pool = self.owner._side_pool()
        lane = pool[self._side_i % len(pool)]
        self._side_i += 1
        lane.stream.wait_stream(self.cur.stream)
        self._side_used[id(lane)] = lane
        with _LaneScope(self, lane):
            fn()

    def _flush_deferred(self):
        """Issue the queued leaf launches on side lanes that start after the main lane's work so far (flat fork from
        main; joined at the end of run_backward)."""
        fns = [it[1] for it in self._deferred]
        self._deferred = []
        pool = self.owner._side_pool()
        k = min(len(pool), int(os.environ.get('HRF_WGRAD_FLUSH_LANES', os.environ.get('HRF_WGRAD_LANES', '4'))))
        group = os.environ.get('HRF_WGRAD_GROUP', '1') != '0'
        for j in range(k):
            lane = pool[(self._side_i + j) % len(pool)]
            lane.stream.wait_stream(self.main.stream)
            self._side_used[id(lane)] = lane
            with _LaneScope(self, lane):
                if group:
                    self.L.hrf_wgrad_group_begin()
                try:
                    for fn in fns[j::k]:
                        fn()
                finally:
                    if group:
                        self.L.hrf_wgrad_group_end(self.stream)
        self._side_i += k

    def _run_tape(self, strand):
        """Reverse pass over one strand's tape (runs inside that strand's coroutine, or at top level for the root)."""
        tape = strand.tape
        flush_n = self._flush_n if strand is self.root else 0
        while tape:
            e = tape.pop()
            if e[0] == 'J':                     # reverse of a join = fork
                for k in e[2]:
                    if self.multi and k.stream is not None:
                        k.stream.wait_stream(e[1].stream)
                if self.multi and e[2] and e[2][0].stream is not None:
                    self._fork_anchor(e[1], force=len(e) > 3 and e[3])
                    self._lane_stamp('fork', e[1], e[2])
            elif e[0] == 'F':                   # reverse of a fork = join
                if self.multi and e[2] and e[2][0].stream is not None:
                    self._lane_stamp('join', e[1], e[2])
                for k in e[2]:
                    if self.multi and k.stream is not None:
                        e[1].stream.wait_stream(k.stream)
                if flush_n and e[1] is self.main and len(self._deferred) >= flush_n:
                    self._flush_deferred()
            elif e[0] == 'P':                   # sibling strands: their tapes run in lock-step, like their forward bodies
                kids = e[2]
                self.parallel([k.lane for k in kids], [lambda k=k: self._run_tape(k) for k in kids], kids=kids)
            else:
                fn, lane, sync = e
                if sync is not None and self.coll and sync.train and not sync.bx_done:
                    self.sync_wait_bwd(sync, lane)
                with _LaneScope(self, lane):
                    fn()

    def run_backward(self):
        eng = self.owner._engine()
        if getattr(self, 'gen', None) is not None and self.gen != getattr(eng, 'gen', self.gen):
            raise _lib.HRFuserHipError(
                'backward of a forward pass that is no longer the latest one of this module: every forward re-uses the '
                'BatchNorm statistics slots and step buffers of its engine, so `net(x1); net(x2); loss.backward()` would '
                'back-propagate x1 with the statistics of x2.  Run backward before the next forward of the same module '
                '(or use two module instances).')
        use_keep_list(self.owner._engine().keep)
        self.owner._engine().fs_prepare()
        # HRF_WGRAD=flush: every time all lanes are joined into the main lane and enough weight gradients are queued,
        # issue them (grouped) on low-priority side lanes forked from main - they fill the CUs the latency-bound
        # data-gradient chain leaves idle instead of forming a phase of their own at the end
        self._flush_n = int(os.environ.get('HRF_WGRAD_FLUSH', '48')) if (self.multi and os.environ.get('HRF_WGRAD', 'defer') == 'flush') else 0
        entry = torch.cuda.current_stream() if self.multi else None      # (autograd calls this from its own thread / stream)
        if self.multi:
            self.main.stream.wait_stream(entry)
        # the strand tree of the forward pass, in reverse: 'P' entries re-run the sibling strands' tapes in lock-step
        # (merged data-gradient launches; SyncBN backward exchanges parked and flushed as ONE packed collective)
        self.strand = self.root
        self._run_tape(self.root)
        if self._deferred and os.environ.get('HRF_DEBUG_SKIP_WGRAD') == '1':
            self._deferred = []                 # timing experiments only: drops the weight-gradient phase
        eng0 = self.owner._engine()
        if eng0.rpb_pending():
            # every attn_block_bwd has left its dS planes: ONE gather launch for all relative-position-bias tables, a leaf of the
            # weight-gradient phase (52 launches of its own for HRFuser-T before)
            if self._deferred:
                self._deferred.append((eng0.fs_rpb_bytes, lambda: eng0.rpb_grad_now(self.L, self.stream), None, None))
            else:
                eng0.rpb_grad_now(self.L, self.stream)
        if self._deferred and eng0.fs_used and os.environ.get('HRF_FOLD_LEAF', '1') != '0':
            # the per-window slots of the fused attention blocks are complete (every attn_block_bwd has run) and their targets
            # in the gradient arena are touched by no other leaf: the fold (370 MB for HRFuser-T) joins the weight-gradient
            # phase as one more leaf instead of running alone on the main lane behind it
            self._deferred.append((4.0 * eng0.fs_bytes(), lambda: eng0.fold_slots_now(self.L, self.stream), None, None))
        comm = None
        if self._deferred:
            items, self._deferred = self._deferred, []
            k = int(os.environ.get('HRF_WGRAD_LANES', '4'))
            group = os.environ.get('HRF_WGRAD_GROUP', '1') != '0'
            self.strand = self.root
            for r, (its, ready) in enumerate(self._exchange_rounds(items)):
                lanes = self.fork(k)
                parts = _balance(its, k) if os.environ.get('HRF_WGRAD_BALANCE', '1') != '0' else \
                    [[it[1] for it in its[j::k]] for j in range(k)]
                for j in range(k):
                    with _LaneScope(self, lanes[j]):
                        # the dense weight gradients of a lane are queued and issued as a few grouped launches
                        # (up to 16 same-variant problems each, include/hrfuser_hip.h); everything else launches at once
                        if group:
                            self.L.hrf_wgrad_group_begin()
                        try:
                            for fn in parts[j]:
                                fn()
                        finally:
                            if group:
                                self.L.hrf_wgrad_group_end(self.stream)
                self.join(lanes)
                xs = self.owner.__dict__.get('_exchange_stamps')      # measurement aid (tools/exchange_overlap.py)
                if xs is not None:
                    xs.take(self, 'leaves_done', f'group {r}', stream=self.main.ptr)
                if self.exchange is not None:
                    if r == 0:
                        # HRF_WGRAD=flush left leaves on the side lanes (joined into main only behind this loop): the folds
                        # and the bucket all-reduces below read the arena and the replicated accumulators those kernels are
                        # still writing - join them FIRST (silently wrong gradients with world > 1 otherwise)
                        if self.multi and self._side_used:
                            for lane in self._side_used.values():
                                self.main.stream.wait_stream(lane.stream)
                            self._side_used = {}
                        # every leaf that goes through the replicated accumulators or the per-window slots ran in round 0:
                        # after these folds a slice of the arena is final as soon as its own dense leaves are done
                        with _LaneScope(self, self.main):
                            eng0.fold_grads(self.L, self.main.ptr)
                    comm = self._exchange_issue(ready, comm)
            self.root.tape.clear()              # (fork / join recorded markers: the pass is over)
        if self.multi:
            for lane in self._side_used.values():
                self.main.stream.wait_stream(lane.stream)
            self._side_used = {}
        with _LaneScope(self, self.main):
            self.owner._engine().fold_grads(self.L, self.main.ptr)
            st = self.owner.__dict__.get('_stage_stamps')
            if st is not None:
                st.take(self, 'bwd', 'weight_gradients')      # end of the deferred weight-gradient phase + folds
        if self.exchange is not None:
            if self.n_grad_collectives == 0:                  # nothing was deferred (frozen weights, HRF_WGRAD=inline / flush)
                comm = self._exchange_issue(list(self.exchange[0]), comm)
            if comm is not None and comm.stream is not None:
                self.main.stream.wait_stream(comm.stream)
                self._free.append(comm)
        if self.multi:
            torch.cuda.set_stream(entry)
            entry.wait_stream(self.main.stream)

    def _exchange_rounds(self, items):
        """-> [(leaf items, arena slices that are final after them)]: without a gradient exchange one round with everything.
        With one (Ctx.exchange = (buckets, fn, rounds)): the buckets are dealt to `rounds` consecutive groups; round 0 carries
        every leaf without a known arena target (depthwise / LayerNorm / relative-position-bias gradients through the
        replicated accumulators, the slot fold, the neck) plus the dense leaves of the first group, round r the dense leaves
        whose parameter lies in group r.  The reference overlaps its bucketed all-reduce with the backward through DDP's hooks
        (mmdet/apis/train.py:113-121, MMDistributedDataParallel); here the weight gradients are LEAVES deferred behind the
        data-gradient chain, so the overlap is inside that phase: all-reduce of group r beside the leaves of group r + 1."""
        if self.exchange is None:
            return [(items, [])]
        buckets, _, rounds = self.exchange
        rounds = max(1, min(int(rounds), len(buckets)))
        per = (len(buckets) + rounds - 1) // rounds
        groups = [buckets[g * per:(g + 1) * per] for g in range(rounds)]
        groups = [g for g in groups if g]
        offs = self.owner._engine()._poffs
        out = [([], list(g)) for g in groups]
        for it in items:
            tgt = it[3] if len(it) > 3 else None
            o = offs.get(id(tgt)) if tgt is not None else None
            r = 0
            if o is not None:
                for gi, g in enumerate(groups):
                    if g[0][0] <= o < g[-1][1]:
                        r = gi
                        break
            out[r][0].append(it)
        return out

    def _exchange_issue(self, ready, comm):
        """All-reduce the arena slices `ready` on the communication lane (a direct child of the main lane, like every lane),
        behind everything the main lane has joined so far."""
        if not ready:
            return comm
        fn = self.exchange[1]
        if self.multi:
            if comm is None:
                comm = self._free.pop() if self._free else self.owner._lane_pool(grow=True)
            comm.stream.wait_stream(self.main.stream)
            xs = self.owner.__dict__.get('_exchange_stamps')
            with _LaneScope(self, comm):
                for a, b in ready:
                    if xs is not None:
                        xs.take(self, 'allreduce_begin', f'[{a}:{b})')
                    fn(a, b)
                    if xs is not None:
                        xs.take(self, 'allreduce_end', f'[{a}:{b})')
                    self.n_grad_collectives += 1
        else:
            for a, b in ready:
                fn(a, b)
                self.n_grad_collectives += 1
        return comm


def _balance(items, k, chunk=None):
    """Longest-processing-time-first split of (cost, fn[, key]) leaf launches over k lanes (the lanes of the deferred
    weight-gradient phase run concurrently; the phase ends with the slowest one).  Items with the same `key` map to the same
    kernel variant of the grouped weight-gradient launches (hrf_wgrad_group_end issues up to 16 of them as ONE launch, per
    lane): they travel in chunks of `chunk` (8: 12.73 ms; 16: 12.76; 1 = every item on its own: 12.81, same box) so that a variant is not
    scattered over all lanes as 4 small launches."""
    if chunk is None:
        chunk = int(os.environ.get('HRF_WGRAD_CHUNK', '8'))
    units, byk = [], {}
    for it in items:
        key = it[2] if len(it) > 2 else None
        if key is None or chunk <= 1:
            units.append((it[0], [it[1]]))
        else:
            byk.setdefault(key, []).append(it)
    for grp in byk.values():
        grp.sort(key=lambda t: -t[0])
        for i in range(0, len(grp), chunk):
            part = grp[i:i + chunk]
            units.append((sum(t[0] for t in part), [t[1] for t in part]))
    loads = [0.0] * k
    parts = [[] for _ in range(k)]
    for cost, fns in sorted(units, key=lambda t: -t[0]):
        j = loads.index(min(loads))
        loads[j] += cost
        parts[j].extend(fns)
    return parts


class _LaneScope:
    """`with ctx.on(lane)`: launches (and torch ops) go to `lane` inside the block.  Explicit set_stream calls instead of
    torch.cuda.stream(): strands are suspended inside such blocks, and each resume re-establishes its own lane."""

    def __init__(self, ctx, lane):
        self.ctx, self.lane = ctx, lane

    def __enter__(self):
        c = self.ctx
        self.prev = c.cur
        c.cur, c.stream = self.lane, self.lane.ptr
        if c.multi and self.lane.stream is not None:
            torch.cuda.set_stream(self.lane.stream)
        return self.lane

    def __exit__(self, *exc):
        c = self.ctx
        c.cur, c.stream = self.prev, self.prev.ptr
        if c.multi and self.prev.stream is not None:
            torch.cuda.set_stream(self.prev.stream)
        return False


# Tensors created inside ops are kept alive until the next forward of the SAME engine begins: with
# several lanes a buffer may still be read on a sibling stream after its Python owner dropped it.
# Each engine (backbone, neck, a block harness) owns its list; the one in use is switched at the start
# of its forward / backward so that a neck step never drops the backbone's buffers.
_KEEP = []


def use_keep_list(lst):
    global _KEEP
    _KEEP = lst


def _keep(t):
    _KEEP.append(t)
    return t


def _new_like(t):
    return _keep(torch.empty_like(t))


def _new(shape, device):
    return _keep(torch.empty(shape, device=device, dtype=torch.float32))


def release_step_buffers():
    _KEEP.clear()


# ----------------------------------------------------------------------------- input descriptors
def _nhwc_strides(B, H, W, C):
    return (H * W * C, W * C, C, 1)


def _src_desc(src):
    """-> tensor, strides(sB,sY,sX,sC), (B,H,W,C), tf_mode, scale, shift, rowstat"""
    if isinstance(src, Act):
        B, H, W, C = src.t.shape
        return src.t, _nhwc_strides(B, H, W, C), (B, H, W, C), TF_NONE, None, None, None
    if isinstance(src, Lazy):
        B, H, W, C = src.raw.shape
        return src.raw, _nhwc_strides(B, H, W, C), (B, H, W, C), src.mode, src.st.scale, src.st.shift, None
    if isinstance(src, LNIn):
        B, H, W, C = src.act.t.shape
        return src.act.t, _nhwc_strides(B, H, W, C), (B, H, W, C), TF_LN, src.ln.weight, src.ln.bias, src.rowstat
    if isinstance(src, RawInput):
        B, C, H, W = src.t.shape
        sB, sC, sY, sX = src.t.stride()          # NCHW-contiguous or channels-last storage, read in place
        return src.t, (sB, sY, sX, sC), (B, H, W, C), TF_NONE, None, None, None
    raise TypeError(type(src))


def _needs_grad(src):
    if isinstance(src, (Act, RawInput)):
        return src.needs_grad
    if isinstance(src, LNIn):
        return src.act.needs_grad
    return True


# ----------------------------------------------------------------------------- BatchNorm plumbing
def _collectives(ctx):
    return ctx.coll


def bn_forward(ctx, bn, raw, stats):
    """Turn (raw conv output, accumulated sums) into a BNState.  Train mode without SyncBN: nothing is launched - the
    finalize is left to the consumer's prologue (st.pending, take_fin)."""
    st = BNState()
    C = raw.shape[-1]
    st.bn, st.C, st.raw, st.du, st.coef, st.pending, st.lane, st.bx_done = bn, C, raw, None, None, None, None, False
    st.packed = st.bpacked = None
    st.count_ptr = None
    st.fin_lane = None
    slot = ctx.owner._bn_slot(bn)
    st.train = bool(ctx.training and bn.training)
    if st.train:
        st.stats, st.gstats = slot['stats'], slot['gstats']
        st.scale, st.shift, st.mean, st.invstd = slot['scale'], slot['shift'], slot['mean'], slot['invstd']
        rows = raw.numel() // C
        st.count = float(rows * ctx.world)
        if _collectives(ctx):
            ctx.sync_wait(st)                           # batched with the exchanges of the sibling lanes (Ctx.parallel)
        elif _FIN_ONLOAD:
            st.pending = 'fwd'
        else:
            _finalize_now(ctx, st)
    else:
        st.stats, st.gstats = None, slot['gstats']
        st.scale, st.shift, st.mean, st.invstd = ctx.owner._bn_eval_affine(bn)
        st.count = float(raw.numel() // C)
    return st


def _finalize_now(ctx, st, update_running=True):
    bn = st.bn
    mom = bn.momentum if bn.momentum is not None else 0.1
    track = 1 if (bn.track_running_stats and update_running) else 0
    if st.packed is not None:                           # SyncBN: the folded, all-reduced sums
        P = _lib._ptr
        fin = _lib.BnFin(None, P(bn.weight), P(bn.bias), P(bn.running_mean), P(bn.running_var), P(st.scale), P(st.shift),
                         P(st.mean), P(st.invstd), st.count, float(bn.eps), float(mom), track, 1, st.C,
                         1, st.count_ptr)
        ctx.L.hrf_bn_finalize_packed(fin, 1, st.packed[0].data_ptr() + 8 * st.packed[1], ctx.stream)
    else:
        ctx.L.hrf_bn_finalize(st.stats, bn.weight, bn.bias, bn.running_mean, bn.running_var, st.count,
                              float(bn.eps), float(mom), track,
                              st.scale, st.shift, st.mean, st.invstd, st.C, ctx.stream)
    st.pending = None
    st.fin_lane = ctx.cur


def take_fin(ctx, st, limit=FIN_MAXC):
    """hrf_bn_fin_t for a launch that consumes BN(st.raw), or None when scale/shift are already in memory.  The first
    taker is the designated writer (scale/shift/mean/invstd, running statistics); a BatchNorm wider than `limit` is
    finalised by its own launch instead."""
    if st is None or st.pending is None:
        return None
    if st.C > limit:
        # a later consumer with a smaller `limit` than the designated writer's: scale / shift are in memory already
        # (ordered by the lane or the join between the two consumers) - finalising again would apply the running-statistics
        # momentum update twice in one step
        if st.pending == 'fwd':
            _finalize_now(ctx, st)
        elif st.fin_lane is not None and st.fin_lane is not ctx.cur:
            # ... but only a consumer that is stream-ordered behind the writer may rely on that: on a SIBLING lane (forked after
            # the producer, no join in between) the writer's kernel may not have run yet.  Finalise again on this lane without
            # touching the running statistics: both launches store identical scale / shift / mean / invstd
            _finalize_now(ctx, st, update_running=False)
            st.pending = 'written'
        else:
            st.pending = None
        return None
    bn = st.bn
    mom = bn.momentum if bn.momentum is not None else 0.1
    P = _lib._ptr
    stats, copies = (P(st.stats), 0) if st.packed is None else (st.packed[0].data_ptr() + 8 * st.packed[1], 1)
    fin = _lib.BnFin(stats, P(bn.weight), P(bn.bias), P(bn.running_mean), P(bn.running_var),
                     P(st.scale), P(st.shift), P(st.mean), P(st.invstd), st.count, float(bn.eps), float(mom),
                     1 if bn.track_running_stats else 0, 1 if st.pending == 'fwd' else 0, st.C, copies, st.count_ptr)
    st.pending = 'written'
    st.fin_lane = ctx.cur
    return fin


def _src_fin(ctx, src, limit=FIN_MAXC):
    return take_fin(ctx, src.st, limit) if isinstance(src, Lazy) else None


def bn_backward_coef(ctx, st, consumer_follows=True, limit=FIN_MAXC):
    """After a consumer wrote st.du / st.gstats: -> ((cA, cB, cC), bfin).  bfin (hrf_bn_bfin_t) is handed to the
    data-gradient launch of the producing convolution, which derives the coefficients on load, publishes them for
    the weight-gradient kernel and adds dgamma / dbeta; bfin is None when the separate finalize launch ran instead
    (SyncBN, no data-gradient launch to come, BatchNorm wider than `limit`)."""
    slot = ctx.owner._bn_slot(st.bn)
    cA, cB, cC = slot['cA'], slot['cB'], slot['cC']
    st.coef = (cA, cB, cC)
    wg = st.bn.weight.grad if st.bn.weight.requires_grad else None
    bg = st.bn.bias.grad if st.bn.bias.requires_grad else None
    coll = st.train and _collectives(ctx)
    if coll and not st.bx_done:
        # a tape entry that was pushed without `sync=`: exchange now, through the same packed route as everybody else, so that
        # EVERY SyncBN backward normalises by the all-reduced sample count (st.count_ptr) - the stand-alone
        # hrf_bn_bwd_finalize below only knows rows x world
        ctx.flush_bwd([st], [ctx.cur])
    if coll and st.bx_done:                             # exchanged by run_backward's batch (Ctx.flush_bwd): packed sums
        packed, local, off = st.bpacked
        P = _lib._ptr
        lptr = local.data_ptr() + 8 * off if local is not None else None
        bf = _lib.BnBFin(packed.data_ptr() + 8 * off, P(st.bn.weight), P(st.mean), P(st.invstd), P(wg), P(bg), P(cA), P(cB),
                         P(cC), st.count, 1, 1, st.C, 1, lptr, 1.0 / max(1, ctx.world), st.count_ptr)
        if _FIN_ONLOAD and consumer_follows and st.C <= limit:
            return st.coef, bf
        ctx.L.hrf_bn_bwd_finalize_packed(bf, 1, packed.data_ptr() + 8 * off, lptr, ctx.stream)
        return st.coef, None
    if _FIN_ONLOAD and consumer_follows and not coll and st.C <= limit:
        P = _lib._ptr
        return st.coef, _lib.BnBFin(P(st.gstats), P(st.bn.weight), P(st.mean), P(st.invstd), P(wg), P(bg), P(cA), P(cB),
                                    P(cC), st.count, 1 if st.train else 0, 1, st.C)
    local = None
    if coll:
        local = _keep(st.gstats.clone())
        ctx.all_reduce(st.gstats)
    ctx.L.hrf_bn_bwd_finalize(st.gstats, local, st.bn.weight, st.mean, st.invstd, st.count, 1 if st.train else 0,
                              wg, bg, cA, cB, cC, st.C, ctx.stream)
    return st.coef, None


# ----------------------------------------------------------------------------- conv / linear ops
def _conv_out_hw(H, W, KH, stride):
    pad = KH // 2
    return (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KH) // stride + 1


_S2F_MINPIX = int(os.environ.get('HRF_C3X_S2F_MINPIX', '4096'))   # A/B switch: stride-2 forward on the packed engine from this many output pixels
_IM2COL = os.environ.get('HRF_IM2COL', '1') != '0'      # A/B switch: the stem's first convolution through hrf_im2col3x3


def _packed(ctx, weight, direction, strides, B, H, W, Cin):
    """The tap-major pack of a front-end 3x3 convolution (csrc/conv3x_engine.hip; Engine.packed) when the activation is
    dense NHWC - None: the call goes through the OIHW entry point."""
    eng = ctx.owner._engine()
    wp = eng.packed(weight, direction) if hasattr(eng, 'packed') else None
    if wp is None or tuple(strides) != (H * W * Cin, W * Cin, Cin, 1):
        return None
    return wp


def _conv_backward(ctx, src, weight, bias, KH, stride, Cout, dy, ldD, doff, yraw, st=None):
    """Shared backward of every dense conv / linear: dX routed by the source kind, then dW (+db).
    `st`: BNState of the BatchNorm that follows the convolution (dy = st.du, BatchNorm backward applied on load)."""
    L, s = ctx.L, ctx.stream
    x, strides, (B, H, W, Cin), tf, sc, sh, rowstat = _src_desc(src)
    needs = _needs_grad(src)
    # (stride 2: one block walks the four parity classes of a SOURCE tile - worth it from ~64 tiles: 2 x 16 x 24 source pixels
    # measured 34 vs 26 us on the parity-class blocks of conv3_engine.hip)
    wpb = _packed(ctx, weight, 1, strides, B, H, W, Cin) if (KH == 3 and (stride == 1 or B * H * W >= 16384)) else None

    def bwd_data(*args):
        if wpb is not None:
            L.hrf_conv_bwd_data_packed(*args[:-1], wpb, args[-1])
        else:
            L.hrf_conv_bwd_data(*args)
    bfin = None
    cA = cB = cC = None
    if st is not None:
        (cA, cB, cC), bfin = bn_backward_coef(ctx, st, consumer_follows=needs)
    # the data gradient goes first: with `bfin` it is the launch that publishes cA/cB/cC for the weight gradient
    if not needs:
        pass
    elif isinstance(src, Lazy):
        ps = src.st
        ps.du = _new_like(ps.raw)
        bwd_data(dy, ldD, doff, yraw, cA, cB, cC, bfin, weight, KH, stride, Cout, B, H, W, Cin,
                            ps.du, *strides, 0, 1, ps.raw, Cin, ps.scale, ps.shift, _TF2ACT[src.mode],
                            ps.gstats, s)
    elif isinstance(src, LNIn):
        da = _new_like(src.act.t)
        bwd_data(dy, ldD, doff, yraw, cA, cB, cC, bfin, weight, KH, stride, Cout, B, H, W, Cin,
                            da, *strides, 0, 0, None, 0, None, None, 0, None, s)
        g, acc = src.act.grad_target()
        eng = ctx.owner._engine()
        gacc, cs = eng.grad_acc(src.ln.weight)
        bacc, _ = eng.grad_acc(src.ln.bias)
        L.hrf_ln_bwd(da, src.act.t, src.rowstat, src.ln.weight, B * H * W, Cin, g, acc, gacc, bacc, cs, s)
    elif isinstance(src, Act):
        g, acc = src.grad_target()
        bwd_data(dy, ldD, doff, yraw, cA, cB, cC, bfin, weight, KH, stride, Cout, B, H, W, Cin,
                            g, *strides, acc, 0, None, 0, None, None, 0, None, s)
    else:                                   # RawInput (NCHW gradient written through strides)
        if src.grad is None:
            src.grad = _new_like(src.t)
            acc = 0
        else:
            acc = 1
        bwd_data(dy, ldD, doff, yraw, cA, cB, cC, bfin, weight, KH, stride, Cout, B, H, W, Cin,
                            src.grad, *strides, acc, 0, None, 0, None, None, 0, None, s)
    if weight.requires_grad:
        # weight gradients are leaves of the backward graph: issue them on a side lane so they overlap
        # the latency-bound data-gradient chain (operands are never mutated afterwards)
        bgrad = bias.grad if (bias is not None and bias.requires_grad) else None
        xw, sw = x, strides
        if isinstance(src, RawInput) and KH == 3 and src.cols is not None:
            # the forward formed the 3x3 patches as rows (hrf_im2col3x3): grad_weight = dY^T . cols, the 1x1 form on
            # channel-contiguous rows - OIHW memory IS [Cout][9 Cin] (r6: 79 -> ~26 us per sensor stream)
            Ho, Wo = _conv_out_hw(H, W, KH, stride)
            K9 = 9 * Cin
            cols = src.cols
            cost = 4.0 * B * Ho * Wo * (K9 + Cout * (2 if cA is not None else 1))
            ctx.side_launch(lambda: L.hrf_conv_bwd_weight(
                dy, ldD, doff, yraw, cA, cB, cC, cols, *_nhwc_strides(B, Ho, Wo, K9), B, Ho, Wo, K9, 1, 1, Cout,
                TF_NONE, None, None, None, weight.grad, bgrad, ctx.stream), cost=cost,
                key=('conv_w', K9, Cout, 1, 1, TF_NONE, cA is not None), target=weight)
            return
        if isinstance(src, RawInput) and KH == 3:
            # NCHW network input: the pixel-major weight-gradient kernel wants channel-contiguous rows;
            # one 6 MB layout copy (torch, capturable) replaces the 207 us strided LDS kernel by ~40 us
            xw = src.nhwc if src.nhwc is not None else _keep(x.permute(0, 2, 3, 1).contiguous())
            sw = _nhwc_strides(B, H, W, Cin)
        Ho, Wo = _conv_out_hw(H, W, KH, stride)
        # byte-equivalent cost for the lane balancer: operand traffic + flops at ~10 FLOP/B
        cost = 4.0 * B * (H * W * Cin + Ho * Wo * Cout * (2 if cA is not None else 1)) + 0.2 * B * Ho * Wo * Cout * Cin * KH * KH
        # the large 3x3 problems (stems, transition1): LDS-staged kernel, per-split slabs in step-lifetime scratch + fold
        nsc = L.hrf_conv_bwd_weight_scratch(*sw, B, H, W, Cin, KH, stride, Cout, tf, 1 if bgrad is not None else 0) if KH == 3 else 0
        if nsc > 0:
            scratch = _new((nsc,), weight.device)
            ctx.side_launch(lambda: L.hrf_conv_bwd_weight_s(
                dy, ldD, doff, yraw, cA, cB, cC, xw, *sw, B, H, W, Cin, KH, stride, Cout,
                tf, sc, sh, rowstat, weight.grad, bgrad, scratch, ctx.stream), cost=cost,
                key=('conv_w3x', Cin, Cout, KH, stride, tf, cA is not None), target=weight)
            return
        ctx.side_launch(lambda: L.hrf_conv_bwd_weight(
            dy, ldD, doff, yraw, cA, cB, cC, xw, *sw, B, H, W, Cin, KH, stride, Cout,
            tf, sc, sh, rowstat, weight.grad, bgrad, ctx.stream), cost=cost,
            key=('conv_w', Cin, Cout, KH, stride, tf, cA is not None), target=weight)


# ----------------------------------------------------------------------------- GroupNorm (norm_cfg type 'GN')
def is_gn(norm):
    return isinstance(norm, torch.nn.GroupNorm)


_GN_CONST = {}


def _gn_unit_affine(C, device):
    """(ones, zeros) of C floats: the consumers of a GroupNorm output apply their activation through the on-load affine of
    the BatchNorm machinery with scale 1 / shift 0."""
    key = (str(device), C)
    if key not in _GN_CONST:
        _GN_CONST[key] = (torch.ones(C, device=device), torch.zeros(C, device=device))
    return _GN_CONST[key]


def gn_forward(ctx, gn, raw):
    """GroupNorm of a raw convolution output (hrf_gn_moments -> hrf_gn_apply) -> a BNState whose `raw` is the NORMALISED
    pre-activation value gamma*xhat + beta with a unit affine and frozen statistics: every consumer kernel (conv / dw /
    materialize / fuse_sum loaders and their backward epilogues) works on it unchanged, and the gradient they leave in
    st.du is the gradient with respect to the GroupNorm output (gn_backward)."""
    L, s = ctx.L, ctx.stream
    B, H, W, C = raw.shape
    G = gn.num_groups
    mom = _keep(torch.zeros(B * 2 * C, device=raw.device, dtype=torch.float64))
    L.hrf_gn_moments(raw, None, B, H * W, C, mom, s)
    y = _new_like(raw)
    stat = _new((B, G, 2), raw.device)
    L.hrf_gn_apply(raw, mom, gn.weight, gn.bias, float(gn.eps), B, H * W, C, G, y, stat, s)
    st = BNState()
    st.bn, st.C, st.raw, st.du, st.coef, st.pending, st.lane, st.bx_done = gn, C, y, None, None, None, None, True
    st.packed = st.bpacked = None
    st.count_ptr = None
    st.fin_lane = None
    st.train = False
    st.stats = None
    st.gstats = _keep(torch.zeros(_lib.STAT_COPIES * 2 * C, device=raw.device, dtype=torch.float64))   # consumers' moments: unused
    st.scale, st.shift = _gn_unit_affine(C, raw.device)
    st.mean = st.invstd = None
    st.count = float(B * H * W)
    st.gn_raw, st.gn_stat = raw, stat
    return st


def gn_backward(ctx, st):
    """-> gradient with respect to the raw convolution output (the parameter gradients are added in the same launch)."""
    L, s = ctx.L, ctx.stream
    gn, raw = st.bn, st.gn_raw
    B, H, W, C = raw.shape
    gmom = _keep(torch.zeros(B * 2 * C, device=raw.device, dtype=torch.float64))
    L.hrf_gn_moments(st.du, raw, B, H * W, C, gmom, s)
    draw = _new_like(raw)
    wg = gn.weight.grad if gn.weight.requires_grad else None
    bg = gn.bias.grad if gn.bias.requires_grad else None
    L.hrf_gn_bwd(st.du, raw, st.gn_stat, gmom, gn.weight, B, H * W, C, gn.num_groups, draw, wg, bg, s)
    return draw


def conv_bn(ctx, src, conv, bn, mode):
    """conv (k=1|3, dense) + BatchNorm (+ReLU/GELU) -> Lazy.  mode: TF_AFFINE / TF_RELU / TF_GELU."""
    L, s = ctx.L, ctx.stream
    x, strides, (B, H, W, Cin), tf, sc, sh, rowstat = _src_desc(src)
    w, b = conv.weight, conv.bias
    Cout, KH, stride = w.shape[0], w.shape[2], conv.stride[0]
    Ho, Wo = _conv_out_hw(H, W, KH, stride)
    y = _new((B, Ho, Wo, Cout), x.device)
    if is_gn(bn):
        L.hrf_conv_fwd(x, *strides, B, H, W, Cin, w, b, KH, stride, Cout, y, Cout, 0, None, None, 0,
                       tf, sc, sh, rowstat, None, _src_fin(ctx, src), None, 0.0, s)
        gst = gn_forward(ctx, bn, y)
        gout = Lazy(gst, mode)
        if ctx.probe is not None and mode == TF_RELU:
            ctx.probe.append(('lazy', gst))

        def gbwd():
            draw = gn_backward(ctx, gst)
            _conv_backward(ctx, src, w, b, KH, stride, Cout, draw, Cout, 0, None, None)
            gst.du = None
        ctx.push(gbwd)
        return gout
    slot = ctx.owner._bn_slot(bn)
    train = ctx.training and bn.training
    stats = slot['stats'] if train else None
    # deep contraction, few row blocks (the 256 -> 36 stride-2 transition): split over K, partial tiles in step-lifetime scratch
    # packed-weight engine: stride 1, and stride 2 (parity-plane halo) from _S2F_MINPIX output pixels and 32 input channels up
    # (the split-K engine keeps the deep, few-tile 256 -> 36 transition: 54.8 us against 76.8 us on 120 packed-engine blocks)
    nsc = L.hrf_conv_fwd_split_scratch(*strides, B, H, W, Cin, KH, stride, Cout, Cout, 0) if KH == 3 else 0
    wp = None
    if nsc == 0 and KH == 3 and tf != TF_LN and (stride == 1 or (B * Ho * Wo >= _S2F_MINPIX and Cin >= 32)):
        wp = _packed(ctx, w, 0, strides, B, H, W, Cin)
    if nsc > 0:
        L.hrf_conv_fwd_split(x, *strides, B, H, W, Cin, w, b, KH, stride, Cout, y, Cout, 0, None, None, 0,
                             tf, sc, sh, rowstat, stats, _src_fin(ctx, src), None, 0.0, _new((nsc,), x.device), s)
    elif isinstance(src, RawInput) and KH == 3 and 9 * Cin <= 32 and _IM2COL:
        # the stem's first convolution (3 -> 64, stride 2, NCHW network input): the 3x3 patches are formed ONCE as rows
        # [B Ho Wo][9 Cin] (column order = OIHW memory order) and the convolution is a row GEMM on them; the weight gradient
        # of the backward re-uses the same rows (27 strided gathers per output pixel in both kernels before)
        K9 = 9 * Cin
        cols = _new((B, Ho, Wo, K9), x.device)
        L.hrf_im2col3x3(x, *strides, B, H, W, Cin, stride, cols, K9, s)
        src.cols = cols
        L.hrf_conv_fwd(cols, *_nhwc_strides(B, Ho, Wo, K9), B, Ho, Wo, K9, w, b, 1, 1, Cout, y, Cout, 0, None, None, 0,
                       TF_NONE, None, None, None, stats, None, None, 0.0, s)
    elif wp is not None:
        L.hrf_conv_fwd_packed(x, *strides, B, H, W, Cin, w, b, KH, stride, Cout, y, Cout, 0, None, None, 0,
                              tf, sc, sh, rowstat, stats, _src_fin(ctx, src), None, 0.0, wp, s)
    else:
        L.hrf_conv_fwd(x, *strides, B, H, W, Cin, w, b, KH, stride, Cout, y, Cout, 0, None, None, 0,
                       tf, sc, sh, rowstat, stats, _src_fin(ctx, src), None, 0.0, s)
    st = bn_forward(ctx, bn, y, stats)
    out = Lazy(st, mode)
    if ctx.probe is not None and mode == TF_RELU:
        ctx.probe.append(('lazy', st))

    def bwd():
        _conv_backward(ctx, src, w, b, KH, stride, Cout, st.du, Cout, 0, st.raw, st)
        st.du = None
    ctx.push(bwd, sync=st)
    return out


class Plain:
    """Output of a Linear without BatchNorm: (rows, ld) buffer + gradient buffer of the same shape."""
    __slots__ = ('t', 'grad')

    def __init__(self, t):
        self.t, self.grad = t, None


def linear_into(ctx, src, lin, out, off):
    """out.t[:, off:off+Cout] = Linear(src).  `out` is a Plain shared by several projections."""
    L, s = ctx.L, ctx.stream
    x, strides, (B, H, W, Cin), tf, sc, sh, rowstat = _src_desc(src)
    w, b = lin.weight, lin.bias
    Cout = w.shape[0]
    ld = out.t.shape[-1]
    L.hrf_conv_fwd(x, *strides, B, H, W, Cin, w, b, 1, 1, Cout, out.t, ld, off, None, None, 0,
                   tf, sc, sh, rowstat, None, _src_fin(ctx, src), None, 0.0, s)

    def bwd():
        _conv_backward(ctx, src, w, b, 1, 1, Cout, out.grad, ld, off, None, None)
    ctx.push(bwd)


def linear_residual(ctx, o, lin, res, res2=None, drop=None):
    """x_new = res (+ res2) + scale*Linear(o).  `drop` = (mask|None, mscale, rowscale|None) in training
    for nn.Dropout / DropPath; otherwise the residual adds are fused into the GEMM epilogue."""
    L, s = ctx.L, ctx.stream
    B, H, W, C = res.t.shape
    w, b = lin.weight, lin.bias
    rows = B * H * W
    out = Act(_new_like(res.t))
    strides = _nhwc_strides(B, H, W, C)
    if drop is None:
        # the new residual stream is what the next LayerNorm reads: its row statistics come for free
        out.rowstat = (LN_EPS, _new((rows, 2), res.t.device))
        L.hrf_conv_fwd(o.t, *strides, B, H, W, C, w, b, 1, 1, C, out.t, C, 0, res.t,
                       res2.t if res2 is not None else None, C, TF_NONE, None, None, None, None, None,
                       out.rowstat[1], LN_EPS, s)
    else:
        mask, mscale, rowscale = drop
        y = _new_like(res.t)
        L.hrf_conv_fwd(o.t, *strides, B, H, W, C, w, b, 1, 1, C, y, C, 0, None, None, 0,
                       TF_NONE, None, None, None, None, None, None, 0.0, s)
        L.hrf_scale_add(y, mask, mscale, rowscale, H * W, res.t, res2.t if res2 is not None else None,
                        out.t, rows, C, s)

    def bwd():
        g = out.grad
        if drop is None:
            dy = g
        else:
            mask, mscale, rowscale = drop
            dy = _new_like(g)
            L.hrf_scale_add(g, mask, mscale, rowscale, H * W, None, None, dy, rows, C, s)
        if w.requires_grad:
            L.hrf_conv_bwd_weight(dy, C, 0, None, None, None, None, o.t, *strides, B, H, W, C, 1, 1, C,
                                  TF_NONE, None, None, None, w.grad, b.grad if b is not None else None, s)
        og, acc = o.grad_target()
        L.hrf_conv_bwd_data(dy, C, 0, None, None, None, None, None, w, 1, 1, C, B, H, W, C, og, *strides, acc,
                            0, None, 0, None, None, 0, None, s)
        # identity paths: the residual streams receive the output gradient unchanged
        if res2 is not None and res2.needs_grad:
            if res2.grad is None and res.grad is None and res.needs_grad:
                res2.grad = gpu_clone(g)
            else:
                res2.add_grad(g)
        if res.needs_grad:
            res.add_grad(g)
    ctx.push(bwd)
    return out


def ln_input(ctx, act, ln, cache=None):
    """LayerNorm as a transform-on-load: only the (mean, rstd) row statistics are computed."""
    B, H, W, C = act.t.shape
    key = (id(act), float(ln.eps))
    pre = getattr(act, 'rowstat', None)
    if pre is not None and abs(pre[0] - float(ln.eps)) < 1e-12:
        rowstat = pre[1]                       # emitted by the producing kernel's epilogue
    elif cache is not None and key in cache:
        rowstat = cache[key]
    else:
        rowstat = _new((B * H * W, 2), act.t.device)
        ctx.L.hrf_ln_stats(act.t, B * H * W, C, float(ln.eps), rowstat, ctx.stream)
        if cache is not None:
            cache[key] = rowstat
    return LNIn(act, rowstat, ln)


def window_attention(ctx, q, qoff, k, koff, v, voff, kpad, vpad, kbias, kboff, vbias, vboff, rpb, heads, dims):
    """softmax(q k^T d^-1/2 + RPB) v per 7x7 window and head.  q/k/v are Plain projection buffers.
    (kbias, kboff) / (vbias, vboff): bias parameter and element offset receiving the pad-key/value grads."""
    L, s = ctx.L, ctx.stream
    B, H, W, C = dims
    o = Act(_new((B, H, W, C), q.t.device))
    L.hrf_window_attn_fwd(q.t, q.t.shape[-1], qoff, k.t, k.t.shape[-1], koff, v.t, v.t.shape[-1], voff,
                          kpad, vpad, rpb, o.t, C, B, H, W, C, heads, s)

    def bwd():
        for p in (q, k, v):
            if p.grad is None:
                p.grad = _new_like(p.t)
        eng = ctx.owner._engine()
        kacc, cs = eng.grad_acc(kbias)
        vacc, cs2 = eng.grad_acc(vbias)
        racc, cs3 = eng.grad_acc(rpb)
        assert cs == cs2 == cs3
        L.hrf_window_attn_bwd(q.t, q.t.shape[-1], qoff, k.t, k.t.shape[-1], koff, v.t, v.t.shape[-1], voff,
                              kpad, vpad, rpb, o.grad, C,
                              q.grad, q.grad.shape[-1], qoff, k.grad, k.grad.shape[-1], koff,
                              v.grad, v.grad.shape[-1], voff, kacc.reshape(-1)[kboff:], vacc.reshape(-1)[vboff:],
                              racc, cs, B, H, W, C, heads, s)
    ctx.push(bwd)
    return o


# ----------------------------------------------------------------------------- eval-mode CrossFFN in one launch
_FFN_EVAL = os.environ.get('HRF_FFN_EVAL', '1') != '0'
# widest block that takes the one-launch route (the kernel is built for every width hrf_ffn_eval_supported names; measured on
# MI355X it beats dw + fc3 + the fc1 head of the attention launch on the finest branch only, where a launch has 480+ tiles)
_FFN_EVAL_MAXC = int(os.environ.get('HRF_FFN_EVAL_MAXC', '18'))


def ffn_eval_ok(ctx, C, ffn):
    """Can the CrossFFN of this block run as ONE launch (csrc/ffn_eval.hip)?  Only without a tape (an eval forward: nothing
    will back-propagate through it), with every BatchNorm of the CrossFFN frozen (eval mode / norm_eval: running statistics
    as a per-channel affine) and for the widths the kernel is built for.  `norm_eval=True` TRAINING keeps the per-op route."""
    if not _FFN_EVAL or ctx.record or ctx.training or C > _FFN_EVAL_MAXC:
        return False
    l = ffn.layers
    if any(is_gn(l[k]) or l[k].training or l[k].running_mean is None for k in (1, 4, 7)):
        return False
    return l[0].weight.shape[0] == 4 * C and l[3].bias is not None and l[0].bias is not None and l[6].bias is not None and \
        bool(ctx.L.hrf_ffn_eval_supported(C, 4 * C))


def ffn_eval(ctx, x, ln, ffn):
    """x + GELU(BN3(fc3(GELU(BN2(dw(GELU(BN1(fc1(LN(x))))))))))  (hrformer.py:351,267-295,371-372 with frozen BatchNorms; DropPath
    is the identity in eval): one launch, the 4C-wide hidden tensor stays on the chip."""
    B, H, W, C = x.t.shape
    l = ffn.layers
    eng = ctx.owner._engine()
    P = _lib._ptr
    a = _lib.FfnEval()
    a.B, a.H, a.W, a.C, a.hidden = B, H, W, C, 4 * C
    a.x = P(x.t)
    a.ln_g, a.ln_b, a.ln_eps = P(ln.weight), P(ln.bias), float(ln.eps)
    (s1, t1, _, _), (s2, t2, _, _), (s3, t3, _, _) = (eng.bn_eval_affine(l[k]) for k in (1, 4, 7))
    a.w1, a.b1, a.s1, a.t1 = P(l[0].weight), P(l[0].bias), P(s1), P(t1)
    a.wd, a.bd, a.s2, a.t2 = P(l[3].weight), P(l[3].bias), P(s2), P(t2)
    a.w3, a.b3, a.s3, a.t3 = P(l[6].weight), P(l[6].bias), P(s3), P(t3)
    out = Act(_new((B, H, W, C), x.t.device))
    a.out = P(out.t)
    ctx.L.hrf_ffn_eval(a, ctx.stream)
    return out


# ----------------------------------------------------------------------------- fused window-attention block
_ATTN_FUSED = os.environ.get('HRF_ATTN_FUSED', '1') != '0'


def attn_block_ok(ctx, C, heads):
    """Is the one-launch attention block (csrc/attn_block.hip) available for this width, in this mode?"""
    if not _ATTN_FUSED or not ctx.L.hrf_attn_block_supported(C, heads):
        return False
    return (not ctx.record) or bool(ctx.L.hrf_attn_block_bwd_supported(C, heads))


def attn_block(ctx, key, heads, xq, xkv, lnq, lnkv, wq, wk, wv, rpb, out_proj, res, res2=None, drop=None, ffn=None):
    """out = res (+ res2) + drop(out_proj(window_attention(LN_q(xq), LN_kv(xkv)))) and, with ffn = (LayerNorm, 1x1 conv,
    BatchNorm), the CrossFFN head h1 = conv(LN(out)) as a Lazy GELU(BN(.)) - ONE launch per direction.
    wq / wk / wv: (Linear module, first output row): the three projections, possibly rows of one packed qkv Linear.
    key: hashable identity of the layer instance (its parameter-gradient slots live in the engine).  xkv is xq:
    self-attention.  The residual conventions are the reference's: self-attention res is xq; cross-attention res is
    the running sum and res2 the modality map that is also the key/value source."""
    L, s = ctx.L, ctx.stream
    tail = None
    if isinstance(xq, LazyTail):
        # the block input is the previous block's CrossFFN tail: this launch forms the rows and writes them to `xin`
        tail = xq
        assert xkv is tail and res is tail and res2 is None, 'a lazy CrossFFN tail feeds self-attention blocks only'
        xq = xkv = res = Act(_new_like(tail.res.t))
    B, H, W, C = xq.t.shape
    cross = xkv is not xq
    assert res2 is None or res2 is xkv
    dev = xq.t.device
    out = Act(_new((B, H, W, C), dev))
    N1 = 4 * C
    st = None
    h1raw = stats = None
    if ffn is not None:
        ln2, conv1, bn1 = ffn
        assert conv1.weight.shape[0] == N1
        h1raw = _new((B, H, W, N1), dev)
        slot = ctx.owner._bn_slot(bn1)
        stats = slot['stats'] if (ctx.training and bn1.training) else None
    mask, mscale, rowscale = drop if drop is not None else (None, 1.0, None)
    P = _lib._ptr

    def wrow(spec):
        lin, r0 = spec
        return lin.weight.data_ptr() + 4 * r0 * C, (lin.bias.data_ptr() + 4 * r0) if lin.bias is not None else None

    def fill():
        a = _lib.AttnBlock()
        a.B, a.H, a.W, a.C, a.heads = B, H, W, C, heads
        a.xq, a.xkv = P(xq.t), P(xkv.t)
        a.lnq_g, a.lnq_b, a.lnkv_g, a.lnkv_b, a.ln_eps = P(lnq.weight), P(lnq.bias), P(lnkv.weight), P(lnkv.bias), float(lnq.eps)
        (a.wq, a.bq), (a.wk, a.bk), (a.wv, a.bv) = wrow(wq), wrow(wk), wrow(wv)
        a.rpb, a.wo, a.bo = P(rpb), P(out_proj.weight), P(out_proj.bias)
        a.res, a.res2 = P(res.t), P(res2.t) if res2 is not None else None
        a.mask, a.mscale, a.rowscale, a.rows_per_sample = P(mask), float(mscale), P(rowscale), H * W
        a.out = P(out.t)
        if ffn is not None:
            a.ln2_g, a.ln2_b, a.out_eps = P(ln2.weight), P(ln2.bias), float(ln2.eps)
            a.w1, a.b1, a.h1, a.stats1, a.hidden = P(conv1.weight), P(conv1.bias), P(h1raw), P(stats), N1
        if tail is not None:
            ts = tail.lazy.st
            a.tail_res, a.tail_raw, a.tail_scale, a.tail_shift = P(tail.res.t), P(ts.raw), P(ts.scale), P(ts.shift)
            a.tail_rowscale, a.x_out = P(tail.rowscale), P(xq.t)
        return a

    a0 = fill()
    if tail is not None:
        tfin = take_fin(ctx, tail.lazy.st)
        if tfin is not None:
            a0._keep_fin = tfin                         # the struct holds a raw pointer to it
            a0.tail_fin = ctypes.addressof(tfin)
    L.hrf_attn_block_fwd(a0, s)
    h1 = None
    if ffn is not None:
        st = bn_forward(ctx, bn1, h1raw, stats)
        h1 = Lazy(st, TF_GELU)
    if not ctx.record:
        return out, h1
    eng = ctx.owner._engine()
    nwin = B * ((H + 6) // 7) * ((W + 6) // 7)
    entries = []
    if ffn is not None:
        entries += [('w1', conv1.weight, 0, N1 * C), ('b1', conv1.bias, 0, N1), ('g2', ln2.weight, 0, C), ('bt2', ln2.bias, 0, C)]
    entries += [('wo', out_proj.weight, 0, C * C), ('bo', out_proj.bias, 0, C)]
    for nm, (lin, r0) in (('q', wq), ('k', wk), ('v', wv)):
        entries += [('w' + nm, lin.weight, r0 * C, C * C), ('b' + nm, lin.bias, r0, C)]
    entries += [('gq', lnq.weight, 0, C), ('btq', lnq.bias, 0, C)]
    if cross:
        entries += [('gkv', lnkv.weight, 0, C), ('btkv', lnkv.bias, 0, C)]
    rpb_one = rpb.requires_grad and os.environ.get('HRF_RPB_ONE', '1') != '0'      # one gather launch per step (0: one per layer)
    offs = eng.fs_register(key, nwin, entries, rpb if rpb_one else None, heads if rpb_one else 0)

    def bwd():
        a = fill()
        a.gout = P(out.grad)
        if ffn is not None:
            (cA, cB, cC), bfin = bn_backward_coef(ctx, st, consumer_follows=True)
            a.du1, a.cA1, a.cB1, a.cC1 = P(st.du), P(cA), P(cB), P(cC)
            if bfin is not None:
                a._keep_bfin = bfin                     # the struct holds a raw pointer to it
                a.bfin1 = ctypes.addressof(bfin)
        if cross:
            # every gradient buffer gets exactly ONE write from this launch: when the residual row IS the query row (the
            # k = 0 modality of a fusion block: acc starts as x) the residual gradient rides on the dq write (dq_add_res)
            # instead of a second, aliasing dres store (the two stores overwrote each other for M = 1)
            res_is_q = res is xq
            if res.needs_grad and not res_is_q:
                g, acc = res.grad_target()
                a.dres, a.dres_acc = P(g), acc
            if xkv.needs_grad:
                g, acc = xkv.grad_target()
                a.dkv, a.dkv_acc, a.dkv_add_res = P(g), acc, 1 if res2 is not None else 0
            if xq.needs_grad:
                g, acc = xq.grad_target()
                a.dq, a.dq_acc, a.dq_add_res = P(g), acc, 1 if res_is_q else 0
        elif tail is not None:
            # dx of this block IS the gradient of the tail's residual stream; the launch also emits the tail's du and moments
            ts = tail.lazy.st
            g, acc = tail.res.grad_target()
            a.dq, a.dq_acc, a.dq_add_res = P(g), acc, 1
            ts.du = _new_like(ts.raw)
            a.tail_du, a.tail_gstats = P(ts.du), P(ts.gstats)
        elif xq.needs_grad:
            assert res is xq
            g, acc = xq.grad_target()
            a.dq, a.dq_acc, a.dq_add_res = P(g), acc, 1
        a.pslot, a.slot_stride = eng.fs_buffer(key), offs['_n']
        if C <= 18:            # the 8-wave form parks role A's gradient rows here (step-lifetime scratch, never read by anyone else)
            a.gx_park = P(_new((nwin * 64 * 32,), dev))
        dsp = None if rpb_one else _new((nwin * heads * 49 * 49,), dev)
        a.ds_plane = eng.fs_plane(key) if rpb_one else P(dsp)
        for nm in ('w1', 'b1', 'g2', 'bt2', 'wo', 'bo', 'wq', 'bq', 'wk', 'bk', 'wv', 'bv', 'gq', 'btq', 'gkv', 'btkv', 'rpb'):
            setattr(a, 'off_' + nm, offs.get(nm, -1))
        L.hrf_attn_block_bwd(a, s)
        if rpb.requires_grad and not rpb_one:       # leaf: gathered from the dS planes with the deferred weight gradients
            racc, cs = eng.grad_acc(rpb)
            ctx.side_launch(lambda: L.hrf_rpb_grad(dsp, nwin, heads, racc, cs, ctx.stream), cost=4.0 * dsp.numel())
        if st is not None:
            st.du = None
    ctx.push(bwd, sync=st)
    return out, h1


# ----------------------------------------------------------------------------- depthwise conv
def dwconv_bn(ctx, src, conv, bn, mode):
    L, s = ctx.L, ctx.stream
    x, _, (B, H, W, C), tf, sc, sh, _ = _src_desc(src)
    w, b = conv.weight, conv.bias
    stride = conv.stride[0]
    Ho, Wo = _conv_out_hw(H, W, 3, stride)
    y = _new((B, Ho, Wo, C), x.device)
    if is_gn(bn):
        L.hrf_dwconv_fwd(x, B, H, W, C, w, b, stride, tf, sc, sh, y, None, _src_fin(ctx, src, 1 << 30), s)
        gst = gn_forward(ctx, bn, y)
        gout = Lazy(gst, mode)
        if ctx.probe is not None and mode == TF_RELU:
            ctx.probe.append(('lazy', gst))

        def gbwd():
            draw = gn_backward(ctx, gst)
            if isinstance(src, Lazy):
                ps = src.st
                ps.du = _new_like(ps.raw)
                L.hrf_dwconv_bwd_data(draw, None, None, None, None, None, w, stride, B, H, W, C, ps.du, 0, 1, ps.raw,
                                      ps.scale, ps.shift, _TF2ACT[src.mode], ps.gstats, s)
            elif src.needs_grad:
                g, acc = src.grad_target()
                L.hrf_dwconv_bwd_data(draw, None, None, None, None, None, w, stride, B, H, W, C, g, acc, 0, None, None,
                                      None, 0, None, s)
            if w.requires_grad:
                eng = ctx.owner._engine()
                wacc, cs = eng.grad_acc(w)
                bacc = eng.grad_acc(b)[0] if b is not None else None
                ctx.side_launch(lambda: L.hrf_dwconv_bwd_weight(
                    draw, None, None, None, None, x, B, H, W, C, stride, tf, sc, sh, wacc, bacc, cs, ctx.stream),
                    cost=4.0 * B * H * W * C * (1.0 + 2.0 / (stride * stride)), key=('dw_w', stride))
            gst.du = None
        ctx.push(gbwd)
        return gout
    slot = ctx.owner._bn_slot(bn)
    train = ctx.training and bn.training
    stats = slot['stats'] if train else None
    L.hrf_dwconv_fwd(x, B, H, W, C, w, b, stride, tf, sc, sh, y, stats, _src_fin(ctx, src, 1 << 30), s)
    st = bn_forward(ctx, bn, y, stats)
    out = Lazy(st, mode)
    if ctx.probe is not None and mode == TF_RELU:
        ctx.probe.append(('lazy', st))

    def bwd():
        needs = isinstance(src, Lazy) or src.needs_grad
        (cA, cB, cC), bfin = bn_backward_coef(ctx, st, consumer_follows=needs, limit=1 << 30)
        # data gradient first: with `bfin` it publishes cA/cB/cC for the weight-gradient launch below
        fused_wg = False
        if isinstance(src, Lazy):
            ps = src.st
            ps.du = _new_like(ps.raw)
            fused_wg = stride == 1 and w.requires_grad and _DW_FUSED_WG
            if fused_wg:                          # weight / bias gradient from the same pass (no leaf launch)
                eng = ctx.owner._engine()
                wacc, cs = eng.grad_acc(w)
                bacc = eng.grad_acc(b)[0] if b is not None else None
                L.hrf_dwconv_bwd_data_weight(st.du, st.raw, cA, cB, cC, bfin, w, B, H, W, C, ps.du, ps.raw, ps.scale,
                                             ps.shift, _TF2ACT[src.mode], ps.gstats, wacc, bacc, cs, s)
            else:
                L.hrf_dwconv_bwd_data(st.du, st.raw, cA, cB, cC, bfin, w, stride, B, H, W, C, ps.du, 0, 1, ps.raw,
                                      ps.scale, ps.shift, _TF2ACT[src.mode], ps.gstats, s)
        elif src.needs_grad:
            g, acc = src.grad_target()
            L.hrf_dwconv_bwd_data(st.du, st.raw, cA, cB, cC, bfin, w, stride, B, H, W, C, g, acc, 0, None, None,
                                  None, 0, None, s)
        if w.requires_grad and not fused_wg:
            du_ = st.du
            eng = ctx.owner._engine()
            wacc, cs = eng.grad_acc(w)
            bacc = eng.grad_acc(b)[0] if b is not None else None
            ctx.side_launch(lambda: L.hrf_dwconv_bwd_weight(
                du_, st.raw, cA, cB, cC, x, B, H, W, C, stride, tf, sc, sh, wacc, bacc, cs, ctx.stream),
                cost=4.0 * B * H * W * C * (1.0 + 2.0 / (stride * stride)), key=('dw_w', stride))
        st.du = None
    ctx.push(bwd, sync=st)
    return out


# ----------------------------------------------------------------------------- materialisation
def materialize(ctx, lazy, act, res=None, lazy2=None, act_first=False, rowscale=None):
    """Store act(BN(raw)) [+res] [+BN2(raw2)] as an NHWC tensor.

    act_first=True : out = res + rowscale*act(BN(raw))           (CrossFFN tail, DropPath scale)
    act_first=False: out = act(BN(raw) + res + BN2(raw2))        (Bottleneck tail / transition ReLU)
    """
    L, s = ctx.L, ctx.stream
    if not act_first and act not in (ACT_NONE, ACT_RELU):
        raise _lib.HRFuserHipError('materialize: an activation applied last must be ReLU or none (its backward uses the output mask)')
    st = lazy.st
    B, H, W, C = st.raw.shape
    rows = B * H * W
    out = Act(_new_like(st.raw))
    st2 = lazy2.st if lazy2 is not None else None
    rs_out = None
    if act_first:                              # CrossFFN tail = input of the next block's norm1
        out.rowstat = (LN_EPS, _new((rows, 2), st.raw.device))
        rs_out = out.rowstat[1]
    L.hrf_affine_act_res(st.raw, st.scale, st.shift, st2.raw if st2 else None, st2.scale if st2 else None,
                         st2.shift if st2 else None, res.t if res is not None else None, rowscale, H * W,
                         act, 1 if act_first else 0, out.t, rows, C, rs_out, LN_EPS, take_fin(ctx, st), take_fin(ctx, st2), s)
    if ctx.probe is not None and act == ACT_RELU and not act_first:
        ctx.probe.append(('out', out.t))

    def bwd():
        g = _new_like(out.t)
        if act_first:
            assert act == ACT_GELU and st2 is None
            L.hrf_act_bwd(out.grad, None, st.raw, st.scale, st.shift, rowscale, H * W, 1, g, None, None,
                          st.gstats, None, None, rows, C, s)
            st.du = g
            if res is not None and res.needs_grad:
                res.add_grad(out.grad)
        else:
            mode = 0 if act == ACT_RELU else 2
            L.hrf_act_bwd(out.grad, out.t, st.raw, None, None, None, 1, mode, g, st2.raw if st2 else None, None,
                          st.gstats, st2.gstats if st2 else None, None, rows, C, s)
            st.du = g
            if st2 is not None:
                st2.du = g
            if res is not None and res.needs_grad:
                # g stays alive as st.du for the producer conv's backward -> never alias it
                if res.grad is None:
                    res.grad = gpu_clone(g)
                else:
                    gpu_add_(res.grad, g)
    ctx.push(bwd)
    return out


def fuse_sum(ctx, dims, terms):
    """HRModule exchange for one output branch: ReLU(sum terms).  terms: list of
    ('id', Act) | ('same', Lazy) | ('up', Lazy: bilinear) | ('near', Lazy: nearest by an integer factor)
    (hrnet.py:192-206)."""
    L, s = ctx.L, ctx.stream
    B, H, W, C = dims
    dev = terms[0][1].t.device if isinstance(terms[0][1], Act) else terms[0][1].raw.device
    out = Act(_new((B, H, W, C), dev))
    args = []
    fins = (_lib.BnFin * 4)()
    any_fin = False
    for k, (kind, t) in enumerate(terms):
        if kind == 'id':
            args += [1, t.t, None, None, 0, 0]
            continue
        fin = take_fin(ctx, t.st, FIN_MAXC // 2)
        if fin is not None:
            fins[k] = fin
            any_fin = True
        if kind == 'same':
            args += [2, t.raw, t.st.scale, t.st.shift, 0, 0]
        else:
            args += [3 if kind == 'up' else 4, t.raw, t.st.scale, t.st.shift, t.raw.shape[1], t.raw.shape[2]]
    for _ in range(4 - len(terms)):
        args += [0, None, None, None, 0, 0]
    L.hrf_fuse_sum(*args, out.t, B, H, W, C, fins if any_fin else None, s)
    if ctx.probe is not None:
        ctx.probe.append(('out', out.t))

    def bwd():
        g = _new_like(out.t)
        same = [t for kind, t in terms if kind == 'same']
        assert len(same) <= 3
        ys = [t.raw for t in same] + [None] * (3 - len(same))
        sts = [t.st.gstats for t in same] + [None] * (3 - len(same))
        L.hrf_act_bwd(out.grad, out.t, ys[0], None, None, None, 1, 0, g, ys[1], ys[2], sts[0], sts[1], sts[2],
                      B * H * W, C, s)
        # g doubles as st.du of every 'same' term and is read later by their conv backward (possibly
        # on another lane): it may only be aliased as the identity gradient when no such reader exists
        can_alias = len(same) == 0
        for kind, t in terms:
            if kind == 'id':
                if t.needs_grad:
                    if t.grad is None:
                        t.grad = g if can_alias else gpu_clone(g)
                        can_alias = False
                    else:
                        gpu_add_(t.grad, g)
            elif kind == 'same':
                t.st.du = g
            else:
                st = t.st
                st.du = _new_like(st.raw)
                adj = L.hrf_bilinear_up_bwd if kind == 'up' else L.hrf_nearest_up_bwd
                adj(g, C, 0, B, H, W, C, st.raw, st.raw.shape[1], st.raw.shape[2], st.du, st.gstats, s)
    ctx.push(bwd)
    return out


def collect_relu_masks(ctx):
    """Test instrumentation: the sign mask (logical NCHW, bool) of every ReLU the forward applied, in issue order.
    On-load ReLUs are re-evaluated with the library's own affine + ReLU kernel so the mask is the one the kernels used."""
    masks = []
    for kind, obj in ctx.probe:
        if kind == 'out':
            t = obj
        else:
            st = obj
            B, H, W, C = st.raw.shape
            t = torch.empty_like(st.raw)
            ctx.L.hrf_affine_act_res(st.raw, st.scale, st.shift, None, None, None, None, None, H * W, ACT_RELU, 0, t,
                                     B * H * W, C, None, 0.0, None, None, ctx.stream)
        masks.append((t > 0).permute(0, 3, 1, 2))
    return masks
