"""Peer-to-peer SyncBN exchange (csrc/p2p_exchange.hip): the host side - inbox allocation, IPC handle exchange, slot table.

What the reference gets from torch.nn.SyncBatchNorm (norm_cfg type SyncBN, configs/_base_/models/
cascade_rcnn_hrfuser_fpn_nus_clr_fusion.py:2; wrapped by MMDistributedDataParallel, mmdet/apis/train.py:113-121) is, per
BatchNorm layer and direction, an exchange of 2*C moments + a sample count between the ranks.  With RCCL those are
collectives of one communicator that must run in one order on every rank (the lanes hop to the main lane for each);
here every layer owns a slot in an inbox every rank exposes to its peers and an exchange is ONE launch on the lane that needs
it (include/hrfuser_hip.h: hrf_p2p_exchange).  HRF_SYNC_P2P (the same on every rank: part of the schedule fingerprint): unset
= auto (groups of more than one rank use it after a collective handshake and fall back TOGETHER to the communicator schedule
if any rank cannot), 1 = on, 0 = off.  The communicator schedule stays the fallback and the parity reference
(bench.py sync_ab times both), and the gradient all-reduce at the end of a step stays a collective.

Nothing here computes: allocation, `torch.distributed` object all-gather of the 64-byte IPC handles, pointer tables.
"""
import ctypes
import os

import torch

from . import _lib


def mode():
    """HRF_SYNC_P2P: '1' = on (a failure to set it up raises), '0' = off (collectives through the communicator), unset /
    'auto' = on for a group of more than one rank AFTER a collective handshake (IPC mapping + a few verified test exchanges
    with a short time-out); any rank failing it sends every rank back to the communicator schedule with a warning."""
    v = os.environ.get('HRF_SYNC_P2P', 'auto').strip().lower()
    return {'1': 'on', 'on': 'on', '0': 'off', 'off': 'off'}.get(v, 'auto')


def wanted(world):
    m = mode()
    return m == 'on' or (m == 'auto' and world > 1)


class P2PExchange:
    """Inbox of this rank + the mapped inboxes of its peers + the static slot of every BatchNorm (layer, direction)."""

    def __init__(self, lib, bns, group, world, rank, device, timeout_s=None):
        """LOCAL part only (slot table, inbox allocation): nothing here talks to the other ranks - `create` drives the collective
        steps so that a rank whose local step fails still takes part in every collective the others are in."""
        self.lib, self.world, self.rank, self.device, self.group = lib, int(world), int(rank), device, group
        if not (1 <= self.world <= 8):
            raise _lib.HRFuserHipError(f'peer-to-peer SyncBN exchange: 1..8 ranks of one node, got {world}')
        off = 0
        self.slots = {}
        for i, m in enumerate(bns):
            C = m.num_features
            self.slots[id(m)] = ((off, 2 * i), (off + 2 * C + 1, 2 * i + 1))       # (slot_off, slot_id) forward, backward
            off += 2 * (2 * C + 1)
        self.test_slot = (off, 2 * len(bns))                                       # one more slot (C = 1): the handshake
        off += 3
        self.slot_doubles, self.nslots = off, 2 * len(bns) + 1
        self.data_bytes = self.world * 2 * self.slot_doubles * 8
        self.bytes = self.data_bytes + self.world * 2 * self.nslots * 8
        self.base, self.handle, self.opened, self.ctx = None, None, [], None
        self.peers = [None] * self.world
        self.exchanges = 0
        # how long an exchange waits for a peer before the step is declared lost (NaN statistics + error word) - a rank that is
        # merely slow (a checkpoint, a data-loader stall) is waited for
        # (r6: 600 s - torch's NCCL watchdog default - instead of 1800: a dead peer holds the GPU in a spin-wait for that long, and
        # Trainer.check() / .item() / synchronize() block the host behind it; long evaluation pauses between steps do NOT count -
        # the wait only runs while a step's exchange kernel is resident.  bench.py sets 60 s for its first-contact runs.)
        self.timeout_s = float(os.environ.get('HRF_P2P_TIMEOUT_S', '600')) if timeout_s is None else float(timeout_s)
        self.retired = False           # replaced by a newer context (Engine.p2p_context) while captures still pinned it
        self._order = {}               # first step: {stream: [slot ids in enqueue order]} (verify_order), None once verified
        self.pins = 0                  # captured graphs that carry this context's pointers (Engine.p2p_context never closes those)
        self._poll = None              # (pinned host word, event) of the last non-blocking read of the error word
        base = ctypes.c_void_p()
        handle = (ctypes.c_char * 64)()
        lib.hrf_p2p_alloc(self.bytes, ctypes.addressof(base), ctypes.addressof(handle) if self.world > 1 else None)
        self.base, self.handle = base.value, bytes(handle.raw)
        self.peers[self.rank] = self.base

    def open_peers(self, everyone):
        """Map the peers' inboxes (`everyone`: the all-gathered (handle, pid, bytes) of every rank; None where a rank failed)."""
        for p, ent in enumerate(everyone):
            if p == self.rank:
                continue
            if ent is None:
                raise _lib.HRFuserHipError(f'peer-to-peer SyncBN exchange: rank {p} could not allocate its inbox')
            h, pid, nbytes = ent
            if nbytes != self.bytes:
                raise _lib.HRFuserHipError(f'peer-to-peer SyncBN exchange: rank {p} built a different slot table '
                                           f'({nbytes} vs {self.bytes} bytes) - the ranks do not run the same model')
            if pid == os.getpid():
                raise _lib.HRFuserHipError('peer-to-peer SyncBN exchange: two ranks in one process')
            q = ctypes.c_void_p()
            hb = (ctypes.c_char * 64).from_buffer_copy(h)
            self.lib.hrf_p2p_open(ctypes.addressof(hb), ctypes.addressof(q))
            self.peers[p] = q.value
            self.opened.append(q.value)

    def finish(self):
        """The kernel-side context, once every inbox is mapped."""
        self.gen = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int64, device=self.device)
        c = _lib.P2p()
        c.world, c.rank = self.world, self.rank
        for p in range(self.world):
            c.inbox[p] = self.peers[p]
            c.flags[p] = self.peers[p] + self.data_bytes
        c.slot_doubles, c.nslots = self.slot_doubles, self.nslots
        c.gen, c.err = self.gen.data_ptr(), self.err.data_ptr()
        c.timeout_ticks = int(self.timeout_s * 1e8)
        self.ctx = c

    @classmethod
    def create(cls, lib, bns, group, world, rank, device, strict):
        """-> a ready exchange context, or None when the ranks agreed to use the communicator schedule instead.  COLLECTIVE: every
        rank walks through the same sequence of collectives (object all-gather of the handles, agreement all-reduces) whatever
        fails locally in between; only the AGREED outcome raises (strict) or falls back (auto) - on every rank alike."""
        import torch.distributed as dist
        why, px = '', None

        def agree(ok):
            if world <= 1:
                return ok
            dev = device if dist.get_backend(group) == 'nccl' else torch.device('cpu')
            t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t))
        try:
            px = cls(lib, bns, group, world, rank, device)
        except Exception as e:
            why = f'{type(e).__name__}: {str(e)[:200]}'
        if world > 1:
            everyone = [None] * world
            dist.all_gather_object(everyone, (px.handle, os.getpid(), px.bytes) if px is not None else None, group=group)
            if px is not None:
                try:
                    px.open_peers(everyone)
                except Exception as e:
                    why = f'{type(e).__name__}: {str(e)[:200]}'
        ok = agree(px is not None and not why)            # (also the barrier: nobody pushes into an inbox that is not mapped everywhere)
        if ok:
            px.finish()
            if world > 1:
                try:
                    px.handshake()
                except Exception as e:
                    why = f'{type(e).__name__}: {str(e)[:200]}'
                ok = agree(not why)
        if ok:
            return px
        if px is not None:
            px.close()
        msg = 'SyncBN: the peer-to-peer exchange is not available on every rank' + (f' (this rank: {why})' if why else '')
        if strict:
            raise _lib.HRFuserHipError(msg + ' and HRF_SYNC_P2P=1 demands it')
        import warnings
        warnings.warn(msg + '; using the collective schedule (HRF_SYNC_P2P=0 silences this)')
        return None

    def handshake(self, rounds=6, timeout_s=5.0):
        """A few verified test exchanges on the reserved slot (both parities, a short time-out): every rank contributes
        (rank + 1) * generation per moment copy and must read back the sum over the ranks.  Raises on a time-out or a wrong
        value - before the first real exchange depends on the mapping."""
        keep = self.ctx.timeout_ticks
        self.ctx.timeout_ticks = int(timeout_s * 1e8)
        try:
            stream = _lib.stream_ptr()
            K = _lib.STAT_COPIES
            src = torch.zeros(K * 2, dtype=torch.float64, device=self.device)
            out = torch.zeros(3, dtype=torch.float64, device=self.device)
            n = 1
            for _ in range(rounds):
                self.tick(stream)
                g = int(self.gen.item())
                src.fill_(float((self.rank + 1) * g))
                self.lib.hrf_p2p_exchange(self.ctx, (ctypes.c_void_p * n)(src.data_ptr()), (ctypes.c_int * n)(1), n,
                                          (ctypes.c_double * n)(float(self.rank + 7)), (ctypes.c_long * n)(self.test_slot[0]),
                                          (ctypes.c_int * n)(self.test_slot[1]), out, 0, stream)
                got = out.cpu().tolist()
                self.check()
                tri = self.world * (self.world + 1) // 2
                want = [float(K * g * tri)] * 2 + [float(7 * self.world + tri - self.world)]
                if got != want:
                    raise _lib.HRFuserHipError(f'peer-to-peer SyncBN exchange: handshake generation {g} read {got}, expected {want}')
        finally:
            self.ctx.timeout_ticks = keep

    def tick(self, stream):
        """Once per training step, before the first exchange (a launch: part of a captured step)."""
        self.lib.hrf_p2p_tick(self.gen, stream)

    def exchange(self, sts, backward, rows, packed, stream):
        """One launch: fold + push + wait + reduce for the BatchNorm states `sts` (forward moments or backward sums)."""
        n = len(sts)
        k = 1 if backward else 0
        ptrs = (ctypes.c_void_p * n)(*[(st.gstats if backward else st.stats).data_ptr() for st in sts])
        cs = (ctypes.c_int * n)(*[st.C for st in sts])
        rw = (ctypes.c_double * n)(*[float(st.raw.numel() // st.C) for st in sts]) if rows else None
        so = (ctypes.c_long * n)(*[self.slots[id(st.bn)][k][0] for st in sts])
        si = (ctypes.c_int * n)(*[self.slots[id(st.bn)][k][1] for st in sts])
        self.lib.hrf_p2p_exchange(self.ctx, ptrs, cs, n, rw, so, si, packed, 0, stream)
        self.exchanges += 1
        if self._order is not None:
            self._order.setdefault(stream, []).extend(int(v) for v in si)

    def verify_order(self):
        """LIVENESS of the exchange.  An exchange is a launch that spin-waits for its peers' flags; launches of one HIP stream
        (one executor stream of a captured graph) run in order.  Two ranks that enqueued exchanges X and Y of one stream in
        OPPOSITE order would wait for each other until the time-out.  This cannot happen because (i) every rank runs the same
        program on the same schedule fingerprint (runtime.sync_fingerprint, checked at set_sync_group), so the sequence of
        exchanges per lane is the same everywhere, and (ii) the stream of a launch is a function of the program alone (lanes
        are forked structurally; the hipGraph executor assigns its streams from the graph's structure).  (i) is ASSERTED here
        instead of assumed: at the start of the second training step every rank all-gathers a digest of the per-lane slot
        sequences of its first step and raises if they differ - before the first captured replay depends on it."""
        if self._order is None:
            return
        log, self._order = self._order, None
        if self.world <= 1 or not log:
            return
        import hashlib
        import torch.distributed as dist
        seqs = [tuple(v) for v in log.values()]          # lanes in order of first use (stream handles differ between ranks)
        mine = hashlib.sha256(repr(seqs).encode()).hexdigest()
        everyone = [None] * self.world
        dist.all_gather_object(everyone, (mine, len(seqs), sum(len(v) for v in seqs)), group=self.group)
        if any(e != everyone[0] for e in everyone):
            raise _lib.HRFuserHipError(
                'peer-to-peer SyncBN exchange: the ranks enqueue their exchanges in different orders per lane '
                f'(digest, lanes, exchanges per rank: {[(e[0][:12], e[1], e[2]) for e in everyone]}) - they would dead-lock '
                'until the time-out; the ranks do not run the same model / schedule')

    def poll(self):
        """Step-boundary check that never blocks the host: raise if the error word READ AT THE PREVIOUS CALL was set, then queue
        the next read (a pinned 8-byte copy + an event on the current stream).  Trainer.step / replay and the autograd bridge
        call it once per step, so a lost exchange surfaces one step later at the latest (its statistics are NaN meanwhile)."""
        if self.ctx is None:
            return
        if self.err.device.type != 'cuda':               # the CPU emulator of the tests: launches are synchronous
            return self._raise_if(int(self.err.item()))
        if torch.cuda.is_current_stream_capturing():
            return
        if self._poll is not None:
            host, ev = self._poll
            if not ev.query():
                return                                   # the previous read has not landed yet: look again next step
            self._raise_if(int(host[0]))
        host = self._poll[0] if self._poll is not None else torch.zeros(1, dtype=torch.int64).pin_memory()
        host.copy_(self.err, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._poll = (host, ev)

    def check(self):
        """Raise if an exchange timed out (reads one word from the device: call at a point that synchronises anyway)."""
        self._raise_if(int(self.err.item()))

    def _raise_if(self, e):
        if e:
            src, slot = (e >> 32) - 1, (e & 0xffffffff) - 1
            raise _lib.HRFuserHipError(
                f'peer-to-peer SyncBN exchange timed out on rank {self.rank}: rank {src} never delivered slot {slot} '
                f'(BatchNorm {slot // 2}, {"backward" if slot & 1 else "forward"}) of generation {int(self.gen.item())} within '
                f'{self.timeout_s:g} s (HRF_P2P_TIMEOUT_S) - a peer died, runs a different model / schedule, or HRF_SYNC_P2P differs '
                'between the ranks; the BatchNorm statistics of that step are NaN')

    def pin(self):
        """a captured graph carries this context's inbox / flag / counter pointers by value"""
        self.pins += 1

    def unpin(self):
        """the capture that pinned the context was dropped; a RETIRED context (replaced in Engine.p2p_context) whose last pin goes
        is closed - its IPC mappings and its inbox are released (pins only ever grew)"""
        self.pins = max(0, self.pins - 1)
        if self.pins == 0 and self.retired:
            self.close()

    def close(self):
        for q in self.opened:
            try:
                self.lib.hrf_p2p_close(q)
            except Exception:
                pass
        self.opened = []
        if self.base:
            try:
                self.lib.hrf_p2p_free(self.base)
            except Exception:
                pass
            self.base = None
