"""Multi-GPU sharded radix sort: one process per GPU, MSB-bucket exchange over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).  No reference counterpart -- the reference drives exactly
one device (Adl/Adl.h:90-94); SURVEY.md section 8e describes this step.

Radix order is decided first by the most significant bits, so rank g of G owns a contiguous range of top-byte
values: with balance=False the keys whose top log2(G) bits equal g; with balance=True (default) a range chosen
from the GLOBAL top-byte histogram so that every rank receives about the same number of keys whatever the key
distribution (balanced splitters; rank order stays key order).  Per sort:
  1. local stable partition of the rank's keys by their top byte (adlhip_partition_top_byte_u32: one count ->
     scan -> scatter pass), which also yields the 256 top-byte totals; balance=True: ONE 2-KiB all-reduce of the
     totals, then every rank cuts the byte range at the same G-1 places (choose_splitters, device-side torch
     ops, no host sync) and folds its own totals into G segment sizes;
  2. ONE small all-gather of the G segment sizes (G x G matrix) -- the only host sync;
  3. ONE all_to_all_single of the segments with those split sizes.  On a fully connected xGMI node
     every pair of GPUs has its own link, so all G-1 links of a GPU carry traffic at once;
  4. local radix sort of the received keys (adlhip_radix_sort_u32).
Concatenating the ranks' outputs in rank order is the globally sorted array.  Receive buffers are
sized from the exchanged counts, never from n/G.

Two drivers over the same steps:
  * ShardedRadixSort.sort(keys)          -- one batch, steps 1-4 back to back on the caller's stream;
  * ShardedRadixSort.sort_stream(batches) -- a stream of independent batches, two-stage pipeline: steps
    1-3 of batch i+1 ("exchange stage", its own HIP stream) run while step 4 of batch i ("sort stage",
    a second HIP stream) runs.  The exchange is xGMI-bound and leaves HBM and most CUs idle, the local
    sort is HBM-bound and leaves xGMI idle, so the two overlap almost fully; the only host sync (the
    count matrix of batch i+1) is taken AFTER batch i's local sort has been enqueued.

The device work is delegated to a `backend` object so that the host logic above (split arithmetic,
exchange, ordering, pipelining) can be exercised on CPU with gloo in the tests.  The product backend
is HipBackend; there is no CPU backend in this package.
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .adl import Buffer, Config, DeviceUtils
from .pprims import ELEM_KV32, ELEM_U32, Pprims


def choose_splitters(global_totals, num_ranks):
    """Cut the 256 top-byte values into `num_ranks` contiguous ranges of near-equal population.
    global_totals: int64 tensor [256] (the all-reduced top-byte histogram, on any device).  Returns an int64 tensor
    bounds[num_ranks + 1] on the same device, bounds[0] = 0, bounds[-1] = 256, non-decreasing: rank g owns the top
    bytes [bounds[g], bounds[g+1]).  Every cut is the byte boundary whose cumulative count is nearest to
    g * total / num_ranks, so a rank's share differs from the mean by at most the population of one byte value
    (uniform keys: exactly the fixed top-bits ownership).  Pure tensor ops: deterministic, identical on every
    rank, no host synchronisation."""
    G = int(num_ranks)
    t = global_totals.to(torch.int64)
    cum = torch.cumsum(t, 0)                                     # cum[b] = keys with top byte <= b
    total = cum[-1]
    g = torch.arange(1, G, dtype=torch.int64, device=t.device)
    target = (total * g) // G
    idx = torch.searchsorted(cum, target).clamp(max=255)         # first byte b with cum[b] >= target
    hi = cum[idx]
    lo = torch.where(idx > 0, cum[(idx - 1).clamp(min=0)], torch.zeros_like(hi))
    cut = torch.where((hi - target) <= (target - lo), idx + 1, idx)     # boundary after byte idx, or before it
    cut = torch.cummax(cut, 0).values if G > 1 else cut
    zero = torch.zeros(1, dtype=torch.int64, device=t.device)
    return torch.cat([zero, cut, zero + 256])


class _Stage:
    """One pipeline stage on one GPU: a HIP stream (torch's) + an adlhip device bound to it + the
    Pprims object that owns the stage's sort scratch."""

    def __init__(self, local_rank, stream):
        self.stream = stream            # torch.cuda.Stream, or None = torch's current stream at construction
        raw = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        self.device = DeviceUtils.allocate(cfg=Config(local_rank), stream=raw)
        self.pprims = Pprims()
        self.work = None                # partition scratch
        self.reserved = 0               # bytes of elements the sort scratch has been sized for

    def close(self):
        self.pprims.close()
        if self.work is not None:
            self.work.release()
            self.work = None
        DeviceUtils.deallocate(self.device)


class HipBackend:
    """Device half of the sharded sort on one MI355X: torch owns the memory, libadlhip.so does the work
    on torch streams (so RCCL collectives and sort kernels are ordered by stream order and events,
    never by host syncs).  Stages: `caller` (torch's current stream at construction; sort()),
    `exchange` and `sorting` (two extra streams, created on first use; sort_stream())."""

    pipeline_depth = 3      # receive slots: the caller may hold 2 results while the third slot is being filled

    def __init__(self, local_rank=0):
        import ctypes
        self._ct = ctypes
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        self.torch_device = torch.device("cuda", local_rank)
        self._caller = _Stage(local_rank, None)
        self._exchange = None
        self._sorting = None
        self._cur = self._caller
        self._params = {}
        self._recv = {}
        self._part = {}

    # the caller-stream device doubles as the handle for knobs (bench.py: sort.algo, sort.digit_bits)
    @property
    def device(self):
        return self._caller.device

    @property
    def pprims(self):
        return self._caller.pprims

    def setParam(self, name, value):
        """Set a libadlhip knob on every stage (also on stages created later)."""
        self._params[name] = int(value)
        for st in (self._caller, self._exchange, self._sorting):
            if st is not None:
                st.device.setParam(name, value)

    def _stage(self, which):
        st = getattr(self, which)
        if st is None:
            # the exchange stage feeds the xGMI links, the scarcer resource: its (short) partition kernels go first so that
            # the next all-to-all starts as early as possible under the running local sort
            prio = -1 if which == "_exchange" else 0
            st = _Stage(self.local_rank, torch.cuda.Stream(device=self.torch_device, priority=prio))
            for k, v in self._params.items():
                st.device.setParam(k, v)
            setattr(self, which, st)
        return st

    def close(self):
        torch.cuda.synchronize(self.torch_device)
        self._recv.clear()
        self._part.clear()
        for st in (self._caller, self._exchange, self._sorting):
            if st is not None:
                st.close()
        self._caller = self._exchange = self._sorting = None

    # ---- pipeline plumbing (sort_stream) --------------------------------------------------------
    @contextlib.contextmanager
    def exchange_scope(self, after_caller=False):
        """Work enqueued inside goes to the exchange stage's stream; with after_caller, after everything
        the caller has enqueued so far on its own stream (the batch's keys, reads of earlier results)."""
        st = self._stage("_exchange")
        if after_caller:
            st.stream.wait_stream(torch.cuda.current_stream(self.torch_device))
        prev, self._cur = self._cur, st
        try:
            with torch.cuda.stream(st.stream):
                yield
        finally:
            self._cur = prev

    @contextlib.contextmanager
    def sort_scope(self, after=None):
        """Work enqueued inside goes to the sort stage's stream, after event `after`."""
        st = self._stage("_sorting")
        if after is not None:
            st.stream.wait_event(after)
        prev, self._cur = self._cur, st
        try:
            with torch.cuda.stream(st.stream):
                yield
        finally:
            self._cur = prev

    def event(self, timing=False):
        """Record an event on the stream of the scope we are in."""
        ev = torch.cuda.Event(enable_timing=timing)
        ev.record(torch.cuda.current_stream(self.torch_device))
        return ev

    def wait(self, ev):
        """The stream of the scope we are in waits for `ev` (device-side, the host does not block)."""
        if ev is not None:
            torch.cuda.current_stream(self.torch_device).wait_event(ev)

    def _slot_buffer(self, pool, slot, n, dtype=torch.int32):
        slot = (slot, dtype)
        t = pool.get(slot)
        if t is None or t.numel() < n:
            # grow with headroom so that batch-to-batch jitter of the received count does not reallocate;
            # the old tensor goes back to torch's allocator, which keeps it alive for the streams that used it
            if t is not None:
                for st in (self._exchange, self._sorting):
                    if st is not None:
                        t.record_stream(st.stream)
            # receive slots: balanced splitters keep a rank's share within ~1.25x of the mean, so a quarter of headroom
            # means later batches of the same size never reallocate; partition slots hold exactly the batch
            cap = int(n) + int(n) // (4 if pool is self._recv else 16) + 1024
            t = torch.empty(cap, dtype=dtype, device=self.torch_device)
            pool[slot] = t
        return t[:n]

    def reserve(self, n, dtype=torch.int32):
        """Allocate everything a sort_stream over batches of about n elements per rank needs -- the receive and partition
        slots, both stages' scratch -- now, so that the first batches do not pay for it (allocations sync the device)."""
        n = int(n)
        cap = n + n // 4 + 1024
        for slot in range(self.pipeline_depth):
            self.recv_buffer(slot, cap, dtype)
            self.part_buffer(slot, n, dtype)
        kind = ELEM_U32 if dtype == torch.int32 else ELEM_KV32
        ex, so = self._stage("_exchange"), self._stage("_sorting")
        if cap * torch.empty(0, dtype=dtype).element_size() > so.reserved:
            so.reserved = cap * torch.empty(0, dtype=dtype).element_size()
            so.pprims.reserve(so.device, kind, cap)
        ct = self._ct
        tb, wb = ct.c_size_t(), ct.c_size_t()
        check(_lib.load().adlhip_radix_sort_scratch_bytes(ex.device._h, kind, n, ct.byref(tb), ct.byref(wb)), "scratch_bytes")
        if ex.work is None or ex.work.getSize() < wb.value:
            if ex.work is not None:
                DeviceUtils.waitForCompletion(ex.device)
                ex.work.release()
            ex.work = Buffer(ex.device, wb.value + wb.value // 8, np.uint8)

    def recv_buffer(self, slot, n, dtype=torch.int32):
        """Persistent receive buffer of pipeline slot `slot`, at least n elements (view of exactly n)."""
        return self._slot_buffer(self._recv, slot, n, dtype)

    def part_buffer(self, slot, n, dtype=torch.int32):
        return self._slot_buffer(self._part, slot, n, dtype)

    # ---- device work -----------------------------------------------------------------------------
    # element kinds: int32 tensors hold u32 keys, int64 tensors hold {u32 key (low dword), u32 value} pairs
    def empty(self, n, dtype=torch.int32):
        return torch.empty(int(n), dtype=dtype, device=self.torch_device)

    def _wrap(self, device, t):
        b = Buffer(dtype=np.uint32 if t.dtype == torch.int32 else np.uint64)
        b.setRawPtr(device, t.data_ptr(), t.numel())
        return b

    def partition_msb(self, keys, num_buckets, out=None):
        """keys: int32 CUDA tensor holding u32 bit patterns, or int64 CUDA tensor holding {key, value} pairs.
        Returns (partitioned tensor, int32 CUDA tensor of num_buckets segment sizes).  Runs on the stage of the
        scope we are in."""
        ct = self._ct
        st = self._cur
        n = keys.numel()
        if keys.dtype not in (torch.int32, torch.int64):
            raise TypeError("sharded sort takes int32 (u32 keys) or int64 ({key, value} pairs) tensors, got %s" % keys.dtype)
        pairs = keys.dtype == torch.int64
        if out is None:
            out = self.empty(n, keys.dtype)
        counts = torch.zeros(num_buckets, dtype=torch.int32, device=self.torch_device)
        lib = _lib.load()
        tb = ct.c_size_t()
        wb = ct.c_size_t()
        check(lib.adlhip_radix_sort_scratch_bytes(st.device._h, 1 if pairs else 0, n, ct.byref(tb), ct.byref(wb)), "scratch_bytes")
        if st.work is None or st.work.getSize() < wb.value:
            if st.work is not None:
                DeviceUtils.waitForCompletion(st.device)
                st.work.release()
            st.work = Buffer(st.device, wb.value + wb.value // 8, np.uint8)
        fn = lib.adlhip_partition_msb_kv32 if pairs else lib.adlhip_partition_msb_u32
        check(fn(st.device._h, ct.c_void_p(keys.data_ptr()), ct.c_void_p(out.data_ptr()), ct.c_void_p(counts.data_ptr()),
                 st.work.ptr(), st.work.getSize(), n, int(num_buckets)), "adlhip_partition_msb")
        return out, counts


    def partition_top_byte(self, keys, out=None):
        """As partition_msb, but ordered by the whole top byte and with the 256 top-byte totals (int32 CUDA tensor)
        instead of per-bucket counts: the caller cuts the byte range wherever the global histogram says."""
        ct = self._ct
        st = self._cur
        n = keys.numel()
        if keys.dtype not in (torch.int32, torch.int64):
            raise TypeError("sharded sort takes int32 (u32 keys) or int64 ({key, value} pairs) tensors, got %s" % keys.dtype)
        pairs = keys.dtype == torch.int64
        if out is None:
            out = self.empty(n, keys.dtype)
        totals = torch.empty(256, dtype=torch.int32, device=self.torch_device)
        lib = _lib.load()
        tb = ct.c_size_t()
        wb = ct.c_size_t()
        check(lib.adlhip_radix_sort_scratch_bytes(st.device._h, 1 if pairs else 0, n, ct.byref(tb), ct.byref(wb)), "scratch_bytes")
        if st.work is None or st.work.getSize() < wb.value:
            if st.work is not None:
                DeviceUtils.waitForCompletion(st.device)
                st.work.release()
            st.work = Buffer(st.device, wb.value + wb.value // 8, np.uint8)
        fn = lib.adlhip_partition_top_byte_kv32 if pairs else lib.adlhip_partition_top_byte_u32
        check(fn(st.device._h, ct.c_void_p(keys.data_ptr()), ct.c_void_p(out.data_ptr()), ct.c_void_p(totals.data_ptr()),
                 st.work.ptr(), st.work.getSize(), n), "adlhip_partition_top_byte")
        return out, totals

    def local_sort(self, keys):
        """In-place ascending sort (by key, stable) of an int32 CUDA tensor holding u32 bit patterns or an int64
        CUDA tensor holding {key, value} pairs, on the stage of the scope we are in."""
        n = keys.numel()
        if n:
            st = self._cur
            nbytes = n * keys.element_size()
            if nbytes > st.reserved:
                # received counts jitter from batch to batch: size the scratch with headroom once instead of
                # re-growing it (a device-wide sync + hipMalloc) whenever a slightly larger batch arrives
                cap = n + n // 16 + 1024
                st.reserved = cap * keys.element_size()
                st.pprims.reserve(st.device, ELEM_U32 if keys.dtype == torch.int32 else ELEM_KV32, cap)
            st.pprims.radixSort(st.device, self._wrap(st.device, keys), n)
            # the stage's stream is synchronised by torch, never by adlhip_sync: pick up device-side faults (look-back
            # time-out) of earlier, completed batches here -- stream-ordered, non-blocking
            st.device.checkFault()
        return keys


class ShardedRadixSort:
    """Host logic of the MSB-bucket sharded sort of u32 keys (int32 tensors) or {u32 key, u32 value} pairs (int64
    tensors, key in the low dword; stable: equal keys keep the order (source rank, position in the rank's shard)).
    `backend` supplies empty / partition_msb / local_sort
    (and, for sort_stream, the pipeline plumbing: exchange_scope / sort_scope / event / wait /
    recv_buffer / part_buffer / pipeline_depth)."""

    def __init__(self, backend, group=None, rehearse_buckets=0, balance=True):
        self.backend = backend
        self.group = group
        # balanced splitters: ownership follows the global top-byte histogram
        self.balance = bool(balance) and hasattr(backend, "partition_top_byte")
        self.last_bounds = None     # int64 tensor [G+1] of the last balanced sort: rank g owns top bytes [b[g], b[g+1])
        # development aid: partition into this many buckets even if the world is smaller (the counts are folded back
        # to one per rank), so that a 1-GPU box pays the partition pass an 8-GPU rank would pay
        self.rehearse_buckets = int(rehearse_buckets)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world & (self.world - 1):
            raise ValueError("world size must be a power of two (MSB buckets), got %d" % self.world)
        if self.world > 256:
            raise ValueError("at most 256 ranks")
        self.last_splits = None

    def _partition(self, keys, out=None):
        """Partition by the top bits into one contiguous segment per rank; returns (partitioned, counts[G])."""
        G = self.world
        if self.balance and not self.rehearse_buckets:
            part, totals = self.backend.partition_top_byte(keys, out=out)
            glob = totals.to(torch.int64)
            if dist.is_initialized() and G > 1:
                dist.all_reduce(glob, group=self.group)          # 2 KiB: the global top-byte histogram
            bounds = choose_splitters(glob, G)
            lc = torch.cat([torch.zeros(1, dtype=torch.int64, device=totals.device), torch.cumsum(totals.to(torch.int64), 0)])
            counts = (lc[bounds[1:]] - lc[bounds[:-1]]).to(totals.dtype)
            self.last_bounds = bounds
            return part, counts
        B = max(G, self.rehearse_buckets)
        part, counts = self.backend.partition_msb(keys, B, out=out)
        if B != G:
            counts = counts.view(G, B // G).sum(dim=1, dtype=counts.dtype)
        return part, counts

    def _splits(self, matrix):
        """Count matrix (row r = rank r's send counts) -> (send_splits, recv_splits) of this rank.  The
        .cpu() is the only host sync of a sort; it waits for the stream the matrix was gathered on."""
        G = self.world
        m = matrix.cpu().view(G, G)
        send_splits = [int(x) for x in m[self.rank]]
        recv_splits = [int(x) for x in m[:, self.rank]]  # segments arrive in source-rank order
        self.last_splits = (send_splits, recv_splits)
        return send_splits, recv_splits

    def sort(self, keys, force_exchange=False, marks=None):
        """keys: this rank's shard.  Returns this rank's slice of the globally sorted sequence
        (all keys whose top log2(world) bits equal the rank, ascending).  force_exchange runs the
        partition + collectives even with one rank (used to exercise the code path on a 1-GPU box).
        marks: optional list that receives five backend events (start, partitioned, counts known,
        exchanged, sorted) for a per-stage breakdown."""
        be = self.backend
        G = self.world
        if G == 1 and not (force_exchange and dist.is_initialized()):
            return be.local_sort(keys)
        mark = (lambda: marks.append(be.event(True))) if marks is not None else (lambda: None)
        mark()
        part, counts = self._partition(keys)
        mark()
        # one collective for all split sizes: row r of the matrix = rank r's send counts
        matrix = torch.empty(G * G, dtype=counts.dtype, device=counts.device)
        dist.all_gather_into_tensor(matrix, counts, group=self.group)
        send_splits, recv_splits = self._splits(matrix)
        mark()
        recv = be.empty(sum(recv_splits), keys.dtype)
        dist.all_to_all_single(recv, part, recv_splits, send_splits, group=self.group)
        mark()
        be.local_sort(recv)
        mark()
        return recv

    def sort_stream(self, batches, force_exchange=False):
        """Generator: feeds an iterable of independent batches (this rank's shard of each) through the
        two-stage pipeline and yields, in order, this rank's slice of each batch's globally sorted
        sequence.  Batch i+1's partition + exchange overlap batch i's local sort.

        Every rank must feed the same number of batches, and a batch's keys must stay untouched until its
        result has been yielded.  A yielded tensor lives in one of `backend.pipeline_depth` rotating
        receive slots: at most pipeline_depth - 1 results may be held at a time (pulling result k recycles
        the slot of result k - (pipeline_depth - 1); reads of it enqueued on the caller's stream before
        that pull are safe).  A result is ready on the caller's current stream: no host sync is needed to
        use it."""
        be = self.backend
        G = self.world
        if G == 1 and not (force_exchange and dist.is_initialized()):
            for keys in batches:
                yield be.local_sort(keys)
            return
        depth = int(be.pipeline_depth)
        sorted_ev = [None] * depth      # slot -> event: the local sort that last used the slot has finished
        arrived = None                  # (recv tensor, event: its all-to-all has completed, slot)
        for i, keys in enumerate(batches):
            slot = i % depth
            with be.exchange_scope(after_caller=True):
                part, counts = self._partition(keys, out=be.part_buffer(slot, keys.numel(), keys.dtype))
                matrix = torch.empty(G * G, dtype=counts.dtype, device=counts.device)
                dist.all_gather_into_tensor(matrix, counts, group=self.group)
            out = None
            if arrived is not None:
                # enqueue the previous batch's local sort BEFORE the host blocks on this batch's counts
                out = self._sort_arrived(arrived, sorted_ev)
            with be.exchange_scope():
                send_splits, recv_splits = self._splits(matrix)
                be.wait(sorted_ev[slot])        # the slot's previous occupant has been sorted (and handed out)
                recv = be.recv_buffer(slot, sum(recv_splits), keys.dtype)
                dist.all_to_all_single(recv, part, recv_splits, send_splits, group=self.group)
                arrived = (recv, be.event(), slot)
            if out is not None:
                yield out
        if arrived is not None:
            yield self._sort_arrived(arrived, sorted_ev)

    def _sort_arrived(self, arrived, sorted_ev):
        be = self.backend
        recv, ev, slot = arrived
        with be.sort_scope(after=ev):
            be.local_sort(recv)
            sorted_ev[slot] = be.event()
        be.wait(sorted_ev[slot])        # outside the scopes = the caller's stream: the result is ready there
        return recv
