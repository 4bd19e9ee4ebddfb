"""Multi-GPU sharded radix sort: one process per GPU, MSB-bucket exchange over torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm).  No reference counterpart -- the reference drives exactly
one device (Adl/Adl.h:90-94); SURVEY.md section 8e describes this step.

Radix order is decided first by the most significant bits, so rank g of G owns the keys whose top
log2(G) bits equal g.  Per sort:
  1. local stable partition of the rank's keys into G contiguous segments by their top bits
     (adlhip_partition_msb_u32: one count -> scan -> scatter pass on the top byte);
  2. ONE small all-gather of the G segment sizes (G x G matrix) -- the only host sync;
  3. ONE all_to_all_single of the segments with those split sizes.  On a fully connected xGMI node
     every pair of GPUs has its own link, so all G-1 links of a GPU carry traffic at once;
  4. local radix sort of the received keys (adlhip_radix_sort_u32).
Concatenating the ranks' outputs in rank order is the globally sorted array.  Receive buffers are
sized from the exchanged counts, never from n/G.

The device work is delegated to a `backend` object so that the host logic above (split arithmetic,
exchange, ordering) can be exercised on CPU with gloo in the tests.  The product backend is
HipBackend; there is no CPU backend in this package.
"""
import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .adl import Buffer, Config, DeviceUtils
from .pprims import Pprims


class HipBackend:
    """Device half of the sharded sort on one MI355X: torch owns the memory, libadlhip.so does the work
    on torch's current stream (so RCCL collectives and sort kernels are ordered without extra syncs)."""

    def __init__(self, local_rank=0):
        import ctypes
        self._ct = ctypes
        torch.cuda.set_device(local_rank)
        self.torch_device = torch.device("cuda", local_rank)
        stream = torch.cuda.current_stream().cuda_stream
        self.device = DeviceUtils.allocate(cfg=Config(local_rank), stream=stream)
        self.pprims = Pprims()
        self._counts = None
        self._work = None

    def close(self):
        self.pprims.close()
        if self._work is not None:
            self._work.release()
            self._work = None
        DeviceUtils.deallocate(self.device)

    def empty(self, n):
        return torch.empty(int(n), dtype=torch.int32, device=self.torch_device)

    def _wrap(self, t):
        b = Buffer(dtype=np.uint32)
        b.setRawPtr(self.device, t.data_ptr(), t.numel())
        return b

    def partition_msb(self, keys, num_buckets):
        """keys: int32 CUDA tensor holding u32 bit patterns.  Returns (partitioned tensor, int32 CUDA
        tensor of num_buckets segment sizes)."""
        ct = self._ct
        n = keys.numel()
        out = self.empty(n)
        counts = torch.zeros(num_buckets, dtype=torch.int32, device=self.torch_device)
        lib = _lib.load()
        tb = ct.c_size_t()
        wb = ct.c_size_t()
        check(lib.adlhip_radix_sort_scratch_bytes(self.device._h, 0, n, ct.byref(tb), ct.byref(wb)), "scratch_bytes")
        if self._work is None or self._work.getSize() < wb.value:
            if self._work is not None:
                self._work.release()
            self._work = Buffer(self.device, wb.value, np.uint8)
        check(lib.adlhip_partition_msb_u32(self.device._h, ct.c_void_p(keys.data_ptr()), ct.c_void_p(out.data_ptr()),
                                           ct.c_void_p(counts.data_ptr()), self._work.ptr(), self._work.getSize(),
                                           n, int(num_buckets)), "adlhip_partition_msb_u32")
        return out, counts

    def local_sort(self, keys):
        """In-place ascending sort of an int32 CUDA tensor holding u32 bit patterns."""
        if keys.numel():
            self.pprims.radixSort(self.device, self._wrap(keys), keys.numel())
        return keys


class ShardedRadixSort:
    """Host logic of the MSB-bucket sharded sort.  `backend` supplies empty / partition_msb / local_sort."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world & (self.world - 1):
            raise ValueError("world size must be a power of two (MSB buckets), got %d" % self.world)
        if self.world > 256:
            raise ValueError("at most 256 ranks")
        self.last_splits = None

    def sort(self, keys, force_exchange=False):
        """keys: this rank's shard.  Returns this rank's slice of the globally sorted sequence
        (all keys whose top log2(world) bits equal the rank, ascending).  force_exchange runs the
        partition + collectives even with one rank (used to exercise the code path on a 1-GPU box)."""
        G = self.world
        if G == 1 and not (force_exchange and dist.is_initialized()):
            return self.backend.local_sort(keys)
        part, counts = self.backend.partition_msb(keys, G)
        # one collective for all split sizes: row r of the matrix = rank r's send counts
        matrix = torch.empty(G * G, dtype=counts.dtype, device=counts.device)
        dist.all_gather_into_tensor(matrix, counts, group=self.group)
        m = matrix.cpu().view(G, G)                      # the only host sync of the sort
        send_splits = [int(x) for x in m[self.rank]]
        recv_splits = [int(x) for x in m[:, self.rank]]  # segments arrive in source-rank order
        self.last_splits = (send_splits, recv_splits)
        recv = self.backend.empty(sum(recv_splits))
        dist.all_to_all_single(recv, part, recv_splits, send_splits, group=self.group)
        return self.backend.local_sort(recv)
