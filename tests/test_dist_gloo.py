"""CPU-only, world_size 2 over gloo: the host logic of the multi-GPU MSB-bucket sharded sort
(oclradixsort_amd/dist.py: split arithmetic, all-gather of the count matrix, all_to_all_single with
ragged splits, source-rank order of the received segments).  The device work is supplied by a numpy
test backend defined HERE (tests only) -- the product package has no CPU backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


class NumpyBackend:
    """Stands in for HipBackend: same three methods, on CPU tensors, using the oracle."""

    def empty(self, n):
        return torch.empty(int(n), dtype=torch.int32)

    def partition_msb(self, keys, num_buckets):
        k = keys.numpy().view(np.uint32)
        lg = num_buckets.bit_length() - 1
        bucket = (k >> np.uint32(32 - lg)).astype(np.int64) if lg else np.zeros(k.size, dtype=np.int64)
        order = np.argsort(bucket, kind="stable")
        counts = np.bincount(bucket, minlength=num_buckets).astype(np.int32)
        return torch.from_numpy(k[order].view(np.int32).copy()), torch.from_numpy(counts)

    def local_sort(self, keys):
        import oracle
        k = keys.numpy().view(np.uint32)
        k[:] = oracle.sort_u32(k)
        return keys


def _worker(rank, world, port, n_per_rank, skew, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle
        from oclradixsort_amd.dist import ShardedRadixSort
        keys = oracle.keys_u32(n_per_rank, seed=77, first_index=rank * n_per_rank)
        if skew:   # most keys in the low bucket: ragged exchange, one rank receives almost everything
            keys = np.where(np.arange(n_per_rank) % 10 != 0, keys >> np.uint32(3), keys).astype(np.uint32)
        sorter = ShardedRadixSort(NumpyBackend())
        got = sorter.sort(torch.from_numpy(keys.view(np.int32).copy()))
        np.save(os.path.join(out_dir, "out_%d.npy" % rank), got.numpy().view(np.uint32))
        np.save(os.path.join(out_dir, "in_%d.npy" % rank), keys)
        send, recv = sorter.last_splits
        assert sum(send) == n_per_rank and sum(recv) == got.numel()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("skew", [False, True], ids=["uniform", "skewed"])
def test_sharded_sort_world2_gloo(tmp_path, skew):
    import oracle
    world, n = 2, 50021
    mp.spawn(_worker, args=(world, _free_port(), n, skew, str(tmp_path)), nprocs=world, join=True)
    ins = [np.load(tmp_path / ("in_%d.npy" % r)) for r in range(world)]
    outs = [np.load(tmp_path / ("out_%d.npy" % r)) for r in range(world)]
    want = oracle.sort_u32(np.concatenate(ins))
    got = np.concatenate(outs)                       # rank order == global order
    assert np.array_equal(got, want)
    for r, o in enumerate(outs):                     # bucket ownership: top bit == rank
        assert o.size == 0 or ((o >> np.uint32(31)) == r).all()


def test_world_size_must_be_power_of_two():
    from oclradixsort_amd.dist import ShardedRadixSort
    s = ShardedRadixSort(NumpyBackend())             # not initialised: world 1 -> plain local sort
    import oracle
    k = oracle.keys_u32(1000, 3)
    got = s.sort(torch.from_numpy(k.view(np.int32).copy()))
    assert np.array_equal(got.numpy().view(np.uint32), oracle.sort_u32(k))
