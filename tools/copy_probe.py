#!/usr/bin/env python3
"""Ceiling check: device-to-device copy of 64Mi u32 (256 MiB read + 256 MiB written) with torch's copy kernel, and a strided
gather of 4 KiB chunks out of 5 KiB slabs (the large sort's finish reads that shape)."""
import torch, time
n = 1 << 26
a = torch.randint(0, 2**31 - 1, (n,), dtype=torch.int32, device="cuda")
b = torch.empty_like(a)
slab = torch.randint(0, 2**31 - 1, (65536, 1280), dtype=torch.int32, device="cuda")
out = torch.empty((65536, 1024), dtype=torch.int32, device="cuda")
def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
t = timeit(lambda: b.copy_(a)); print("linear copy 64Mi u32        %.4f ms  %.0f GB/s read+write" % (t, 2 * n * 4 / t / 1e6))
t = timeit(lambda: out.copy_(slab[:, :1024])); print("slab gather 65536 x 1024/1280 %.4f ms  %.0f GB/s read+write" % (t, 2 * n * 4 / t / 1e6))
t = timeit(lambda: b.fill_(1)); print("fill 64Mi u32               %.4f ms  %.0f GB/s write" % (t, n * 4 / t / 1e6))
t = timeit(lambda: a.sum()); print("sum 64Mi i32                %.4f ms  %.0f GB/s read" % (t, n * 4 / t / 1e6))
