#!/usr/bin/env python3
"""Work-buffer levels (adlhip_radix_sort_scratch_bytes_for): size and speed of whole-key sorts with the full-speed (1), the lean (2)
and the minimum (0) work buffer.   python tools/r3_lean_scratch.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oclradixsort_amd import Buffer, DeviceUtils, _lib
from oclradixsort_amd._lib import check
lib = _lib.load()
d = DeviceUtils.allocate()
d.setParam("sort.algo", -1)
def sizes(kind, n, bits, level):
    tb, wb = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.adlhip_radix_sort_scratch_bytes_for(d._h, kind, n, bits, level, ctypes.byref(tb), ctypes.byref(wb)), "sizes")
    return wb.value
print("%-4s %12s %6s %12s %10s %10s" % ("kind", "n", "level", "work MB", "ms/sort", "G/s"))
for kind, name, dtype, bits, ns in ((0, "u32", np.uint32, 32, (1 << 22, 1 << 24, 1 << 26, 1 << 27)), (2, "u64", np.uint64, 64, (1 << 22, 1 << 24, 1 << 25))):
    for n in ns:
        keys = [oracle.keys_u32(n, seed=s) if kind == 0 else oracle.keys_u64(n, seed=s) for s in (1, 2, 3, 4)]
        fn = lib.adlhip_radix_sort_u32 if kind == 0 else lib.adlhip_radix_sort_u64
        bufs = [Buffer(d, n, dtype) for _ in keys]
        tmp = Buffer(d, n, dtype)
        for level in (1, 2, 0):
            wb = sizes(kind, n, bits, level)
            work = Buffer(d, wb, np.uint8)
            best = 1e9
            for rep in range(4):
                for b, k in zip(bufs, keys): b.write(k)
                DeviceUtils.waitForCompletion(d)
                t0 = time.perf_counter()
                for b in bufs: check(fn(d._h, b.ptr(), tmp.ptr(), work.ptr(), wb, n, bits), "sort")
                DeviceUtils.waitForCompletion(d)
                best = min(best, (time.perf_counter() - t0) / len(bufs))
            out = bufs[0].toHost()
            assert np.all(out[1:] >= out[:-1])
            print("%-4s %12d %6d %12.1f %10.3f %10.1f" % (name, n, level, wb / 1e6, best * 1e3, n / best / 1e9), flush=True)
            work.release()
        for b in bufs: b.release()
        tmp.release()
DeviceUtils.deallocate(d)
