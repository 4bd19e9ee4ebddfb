#!/usr/bin/env python3
"""Timing of adlhip_segment_sort alone (pass C of the hybrid sort): 16384 segments of ~n/16384 elements whose
low `bits` bits are random.  usage: python tools/segment_sort_bench.py [--n N] [--kind 0|1] [--bits 18] [--cap 8192]"""
import argparse, ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Stopwatch, _lib
from oclradixsort_amd._lib import check
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 26)
ap.add_argument("--kind", type=int, default=0)
ap.add_argument("--bits", type=int, default=18)
ap.add_argument("--cap", type=int, default=0)
ap.add_argument("--segments", type=int, default=16384)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
lib = _lib.load()
d = DeviceUtils.allocate()
S, n = a.segments, a.n
rng = np.random.RandomState(1)
sizes = rng.multinomial(n, np.full(S, 1.0 / S)).astype(np.int64)      # what uniform keys give
starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
cap = a.cap or (1 << int(np.ceil(np.log2(sizes.max()))))
sb = Buffer(d, S + 1, np.uint32); sb.write(starts)
dtype = np.uint32 if a.kind == 0 else np.uint64
bufs = [Buffer(d, n, dtype) for _ in range(4)]
for i, b in enumerate(bufs): b.generate(n, seed=10 + i, kind=0 if a.kind == 0 else 1)
DeviceUtils.waitForCompletion(d)
def run():
    sw = Stopwatch(d); sw.start()
    for i in range(a.steps):
        check(lib.adlhip_segment_sort(d._h, a.kind, bufs[i % 4].ptr(), sb.ptr(), S, cap, a.bits), "segment_sort")
    sw.stop()
    return sw.getMs() / a.steps
run()
for i, b in enumerate(bufs): b.generate(n, seed=20 + i, kind=0 if a.kind == 0 else 1)
DeviceUtils.waitForCompletion(d)
ms = run()
esz = 4 if a.kind == 0 else 8
print("segment_sort kind=%d n=%d segments=%d (max %d, tile %d) low_bits=%d: %.1f us  %.1f GB/s read+write  %.1f Gelem/s"
      % (a.kind, n, S, sizes.max(), cap, a.bits, ms * 1e3, 2.0 * n * esz / ms / 1e6, n / ms / 1e6))
for b in bufs + [sb]: b.release()
DeviceUtils.deallocate(d)
