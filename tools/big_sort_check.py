#!/usr/bin/env python3
"""One-off check of the n >= 2^30 path (three-kernel passes, 32-bit indices close to their limit): sort n = 2^30 + 12345
u32 keys, verify sortedness and the multiset (count, sum, xor) chunk by chunk against the index-addressable generator."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
n = (1 << 30) + 12345
d = DeviceUtils.allocate(); p = Pprims()
b = Buffer(d, n, np.uint32)
b.generate(n, seed=98)
p.radixSort(d, b, n)          # warm-up: the scratch buffers (4 GiB + tables) are allocated here
b.generate(n, seed=99)
DeviceUtils.waitForCompletion(d)
sw = Stopwatch(d); sw.start()
p.radixSort(d, b, n)
sw.stop(); DeviceUtils.waitForCompletion(d)
print("sorted %d keys in %.2f ms (%.1f Gkeys/s)" % (n, sw.getMs(), n / sw.getMs() / 1e6), flush=True)
CH = 1 << 26
s_in = np.uint64(0); x_in = np.uint64(0); s_out = np.uint64(0); x_out = np.uint64(0)
prev_last = None
ok = True
chunk = np.empty(CH, dtype=np.uint32)
t0 = time.time()
for off in range(0, n, CH):
    m = min(CH, n - off)
    b.read(chunk[:m], m, off); DeviceUtils.waitForCompletion(d)
    c = chunk[:m]
    ok &= bool(np.all(c[1:] >= c[:-1]))
    if prev_last is not None: ok &= bool(c[0] >= prev_last)
    prev_last = c[-1]
    s_out += c.astype(np.uint64).sum(dtype=np.uint64); x_out ^= np.bitwise_xor.reduce(c.astype(np.uint64))
    k = oracle.keys_u32(m, seed=99, first_index=off)
    s_in += k.astype(np.uint64).sum(dtype=np.uint64); x_in ^= np.bitwise_xor.reduce(k.astype(np.uint64))
    print("  chunk at %d ok=%s (%.0f s)" % (off, ok, time.time() - t0), flush=True)
print("sorted:", ok, " multiset (sum, xor) equal:", bool(s_in == s_out and x_in == x_out))
b.release(); p.close(); DeviceUtils.deallocate(d)
assert ok and s_in == s_out and x_in == x_out
