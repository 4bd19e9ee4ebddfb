#!/usr/bin/env python3
"""Sort time on non-uniform inputs (64Mi u32): sorted, reverse, all-equal, 16 / 256 / 4096 distinct values, low byte only, one heavy top byte.
Columns: the automatic choice -- the first and the 8th sort of a FRESH handle whose scratch was reserved beforehand (round 4: nothing is
remembered between sorts, so the two differ only by first-use costs of the kernels' code) -- then forced (algo, rank) pairs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
n = 1 << 26
d = DeviceUtils.allocate(); p = Pprims()
base = Buffer(d, n, np.uint32); work = Buffer(d, n, np.uint32)
def make(kind):
    base.generate(n, seed=9)
    if kind == "uniform": return
    if kind == "sorted": p.radixSort(d, base, n); return
    h = None
    if kind == "all_equal": a = np.full(n, 0x12345678, dtype=np.uint32)
    elif kind == "16_values": a = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(28)) * np.uint32(0x11111111)
    elif kind == "low_byte": a = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) >> np.uint32(24)
    elif kind == "256_values": a = ((np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) >> np.uint32(24)) * np.uint32(0x01010101) ^ np.uint32(0x5a5a0000)
    elif kind == "heavy_top_byte":   # 90 % of the keys under one top byte
        base.generate(n, seed=9); a = base.toHost(); a = np.where(np.arange(n) % 10 != 0, (a >> np.uint32(8)) | np.uint32(0x37000000), a).astype(np.uint32)
    elif kind == "4096_values": a = ((np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) >> np.uint32(20)) * np.uint32(0x00100801)
    elif kind == "reverse":
        p.radixSort(d, base, n); a = base.toHost()[::-1].copy()
    base.write(a); DeviceUtils.waitForCompletion(d)
print("%-12s %s" % ("input", "ms/sort by (algo, rank)"))
for kind in ("uniform", "sorted", "reverse", "all_equal", "16_values", "low_byte", "256_values", "heavy_top_byte", "4096_values"):
    make(kind)
    row = []
    d2 = DeviceUtils.allocate(); p2 = Pprims()   # a fresh handle
    p2.reserve(d2, 0, n)
    w2 = Buffer(d2, n, np.uint32)
    times = []
    for t in range(8):
        if t == 0: host = base.toHost()
        w2.write(host, n); DeviceUtils.waitForCompletion(d2)
        sw = Stopwatch(d2); sw.start(); p2.radixSort(d2, w2, n); sw.stop()
        times.append(sw.getMs())
    row.append("auto 1st %.3f 8th %.3f" % (times[0], times[7]))
    w2.release(); p2.close(); DeviceUtils.deallocate(d2)
    for algo, rank in ((0, 1), (0, 0), (1, 1), (1, 0)):
        d.setParam("sort.algo", algo); d.setParam("sort.rank", rank)
        best = 1e9
        for t in range(3):
            work.write(base, n)
            DeviceUtils.waitForCompletion(d)
            sw = Stopwatch(d); sw.start(); p.radixSort(d, work, n); sw.stop()
            best = min(best, sw.getMs())
        row.append("a%d/r%d %.3f" % (algo, rank, best))
        if "--profile" in sys.argv and rank == 1:   # per-kernel breakdown of one more sort
            work.write(base, n); DeviceUtils.waitForCompletion(d)
            d.toggleProfiling(True); d.profile(reset=True)
            p.radixSort(d, work, n); DeviceUtils.waitForCompletion(d)
            prof = d.profile(reset=True); d.toggleProfiling(False)
            row.append("[" + " ".join("%s=%.3f" % (k, ms / c) for k, (c, ms) in prof.items()) + "]")
    print("%-12s %s" % (kind, "   ".join(row)), flush=True)
base.release(); work.release(); p.close(); DeviceUtils.deallocate(d)
