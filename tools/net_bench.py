#!/usr/bin/env python3
"""Time of the large sort's safety net (the cooperative LSD sort inside the offsets kernel): 64 Mi u32 keys that do not fit the
slabs, the large sort forced ("sort.msd2" = 2), every result compared with numpy's sort; then u64 keys and pairs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
d = DeviceUtils.allocate(); p = Pprims()
d.setParam("sort.msd2", 2)
i = np.arange(n, dtype=np.uint32)
h = i * np.uint32(2654435761)
kinds = {
    "all_equal": np.full(n, 0x12345678, dtype=np.uint32),
    "16_values": (h >> np.uint32(28)) * np.uint32(0x11111111),
    "256_values": ((h >> np.uint32(24)) * np.uint32(0x01010101)) ^ np.uint32(0x5a5a0000),
    "4096_values": (h >> np.uint32(20)) * np.uint32(0x00100801),
    "heavy_top_byte": np.where(i % 10 != 0, (h >> np.uint32(8)) | np.uint32(0x37000000), h * np.uint32(40503)).astype(np.uint32),
}
for dtype, name in ((np.uint32, "u32"), (np.uint64, "u64"), (None, "kv32")):
    for kind, a in kinds.items():
        if dtype is None:
            host = a.astype(np.uint64) | (i.astype(np.uint64) << np.uint64(32))
            want = host[np.argsort(a, kind="stable")]
        elif dtype == np.uint64:
            host = (a.astype(np.uint64) << np.uint64(32)) | a.astype(np.uint64)
            want = np.sort(host)
        else:
            host = a
            want = np.sort(a)
        b = Buffer(d, n, host.dtype)
        times = []
        for t in range(3):
            b.write(host); DeviceUtils.waitForCompletion(d)
            sw = Stopwatch(d); sw.start()
            (p.radixSort64 if dtype == np.uint64 else p.radixSort)(d, b, n)
            sw.stop(); times.append(sw.getMs())
        ok = np.array_equal(b.toHost(), want)
        d.toggleProfiling(True); d.profile(reset=True)
        b.write(host); (p.radixSort64 if dtype == np.uint64 else p.radixSort)(d, b, n); DeviceUtils.waitForCompletion(d)
        prof = d.profile(reset=True); d.toggleProfiling(False)
        print("%-5s %-15s %s  %s  [%s]" % (name, kind, " ".join("%.3f" % t for t in times), "OK" if ok else "MISMATCH",
                                         " ".join("%s=%.3f" % (k, ms / c) for k, (c, ms) in prof.items())), flush=True)
        b.release()
p.close(); DeviceUtils.deallocate(d)
