#!/usr/bin/env python3
"""Large keys-only sort: ms/sort for u32 keys with "sort.msd2" forced on (2) and off (0), fresh random keys per sort.
MSD2CURVE_SHIFT=s sorts keys >> s (top bits unused, as in a rank of a multi-GPU sort), MSD2CURVE_MODES=1,0 picks the modes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
import oracle
d = DeviceUtils.allocate(); p = Pprims()
shift = int(os.environ.get("MSD2CURVE_SHIFT", "0"))
modes = [int(m) for m in os.environ.get("MSD2CURVE_MODES", "2,0").split(",")]
sizes = [int(float(a)) for a in sys.argv[1:]] or [1 << 22, 6 << 20, 1 << 23, 12 << 20, 1 << 24, 24 << 20, 1 << 25, 48 << 20, 1 << 26]
print("%12s " % "n" + " ".join("%12s" % ("msd2=%d ms" % m) for m in modes) + " " + " ".join("%10s" % "Gkeys/s" for m in modes) + ("   keys >> %d" % shift if shift else ""))
for n in sizes:
    reps = 4
    bufs = [Buffer(d, n, np.uint32) for _ in range(reps)]
    masters = []
    if shift:
        for i in range(reps):
            m = Buffer(d, n, np.uint32)
            m.write(oracle.keys_u32(n, seed=i) >> np.uint32(shift), n)
            masters.append(m)
        DeviceUtils.waitForCompletion(d)
    res = []
    for mode in modes:
        d.setParam("sort.msd2", mode)
        best = 1e9
        for trial in range(5):
            for i, b in enumerate(bufs):
                if shift: b.write(masters[i], n)
                else: b.generate(n, seed=trial * 100 + i, kind=0)
            DeviceUtils.waitForCompletion(d)
            sw = Stopwatch(d); sw.start()
            for r in range(reps): p.radixSort(d, bufs[r], n)
            sw.stop()
            best = min(best, sw.getMs() / reps)
        res.append(best)
    print("%12d " % n + " ".join("%12.4f" % r for r in res) + " " + " ".join("%10.2f" % (n / r / 1e6) for r in res), flush=True)
    for b in bufs + masters: b.release()
d.checkFault()
p.close(); DeviceUtils.deallocate(d)
