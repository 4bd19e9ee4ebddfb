import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
for n in (1024, 2048, 3000, 4096, 4097, 6000, 8192, 12000, 16384, 16385):
    bufs = [Buffer(d, n, np.uint32) for _ in range(8)]
    for b in bufs: b.generate(n, seed=n)
    DeviceUtils.waitForCompletion(d)
    best = 1e9
    for t in range(3):
        sw = Stopwatch(d); sw.start()
        for r in range(200): p.radixSort(d, bufs[r % 8], n)
        sw.stop(); best = min(best, sw.getMs() / 200)
    d.toggleProfiling(True); d.profile(reset=True)
    p.radixSort(d, bufs[0], n)
    prof = d.profile(reset=True); d.toggleProfiling(False)
    print(n, "%.1f us" % (best * 1e3), {k: round(v[1] * 1e3, 1) for k, v in prof.items()})
    for b in bufs: b.release()
