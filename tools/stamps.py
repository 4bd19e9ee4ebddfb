#!/usr/bin/env python3
"""Phase timing of the tile body from the diagnostic stamp build (make -C oclradixsort_amd/csrc stamps).
   ADLHIP_LIB=oclradixsort_amd/lib/libadlhip_stamps.so python tools/stamps.py [--configs a:b:t:r,...]
Reads SHARES, not absolute run time (the stamps themselves serialise a little)."""
import argparse, ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 26)
ap.add_argument("--configs", default="0:8:6:1,0:8:1:1")
args = ap.parse_args()
n = args.n
d = DeviceUtils.allocate(); p = Pprims()
lib = _lib.load()
lib.adlhip_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
TILES_MAX = (n + 4095) // 4096
stamps = Buffer(d, TILES_MAX * 16, np.uint64)
old_names = {(11, 0): "ticket", (0, 1): "load", (1, 2): "rank", (2, 3): "barA", (3, 4): "perbin+scan", (4, 5): "lookback",
             (5, 6): "bar", (6, 7): "lds-scatter", (7, 8): "bar", (8, 9): "write-out", (9, 10): "bar"}
new_names = {(0, 2): "load+rank", (2, 3): "barA", (3, 4): "bookkeeping(w0)", (4, 5): "barB", (5, 6): "lds-scatter",
             (6, 7): "lookback(w0)", (7, 9): "barC", (9, 10): "write-out"}
tiles_of = {0: 4096, 1: 8192, 2: 16384, 3: 4096, 4: 8192, 5: 8192, 6: 16384}
buf = Buffer(d, n, np.uint32)
for cfg in args.configs.split(","):
    algo, bits, tile, rank = (int(x) for x in cfg.split(":"))
    d.setParam("sort.algo", algo); d.setParam("sort.digit_bits", bits); d.setParam("sort.tile", tile); d.setParam("sort.rank", rank)
    buf.generate(n, seed=5)
    stamps.clear()
    assert lib.adlhip_debug_set_stamp_buffer(d._h, stamps.ptr()) == 0
    p.radixSort(d, buf, n, 8)          # ONE 8-bit pass so the stamps belong to a single kernel launch
    s = stamps.toHost().reshape(-1, 16)
    nt = (n + tiles_of[tile] - 1) // tiles_of[tile] + 16
    s = s[:nt].astype(np.int64)
    print("config %s: %d tiles of %d keys" % (cfg, nt, tiles_of[tile]))
    names = new_names if algo in (0, 2) else old_names
    for (a, b), nm in names.items():
        ok = (s[:, a] > 0) & (s[:, b] > 0)
        if not ok.any():
            continue
        dlt = (s[ok, b] - s[ok, a])
        print("   %-28s mean %8.0f  p50 %8.0f  p90 %8.0f cycles" % (nm + "(%d>%d)" % (a, b), dlt.mean(), np.median(dlt), np.percentile(dlt, 90)))
    ok = (s[:, 0] > 0) & (s[:, 10] > 0)
    life = (s[ok, 10] - s[ok, 0])
    print("   tile lifetime (0>10) mean %.0f cycles" % life.mean())
buf.release(); stamps.release(); p.close(); DeviceUtils.deallocate(d)
