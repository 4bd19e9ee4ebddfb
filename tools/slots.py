#!/usr/bin/env python3
"""Slot occupancy of the one-sweep pass kernel from the diagnostic stamp build: for every tile the time of its
workgroup's FIRST instruction, of the first instruction after the ticket, and of the last write-out instruction, plus
the CU it ran on.  Prints how long a CU's workgroup slots sit between one tile's last instruction and the next
tile's first (stores draining + dispatch), the ticket latency, and the chip-wide occupancy curve.
   make -C oclradixsort_amd/csrc stamps
   ADLHIP_LIB=$PWD/oclradixsort_amd/lib/libadlhip_stamps.so python tools/slots.py [--tile 6]"""
import argparse, ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, _lib
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 26)
ap.add_argument("--tile", type=int, default=6)
ap.add_argument("--slots", type=int, default=2, help="workgroups per CU the variant fits")
args = ap.parse_args()
n = args.n
d = DeviceUtils.allocate(); p = Pprims()
lib = _lib.load()
lib.adlhip_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
stamps = Buffer(d, ((n + 4095) // 4096 + 64) * 16, np.uint64)
d.setParam("sort.algo", 0); d.setParam("sort.tile", args.tile)
buf = Buffer(d, n, np.uint32)
buf.generate(n, seed=5)
stamps.clear()
assert lib.adlhip_debug_set_stamp_buffer(d._h, stamps.ptr()) == 0
p.radixSort(d, buf, n, 8)          # ONE 8-bit pass: the stamps belong to a single launch
s = stamps.toHost().reshape(-1, 16).astype(np.int64)
s = s[(s[:, 11] > 0) & (s[:, 10] > 0)]
print("%d tiles stamped" % len(s))
entry, after_ticket, end = s[:, 11], s[:, 0], s[:, 10]
rt0, rt1 = s[:, 13], s[:, 14]
hw = s[:, 12]
cu_key = ((hw >> 32) & 0xf) * 65536 + ((hw >> 8) & 0xff)      # (xcc, se/sh/cu)
clk = np.median((end - entry) / np.maximum(rt1 - rt0, 1))      # shader cycles per realtime tick
print("CUs seen: %d; shader clock / realtime clock = %.1f (realtime = 100 MHz -> %.2f GHz)" % (len(np.unique(cu_key)), clk, clk / 10))
us = lambda cyc: cyc / (clk * 100.0)
print("ticket (first instruction -> tile known): mean %.2f us  p50 %.2f  p90 %.2f" % (us((after_ticket - entry).mean()),
      us(np.median(after_ticket - entry)), us(np.percentile(after_ticket - entry, 90))))
print("tile lifetime incl. ticket: mean %.2f us" % us((end - entry).mean()))
# per CU: sort tiles by entry; a tile's predecessor in its slot = the tile that ended most recently before it entered
gaps = []
conc = []
for k in np.unique(cu_key):
    m = cu_key == k
    e, x = entry[m], end[m]
    o = np.argsort(e)
    e, x = e[o], x[o]
    free = []          # end times of finished tiles whose slot has not been re-used yet
    import heapq
    running = []       # min-heap of end times of resident tiles
    for i in range(len(e)):
        while running and running[0] <= e[i]:
            free.append(heapq.heappop(running))
        if free:
            # the slot that became free EARLIEST is the one a waiting workgroup takes first
            f = min(free); free.remove(f)
            gaps.append(e[i] - f)
        heapq.heappush(running, x[i])
        conc.append(len(running))
gaps = np.array(gaps)
print("slot turnaround (a tile's last instruction -> next tile's first instruction on that CU): mean %.2f us  p50 %.2f  p90 %.2f  (n=%d)"
      % (us(gaps.mean()), us(np.median(gaps)), us(np.percentile(gaps, 90)), len(gaps)))
print("resident tiles per CU when a tile starts: mean %.2f max %d" % (np.mean(conc), np.max(conc)))
# chip-wide occupancy over (real) time
t0, t1 = rt0.min(), rt1.max()
span = t1 - t0
print("kernel span (first entry -> last end): %.1f us" % (span / 100.0))
grid = np.linspace(t0, t1, 41)
occ = [(np.sum((rt0 <= g) & (rt1 > g))) for g in grid]
print("resident tiles over time (40 steps): " + " ".join(str(o) for o in occ))
buf.release(); stamps.release(); p.close(); DeviceUtils.deallocate(d)
