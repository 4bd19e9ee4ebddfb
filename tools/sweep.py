#!/usr/bin/env python3
"""Sweep sort variants on one GPU (diagnostic): prints ms/sort, Gkeys/s and per-kernel average ms.
usage: python tools/sweep.py [--n N] [--kind u32|kv|u64] [--steps K] [--configs algo:bits:tile:rank,...]"""
import argparse, os, sys, itertools, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 26)
ap.add_argument("--kind", default="u32")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--configs", default="")
ap.add_argument("--verify", action="store_true")
ap.add_argument("--param", action="append", default=[], help="name=value passed to adlhip_set_param")
args = ap.parse_args()
n, K = args.n, args.steps
d = DeviceUtils.allocate()
for kv in args.param:
    k, v = kv.split('='); d.setParam(k, int(v))
p = Pprims()
dtype = np.uint32 if args.kind in ("u32", "soa") else np.uint64
gen_kind = {"u32": 0, "kv": 1, "u64": 2, "soa": 0}[args.kind]
bufs = [Buffer(d, n, dtype) for _ in range(K)]
vbufs = [Buffer(d, n, np.uint32) for _ in range(K)] if args.kind == "soa" else []
if args.configs:
    configs = [tuple(int(x) for x in c.split(":")) for c in args.configs.split(",")]
else:
    configs = [(a, 8, t, r) for a in (0, 1) for t in range(6) for r in (1, 0)]
want = None
if args.verify:
    import oracle
    if args.kind == "u32": want = oracle.sort_u32(oracle.keys_u32(n, 1000))
    elif args.kind == "kv": want = oracle.sort_kv32(oracle.pairs_kv32(n, 1000))
    elif args.kind == "soa":
        pr = oracle.keys_u32(n, 1000).astype(np.uint64) | (oracle.keys_u32(n, 5000).astype(np.uint64) << np.uint64(32))
        want = (oracle.sort_kv32(pr) & np.uint64(0xffffffff)).astype(np.uint32)
    else: want = oracle.sort_u64(oracle.keys_u64(n, 1000))
print("%-28s %9s %9s  %s" % ("algo:bits:tile:rank", "ms/sort", "G/s", "per-kernel avg ms"))
for (algo, bits, tile, rank) in configs:
    d.setParam("sort.algo", algo); d.setParam("sort.digit_bits", bits); d.setParam("sort.tile", tile)
    try:
        d.setParam("sort.rank", rank)
    except Exception as e:
        print("rank", rank, "unavailable:", e); continue
    def run(profile):
        for i, b in enumerate(bufs):
            b.generate(n, seed=1000 + i, kind=gen_kind)
        for i, b in enumerate(vbufs):
            b.generate(n, seed=5000 + i, kind=0)
        DeviceUtils.waitForCompletion(d)
        sw = Stopwatch(d)
        if profile:
            d.toggleProfiling(True); d.profile(reset=True)
        sw.start()
        for i, b in enumerate(bufs):
            if args.kind == "u64": p.radixSort64(d, b, n)
            elif args.kind == "soa": p.radixSortSoA(d, b, vbufs[i], n)
            else: p.radixSort(d, b, n)
        sw.stop()
        ms = sw.getMs() / K
        prof = None
        if profile:
            prof = d.profile(reset=True); d.toggleProfiling(False)
        return ms, prof
    try:
        run(False)                       # warm-up
        ms, _ = run(False)
        ok = ""
        if want is not None:
            ok = " OK" if np.array_equal(bufs[0].toHost(), want) else " MISMATCH"
        _, prof = run(True)
        ks = " ".join("%s=%.3f" % (k, v[1] / v[0]) for k, v in prof.items())
        print("%-28s %9.3f %9.2f  %s%s" % ("%d:%d:%d:%d" % (algo, bits, tile, rank), ms, n / ms / 1e6, ks, ok), flush=True)
    except Exception as e:
        print("%-28s FAILED %s" % ("%d:%d:%d:%d" % (algo, bits, tile, rank), e), flush=True)
for b in bufs + vbufs: b.release()
p.close(); DeviceUtils.deallocate(d)
