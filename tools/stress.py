#!/usr/bin/env python3
"""Soak test of the inter-workgroup look-back protocol: many back-to-back one-sweep sorts of random sizes,
element kinds and key distributions, every result checked (sortedness + multiset checksum on the device
result copied back, full oracle comparison for the smaller ones), fault word checked at every sync.  Every eighth sort
is followed -- while later sorts are already queued -- by the LDS-order self-test on a SECOND handle (the hardware
behaviour "sort.rank" = 1 rests on, adlhip_selftest_lds_order), and one size class reaches past 64 Mi keys (the
pointer-store write-out).  Half of the sorts take the automatic choice (mid-size sort, large sort with its look-back /
cursor passes and the safety net inside its offsets kernel), with "sort.msd2" forced on for some and keys shifted down by a random number of bits.
   python tools/stress.py [--seconds 60]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
import ctypes
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, _lib
from oclradixsort_amd._lib import check
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--skip-before", type=int, default=0, help="replay: draw the first iterations without running them")
ap.add_argument("--stop-after", type=int, default=1 << 30); ap.add_argument("--verbose", action="store_true")
ap.add_argument("--seed", type=int, default=2026)
ap.add_argument("--mix-rank", action="store_true", help="also draw sort.rank per sort (0 = ballot ranking: the stable large sort's ballot variants)")
args = ap.parse_args()
mix_rng = np.random.RandomState(args.seed + 1)
skip_before, verbose = args.skip_before, args.verbose
d = DeviceUtils.allocate(); p = Pprims()
d2 = DeviceUtils.allocate(); selftests = 0
rng = np.random.RandomState(args.seed)
t_end = time.time() + args.seconds
it = 0; elems = 0; t_mark = time.time()
def checks(a):
    a64 = a.astype(np.uint64)
    return int(a64.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(a64)) if a.size else 0
while time.time() < t_end and it < args.stop_after:
    it += 1
    kind = rng.choice(["u32", "kv", "u64", "soa", "soaw"])
    n = int(2 ** rng.uniform(10, 25.5)) + int(rng.randint(0, 1000))
    if it % 7 == 0: n = int(2 ** rng.uniform(21, 26.3))   # more of the large sort's range
    if kind == "u32" and it % 23 == 0: n = (1 << 26) + int(rng.randint(1, 1 << 22))     # past 256 MiB: pointer stores
    if kind == "u32" and it % 37 == 0: n = int(2 ** rng.uniform(27.1, 28.6))            # 140 Mi ... 400 Mi keys: segments finished by a workgroup each
    algo = int(rng.choice([0, 0, 1, -1, -1, -1])); bits = int(rng.choice([8, 8, 8, 4])); tile = int(rng.choice([-1, -1, 0, 1, 2, 5]))
    if algo < 0: bits, tile = 8, -1   # what the automatic paths run with
    d.setParam("sort.algo", algo); d.setParam("sort.digit_bits", bits); d.setParam("sort.tile", tile)
    if args.mix_rank: d.setParam("sort.rank", int(mix_rng.choice([1, 1, 0])))   # ballot ranking on a third of the sorts (own generator: replays stay aligned)
    d.setParam("sort.msd2", int(rng.choice([1, 1, 1, 2, 3, 4, 5])))   # automatic or a forced form
    dist = rng.choice(["uniform", "lowbits", "fewvals", "sortedish", "shifted", "shifted", "heavy", "vals4096"])
    shift = int(rng.randint(1, 20))
    msd2_mode = d.getParam("sort.msd2")
    if it < skip_before:   # replay: only the draws of the generator (they do not depend on any result)
        if kind == "u32":
            if it % 5 == 0 and n < (1 << 22): rng.choice([16, 20, 24, 28])
            else: rng.randint(1, 4)
        continue
    if verbose: print("it %d %s n=%d algo=%d bits=%d tile=%d msd2=%d dist=%s shift=%d" % (it, kind, n, algo, bits, tile, msd2_mode, dist, shift), flush=True)
    if kind in ("u32", "soa", "kv", "soaw"):
        k = oracle.keys_u32(n, seed=it)
        if dist == "heavy": k = np.where(np.arange(n) % 10 != 0, (k >> np.uint32(8)) | np.uint32(0x37000000), k).astype(np.uint32)
        elif dist == "vals4096": k = (k >> np.uint32(20)) * np.uint32(0x00100801)
        if dist == "lowbits": k &= np.uint32(0xffff)
        elif dist == "fewvals": k = (k % np.uint32(5)) * np.uint32(0x01010101)
        elif dist == "sortedish": k = np.sort(k)
        elif dist == "shifted": k >>= np.uint32(shift)
    if kind == "u32":
        b = Buffer(d, n, np.uint32); b.write(k)
        if it % 5 == 0 and n < (1 << 22):   # a sort on part of the key (stable form of the large sort above 2 Mi keys)
            sb = int(rng.choice([16, 20, 24, 28]))
            p.radixSort(d, b, n, sb)
            out = b.toHost(); b.release()
            assert np.array_equal(out, oracle.sort_u32_bits(k, sb)), (it, kind, n, sb, dist)
            elems += n
            continue
        reps = int(rng.randint(1, 4))
        for _ in range(reps): p.radixSort(d, b, n)          # re-sorting sorted data stresses the low-entropy paths
        if it % 8 == 0:   # the ranking's hardware assumption, checked on another stream while these sorts run
            mism = ctypes.c_uint32(1)
            check(_lib.load().adlhip_selftest_lds_order(d2._h, 256, ctypes.byref(mism)), "selftest")
            assert mism.value == 0, ("lds order self-test", it)
            selftests += 1
        out = b.toHost(); b.release()
        assert np.all(out[1:] >= out[:-1]) and checks(out) == checks(k), (it, kind, n, algo, bits, tile, dist)
        if n < (1 << 22): assert np.array_equal(out, oracle.sort_u32(k)), (it, kind, n)
    elif kind == "kv":
        pr = k.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
        b = Buffer(d, n, np.uint64); b.write(pr); p.radixSort(d, b, n); out = b.toHost(); b.release()
        kk = out & np.uint64(0xffffffff); vv = out >> np.uint64(32)
        assert np.all(kk[1:] >= kk[:-1]); same = kk[1:] == kk[:-1]
        assert np.all(vv[1:][same] > vv[:-1][same]) and checks(out) == checks(pr), (it, kind, n, algo, bits, tile, dist)
    elif kind == "soa":
        v = np.arange(n, dtype=np.uint32)
        kb = Buffer(d, n, np.uint32); vb = Buffer(d, n, np.uint32); kb.write(k); vb.write(v)
        p.radixSortSoA(d, kb, vb, n); ok, ov = kb.toHost(), vb.toHost(); kb.release(); vb.release()
        assert np.all(ok[1:] >= ok[:-1]); same = ok[1:] == ok[:-1]
        assert np.all(ov[1:][same] > ov[:-1][same]) and np.array_equal(k[ov], ok), (it, kind, n, algo, bits, tile, dist)
    elif kind == "soaw":   # wide values / 64-bit keys on separate arrays (adlhip_radix_sort_soa): index sort + gather
        n = min(n, 1 << 23)
        k = k[:n]
        kd = np.uint64 if it % 2 else np.uint32
        vd = [np.uint64, np.uint32, np.dtype([("a", "<u8"), ("b", "<u8")])][it % 3]
        keys = k.astype(np.uint64) * np.uint64(0x100000001) >> np.uint64(it % 29) if kd == np.uint64 else k
        vals = np.zeros(n, dtype=vd)
        vals.view(np.uint32).reshape(n, -1)[:, 0] = np.arange(n, dtype=np.uint32)
        kb = Buffer(d, n, kd); vb = Buffer(d, n, vd); kb.write(keys); vb.write(vals)
        p.radixSortSoA(d, kb, vb, n); ok, ov = kb.toHost(), vb.toHost(); kb.release(); vb.release()
        idx = ov.view(np.uint32).reshape(n, -1)[:, 0].astype(np.int64)
        assert np.all(ok[1:] >= ok[:-1]); same = ok[1:] == ok[:-1]
        assert np.all(idx[1:][same] > idx[:-1][same]) and np.array_equal(keys[idx], ok), (it, kind, n, kd, vd, dist)
    else:
        k64 = oracle.keys_u64(n, seed=it)
        if dist == "lowbits": k64 &= np.uint64(0xffffff)
        elif dist == "shifted": k64 >>= np.uint64(2 * shift)
        elif dist == "fewvals": k64 = (k64 % np.uint64(7)) * np.uint64(0x0101010101010101)
        b = Buffer(d, n, np.uint64); b.write(k64); p.radixSort64(d, b, n); out = b.toHost(); b.release()
        assert np.all(out[1:] >= out[:-1]) and checks(out) == checks(k64), (it, kind, n, algo, bits, tile, dist)
    elems += n
    if time.time() - t_mark > 60:   # a sign of life for the runner
        t_mark = time.time(); print("... %d sorts, %.0f M elements" % (it, elems / 1e6), flush=True)
print("stress ok: %d sorts, %.1f M elements, %.0f s, no mismatch, no look-back fault, %d LDS-order self-tests beside running sorts: 0 mismatches"
      % (it, elems / 1e6, args.seconds, selftests))
print("safety net of the large sort: ran %d times, %d of them sorted by counting" % (d.getParam("stat.net_runs"), d.getParam("stat.net_counting")))
p.close(); DeviceUtils.deallocate(d); DeviceUtils.deallocate(d2)
