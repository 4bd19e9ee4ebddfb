#!/usr/bin/env python3
"""Generate tests/golden/* from the REAL reference (oracle/_ref/libref.so).

Run in the development container, where /root/reference exists:
    make -C oracle && python tests/golden/make_golden.py

Every expected output below comes from the reference's own code --
Tahoe::RadixSort::sort (Tahoe/Algorithm/Sort/RadixSort.cpp:10-104) and, for the `host_path`
rows, Pprims::radixSort on an Adl TYPE_HOST device (Tahoe/ParallelPrimitives/Pprims.cpp:202-212,
306-316) -- never from the restatement in oracle/radixsort_oracle.c.  Inputs follow the Demo recipe
(UnitTest/main.cpp:76-86, 105-205: srand(123) per size, getRandom) plus adversarial cases the
reference's own test lacks.  The scan expectation is the test's sequential
running sum (main.cpp:193-199), computed here with numpy.

Outputs (data only):
  demo_table.json          per-size pins for all 11 Demo sizes x {Sort32, SortKeyValue, Scan}:
                           n, first/last input and output elements, FNV-1a-64 of input and output
  demo_small.npz           full input+output vectors for Sort32 n=1024,2048; KV m=1037,2087; Scan n=1024
  adversarial.npz          full input+output vectors for small adversarial cases
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import oracle  # noqa: E402


def fnv(a):
    return "%016x" % oracle.fnv1a64(a)


def main():
    assert oracle.have_ref(), "build oracle/_ref/libref.so first (make -C oracle)"
    table = {"recipe": "UnitTest/main.cpp:76-86,105-205 srand(123) per size; fnv = FNV-1a-64 of raw LE bytes",
             "sort32": [], "sortkv": [], "scan": []}
    small = {}

    n = 1024
    m = 1024
    while n < 2 * 1024 * 1024:
        # --- Demo.Sort32 (main.cpp:113-141)
        a = oracle.demo_u32(n)
        out = oracle.ref_sort_u32(a)
        out_host = oracle.ref_sort_u32(a, host_path=True)
        assert (out == out_host).all()
        table["sort32"].append({"n": n, "in_first3": [int(x) for x in a[:3]], "out_first": int(out[0]),
                                "out_last": int(out[-1]), "fnv_in": fnv(a), "fnv_out": fnv(out)})
        if n <= 2048:
            small["sort32_in_%d" % n] = a
            small["sort32_out_%d" % n] = out
        # --- Demo.SortKeyValue (main.cpp:142-172): testSize += 13 inside the doubling loop
        m = m + 13
        p = oracle.demo_kv32(m)
        pout = oracle.ref_sort_kv32(p)
        pout_host = oracle.ref_sort_kv32(p, host_path=True)
        assert (pout == pout_host).all()
        keys = (pout & 0xffffffff).astype(np.uint32)
        dup = int((keys[1:] == keys[:-1]).sum())
        table["sortkv"].append({"n": m, "out_first": [int(pout[0] & 0xffffffff), int(pout[0] >> 32)],
                                "out_last": [int(pout[-1] & 0xffffffff), int(pout[-1] >> 32)],
                                "dup_key_pairs": dup, "fnv_in": fnv(p), "fnv_out": fnv(pout)})
        if m <= 2100:
            small["sortkv_in_%d" % m] = p
            small["sortkv_out_%d" % m] = pout
        # --- Demo.Scan (main.cpp:173-205); expectation = the test's own running sum
        s = oracle.demo_scan(n)
        ex = np.concatenate([[0], np.cumsum(s.astype(np.uint64))[:-1]]).astype(np.uint32)
        total = int(s.astype(np.uint64).sum() & 0xffffffff)
        table["scan"].append({"n": n, "in_first4": [int(x) for x in s[:4]], "total": total,
                              "fnv_in": fnv(s), "fnv_out": fnv(ex)})
        if n == 1024:
            small["scan_in_1024"] = s
            small["scan_out_1024"] = ex
        n *= 2
        m *= 2

    with open(os.path.join(HERE, "demo_table.json"), "w") as f:
        json.dump(table, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "demo_small.npz"), **small)

    # --- adversarial small cases through the reference
    rng = np.random.RandomState(20260101)
    adv = {}

    def add32(name, a):
        a = np.asarray(a, dtype=np.uint32)
        adv["u32_in_" + name] = a
        adv["u32_out_" + name] = oracle.ref_sort_u32(a) if a.size else a.copy()

    def addkv(name, keys):
        keys = np.asarray(keys, dtype=np.uint64)
        p = keys | (np.arange(keys.size, dtype=np.uint64) << 32)
        adv["kv_in_" + name] = p
        adv["kv_out_" + name] = oracle.ref_sort_kv32(p) if p.size else p.copy()

    cases = {
        "n1": [0xdeadbeef],
        "n2": [5, 3],
        "n255": rng.randint(0, 2**32, 255, dtype=np.uint64),
        "n257": rng.randint(0, 2**32, 257, dtype=np.uint64),
        "all_equal_1000": np.full(1000, 0x12345678),
        "all_ones_513": np.full(513, 0xffffffff),
        "all_zero_300": np.zeros(300),
        "two_values_2000": rng.randint(0, 2, 2000) * 0x80000001,
        "low8_3000": rng.randint(0, 256, 3000),
        "high8_3000": rng.randint(0, 256, 3000).astype(np.uint64) << 24,
        "sorted_1500": np.sort(rng.randint(0, 2**32, 1500, dtype=np.uint64)),
        "reverse_1500": np.sort(rng.randint(0, 2**32, 1500, dtype=np.uint64))[::-1],
        "with_max_keys_777": np.where(rng.rand(777) < 0.3, 0xffffffff, rng.randint(0, 2**32, 777, dtype=np.uint64)),
        "digit_256_same_4099": (rng.randint(0, 2**28, 4099, dtype=np.uint64) << 4) | 0x7,
    }
    for name, vals in cases.items():
        add32(name, vals)
        addkv(name, vals)
    np.savez_compressed(os.path.join(HERE, "adversarial.npz"), **adv)
    print("wrote demo_table.json, demo_small.npz, adversarial.npz")


if __name__ == "__main__":
    main()
