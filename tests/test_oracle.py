"""CPU-only: pins the oracle (oracle/radixsort_oracle.c) against the golden vectors produced by the
reference (tests/golden/make_golden.py) and, where oracle/_ref/libref.so is present, against the
reference itself on fresh seeded inputs."""
import json
import os

import numpy as np
import pytest

import oracle


@pytest.fixture(scope="module")
def table(golden_dir):
    with open(os.path.join(golden_dir, "demo_table.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def small(golden_dir):
    return np.load(os.path.join(golden_dir, "demo_small.npz"))


@pytest.fixture(scope="module")
def adversarial(golden_dir):
    return np.load(os.path.join(golden_dir, "adversarial.npz"))


def test_demo_inputs_match_golden_prefix(table):
    # the Demo recipe depends on glibc rand(); make sure this host reproduces the recorded inputs
    row = table["sort32"][0]
    a = oracle.demo_u32(row["n"])
    assert [int(x) for x in a[:3]] == row["in_first3"]
    assert "%016x" % oracle.fnv1a64(a) == row["fnv_in"]


def test_sort32_full_vectors(small):
    for n in (1024, 2048):
        got = oracle.sort_u32(small["sort32_in_%d" % n])
        assert np.array_equal(got, small["sort32_out_%d" % n])


def test_sortkv_full_vectors(small):
    for m in (1037, 2087):
        got = oracle.sort_kv32(small["sortkv_in_%d" % m])
        assert np.array_equal(got, small["sortkv_out_%d" % m])


def test_scan_full_vector(small):
    got, total = oracle.exclusive_scan_u32(small["scan_in_1024"])
    assert np.array_equal(got, small["scan_out_1024"])
    assert total == int(small["scan_in_1024"].astype(np.uint64).sum())


def test_demo_table_all_sizes(table):
    """All 11 Demo sizes x 3 primitives: hashes of the oracle's outputs equal the reference's."""
    for row in table["sort32"]:
        a = oracle.demo_u32(row["n"])
        assert "%016x" % oracle.fnv1a64(a) == row["fnv_in"]
        out = oracle.sort_u32(a)
        assert int(out[0]) == row["out_first"] and int(out[-1]) == row["out_last"]
        assert "%016x" % oracle.fnv1a64(out) == row["fnv_out"]
    for row in table["sortkv"]:
        p = oracle.demo_kv32(row["n"])
        assert "%016x" % oracle.fnv1a64(p) == row["fnv_in"]
        out = oracle.sort_kv32(p)
        assert [int(out[0] & 0xffffffff), int(out[0] >> 32)] == row["out_first"]
        assert [int(out[-1] & 0xffffffff), int(out[-1] >> 32)] == row["out_last"]
        assert "%016x" % oracle.fnv1a64(out) == row["fnv_out"]
    for row in table["scan"]:
        s = oracle.demo_scan(row["n"])
        assert [int(x) for x in s[:4]] == row["in_first4"]
        out, total = oracle.exclusive_scan_u32(s)
        assert total == row["total"]
        assert "%016x" % oracle.fnv1a64(out) == row["fnv_out"]


def test_adversarial_vectors(adversarial):
    names = sorted(k[len("u32_in_"):] for k in adversarial.files if k.startswith("u32_in_"))
    assert len(names) >= 10
    for nm in names:
        assert np.array_equal(oracle.sort_u32(adversarial["u32_in_" + nm]), adversarial["u32_out_" + nm]), nm
        assert np.array_equal(oracle.sort_kv32(adversarial["kv_in_" + nm]), adversarial["kv_out_" + nm]), nm


def test_oracle_is_the_unique_stable_sort():
    rng = np.random.RandomState(7)
    for n in (0, 1, 2, 63, 64, 65, 1000, 4097, 100003):
        k = rng.randint(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(oracle.sort_u32(k), np.sort(k, kind="stable"))
        keys = rng.randint(0, 50, n).astype(np.uint64)          # many duplicates -> stability matters
        pairs = keys | (np.arange(n, dtype=np.uint64) << 32)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(oracle.sort_kv32(pairs), pairs[order])
        k64 = rng.randint(0, 2**63, n, dtype=np.uint64) * 2 + rng.randint(0, 2, n).astype(np.uint64)
        assert np.array_equal(oracle.sort_u64(k64), np.sort(k64))


def test_partial_bits_oracle():
    rng = np.random.RandomState(11)
    k = rng.randint(0, 2**32, 5000, dtype=np.uint64).astype(np.uint32)
    for bits in range(4, 33, 4):
        mask = np.uint32((1 << bits) - 1) if bits < 32 else np.uint32(0xffffffff)
        order = np.argsort(k & mask, kind="stable")
        assert np.array_equal(oracle.sort_u32_bits(k, bits), k[order]), bits
    p = (k.astype(np.uint64)) | (np.arange(k.size, dtype=np.uint64) << 32)
    for bits in (4, 12, 20, 32):
        mask = np.uint64((1 << bits) - 1)
        order = np.argsort(p & mask, kind="stable")
        assert np.array_equal(oracle.sort_e64_bits(p, bits), p[order]), bits


@pytest.mark.skipif(not oracle.have_ref(), reason="oracle/_ref/libref.so not built (needs /root/reference)")
def test_restatement_equals_reference_on_seeded_inputs():
    for n in (1, 255, 256, 257, 4096, 70001, 1 << 20):
        k = oracle.keys_u32(n, seed=123)
        assert np.array_equal(oracle.sort_u32(k), oracle.ref_sort_u32(k))
        assert np.array_equal(oracle.sort_u32(k), oracle.ref_sort_u32(k, host_path=True))
        p = oracle.pairs_kv32(n, seed=5)
        p = (p & np.uint64(0xffffffff000000ff))     # 8-bit keys: heavy duplicates, original index as value
        assert np.array_equal(oracle.sort_kv32(p), oracle.ref_sort_kv32(p))
        assert np.array_equal(oracle.sort_kv32(p), oracle.ref_sort_kv32(p, host_path=True))


def test_splitmix_generator_is_index_addressable():
    a = oracle.keys_u32(1000, seed=123, first_index=0)
    b = oracle.keys_u32(400, seed=123, first_index=600)
    assert np.array_equal(a[600:], b)
    p = oracle.pairs_kv32(100, seed=123, first_index=50)
    assert np.array_equal((p >> 32).astype(np.uint32), np.arange(50, 150, dtype=np.uint32))
    assert np.array_equal((p & 0xffffffff).astype(np.uint32), oracle.keys_u32(100, 123, 50))


def test_soa_oracle_is_the_pair_sort_on_separate_arrays():
    """oracle_radix_sort_soa (separate key / value arrays, wide values) pinned: for 4-byte keys and values it must
    give what the restated pair sort gives on the packed pairs -- which the golden vectors and libref.so pin to the
    reference's RadixSort::sort(SortData*) --, and for every width what numpy's stable argsort gives."""
    rng = np.random.RandomState(5)
    for n in (0, 1, 2, 257, 4099, 100003):
        for keys in (oracle.keys_u32(n, seed=n + 1), rng.randint(0, 9, n).astype(np.uint32)):
            vals = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) ^ np.uint32(0x5bd1e995)
            k, v = oracle.sort_soa(keys, vals)
            pairs = keys.astype(np.uint64) | (vals.astype(np.uint64) << np.uint64(32))
            want = oracle.sort_kv32(pairs)
            assert np.array_equal(k, (want & np.uint64(0xffffffff)).astype(np.uint32))
            assert np.array_equal(v, (want >> np.uint64(32)).astype(np.uint32))
            if oracle.have_ref() and n:
                ref = oracle.ref_sort_kv32(pairs)
                assert np.array_equal(k, (ref & np.uint64(0xffffffff)).astype(np.uint32))
                assert np.array_equal(v, (ref >> np.uint64(32)).astype(np.uint32))
            for bits in (4, 12, 20, 28):
                k, v = oracle.sort_soa(keys, vals, bits)
                want = oracle.sort_e64_bits(pairs, bits)
                assert np.array_equal(k, (want & np.uint64(0xffffffff)).astype(np.uint32))
                assert np.array_equal(v, (want >> np.uint64(32)).astype(np.uint32))
    for kdt, bits_list in ((np.uint32, (32, 16)), (np.uint64, (64, 40, 32, 8))):
        for vdt in (np.uint32, np.uint64, np.dtype([("a", "<u8"), ("b", "<u8")])):
            n = 50021
            keys = (rng.randint(0, 1 << 30, n).astype(np.uint64) * np.uint64(0x100000001b3)).astype(kdt) >> kdt(3)
            keys[::7] = keys[3]                                        # duplicates
            vals = np.zeros(n, dtype=vdt)
            vals.view(np.uint8).reshape(n, -1)[:] = rng.randint(0, 256, (n, np.dtype(vdt).itemsize)).astype(np.uint8)
            for bits in bits_list:
                mask = kdt((1 << bits) - 1) if bits < 64 else kdt(0xffffffffffffffff)
                order = np.argsort(keys & mask, kind="stable")
                k, v = oracle.sort_soa(keys, vals, bits)
                assert np.array_equal(k, keys[order]) and v.tobytes() == vals[order].tobytes(), (kdt, vdt, bits)
