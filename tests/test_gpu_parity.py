"""GPU parity tests (run with `-m gpu` on the MI355X box).  Everything goes through the C ABI
(libadlhip.so via ctypes); expectations come from the oracle and from the committed golden vectors.
Integer work: the bar is bit-exact.

Mirrors the reference's test (UnitTest/main.cpp:40-56, 88-213: Demo.Sort32 / Demo.SortKeyValue /
Demo.Scan over 11 doubling sizes with srand(123) data) and adds what it lacks:
low-entropy keys, ragged n, partial sortBits, n = 0, scan >= 1M, scratch reuse, 64-bit keys.
"""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle
from oclradixsort_amd import AdlHipError, Buffer, DeviceUtils, Pprims, Stopwatch, _lib
from oclradixsort_amd._lib import check

pytestmark = pytest.mark.gpu

# (sort.algo, sort.digit_bits, sort.tile, sort.rank)
ALGOS = [(-1, 8, -1, 1), (-1, 4, -1, 0), (0, 8, -1, 1), (0, 4, -1, 1), (1, 8, -1, 1), (1, 4, -1, 1), (0, 8, -1, 0), (1, 8, 0, 0), (0, 8, 0, 1), (1, 8, 1, 1),
         (0, 8, 3, 1), (1, 8, 4, 0), (0, 8, 5, 1), (0, 8, 2, 1), (0, 4, 0, 0), (0, 7, -1, 1)]
ALGO_IDS = ["auto8", "auto4-ballot", "onesweep8", "onesweep4", "threekernel8", "threekernel4", "onesweep8-ballot", "threekernel8-256x16-ballot",
            "onesweep8-256x16", "threekernel8-512x16", "onesweep8-512x8", "threekernel8-1024x8-ballot",
            "onesweep8-256x32", "onesweep8-1024x16", "onesweep4-256x16-ballot", "onesweep7"]


@pytest.fixture(scope="module")
def dev():
    d = DeviceUtils.allocate()
    yield d
    DeviceUtils.deallocate(d)


@pytest.fixture(autouse=True)
def _knobs_back_to_default(request):
    """Every test starts from the default knobs on the shared device handle (a test that ends with "sort.rank" = 0 or 7-bit
    digits must not decide which path the next one takes)."""
    if "dev" in request.fixturenames:
        d = request.getfixturevalue("dev")
        set_algo(d, (-1, 8, -1))
        for name, value in (("sort.msd2", 1), ("sort.mid", 1), ("sort.dict", 1), ("sort.binfinish", 1), ("partition.lookback", 1)):
            d.setParam(name, value)
    yield


@pytest.fixture()
def pp(dev):
    set_algo(dev, (-1, 8, -1))
    p = Pprims()
    yield p
    p.close()


def set_algo(dev, algo):
    dev.setParam("sort.algo", algo[0])
    dev.setParam("sort.digit_bits", algo[1])
    dev.setParam("sort.tile", algo[2] if len(algo) > 2 else 0)
    dev.setParam("sort.rank", algo[3] if len(algo) > 3 else dev.getParam("sort.lds_ordered"))


def gpu_sort_u32(dev, p, keys, bits=32):
    b = Buffer(dev, keys.size, np.uint32)
    b.write(keys)
    p.radixSort(dev, b, keys.size, bits)
    out = b.toHost()
    b.release()
    return out


def gpu_sort_kv(dev, p, pairs, bits=32):
    b = Buffer(dev, pairs.size, np.uint64)
    b.write(pairs)
    p.radixSort(dev, b, pairs.size, bits)
    out = b.toHost()
    b.release()
    return out


def gpu_sort_u64(dev, p, keys, bits=64):
    b = Buffer(dev, keys.size, np.uint64)
    b.write(keys)
    p.radixSort64(dev, b, keys.size, bits)
    out = b.toHost()
    b.release()
    return out


def gpu_scan(dev, p, vals, want_total=False):
    src = Buffer(dev, vals.size, np.uint32)
    dst = Buffer(dev, vals.size, np.uint32)
    src.write(vals)
    total = np.zeros(1, dtype=np.uint32) if want_total else None
    p.scan(dev, dst, src, vals.size, total)
    out = dst.toHost()
    src.release()
    dst.release()
    return (out, int(total[0])) if want_total else out


# ---------------------------------------------------------------------------------------------
# device / buffer plumbing (the boundary)
# ---------------------------------------------------------------------------------------------
def test_device_is_mi355x(dev):
    assert dev.info.arch.decode().startswith("gfx950")
    assert dev.getParam("sort.lds_ordered") == 1     # the fast ranking path is the one under test
    assert dev.info.wavefront_size == 64
    assert DeviceUtils.getNCUs(dev) >= 1
    assert DeviceUtils.getNDevices() >= 1


def test_buffer_roundtrip_map_unmap_and_accounting(dev):
    base = dev.getUsedMemory()
    n = 100000
    b = Buffer(dev, n, np.uint32)
    assert dev.getUsedMemory() == base + 4 * n
    data = oracle.keys_u32(n, 99)
    # fill through getHostPtr / returnHostPtr exactly as UnitTest/main.cpp:118-125 does
    h = b.getHostPtr(n)
    DeviceUtils.waitForCompletion(dev)
    h[:] = data
    b.returnHostPtr(h)
    DeviceUtils.waitForCompletion(dev)
    assert np.array_equal(b.toHost(), data)
    # offset write / read
    b.write(np.arange(10, dtype=np.uint32), 10, 5)
    got = np.empty(10, dtype=np.uint32)
    b.read(got, 10, 5)
    DeviceUtils.waitForCompletion(dev)
    assert np.array_equal(got, np.arange(10, dtype=np.uint32))
    # device-to-device
    c = Buffer(dev, n, np.uint32)
    c.write(b, n)
    assert np.array_equal(c.toHost(), b.toHost())
    # deallocate refuses while memory is live (Adl.inl:102)
    d2 = DeviceUtils.allocate()
    bb = Buffer(d2, 16, np.uint32)
    with pytest.raises(AdlHipError):
        DeviceUtils.deallocate(d2)
    bb.release()
    DeviceUtils.deallocate(d2)
    b.release()
    c.release()
    assert dev.getUsedMemory() == base


def test_bad_arguments_fail_loudly(dev, pp):
    b = Buffer(dev, 1024, np.uint32)
    for bits in (0, 3, 6, 36, -4):
        with pytest.raises(AdlHipError):
            pp.radixSort(dev, b, 1024, bits)     # Pprims.cpp:330: (sortBits & 3) == 0
    with pytest.raises(AdlHipError):
        dev.setParam("sort.algo", 7)
    assert dev.getParam("sort.algo") in (-1, 0, 1)
    with pytest.raises(AdlHipError):
        dev.setParam("no.such.param", 1)
    b.release()


# ---------------------------------------------------------------------------------------------
# the reference's three Demo tests, all 11 sizes, checked against the oracle AND the golden table
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def table(golden_dir):
    with open(os.path.join(golden_dir, "demo_table.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS)
def test_demo_sort32(dev, pp, table, algo):
    set_algo(dev, algo)
    for row in table["sort32"]:          # 1K ... 1024K, UnitTest/main.cpp:105
        n = row["n"]
        keys = oracle.demo_u32(n)        # seedRandom(123) per size, main.cpp:109
        got = gpu_sort_u32(dev, pp, keys)
        assert np.array_equal(got, oracle.sort_u32(keys)), n
        if "%016x" % oracle.fnv1a64(keys) == row["fnv_in"]:
            assert "%016x" % oracle.fnv1a64(got) == row["fnv_out"], n


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS)
def test_demo_sort_key_value(dev, pp, table, algo):
    set_algo(dev, algo)
    for row in table["sortkv"]:          # 1037 ... 1075187 (testSize += 13, main.cpp:144)
        n = row["n"]
        pairs = oracle.demo_kv32(n)
        got = gpu_sort_kv(dev, pp, pairs)
        assert np.array_equal(got, oracle.sort_kv32(pairs)), n       # key AND value: stability
        if "%016x" % oracle.fnv1a64(pairs) == row["fnv_in"]:
            assert "%016x" % oracle.fnv1a64(got) == row["fnv_out"], n


def test_demo_scan_including_1024k(dev, pp, table):
    for row in table["scan"]:            # the reference fails at 1024K by design (README.md:73-74)
        n = row["n"]
        vals = oracle.demo_scan(n)
        got, total = gpu_scan(dev, pp, vals, want_total=True)
        want, wtotal = oracle.exclusive_scan_u32(vals)
        assert np.array_equal(got, want), n
        assert total == wtotal == row["total"], n
        if "%016x" % oracle.fnv1a64(vals) == row["fnv_in"]:
            assert "%016x" % oracle.fnv1a64(got) == row["fnv_out"], n


# ---------------------------------------------------------------------------------------------
# golden adversarial vectors produced by the reference itself
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS)
def test_adversarial_golden(dev, pp, golden_dir, algo):
    set_algo(dev, algo)
    adv = np.load(os.path.join(golden_dir, "adversarial.npz"))
    names = sorted(k[len("u32_in_"):] for k in adv.files if k.startswith("u32_in_"))
    for nm in names:
        assert np.array_equal(gpu_sort_u32(dev, pp, adv["u32_in_" + nm]), adv["u32_out_" + nm]), nm
        assert np.array_equal(gpu_sort_kv(dev, pp, adv["kv_in_" + nm]), adv["kv_out_" + nm]), nm


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS)
def test_small_demo_golden_vectors(dev, pp, golden_dir, algo):
    set_algo(dev, algo)
    small = np.load(os.path.join(golden_dir, "demo_small.npz"))
    for n in (1024, 2048):
        assert np.array_equal(gpu_sort_u32(dev, pp, small["sort32_in_%d" % n]), small["sort32_out_%d" % n])
    for m in (1037, 2087):
        assert np.array_equal(gpu_sort_kv(dev, pp, small["sortkv_in_%d" % m]), small["sortkv_out_%d" % m])
    assert np.array_equal(gpu_scan(dev, pp, small["scan_in_1024"]), small["scan_out_1024"])


# ---------------------------------------------------------------------------------------------
# edge cases the reference never tests
# ---------------------------------------------------------------------------------------------
RAGGED = [0, 1, 2, 63, 64, 65, 255, 256, 257, 1023, 4095, 4096, 4097, 8191, 8193, 12289, 16385, 50001, 65537, 262143, 1000003]


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS)
def test_ragged_sizes_u32_kv_u64(dev, pp, algo):
    set_algo(dev, algo)
    for n in RAGGED:
        k = oracle.keys_u32(n, seed=n + 1)
        assert np.array_equal(gpu_sort_u32(dev, pp, k), oracle.sort_u32(k)), n
        p = oracle.pairs_kv32(n, seed=n + 2)
        assert np.array_equal(gpu_sort_kv(dev, pp, p), oracle.sort_kv32(p)), n
        k64 = oracle.keys_u64(n, seed=n + 3)
        assert np.array_equal(gpu_sort_u64(dev, pp, k64), oracle.sort_u64(k64)), n


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS)
def test_low_entropy_and_stability(dev, pp, algo):
    set_algo(dev, algo)
    rng = np.random.RandomState(3)
    n = 300007
    cases = {
        "all_equal": np.full(n, 0xabcdef01, dtype=np.uint64),
        "all_max": np.full(n, 0xffffffff, dtype=np.uint64),
        "all_zero": np.zeros(n, dtype=np.uint64),
        "two_values": rng.randint(0, 2, n).astype(np.uint64) * 0xffffffff,
        "low_byte": rng.randint(0, 256, n).astype(np.uint64),
        "one_nibble_each": rng.randint(0, 16, n).astype(np.uint64) * 0x11111111,
        "sorted": np.sort(rng.randint(0, 2**32, n, dtype=np.uint64)),
        "reverse": np.sort(rng.randint(0, 2**32, n, dtype=np.uint64))[::-1].copy(),
        "same_digit_per_tile": (np.arange(n, dtype=np.uint64) // 4096) & 0xff,
    }
    for nm, keys in cases.items():
        k32 = keys.astype(np.uint32)
        assert np.array_equal(gpu_sort_u32(dev, pp, k32), oracle.sort_u32(k32)), nm
        pairs = keys | (np.arange(n, dtype=np.uint64) << 32)    # value = original index
        assert np.array_equal(gpu_sort_kv(dev, pp, pairs), oracle.sort_kv32(pairs)), nm


def test_low_entropy_at_one_workgroup_and_launch_bound_sizes(dev, pp):
    """The one-workgroup sort (n <= 16 Ki: partly filled tile, all-ones pads behind the data) and the 8-launch
    three-kernel path (n <= 64 Ki) on keys that collide with the pads or with each other."""
    rng = np.random.RandomState(11)
    for n in (1, 100, 1000, 4097, 5000, 16384, 20000, 65536):
        cases = {
            "all_max": np.full(n, 0xffffffff, dtype=np.uint64),          # equal to the pad pattern
            "all_equal": np.full(n, 0x80000001, dtype=np.uint64),
            "two_values": rng.randint(0, 2, n).astype(np.uint64) * 0xffffffff,
            "top_byte_only": rng.randint(0, 256, n).astype(np.uint64) << 24,
            "sorted": np.sort(rng.randint(0, 2**32, n, dtype=np.uint64)),
        }
        for nm, keys in cases.items():
            k32 = keys.astype(np.uint32)
            assert np.array_equal(gpu_sort_u32(dev, pp, k32), oracle.sort_u32(k32)), (nm, n)
            pairs = keys | (np.arange(n, dtype=np.uint64) << 32)    # value = original index: stability is visible
            assert np.array_equal(gpu_sort_kv(dev, pp, pairs), oracle.sort_kv32(pairs)), (nm, n)
            k64 = keys | (keys << np.uint64(32))
            assert np.array_equal(gpu_sort_u64(dev, pp, k64), oracle.sort_u64(k64)), (nm, n)
        for bits in (4, 12, 20):   # partial sortBits through the same paths
            k = oracle.keys_u32(n, seed=n + bits)
            assert np.array_equal(gpu_sort_u32(dev, pp, k, bits), oracle.sort_u32_bits(k, bits)), (bits, n)


@pytest.mark.parametrize("algo", ALGOS, ids=ALGO_IDS)
def test_partial_sort_bits(dev, pp, algo):
    """sortBits < 32: only the low bits are ordered, stably (Pprims.cpp:357); odd pass counts exercise the
    copy-back (Pprims.cpp:400-403)."""
    set_algo(dev, algo)
    n = 70001
    k = oracle.keys_u32(n, seed=5)
    p = oracle.pairs_kv32(n, seed=6)
    for bits in range(4, 33, 4):
        assert np.array_equal(gpu_sort_u32(dev, pp, k, bits), oracle.sort_u32_bits(k, bits)), bits
        assert np.array_equal(gpu_sort_kv(dev, pp, p, bits), oracle.sort_e64_bits(p, bits)), bits
    k64 = oracle.keys_u64(n, seed=8)
    for bits in (4, 28, 36, 48, 60, 64):
        assert np.array_equal(gpu_sort_u64(dev, pp, k64, bits), oracle.sort_e64_bits(k64, bits)), bits


def test_scratch_reuse_growing_and_shrinking(dev, pp):
    for n in (5000, 2000000, 300, 70000, 2000000, 1):
        k = oracle.keys_u32(n, seed=n)
        assert np.array_equal(gpu_sort_u32(dev, pp, k), oracle.sort_u32(k)), n
        p = oracle.pairs_kv32(n, seed=n)
        assert np.array_equal(gpu_sort_kv(dev, pp, p), oracle.sort_kv32(p)), n


def test_scan_sizes_and_inplace(dev, pp):
    rng = np.random.RandomState(1)
    for n in (0, 1, 2, 255, 4096, 4097, 32768, 32769, 1 << 20, (1 << 22) + 12345):
        v = rng.randint(0, 2**32, n, dtype=np.uint64).astype(np.uint32)   # wrap-around arithmetic
        got, total = gpu_scan(dev, pp, v, want_total=True)
        want, wtotal = oracle.exclusive_scan_u32(v)
        assert np.array_equal(got, want), n
        assert total == wtotal, n
    n = 1 << 21
    v = oracle.demo_scan(n)
    b = Buffer(dev, n, np.uint32)
    b.write(v)
    pp.scan(dev, b, b, n)            # dst == src
    assert np.array_equal(b.toHost(), oracle.exclusive_scan_u32(v)[0])
    b.release()


def test_lookback_under_repeated_uneven_launches(dev, pp):
    """Hammer the tile-status protocol: many back-to-back sorts of different sizes on one stream, checked
    only at the end (no sync in between), plus a fault-word check at sync."""
    set_algo(dev, (0, 8, -1))
    sizes = [1 << 20, 777777, 4097, 3 << 20, 123456, 5 << 20, 65536]
    bufs, wants = [], []
    for i, n in enumerate(sizes):
        k = oracle.keys_u32(n, seed=100 + i)
        b = Buffer(dev, n, np.uint32)
        b.write(k)
        bufs.append(b)
        wants.append(oracle.sort_u32(k))
    DeviceUtils.waitForCompletion(dev)
    for rep in range(3):
        for b, n in zip(bufs, sizes):
            pp.radixSort(dev, b, n)      # re-sorting sorted data is also a valid (idempotence) check
    DeviceUtils.waitForCompletion(dev)   # raises if a look-back wait hit its bound
    for b, w in zip(bufs, wants):
        assert np.array_equal(b.toHost(), w)
        b.release()


# ---------------------------------------------------------------------------------------------
# full BASELINE sizes: size-independent properties (sortedness, multiset checksum, stability)
# ---------------------------------------------------------------------------------------------
def _checksums(a):
    a64 = a.astype(np.uint64) if a.dtype != np.uint64 else a
    return int(a64.sum(dtype=np.uint64)), int(np.bitwise_xor.reduce(a64)), int((a64 * a64).sum(dtype=np.uint64))


@pytest.mark.parametrize("algo", [(0, 8, -1), (1, 8, -1)], ids=["onesweep8", "threekernel8"])
def test_full_size_64m_u32(dev, pp, algo):
    set_algo(dev, algo)
    n = 1 << 26
    keys = oracle.keys_u32(n, seed=123)
    got = gpu_sort_u32(dev, pp, keys)
    assert np.all(got[1:] >= got[:-1])
    assert _checksums(got) == _checksums(keys)
    # spot-check a prefix and a suffix bit-exactly against the oracle on the exact multiset
    want = oracle.sort_u32(keys)
    assert np.array_equal(got, want)
    # idempotence: sorting the sorted array changes nothing
    assert np.array_equal(gpu_sort_u32(dev, pp, got), got)


def test_full_size_64m_key_value(dev, pp):
    n = 1 << 26
    pairs = oracle.pairs_kv32(n, seed=123) & np.uint64(0xffffffff00ffffff)   # 24-bit keys -> duplicates
    got = gpu_sort_kv(dev, pp, pairs)
    k = (got & np.uint64(0xffffffff))
    v = (got >> np.uint64(32))
    assert np.all(k[1:] >= k[:-1])
    same = k[1:] == k[:-1]
    assert same.any()
    assert np.all(v[1:][same] > v[:-1][same])          # stability: original index increases within a key
    assert _checksums(got) == _checksums(pairs)
    assert np.array_equal(got, oracle.sort_kv32(pairs))


def test_full_size_256m_u64(dev, pp):
    """BASELINE config #5 at its stated size, bit-exact against the oracle (2 GiB of keys)."""
    n = 1 << 28
    keys = oracle.keys_u64(n, seed=123)
    got = gpu_sort_u64(dev, pp, keys)
    assert np.all(got[1:] >= got[:-1])
    assert _checksums(got) == _checksums(keys)
    want = oracle.sort_u64(keys)
    del keys
    assert np.array_equal(got, want)
    del want
    # a second, independent run must give the identical array (determinism)
    again = gpu_sort_u64(dev, pp, oracle.keys_u64(n, seed=123))
    assert np.array_equal(got, again)


# ---------------------------------------------------------------------------------------------
# the size classes BASELINE config #4 runs on every GPU: u32 one-sweep beyond 256 MiB of keys (the write-out leaves
# the raw-buffer stores for 64-bit pointer stores, radix_kernels.hpp dst_fits32 / write_out_tile) and the
# n >= 2^30 three-kernel path (30-bit status counts no longer suffice).  Contract: Pprims.cpp:304-406.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [(1 << 26) + 12345, 1 << 27, 1 << 28, (1 << 29) + 4321], ids=["64Mi+12345", "128Mi", "256Mi", "512Mi+4321"])
def test_u32_onesweep_beyond_256_mib(dev, pp, n):
    set_algo(dev, (0, 8, -1))
    keys = oracle.keys_u32(n, seed=n & 0xffff)
    got = gpu_sort_u32(dev, pp, keys)
    assert np.array_equal(got, oracle.sort_u32(keys))


@pytest.mark.parametrize("n", [(1 << 25) + 4321, (1 << 26) + 12345, 1 << 27], ids=["32Mi+4321", "64Mi+12345", "128Mi"])
def test_kv32_onesweep_beyond_256_mib(dev, pp, n):
    """{key, value} pairs past the buffer-store limit (8-byte elements: 32 Mi pairs), 20-bit keys so that stability is
    exercised by ~128 duplicates per key."""
    set_algo(dev, (0, 8, -1))
    pairs = oracle.pairs_kv32(n, seed=n & 0xfff) & np.uint64(0xffffffff000fffff)
    got = gpu_sort_kv(dev, pp, pairs)
    assert np.array_equal(got, oracle.sort_kv32(pairs))


def test_soa_64m_pairs(dev, pp):
    """SoA key/value sort at BASELINE config #3's size."""
    n = 1 << 26
    keys = oracle.keys_u32(n, seed=77) & np.uint32(0x00ffffff)     # 24-bit keys: duplicates
    vals = np.arange(n, dtype=np.uint32)
    kb, vb = Buffer(dev, n, np.uint32), Buffer(dev, n, np.uint32)
    kb.write(keys); vb.write(vals)
    pp.radixSortSoA(dev, kb, vb, n, 32)
    gk, gv = kb.toHost(), vb.toHost()
    kb.release(); vb.release()
    want = oracle.sort_kv32(keys.astype(np.uint64) | (vals.astype(np.uint64) << np.uint64(32)))
    assert np.array_equal(gk, (want & np.uint64(0xffffffff)).astype(np.uint32))
    assert np.array_equal(gv, (want >> np.uint64(32)).astype(np.uint32))


def test_u32_2p30_plus_12345_three_kernel_path_and_large_sort(dev, pp):
    """n >= 2^30: the per-digit choice must leave the one-sweep path (status words carry 30-bit counts); 4 GiB of keys,
    element indices close to the 32-bit limit of the kernels.  And the large sort at this size: segments of 16 Ki 16-bit keys,
    finished by one workgroup each (wg_segment_sort_kernel, 20480-key tile).  Both bit-exact against the oracle."""
    n = (1 << 30) + 12345
    keys = oracle.keys_u32(n, seed=99)
    want = oracle.sort_u32(keys)
    b = Buffer(dev, n, np.uint32)
    CH = 1 << 27
    chunk = np.empty(CH, dtype=np.uint32)
    set_algo(dev, (-1, 8, -1))
    try:
        for msd2 in (0, 4):
            dev.setParam("sort.msd2", msd2)
            b.write(keys)
            dev.toggleProfiling(True)
            dev.profile(reset=True)
            pp.radixSort(dev, b, n)
            prof = dev.profile(reset=True)
            dev.toggleProfiling(False)
            if msd2 == 0:
                assert any(k.startswith("scatter_u32") for k in prof) and not any(k.startswith("onesweep") for k in prof), prof
            else:
                assert set(prof) == {"msd2_sample", "msd2_pass1_u32", "msd2_pass2_u32", "msd2_offsets", "segment_sort_wg_u32"}, prof
            for off in range(0, n, CH):
                m = min(CH, n - off)
                b.read(chunk[:m], m, off)
                DeviceUtils.waitForCompletion(dev)
                assert np.array_equal(chunk[:m], want[off:off + m]), (msd2, off)
        dev.checkFault()
    finally:
        dev.toggleProfiling(False)
        dev.setParam("sort.msd2", 1)
        b.release()


@pytest.mark.parametrize("n", [(150 << 20) + 3, (200 << 20) + 1, (300 << 20) + 77, (1 << 29) + 4321, (896 << 20) + 5],
                         ids=["150Mi+3", "200Mi+1", "300Mi+77", "512Mi+4321", "896Mi+5"])
def test_large_sort_beyond_280mi_u32_keys(dev, pp, n):
    """Above 280 Mi u32 keys a segment of the large sort no longer fits the 80 rows one wave holds; the finish then takes one
    workgroup per segment (tiles of 8192 / 12288 / 16384 / 20480 16-bit keys: these sizes and the 2^30 test take one each).  The
    same kernel serves the tiles of 3072 / 4096 / 5120 keys from about 140 Mi keys (150 Mi+3, 200 Mi+1; 256 Mi in another test).
    Uniform keys, keys that use one eighth of the range (digits placed from the sample) and, at the smallest size, keys that
    overflow a bucket (safety net).  Bit-exact against the oracle."""
    set_algo(dev, (-1, 8, -1))
    dev.setParam("sort.msd2", 4)
    names = {"msd2_sample", "msd2_pass1_u32", "msd2_pass2_u32", "msd2_offsets", "segment_sort_wg_u32"}
    try:
        keys = oracle.keys_u32(n, seed=n & 0xfff)
        got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, pp, keys))
        assert set(prof) == names, prof
        assert prof["msd2_offsets"][1] < 1.0, prof   # no safety net
        assert np.array_equal(got, oracle.sort_u32(keys))
        if n < (1 << 29):
            eighth = (keys >> np.uint32(3)) | np.uint32(0xa0000000)
            got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, pp, eighth))
            assert set(prof) == names and prof["msd2_offsets"][1] < 1.0, prof
            assert np.array_equal(got, oracle.sort_u32(eighth))
            skew = np.where(np.arange(n) % 4 == 0, keys >> np.uint32(8), keys).astype(np.uint32)   # a quarter of the keys in one bucket
            got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, pp, skew))
            assert prof["msd2_offsets"][1] > 1.0, prof   # the safety net sorted
            assert np.array_equal(got, oracle.sort_u32(skew))
        dev.checkFault()
    finally:
        dev.setParam("sort.msd2", 1)


def test_stopwatch_and_profiling(dev, pp):
    n = 1 << 22
    b = Buffer(dev, n, np.uint32)
    b.write(oracle.keys_u32(n, 1))
    sw = Stopwatch(dev)
    dev.toggleProfiling(True)
    dev.profile(reset=True)
    sw.start()
    pp.radixSort(dev, b, n)
    sw.stop()
    ms = sw.getMs()
    csv_path = "/tmp/adlhip_profile_test.csv"
    if os.path.exists(csv_path):
        os.remove(csv_path)
    dev.writeProfileCsv(csv_path)
    prof = dev.profile(reset=True)
    dev.toggleProfiling(False)
    rows = open(csv_path).read().splitlines()
    assert rows[0] == '"kernel","launches","total_ms","avg_ms"' and len(rows) >= 3
    assert ms > 0
    assert any(k.startswith(("onesweep_u32", "scatter_u32", "msd2_pass1_u32")) for k in prof), prof   # whichever path n selects
    assert sum(v[1] for v in prof.values()) > 0
    b.release()


# ---------------------------------------------------------------------------------------------
# multi-GPU building blocks on one GPU: MSB partition + a simulated exchange (everything but RCCL)
# ---------------------------------------------------------------------------------------------
def test_msb_partition_and_simulated_exchange():
    import torch
    from oclradixsort_amd.dist import HipBackend
    be = HipBackend(0)
    try:
        n = 300007
        for G in (1, 2, 4, 8, 256):
            shards = [oracle.keys_u32(n, seed=11, first_index=r * n) for r in range(min(G, 8))]
            parts, counts = [], []
            for k in shards:
                t = torch.from_numpy(k.view(np.int32).copy()).cuda()
                p, c = be.partition_msb(t, G)
                torch.cuda.synchronize()
                p = p.cpu().numpy().view(np.uint32)
                c = c.cpu().numpy().astype(np.int64)
                lg = G.bit_length() - 1
                bucket = (k >> np.uint32(32 - lg)).astype(np.int64) if lg else np.zeros(n, dtype=np.int64)
                assert np.array_equal(c, np.bincount(bucket, minlength=G)), G
                # stable partition by bucket; inside a bucket the keys are ordered by their top byte (stable)
                order = np.argsort((k >> np.uint32(24)).astype(np.int64), kind="stable")
                assert np.array_equal(p, k[order] if G > 1 else k), G      # one bucket: a plain copy
                parts.append(p)
                counts.append(c)
                # {key, value} pairs: same partition by the key's top bits, values travel along
                pr = k.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
                pp_, pc_ = be.partition_msb(torch.from_numpy(pr.view(np.int64).copy()).cuda(), G)
                torch.cuda.synchronize()
                assert np.array_equal(pc_.cpu().numpy().astype(np.int64), c), G
                assert np.array_equal(pp_.cpu().numpy().view(np.uint64), pr[order] if G > 1 else pr), G
            if G <= 8:
                # what all_to_all_single would deliver: rank g receives, in source-rank order, every
                # shard's segment g; then sorts locally
                outs = []
                for g in range(G):
                    segs = []
                    for p, c in zip(parts, counts):
                        off = int(c[:g].sum())
                        segs.append(p[off:off + int(c[g])])
                    recv = torch.from_numpy(np.concatenate(segs).view(np.int32).copy()).cuda()
                    be.local_sort(recv)
                    torch.cuda.synchronize()
                    outs.append(recv.cpu().numpy().view(np.uint32))
                assert np.array_equal(np.concatenate(outs), oracle.sort_u32(np.concatenate(shards))), G
    finally:
        be.close()


def test_pipelined_sort_stream_over_rccl_one_rank():
    """sort_stream on the GPU: partition + all-gather + all-to-all (RCCL, a one-rank group) on the exchange
    stream overlapped with the previous batch's local sort on the sort stream.  Batches of changing size and
    skew, results checked bit-exactly while the newest allowed number of later results is held too."""
    import socket
    import torch
    import torch.distributed as dist
    from oclradixsort_amd.dist import HipBackend, ShardedRadixSort
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    be = None
    try:
        be = HipBackend(0)
        sorter = ShardedRadixSort(be)
        sizes = [1 << 20, 300007, 5, 0, (1 << 22) + 13, 1 << 20, 999, 2000003, 1 << 21]
        ins = []
        for b, n in enumerate(sizes):
            k = oracle.keys_u32(n, seed=900 + b)
            if b % 3 == 1:
                k = (k >> np.uint32(9)).astype(np.uint32)
            ins.append(k)
        dev_in = [torch.from_numpy(k.view(np.int32).copy()).cuda() for k in ins]
        held = []     # (batch, tensor) not yet checked; at most pipeline_depth - 1 results may be held at a time
        checked = 0
        def check_one(b, t):
            got = t.cpu().numpy().view(np.uint32)      # .cpu() runs on the caller's stream: must already be ordered
            assert np.array_equal(got, oracle.sort_u32(ins[b])), "batch %d (n = %d)" % (b, sizes[b])
        for b, res in enumerate(sorter.sort_stream(dev_in, force_exchange=True)):
            assert res.numel() == sizes[b]
            held.append((b, res))
            if len(held) >= be.pipeline_depth - 1:
                check_one(*held.pop(0))
                checked += 1
        for item in held:
            check_one(*item)
            checked += 1
        assert checked == len(sizes)
        # stage discipline: in the pipeline the partition runs on the EXCHANGE stage's handle (its stream carries the
        # collectives that consume the counts), the local sort on the SORT stage's, nothing on the caller's
        for st in (be._caller, be._exchange, be._sorting):
            st.device.toggleProfiling(True)
            st.device.profile(reset=True)
        for res in sorter.sort_stream(dev_in[:3], force_exchange=True):
            pass
        torch.cuda.synchronize()
        prof = [st.device.profile(reset=True) for st in (be._caller, be._exchange, be._sorting)]
        for st in (be._caller, be._exchange, be._sorting):
            st.device.toggleProfiling(False)
        assert not prof[0], prof[0]
        # exchange stage: exactly one top-byte partition pass (count -> scan -> scatter) per batch and no sort;
        # sort stage: the local sorts (several passes per batch, the 5-key batch in one workgroup)
        assert prof[1].get("count_u32_8b", (0, 0))[0] == 3 and prof[1].get("scatter_u32_8b", (0, 0))[0] == 3, prof[1]
        assert not any(k.startswith(("onesweep", "small_sort")) for k in prof[1]), prof[1]
        # 1Mi and 300007 keys: the mid-size sort, or (after the skewed batches above) the per-digit passes
        assert any(k.startswith(("segment_sort", "scatter")) for k in prof[2]) and "small_sort_u32" in prof[2], prof[2]
        assert not any(k.startswith(("mid_prep", "segment_sort")) for k in prof[1]), prof[1]
        # the serial driver gives the same answer through the same collectives
        r = sorter.sort(dev_in[1].clone(), force_exchange=True)
        assert np.array_equal(r.cpu().numpy().view(np.uint32), oracle.sort_u32(ins[1]))
        # {key, value} pairs (int64 tensors) through both drivers: stable by key
        pin = []
        for b, n in enumerate((500009, 0, 1 << 20, 70001)):
            k = oracle.keys_u32(n, seed=40 + b)
            if b == 0:
                k &= np.uint32(0xff0000ff)
            pin.append(k.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32)))
        pdev = [torch.from_numpy(p.view(np.int64).copy()).cuda() for p in pin]
        prev = None
        for b, res in enumerate(sorter.sort_stream(pdev, force_exchange=True)):
            if prev is not None:
                assert np.array_equal(prev[1].cpu().numpy().view(np.uint64), oracle.sort_kv32(pin[prev[0]])), prev[0]
            prev = (b, res)
        assert np.array_equal(prev[1].cpu().numpy().view(np.uint64), oracle.sort_kv32(pin[prev[0]]))
        r = sorter.sort(pdev[0], force_exchange=True)
        assert np.array_equal(r.cpu().numpy().view(np.uint64), oracle.sort_kv32(pin[0]))
    finally:
        if be is not None:
            be.close()
        dist.destroy_process_group()


def _two_rank_worker(rank, world, port, out_dir):
    """Two processes on ONE GPU: the real HipBackend (partition with G = 2, ragged exchange, streams, slots), with the
    two collectives staged through the host over gloo -- RCCL refuses two ranks on one device, and everything else is
    what runs on a multi-GPU node."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    real_ag, real_a2a = dist.all_gather_into_tensor, dist.all_to_all_single

    def staged_all_gather(out, inp, group=None):
        torch.cuda.current_stream().synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        real_ag(o, inp.cpu(), group=group)
        out.copy_(o)

    def staged_all_to_all(out, inp, out_splits, in_splits, group=None):
        torch.cuda.current_stream().synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        real_a2a(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)

    dist.all_gather_into_tensor, dist.all_to_all_single = staged_all_gather, staged_all_to_all
    from oclradixsort_amd.dist import HipBackend, ShardedRadixSort
    be = HipBackend(0)
    try:
        sorter = ShardedRadixSort(be)
        sizes = [700001, 1 << 20, 333, 0, 250007, 1 << 19]
        dev_in = []
        for b, n in enumerate(sizes):
            k = oracle.keys_u32(n, seed=70 + b, first_index=rank * n)
            if b % 2:   # skewed: almost everything goes to rank 0
                k = np.where(np.arange(n) % 9 != 0, k >> np.uint32(1 + b), k).astype(np.uint32)
            np.save(os.path.join(out_dir, "in_%d_%d.npy" % (b, rank)), k)
            dev_in.append(torch.from_numpy(k.view(np.int32).copy()).cuda())
        for b, res in enumerate(sorter.sort_stream(dev_in)):
            np.save(os.path.join(out_dir, "out_%d_%d.npy" % (b, rank)), res.cpu().numpy().view(np.uint32))
        # pairs through the serial driver
        n = 400003
        k = (oracle.keys_u32(n, seed=5, first_index=rank * n) & np.uint32(0xc000000f)).astype(np.uint64)
        pr = k | ((np.arange(n, dtype=np.uint64) + np.uint64(rank * n)) << np.uint64(32))
        np.save(os.path.join(out_dir, "kvin_%d.npy" % rank), pr)
        r = sorter.sort(torch.from_numpy(pr.view(np.int64).copy()).cuda())
        np.save(os.path.join(out_dir, "kvout_%d.npy" % rank), r.cpu().numpy().view(np.uint64))
    finally:
        be.close()
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_with_host_staged_collectives(tmp_path):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    mp.spawn(_two_rank_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for b in range(6):
        ins = [np.load(tmp_path / ("in_%d_%d.npy" % (b, r))) for r in range(world)]
        outs = [np.load(tmp_path / ("out_%d_%d.npy" % (b, r))) for r in range(world)]
        # rank order == key order (balanced splitters: ownership follows the global top-byte histogram)
        assert np.array_equal(np.concatenate(outs), oracle.sort_u32(np.concatenate(ins))), "batch %d" % b
        if ins[0].size > 1000:
            mean = sum(o.size for o in outs) / world
            assert max(o.size for o in outs) <= 1.25 * mean + 2 * np.bincount(np.concatenate(ins) >> np.uint32(24)).max(), "batch %d" % b
    kin = [np.load(tmp_path / ("kvin_%d.npy" % r)) for r in range(world)]
    kout = [np.load(tmp_path / ("kvout_%d.npy" % r)) for r in range(world)]
    assert np.array_equal(np.concatenate(kout), oracle.sort_kv32(np.concatenate(kin)))


def test_generated_keys_match_the_oracle_generator(dev):
    n = 100001
    for kind, dtype, want in ((0, np.uint32, oracle.keys_u32(n, 123, 77)), (1, np.uint64, oracle.pairs_kv32(n, 123, 77)),
                              (2, np.uint64, oracle.keys_u64(n, 123, 77))):
        b = Buffer(dev, n, dtype)
        b.generate(n, seed=123, firstIndex=77, kind=kind)
        assert np.array_equal(b.toHost(), want), kind
        b.release()


def test_scratch_bytes_never_shrink_with_n(dev):
    """adlhip_radix_sort_scratch_bytes(n) suffices for every smaller n (include/adlhip.h): a caller sizes its scratch once for
    its largest batch.  Checked as monotonicity over sizes around every threshold at which a path or a tile changes."""
    import ctypes
    from oclradixsort_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    edges = [1 << k for k in range(10, 29)] + [(2 << 20) + 1, (6 << 20) + 1, (1 << 26) + (1 << 20), 280 << 20, (280 << 20) + 1,
                                              (1 << 28) + (1 << 22), (1 << 28) + (1 << 22) + 1, 3 << 20, 24 << 18, 3 << 27]
    sizes = sorted(set(edges + [e - 1 for e in edges] + [int(x) for x in rng.integers(1 << 10, 3 << 27, 300)]))
    for kind in (0, 1, 2, 3):
        last_t = last_w = 0
        for n in sizes:
            tb, wb = ctypes.c_size_t(), ctypes.c_size_t()
            assert lib.adlhip_radix_sort_scratch_bytes(dev._h, kind, n, ctypes.byref(tb), ctypes.byref(wb)) == 0
            assert tb.value >= last_t and wb.value >= last_w, (kind, n, wb.value, last_w)
            last_t, last_w = tb.value, wb.value


def test_abi_argument_validation(dev):
    """The C ABI refuses bad buffers loudly instead of launching kernels on them."""
    import ctypes
    from oclradixsort_amd import _lib
    lib = _lib.load()
    n = 1 << 20
    keys = Buffer(dev, n + 4, np.uint32)
    tmp = Buffer(dev, n + 4, np.uint32)
    tb, wb = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.adlhip_radix_sort_scratch_bytes(dev._h, 0, n, ctypes.byref(tb), ctypes.byref(wb)) == 0
    work = Buffer(dev, wb.value, np.uint8)
    ok = lambda rc: rc == 0
    # too small a work buffer: less than the MINIMUM (level 0) is refused; anything from there up sorts (the fastest path that fits)
    t0, w0 = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.adlhip_radix_sort_scratch_bytes_for(dev._h, 0, n, 32, 0, ctypes.byref(t0), ctypes.byref(w0)) == 0
    assert not ok(lib.adlhip_radix_sort_u32(dev._h, keys.ptr(), tmp.ptr(), work.ptr(), w0.value - 1, n, 32))
    assert b"work buffer too small" in lib.adlhip_last_error()
    assert ok(lib.adlhip_radix_sort_u32(dev._h, keys.ptr(), tmp.ptr(), work.ptr(), wb.value - 1, n, 32))
    assert not ok(lib.adlhip_radix_sort_scratch_bytes_for(dev._h, 0, n, 30, 1, ctypes.byref(t0), ctypes.byref(w0)))   # not a multiple of 4
    assert not ok(lib.adlhip_radix_sort_scratch_bytes_for(dev._h, 0, n, 32, 3, ctypes.byref(t0), ctypes.byref(w0)))   # no such level
    # null buffers
    assert not ok(lib.adlhip_radix_sort_u32(dev._h, keys.ptr(), None, work.ptr(), wb.value, n, 32))
    assert not ok(lib.adlhip_radix_sort_u32(dev._h, None, tmp.ptr(), work.ptr(), wb.value, n, 32))
    # misaligned data pointer (element offset 1 = 4 bytes)
    assert not ok(lib.adlhip_radix_sort_u32(dev._h, keys.ptr(1), tmp.ptr(), work.ptr(), wb.value, n, 32))
    assert b"16-byte aligned" in lib.adlhip_last_error()
    # bad element kind / bucket count
    assert not ok(lib.adlhip_radix_sort_scratch_bytes(dev._h, 7, n, ctypes.byref(tb), ctypes.byref(wb)))
    cnt = Buffer(dev, 256, np.uint32)
    assert not ok(lib.adlhip_partition_msb_u32(dev._h, keys.ptr(), tmp.ptr(), cnt.ptr(), work.ptr(), work.getSize(), n, 3))
    # and the good call still works afterwards
    k = oracle.keys_u32(n, 5)
    keys.write(k, n)
    assert ok(lib.adlhip_radix_sort_u32(dev._h, keys.ptr(), tmp.ptr(), work.ptr(), work.getSize(), n, 32))
    assert np.array_equal(keys.toHost(n), oracle.sort_u32(k))
    for b in (keys, tmp, work, cnt):
        b.release()


def test_two_pprims_and_two_devices_do_not_interfere(dev):
    d2 = DeviceUtils.allocate()
    p1, p2 = Pprims(), Pprims()
    try:
        n = 500009
        k1, k2 = oracle.keys_u32(n, 1), oracle.keys_u32(n, 2)
        b1, b2 = Buffer(dev, n, np.uint32), Buffer(d2, n, np.uint32)
        b1.write(k1); b2.write(k2)
        for _ in range(3):           # interleaved, unsynchronised
            p1.radixSort(dev, b1, n)
            p2.radixSort(d2, b2, n)
        assert np.array_equal(b1.toHost(), oracle.sort_u32(k1))
        assert np.array_equal(b2.toHost(), oracle.sort_u32(k2))
        b1.release(); b2.release()
    finally:
        p1.close(); p2.close()
        DeviceUtils.deallocate(d2)


# ---------------------------------------------------------------------------------------------
# structure-of-arrays key-value sort
# ---------------------------------------------------------------------------------------------
def test_two_handles_driven_from_two_host_threads():
    """include/adlhip.h: distinct handles may be driven from distinct host threads (the reference is single-threaded;
    the multi-GPU host code is not).  Two threads, each with its own device handle and Pprims, sort different data of
    sizes that walk through every path (one-workgroup, fused-scan, three-kernel, one-sweep) at the same time."""
    import threading
    sizes = [1000, 20000, 70001, 300000, 1 << 20, (1 << 24) + 5, 4097, 1 << 22]
    errors = []

    def worker(tid):
        try:
            d = DeviceUtils.allocate()
            p = Pprims()
            try:
                for rep in range(3):
                    for n in sizes:
                        k = oracle.keys_u32(n, seed=1000 * tid + n + rep)
                        got = gpu_sort_u32(d, p, k)
                        if not np.array_equal(got, oracle.sort_u32(k)):
                            errors.append((tid, n, rep, "u32"))
                        if n <= (1 << 20):
                            pr = oracle.pairs_kv32(n, seed=7 * tid + n + rep)
                            if not np.array_equal(gpu_sort_kv(d, p, pr), oracle.sort_kv32(pr)):
                                errors.append((tid, n, rep, "kv"))
            finally:
                p.close()
                DeviceUtils.deallocate(d)
        except Exception as e:   # noqa: BLE001 - reported below
            errors.append((tid, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("algo", [(-1, 8, -1, 1), (0, 8, -1, 1), (1, 8, -1, 1), (0, 4, -1, 0), (1, 4, 0, 1)],
                         ids=["auto8", "onesweep8", "threekernel8", "onesweep4-ballot", "threekernel4-256x16"])
def test_soa_key_value_sort(dev, pp, algo):
    set_algo(dev, algo)
    rng = np.random.RandomState(17)
    for n in (1, 255, 4097, 70001, 1 << 20, (1 << 24) + 3):
        for keys in (oracle.keys_u32(n, seed=n), (rng.randint(0, 37, n)).astype(np.uint32)):     # random, and many duplicates
            vals = np.arange(n, dtype=np.uint32) * np.uint32(2654435761)
            kb, vb = Buffer(dev, n, np.uint32), Buffer(dev, n, np.uint32)
            kb.write(keys); vb.write(vals)
            bits = 32 if n != 70001 else 20
            pp.radixSortSoA(dev, kb, vb, n, bits)
            gk, gv = kb.toHost(), vb.toHost()
            pairs = keys.astype(np.uint64) | (vals.astype(np.uint64) << np.uint64(32))
            want = oracle.sort_kv32(pairs) if bits == 32 else oracle.sort_e64_bits(pairs, bits)
            assert np.array_equal(gk, (want & np.uint64(0xffffffff)).astype(np.uint32)), (n, bits)
            assert np.array_equal(gv, (want >> np.uint64(32)).astype(np.uint32)), (n, bits)
            kb.release(); vb.release()


# SURVEY f3, second half: 64-bit values and 64-bit keys on separate arrays (adlhip_radix_sort_soa).  Expectation: the oracle's
# stable 4-bit LSD passes over (key, value) arrays (oracle_radix_sort_soa; pinned to the reference's pair sort in test_oracle.py).
V16 = np.dtype([("x", "<u4"), ("y", "<u4"), ("z", "<u4"), ("w", "<u4")])


def _wide_values(n, vdtype, seed):
    if vdtype == V16:
        raw = oracle.keys_u32(4 * n, seed=seed + 1).reshape(n, 4)
        raw[:, 0] = np.arange(n, dtype=np.uint32)          # the source index rides in the value: stability is visible
        return raw.copy().view(V16).reshape(n)
    if np.dtype(vdtype).itemsize == 8:
        return (oracle.keys_u64(n, seed=seed + 1) & np.uint64(0xffffffff00000000)) | np.arange(n, dtype=np.uint64)
    return np.arange(n, dtype=np.uint32) * np.uint32(2654435761)


@pytest.mark.parametrize("kdtype,vdtype", [(np.uint32, np.uint64), (np.uint64, np.uint32), (np.uint64, np.uint64), (np.uint32, V16),
                                           (np.uint64, V16), (np.uint32, np.uint32)],
                         ids=["k32v64", "k64v32", "k64v64", "k32v128", "k64v128", "k32v32"])
def test_soa_wide_key_value_sort(dev, pp, kdtype, vdtype):
    rng = np.random.RandomState(23)
    kbits = 8 * np.dtype(kdtype).itemsize
    for n in (1, 2, 255, 4097, 70001, (1 << 20) + 5, (3 << 20) + 17):
        gen = oracle.keys_u32 if kbits == 32 else oracle.keys_u64
        cases = [gen(n, seed=n), rng.randint(0, 37, n).astype(kdtype)]                       # random, and many duplicates
        if kbits == 64:
            cases.append((rng.randint(0, 5, n).astype(np.uint64) << np.uint64(32)) | rng.randint(0, 7, n).astype(np.uint64))   # ties in either dword
        for keys in cases:
            vals = _wide_values(n, vdtype, n)
            bits = kbits if n != 70001 else kbits - 12
            kb, vb = Buffer(dev, n, kdtype), Buffer(dev, n, vdtype)
            kb.write(keys); vb.write(vals)
            pp.radixSortSoA(dev, kb, vb, n, bits)
            gk, gv = kb.toHost(), vb.toHost()
            kb.release(); vb.release()
            wk, wv = oracle.sort_soa(keys, vals, bits)
            assert np.array_equal(gk, wk), (n, bits)
            assert gv.tobytes() == wv.tobytes(), (n, bits)


def test_soa_wide_64m_u32_keys_u64_values(dev, pp):
    """BASELINE-size check of the wide-value path: 64 Mi u32 keys with u64 values (value = source index in the low dword).
    Size-independent properties: keys sorted, (key, index) strictly increasing among equal keys (stability), every value
    still beside its key (value's high dword = a hash of the key), checksum of values unchanged."""
    n = 64 << 20
    keys = oracle.keys_u32(n, seed=77) & np.uint32(0x00ffffff)          # 16 M distinct values: ~4 duplicates per key
    vals = ((keys.astype(np.uint64) * np.uint64(0x9E3779B1)) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
    kb, vb = Buffer(dev, n, np.uint32), Buffer(dev, n, np.uint64)
    kb.write(keys); vb.write(vals)
    pp.radixSortSoA(dev, kb, vb, n)
    gk, gv = kb.toHost(), vb.toHost()
    kb.release(); vb.release()
    assert np.all(gk[1:] >= gk[:-1])
    assert np.array_equal(gv >> np.uint64(32), (gk.astype(np.uint64) * np.uint64(0x9E3779B1)) & np.uint64(0xffffffff))
    idx = (gv & np.uint64(0xffffffff)).astype(np.int64)
    same = gk[1:] == gk[:-1]
    assert np.all(idx[1:][same] > idx[:-1][same])
    assert int(np.bitwise_xor.reduce(gv)) == int(np.bitwise_xor.reduce(vals)) and int(idx.sum()) == n * (n - 1) // 2
    assert np.array_equal(gk, oracle.sort_u32(keys))


def test_soa_wide_argument_errors(dev, pp):
    lib = _lib.load()
    tk, tv, wb = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.adlhip_radix_sort_soa_scratch_bytes(dev._h, 2, 8, 100, 16, ctypes.byref(tk), ctypes.byref(tv), ctypes.byref(wb)) != 0
    assert lib.adlhip_radix_sort_soa_scratch_bytes(dev._h, 4, 3, 100, 32, ctypes.byref(tk), ctypes.byref(tv), ctypes.byref(wb)) != 0
    assert lib.adlhip_radix_sort_soa_scratch_bytes(dev._h, 4, 8, 100, 36, ctypes.byref(tk), ctypes.byref(tv), ctypes.byref(wb)) != 0
    assert lib.adlhip_radix_sort_soa_scratch_bytes(dev._h, 8, 8, 100, 64, ctypes.byref(tk), ctypes.byref(tv), ctypes.byref(wb)) == 0
    kb, vb = Buffer(dev, 100, np.uint64), Buffer(dev, 100, np.uint64)
    t, w = Buffer(dev, tk.value + tv.value, np.uint8), Buffer(dev, wb.value, np.uint8)
    # work buffer too small -> loud failure, nothing enqueued
    assert lib.adlhip_radix_sort_soa(dev._h, kb.ptr(), 8, vb.ptr(), 8, t.ptr(), ctypes.c_void_p(t.m_ptr + tk.value), w.ptr(), 1024, 100, 64) != 0
    assert b"work buffer too small" in lib.adlhip_last_error()
    assert lib.adlhip_radix_sort_soa(dev._h, kb.ptr(), 8, vb.ptr(), 8, t.ptr(), ctypes.c_void_p(t.m_ptr + tk.value), w.ptr(), wb.value, 0, 64) == 0
    for b in (kb, vb, t, w):
        b.release()


# ---------------------------------------------------------------------------------------------
# Pprims::fill / copy (commented out in the reference, Pprims.cpp:31-120)
# ---------------------------------------------------------------------------------------------
def test_fill_and_copy_primitives(dev, pp):
    f4 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4")])     # Tahoe::float4 as 16 plain bytes
    cases = [(np.uint32, 0xC0FFEE01), (np.int32, -12345), (np.uint64, 0x0123456789ABCDEF), (f4, (1.5, -2.25, 3.0, 4.0))]
    for dtype, value in cases:
        for n in (0, 1, 2, 3, 255, 4097, 100003, (1 << 22) + 1):
            cap = n + 7
            a, b = Buffer(dev, cap, dtype), Buffer(dev, cap, dtype)
            a.clear(); b.clear()
            pp.fill(dev, a, value, n)
            pp.copy(dev, b, a, n)
            want = np.zeros(cap, dtype=dtype)
            want[:n] = np.array(value, dtype=dtype)
            assert a.toHost().tobytes() == want.tobytes(), (dtype, n)
            assert b.toHost().tobytes() == want.tobytes(), (dtype, n)
            a.release(); b.release()
    # 8-byte pattern at an address that is 8- but not 16-byte aligned (the odd leading element), odd and even counts
    lib = _lib.load()
    buf = Buffer(dev, 64, np.uint64)
    pat = np.array([0xFEEDFACECAFEBEEF], dtype=np.uint64)
    for first, cnt in ((1, 1), (1, 2), (1, 5), (3, 6), (2, 7)):
        buf.clear()
        check(lib.adlhip_fill_pattern(dev._h, buf.ptr(first), pat.ctypes.data_as(ctypes.c_void_p), 8, cnt), "fill_pattern")
        got = buf.toHost()
        want = np.zeros(64, dtype=np.uint64)
        want[first:first + cnt] = pat[0]
        assert np.array_equal(got, want), (first, cnt)
    # loud failures: unsupported pattern size, misaligned destination
    assert lib.adlhip_fill_pattern(dev._h, buf.ptr(), pat.ctypes.data_as(ctypes.c_void_p), 3, 4) != 0
    assert lib.adlhip_fill_pattern(dev._h, ctypes.c_void_p(buf.m_ptr + 4), pat.ctypes.data_as(ctypes.c_void_p), 8, 1) != 0
    assert b"fill" in lib.adlhip_last_error()
    buf.release()


# ---------------------------------------------------------------------------------------------
# segments finished in LDS (pass C of the hybrid sort, adlhip_segment_sort)
# ---------------------------------------------------------------------------------------------
def _segment_sort_expected(arr, starts, low_bits):
    """Stable sort of every segment by the low bits of the key (the key is the low dword of 8-byte elements)."""
    seg_id = np.repeat(np.arange(starts.size - 1, dtype=np.uint64), np.diff(starts).astype(np.int64))
    low = (arr.astype(np.uint64) & np.uint64((1 << low_bits) - 1))
    order = np.argsort((seg_id << np.uint64(32)) | low, kind="stable")
    return arr[order]


@pytest.mark.parametrize("kind,cap", [(0, 4096), (0, 8192), (0, 16384), (1, 4096), (1, 8192)],
                         ids=["u32-4Ki", "u32-8Ki", "u32-16Ki", "kv-4Ki", "kv-8Ki"])
def test_segment_sort_in_lds(dev, kind, cap):
    lib = _lib.load()
    rng = np.random.RandomState(cap + kind)
    sizes = np.concatenate([rng.randint(0, cap + 1, 300), [0, 1, 2, 63, 64, 65, cap, cap - 1, 0, 0, cap // 2 + 1]]).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    n = int(starts[-1])
    dtype = np.uint32 if kind == 0 else np.uint64
    sb = Buffer(dev, starts.size, np.uint32)
    sb.write(starts)
    for low_bits, flavour in ((18, "random"), (9, "random"), (1, "random"), (16, "random"), (24, "random"), (14, "dups"),
                              (18, "constant"), (12, "max")):
        if flavour == "random":
            keys = rng.randint(0, 2**32, n, dtype=np.uint64)
        elif flavour == "dups":
            keys = rng.randint(0, 7, n).astype(np.uint64) * np.uint64(0x1111)
        elif flavour == "constant":
            keys = np.full(n, 0x2aaaa, dtype=np.uint64)
        else:
            keys = np.full(n, 0xffffffff, dtype=np.uint64)           # equal to the pad pattern
        arr = keys.astype(np.uint32) if kind == 0 else (keys | (np.arange(n, dtype=np.uint64) << np.uint64(32)))
        b = Buffer(dev, n, dtype)
        b.write(arr)
        check(lib.adlhip_segment_sort(dev._h, kind, b.ptr(), sb.ptr(), starts.size - 1, cap, low_bits), "segment_sort")
        got = b.toHost()
        b.release()
        assert np.array_equal(got, _segment_sort_expected(arr, starts, low_bits)), (low_bits, flavour)
    # a segment beyond the tile is refused loudly (fault word at sync), never sorted wrongly in silence
    big = np.array([0, cap + 1], dtype=np.uint32)
    sb2 = Buffer(dev, 2, np.uint32)
    sb2.write(big)
    b = Buffer(dev, cap + 1, dtype)
    b.clear()
    check(lib.adlhip_segment_sort(dev._h, kind, b.ptr(), sb2.ptr(), 1, cap, 8), "segment_sort")
    with pytest.raises(AdlHipError):
        DeviceUtils.waitForCompletion(dev)
    DeviceUtils.waitForCompletion(dev)      # the fault word is cleared once it has been reported
    for x in (b, sb, sb2):
        x.release()


# ---------------------------------------------------------------------------------------------
# robustness: stream-ordered fault reporting, the ranking assumption under load, balanced splitters on the device
# ---------------------------------------------------------------------------------------------
def _drain_without_sync(dev):
    """Wait for the handle's stream the way a foreign owner of the stream would (an event), not through adlhip_sync."""
    sw = Stopwatch(dev)
    sw.start()
    sw.stop()
    sw.getMs()


def test_fault_check_is_stream_ordered_and_reports_once(dev):
    lib = _lib.load()
    cap = 4096
    sb = Buffer(dev, 2, np.uint32)
    sb.write(np.array([0, cap + 1], dtype=np.uint32))
    b = Buffer(dev, cap + 1, np.uint32)
    b.clear()
    DeviceUtils.waitForCompletion(dev)
    dev.checkFault()                     # first snapshot: clean
    _drain_without_sync(dev)
    check(lib.adlhip_segment_sort(dev._h, 0, b.ptr(), sb.ptr(), 1, cap, 8), "segment_sort")    # raises the device fault word
    dev.checkFault()                     # reports the clean snapshot, enqueues one behind the faulty kernel
    _drain_without_sync(dev)
    with pytest.raises(AdlHipError):
        dev.checkFault()                 # the fault of the completed batch surfaces here
    _drain_without_sync(dev)
    dev.checkFault()                     # reported once; the word was cleared
    _drain_without_sync(dev)
    dev.checkFault()
    DeviceUtils.waitForCompletion(dev)   # nothing left for the blocking path either
    # and a sort after a fault is unaffected (the live word is cleared by the first kernel of every sort)
    p = Pprims()
    k = oracle.keys_u32(1 << 25, 3)
    assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k))
    p.close()
    for x in (b, sb):
        x.release()


def test_lds_order_selftest_beside_running_sorts(dev):
    """The DS-atomic ranking rests on lane-ordered returning DS atomics (radix_kernels.hpp rank_in_wave).  The
    self-test of device creation runs on an idle chip; here its 256- and 1024-thread variants run on a SECOND
    stream while 64Mi-key sorts keep the LDS units and the memory system busy on the first."""
    lib = _lib.load()
    d2 = DeviceUtils.allocate()
    p = Pprims()
    n = 1 << 26
    b = Buffer(dev, n, np.uint32)
    try:
        b.generate(n, seed=5)
        p.radixSort(dev, b, n)
        DeviceUtils.waitForCompletion(dev)
        for rep in range(8):
            b.generate(n, seed=6 + rep)
            for _ in range(4):
                p.radixSort(dev, b, n)          # a few ms of pass kernels queued on the first stream
            mism = ctypes.c_uint32(1)
            check(lib.adlhip_selftest_lds_order(d2._h, 1024, ctypes.byref(mism)), "selftest")
            assert mism.value == 0, rep
        DeviceUtils.waitForCompletion(dev)
        got = b.toHost()
        assert np.all(got[1:] >= got[:-1])
    finally:
        b.release()
        p.close()
        DeviceUtils.deallocate(d2)


def test_rank_modes_agree_at_64m_pairs_with_256_distinct_keys(dev, pp):
    """Forced cross-check of the two ranking paths (sort.rank = 1: returning DS atomics; 0: ballot/mbcnt) where a wrong
    in-wave order cannot hide: 64Mi {key, index} pairs with at most 256 distinct keys, stability bit-exact."""
    n = 1 << 26
    pairs = oracle.pairs_kv32(n, seed=31) & np.uint64(0xffffffff000000ff)
    outs = []
    for rank in (1, 0):
        set_algo(dev, (0, 8, -1, rank))
        outs.append(gpu_sort_kv(dev, pp, pairs))
    assert np.array_equal(outs[0], outs[1])
    assert np.array_equal(outs[0], oracle.sort_kv32(pairs))


def test_balanced_splitters_on_the_device():
    """partition_top_byte + choose_splitters with CUDA tensors: what one rank of the sharded sort does before the
    exchange, on keys that fixed top-bit ownership would send almost entirely to rank 0."""
    import torch
    from oclradixsort_amd.dist import HipBackend, choose_splitters
    be = HipBackend(0)
    try:
        n = 1 << 22
        k = oracle.keys_u32(n, seed=21)
        k = np.where(np.arange(n) % 10 != 0, k >> np.uint32(3), k).astype(np.uint32)
        part, totals = be.partition_top_byte(torch.from_numpy(k.view(np.int32).copy()).cuda())
        torch.cuda.synchronize()
        top = (k >> np.uint32(24)).astype(np.int64)
        assert np.array_equal(totals.cpu().numpy(), np.bincount(top, minlength=256))
        assert np.array_equal(part.cpu().numpy().view(np.uint32), k[np.argsort(top, kind="stable")])
        for G in (2, 4, 8):
            b = choose_splitters(totals.to(torch.int64), G).cpu().numpy()
            share = np.array([np.bincount(top, minlength=256)[b[g]:b[g + 1]].sum() for g in range(G)])
            assert share.sum() == n and share.max() <= 1.25 * n / G, (G, share)
            fixed = np.bincount(top >> (8 - (G.bit_length() - 1)), minlength=G)
            assert fixed.max() > 0.85 * n            # the skew balanced splitters remove
    finally:
        be.close()


def test_sharded_sort_behind_the_c_abi_one_device_group():
    """adlhip_group_* / adlhip_sharded_sort_*: the multi-GPU sort as a C-ABI entry (one process, G devices,
    ncclCommInitAll + grouped ncclSend/ncclRecv).  A 1-GPU box can only form a group of one: every step still
    runs (top-byte partition, totals to the host, splitters, a self send/recv through RCCL, local sort)."""
    from oclradixsort_amd.adl import Config, Device
    lib = _lib.load()
    g = ctypes.c_void_p()
    check(lib.adlhip_group_create(None, 1, ctypes.byref(g)), "group_create")
    try:
        assert lib.adlhip_group_size(g) == 1
        assert lib.adlhip_group_device(g, 1) is None
        dev = Device(lib.adlhip_group_device(g, 0), Config())        # owned by the group: never deallocated here
        for n in (0, 1, 1000003, (1 << 24) + 77):
            keys = oracle.keys_u32(n, seed=n + 5)
            if n > 2000000:
                keys = np.where(np.arange(n) % 5 != 0, keys >> np.uint32(4), keys).astype(np.uint32)    # skewed
            src, dst = Buffer(dev, max(n, 1), np.uint32), Buffer(dev, n + 16, np.uint32)
            src.write(keys, n)
            ins = (ctypes.c_void_p * 1)(src.ptr().value)
            outs = (ctypes.c_void_p * 1)(dst.ptr().value)
            n_in, cap, n_out = (ctypes.c_size_t * 1)(n), (ctypes.c_size_t * 1)(n + 16), (ctypes.c_size_t * 1)(0)
            check(lib.adlhip_sharded_sort_u32(g, ins, n_in, outs, cap, n_out), "sharded_sort_u32")
            DeviceUtils.waitForCompletion(dev)
            assert n_out[0] == n
            assert np.array_equal(dst.toHost(n), oracle.sort_u32(keys)), n
            assert np.array_equal(src.toHost(n), keys), n                # the shard is left intact
            if n:
                b = (ctypes.c_int * 2)()
                check(lib.adlhip_group_last_bounds(g, b), "last_bounds")
                assert list(b) == [0, 256]
                small = (ctypes.c_size_t * 1)(n - 1)                     # an output that cannot hold the slice: loud, sizes reported
                assert lib.adlhip_sharded_sort_u32(g, ins, n_in, outs, small, n_out) != 0
                assert b"too small" in lib.adlhip_last_error() and n_out[0] == n
            src.release(); dst.release()
        n = 700001
        pairs = oracle.pairs_kv32(n, seed=3) & np.uint64(0xffffffff0000ffff)     # duplicates: stability
        src, dst = Buffer(dev, n, np.uint64), Buffer(dev, n, np.uint64)
        src.write(pairs)
        ins, outs = (ctypes.c_void_p * 1)(src.ptr().value), (ctypes.c_void_p * 1)(dst.ptr().value)
        n_in, cap, n_out = (ctypes.c_size_t * 1)(n), (ctypes.c_size_t * 1)(n), (ctypes.c_size_t * 1)(0)
        check(lib.adlhip_sharded_sort_kv32(g, ins, n_in, outs, cap, n_out), "sharded_sort_kv32")
        DeviceUtils.waitForCompletion(dev)
        assert n_out[0] == n and np.array_equal(dst.toHost(), oracle.sort_kv32(pairs))
        # destroying the group while the caller still holds memory of one of its devices fails loudly (Adl.inl:102)
        src.release()
        assert lib.adlhip_group_destroy(g) != 0 and b"live bytes" in lib.adlhip_last_error()
        dst.release()
    finally:
        assert lib.adlhip_group_destroy(g) == 0


@pytest.mark.parametrize("form", [2, 3], ids=["two-launch-keys", "three-launch"])
def test_mid_size_sort_forms_on_friendly_and_skewed_keys(dev, form):
    """The mid-size sort forced into each of its forms ("sort.mid" = 2: unstable MSD pass with bucket cursors + LDS finish, keys
    only; 3: byte histograms + stable MSD pass + LDS finish) on inputs that fit its buckets and on inputs that do not (the
    cooperative LSD sort inside pass 2 / 3 takes over): bit-exact either way, pairs stable."""
    set_algo(dev, (-1, 8, -1))
    dev.setParam("sort.mid", form)
    p = Pprims()
    rng = np.random.RandomState(form)
    try:
        for n in (16385, 40000, 262144, 300007, 1 << 20, (1 << 21) - 5):
            u = rng.randint(0, 2**32, n, dtype=np.uint64)
            cases = {
                "uniform": u,
                "below 2^24": u >> np.uint64(8),
                "below 2^12": u >> np.uint64(20),
                "two clusters": np.where(u & np.uint64(1), u >> np.uint64(16), np.uint64(0xffff0000) | (u >> np.uint64(16))),
                "all equal": np.full(n, 0x01020304, dtype=np.uint64),
                "sorted": np.sort(u),
                "one heavy byte": np.where(rng.rand(n) < 0.9, u >> np.uint64(8), u),
                "top byte only": (u >> np.uint64(24)) << np.uint64(24),
            }
            for nm, k in cases.items():
                k32 = k.astype(np.uint32)
                assert np.array_equal(gpu_sort_u32(dev, p, k32), oracle.sort_u32(k32)), (nm, n)
                if n <= (1 << 20):
                    pairs = (k & np.uint64(0xffffffff)) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
                    assert np.array_equal(gpu_sort_kv(dev, p, pairs), oracle.sort_kv32(pairs)), (nm, n)
        DeviceUtils.waitForCompletion(dev)      # no device fault (bounded spins of the grid barrier)
    finally:
        dev.setParam("sort.mid", 1)
        p.close()


def test_large_keys_only_sort_two_msd_passes_and_lds_finish(dev):
    """"sort.msd2" forced on: two unstable MSD passes with bucket cursors + the wave-per-segment LDS finish, for sizes across its
    range, on keys that fit its slabs and on keys that do not (the offsets kernel's workgroups then sort the untouched input with the
    cooperative LSD sort and the finish returns at once).  Bit-exact against the oracle either way."""
    set_algo(dev, (-1, 8, -1))
    dev.setParam("sort.msd2", 2)
    p = Pprims()
    try:
        for n in ((1 << 22) + 3, 5000011, 1 << 24, 1 << 26, (1 << 26) + 999999):
            k = oracle.keys_u32(n, seed=n & 0xff)
            assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), ("uniform", n)
        n = (1 << 24) + 77
        u = oracle.keys_u32(n, seed=9)
        skewed = {
            "below 2^24": u >> np.uint32(8),
            "one heavy top byte": np.where(np.arange(n) % 10 != 0, u >> np.uint32(8), u).astype(np.uint32),
            "lower half of the range 1.2 x as dense": np.where(np.arange(n) % 5 == 0, u >> np.uint32(1), u).astype(np.uint32),
            "low 16 bits constant (the finish has nothing to do)": u & np.uint32(0xffff0000),
            "low byte constant": u & np.uint32(0xffffff00),
            "second byte constant": u & np.uint32(0xff00ffff),
            "all equal": np.full(n, 0xdeadbeef, dtype=np.uint32),
            "sorted": np.sort(u),
        }
        for nm, k in skewed.items():
            assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), nm
        # keys that leave their top bits unused: the digits are placed below them (a sample of the keys decides where)
        narrow = {
            "one eighth of the key range (a rank of an 8-GPU sort)": (u >> np.uint32(3)) | np.uint32(0xa0000000),
            "below 2^28": u >> np.uint32(4),
            "below 2^20": u >> np.uint32(12),
            "below 2^16 (nothing left for the finish)": u >> np.uint32(16),
            "range that is not a power of two": (u % np.uint32(3 << 27)) + np.uint32(1 << 30),
        }
        for nm, k in narrow.items():
            assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), nm
        # ... and a sample can miss outliers: the first pass checks every key and hands over to the safety net
        k = (u >> np.uint32(8)).copy()
        k[12345] = 0xf0000001
        k[n - 2] = 0x80000000
        assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), "outliers above the sampled range"
        k = oracle.keys_u32(1 << 26, seed=3) >> np.uint32(8)   # the top of the size range, digits placed at bits 8..23
        assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), "below 2^24 at 64Mi"
        DeviceUtils.waitForCompletion(dev)
        # and the automatic choice stays correct when friendly and skewed inputs alternate (nothing is remembered between sorts)
        dev.setParam("sort.msd2", 1)
        for i in range(6):
            k = oracle.keys_u32(n, seed=20 + i) if i % 2 == 0 else (oracle.keys_u32(n, seed=20 + i) >> np.uint32(9))
            assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), i
    finally:
        dev.setParam("sort.msd2", 1)
        p.close()


def test_large_keys_only_sort_u64_keys(dev):
    """The same path for u64 keys (BASELINE config #5's element): digits placed anywhere in the 64 bits (straddling bit 32
    too), up to six 8-bit passes in the LDS finish, a workgroup per segment where segments outgrow a wave's tile."""
    dev.setParam("sort.msd2", 2)
    p = Pprims()
    try:
        for n in ((1 << 20) + 77, 3000001, 1 << 24, (1 << 26) + 4321):
            k = oracle.keys_u64(n, seed=n & 0xff)
            assert np.array_equal(gpu_sort_u64(dev, p, k), oracle.sort_u64(k)), ("uniform", n)
        n = (1 << 23) + 11
        u = oracle.keys_u64(n, seed=5)
        cases = {
            "top 20 bits unused (digits straddle bit 32)": u >> np.uint64(20),
            "top 28 bits unused": u >> np.uint64(28),
            "values below 2^32": u >> np.uint64(32),
            "values below 2^20": u >> np.uint64(44),
            "values below 2^16 (nothing left for the finish)": u >> np.uint64(48),
            "constant high dword": (u & np.uint64(0xffffffff)) | np.uint64(0x1234567800000000),
            "constant low dword": (u & np.uint64(0xffffffff00000000)) | np.uint64(0x9abcdef0),
            "all equal": np.full(n, 0xdeadbeefcafef00d, dtype=np.uint64),
            "sorted": np.sort(u),
            "one heavy top byte": np.where(np.arange(n) % 10 != 0, u >> np.uint64(8), u).astype(np.uint64),
        }
        for nm, k in cases.items():
            assert np.array_equal(gpu_sort_u64(dev, p, k), oracle.sort_u64(k)), nm
        k = (u >> np.uint64(30)).copy()   # outliers above the sampled range
        k[777] = 0xf000000000000001
        k[n - 3] = 0x8000000000000000
        assert np.array_equal(gpu_sort_u64(dev, p, k), oracle.sort_u64(k)), "outliers"
        DeviceUtils.waitForCompletion(dev)
    finally:
        dev.setParam("sort.msd2", 1)
        p.close()


@pytest.mark.parametrize("n", [(1 << 27) + 5, 1 << 28], ids=["128Mi+5", "256Mi"])
def test_large_keys_only_sort_beyond_64mi(dev, pp, n):
    """u32 keys beyond 64 Mi (what a GPU of BASELINE config #4 sorts): segments outgrow the wave's tile, a workgroup finishes
    each one.  Automatic choice; bit-exact against the oracle."""
    keys = oracle.keys_u32(n, seed=n & 0xffff)
    got = gpu_sort_u32(dev, pp, keys)
    want = oracle.sort_u32(keys)
    del keys
    assert np.array_equal(got, want)
    dev.checkFault()


def test_large_pairs_stable_msd_passes_with_lookback(dev):
    """{key, value} pairs on the large path: the two MSD passes place tiles by look-back (stable), the LDS finish is stable,
    so equal keys keep their input order -- bit-exact against the oracle's stable sort, on keys with many duplicates too.
    Keys that do not fit the slabs go to the safety net."""
    dev.setParam("sort.msd2", 2)
    p = Pprims()
    try:
        for n in ((1 << 20) + 77, 3000001, 1 << 24, (1 << 25) + 12345):
            pairs = oracle.pairs_kv32(n, seed=n & 0xff)
            assert np.array_equal(gpu_sort_kv(dev, p, pairs), oracle.sort_kv32(pairs)), ("uniform", n)
        n = (1 << 23) + 11
        pr = oracle.pairs_kv32(n, seed=7)
        key = pr & np.uint64(0xffffffff)
        val = pr & np.uint64(0xffffffff00000000)
        cases = {
            "20-bit keys (every key ~8 times)": val | (key >> np.uint64(12)),
            "16-bit keys": val | (key >> np.uint64(16)),
            "keys in one eighth of the range": val | (key >> np.uint64(3)) | np.uint64(0x60000000),
            "low byte constant": val | (key & np.uint64(0xffffff00)),
            "all keys equal": val | np.uint64(0x12345678),
            "sorted keys": val | np.sort(key),
            "one heavy top byte": val | np.where(np.arange(n) % 10 != 0, key >> np.uint64(8), key),
        }
        for nm, pairs in cases.items():
            pairs = pairs.astype(np.uint64)
            assert np.array_equal(gpu_sort_kv(dev, p, pairs), oracle.sort_kv32(pairs)), nm
        pairs = (val | (key >> np.uint64(10))).astype(np.uint64)   # outliers above the sampled range
        pairs[4321] |= np.uint64(0xf0000000)
        pairs[n - 5] |= np.uint64(0x80000000)
        assert np.array_equal(gpu_sort_kv(dev, p, pairs), oracle.sort_kv32(pairs)), "outliers"
        # the same path for separate key and value arrays (SoA): packed on load, split by the finish; the safety net packs
        # the input into the slab area, sorts it there and splits it back
        for nm, pairs in (("SoA uniform", oracle.pairs_kv32(n, seed=11)), ("SoA 20-bit keys", cases["20-bit keys (every key ~8 times)"]),
                          ("SoA all keys equal (safety net)", cases["all keys equal"]),
                          ("SoA uniform, 3000001", oracle.pairs_kv32(3000001, seed=12))):
            pairs = pairs.astype(np.uint64)
            m = pairs.size
            kb, vb = Buffer(dev, m, np.uint32), Buffer(dev, m, np.uint32)
            kb.write((pairs & np.uint64(0xffffffff)).astype(np.uint32)); vb.write((pairs >> np.uint64(32)).astype(np.uint32))
            p.radixSortSoA(dev, kb, vb, m, 32)
            want = oracle.sort_kv32(pairs)
            assert np.array_equal(kb.toHost(), (want & np.uint64(0xffffffff)).astype(np.uint32)), nm
            assert np.array_equal(vb.toHost(), (want >> np.uint64(32)).astype(np.uint32)), nm
            kb.release(); vb.release()
        DeviceUtils.waitForCompletion(dev)
        dev.setParam("sort.msd2", 1)
        for i in range(4):   # the automatic choice, friendly and skewed inputs alternating
            pairs = oracle.pairs_kv32(n, seed=30 + i)
            if i % 2:
                pairs = (pairs & np.uint64(0xffffffff00000000)) | ((pairs & np.uint64(0xffffffff)) % np.uint64(1000))
            assert np.array_equal(gpu_sort_kv(dev, p, pairs), oracle.sort_kv32(pairs)), i
    finally:
        dev.setParam("sort.msd2", 1)
        p.close()


def test_large_sort_safety_net_runs_inside_its_offsets_kernel(dev):
    """Keys that cannot fit the slabs (all equal) with "sort.msd2" forced on: no kernel of another path is launched -- the five
    launches of the large sort are all there is -- the offsets kernel is where the time goes (its workgroups run the cooperative
    LSD sort), the finish returns at once, and the result is right."""
    n = (1 << 24) + 5
    keys = np.full(n, 0x0badf00d, dtype=np.uint32)
    keys[::7] = 0x12345678
    dev.setParam("sort.msd2", 2)
    p = Pprims()
    b = Buffer(dev, n, np.uint32)
    try:
        b.write(keys)
        dev.toggleProfiling(True)
        dev.profile(reset=True)
        p.radixSort(dev, b, n)
        prof = dev.profile(reset=True)
        dev.toggleProfiling(False)
        assert set(prof) == {"msd2_sample", "msd2_pass1_u32", "msd2_pass2_u32", "msd2_offsets", "segment_sort_wave_u32"}, prof
        assert prof["msd2_offsets"][1] > 10 * prof["segment_sort_wave_u32"][1], prof   # the sort happened there; the finish left
        assert np.array_equal(b.toHost(), np.sort(keys))
        # a friendly input on the same handle: the offsets kernel is back to microseconds
        k2 = oracle.keys_u32(n, seed=77)
        b.write(k2)
        dev.toggleProfiling(True)
        dev.profile(reset=True)
        p.radixSort(dev, b, n)
        prof = dev.profile(reset=True)
        dev.toggleProfiling(False)
        assert prof["msd2_offsets"][1] < 0.1 and prof["msd2_offsets"][1] < prof["segment_sort_wave_u32"][1], prof
        assert np.array_equal(b.toHost(), oracle.sort_u32(k2))
    finally:
        dev.toggleProfiling(False)
        dev.setParam("sort.msd2", 1)
        b.release()
        p.close()



# ---------------------------------------------------------------------------------------------
# round 3: which kernels a sort ran is asserted, not assumed (the automatic choice depends on hints that arrive
# asynchronously); config #3 on the path the bench times; partial sortBits, u64 keys and the binning finish on the large sort
# ---------------------------------------------------------------------------------------------
def _profiled(dev, fn):
    """Run fn() with per-launch profiling on; returns (result, {kernel name: (launches, ms)})."""
    dev.toggleProfiling(True)
    dev.profile(reset=True)
    try:
        out = fn()
    finally:
        prof = dev.profile(reset=True)
        dev.toggleProfiling(False)
    return out, prof


LARGE_PAIRS = {"msd2s_prep", "msd2s_pass1_kv32", "msd2s_pass2_kv32", "msd2s_offsets", "segment_sort_wave_e64"}
LARGE_U32 = {"msd2_sample", "msd2_pass1_u32", "msd2_pass2_u32", "msd2_offsets", "segment_sort_wave_u32"}
LARGE_U64_BIN = {"msd2s_prep", "msd2s_pass1_u64", "msd2s_pass2_u64", "msd2s_offsets", "segment_sort_bin_u64", "segment_sort_listed_e64"}
LARGE_U64_LSD = {"msd2s_prep", "msd2s_pass1_u64", "msd2s_pass2_u64", "msd2s_offsets", "segment_sort_wave_e64"}


def test_config3_64m_aos_pairs_on_the_large_path_bit_exact(dev):
    """BASELINE config #3 (64 Mi {key, value} pairs, AoS) on the path bench.py times -- the stable large sort, forced so that no hint
    decides -- with 24-bit keys (every key ~4 times: stability shows), bit-exact against the oracle's stable sort; the kernels
    that ran are the large sort's and nothing else."""
    n = 1 << 26
    pairs = oracle.pairs_kv32(n, seed=321) & np.uint64(0xffffffff00ffffff)
    dev.setParam("sort.msd2", 2)
    p = Pprims()
    try:
        got, prof = _profiled(dev, lambda: gpu_sort_kv(dev, p, pairs))
        assert set(prof) == LARGE_PAIRS, prof
        assert np.array_equal(got, oracle.sort_kv32(pairs))
    finally:
        dev.setParam("sort.msd2", 1)
        p.close()


def test_size_classes_run_the_kernels_they_claim(dev):
    """Every size class of the u32 sort, forced where a hint could decide otherwise, with the kernel names asserted."""
    set_algo(dev, (-1, 8, -1))
    p = Pprims()
    try:
        cases = [
            (8000, {}, {"small_sort_u32"}),
            (16000, {"sort.mid": 0}, {"small_sort_u32"}),
            (16000, {"sort.mid": 2}, {"mid_bucket_scatter_u32", "segment_sort_u32"}),   # u32 keys above 8 Ki: the two-launch form is faster
            (8193, {"sort.mid": 2}, {"mid_bucket_scatter_u32", "segment_sort_u32"}),
            (300007, {"sort.mid": 2}, {"mid_bucket_scatter_u32", "segment_sort_u32"}),
            (300007, {"sort.mid": 3}, {"mid_prep_u32", "onesweep_u32_8b", "segment_sort_u32"}),
            ((1 << 22) + 5, {"sort.msd2": 2}, LARGE_U32),
            ((1 << 22) + 5, {"sort.msd2": 3}, {"msd2s_prep", "msd2s_pass1_u32", "msd2s_pass2_u32", "msd2s_offsets", "segment_sort_wave_u32"}),
            ((1 << 22) + 5, {"sort.msd2": 5}, {"msd2s_prep", "msd2s_pass1_u32", "msd2h_pass2_u32", "msd2s_offsets", "segment_sort_wave_u32"}),
            (3000001, {"sort.msd2": 5}, {"msd2s_prep", "msd2s_pass1_u32", "msd2h_pass2_u32", "msd2s_offsets", "segment_sort_wave_u32"}),
            ((1 << 22) + 5, {"sort.msd2": 4}, LARGE_U32),
            ((1 << 22) + 5, {"sort.msd2": 0}, {"count_u32_8b", "scan_table", "scatter_u32_8b"}),
            ((1 << 23) + 5, {"sort.msd2": 0}, {"os_hist_u32", "os_hist_reduce", "os_tables", "onesweep_u32_8b"}),
        ]
        for n, knobs, want_names in cases:
            for k, v in knobs.items():
                dev.setParam(k, v)
            try:
                keys = oracle.keys_u32(n, seed=n & 0xff)
                got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, p, keys))
                assert set(prof) == set(want_names), (n, knobs, prof)
                assert np.array_equal(got, oracle.sort_u32(keys)), (n, knobs)
            finally:
                dev.setParam("sort.mid", 1)
                dev.setParam("sort.msd2", 1)
    finally:
        p.close()


def test_large_sort_partial_sort_bits(dev):
    """sortBits < key bits on the large sort (Pprims.cpp:357, :330): the stable form places its digits inside the sorted bits and
    everything is stable, so keys that agree there keep their input order -- bit-exact against the oracle's partial sort for u32
    keys, u64 keys and pairs; skewed keys send it to the safety net, whose last digit is then 4 bits wide and whose odd pass
    counts are copied back (Pprims.cpp:400-403)."""
    dev.setParam("sort.msd2", 2)
    p = Pprims()
    try:
        n = (1 << 22) + 1234
        k = oracle.keys_u32(n, seed=41)
        pr = oracle.pairs_kv32(n, seed=42)
        k64 = oracle.keys_u64(n, seed=43)
        for bits in (16, 20, 24, 28):
            got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, p, k, bits))
            assert "msd2s_pass1_u32" in prof, (bits, prof)
            assert np.array_equal(got, oracle.sort_u32_bits(k, bits)), bits
            got, prof = _profiled(dev, lambda: gpu_sort_kv(dev, p, pr, bits))
            assert "msd2s_pass1_kv32" in prof, (bits, prof)
            assert np.array_equal(got, oracle.sort_e64_bits(pr, bits)), bits
        for bits in (16, 28, 36, 44, 52, 60):
            got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k64, bits))
            assert "msd2s_pass1_u64" in prof, (bits, prof)
            assert np.array_equal(got, oracle.sort_e64_bits(k64, bits)), bits
        # keys whose sorted bits are skewed: the safety net (odd and even pass counts, a 4-bit last digit)
        skew = np.where(np.arange(n) % 8 != 0, k & np.uint32(0xfff000ff), k).astype(np.uint32)
        for bits in (20, 24, 28):
            assert np.array_equal(gpu_sort_u32(dev, p, skew, bits), oracle.sort_u32_bits(skew, bits)), ("skew", bits)
        skew64 = np.where(np.arange(n) % 8 != 0, k64 & np.uint64(0xffffffff000000ff), k64).astype(np.uint64)
        for bits in (36, 44):
            assert np.array_equal(gpu_sort_u64(dev, p, skew64, bits), oracle.sort_e64_bits(skew64, bits)), ("skew64", bits)
        skewp = np.where(np.arange(n) % 8 != 0, pr & np.uint64(0xfffffffffff000ff), pr).astype(np.uint64)
        for bits in (20, 28):
            assert np.array_equal(gpu_sort_kv(dev, p, skewp, bits), oracle.sort_e64_bits(skewp, bits)), ("skewp", bits)
        # a 28-bit sort at BASELINE config #2's size
        big = oracle.keys_u32(1 << 26, seed=44)
        got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, p, big, 28))
        assert "msd2s_pass1_u32" in prof, prof
        assert np.array_equal(got, oracle.sort_u32_bits(big, 28))
        DeviceUtils.waitForCompletion(dev)
    finally:
        dev.setParam("sort.msd2", 1)
        p.close()


def test_u64_keys_stable_passes_binning_finish_and_its_handover(dev):
    """Whole u64 keys on the large sort: stable MSD passes + the binning finish (one counting pass on the top bits below the
    digits, whole-key compares inside the bins).  Segments whose bins fill unevenly -- duplicates, constant bit fields -- are
    handed to the LSD finish's list form.  "sort.binfinish" = 2 takes the binning finish at every size; 0 the LSD finish;
    "sort.msd2" = 4 the cursor passes.  Bit-exact against the oracle every way."""
    p = Pprims()
    try:
        dev.setParam("sort.msd2", 3)        # the stable passes at every size (the automatic choice takes them from 32 Mi keys up)
        dev.setParam("sort.binfinish", 2)
        n = (1 << 23) + 11
        u = oracle.keys_u64(n, seed=5)
        cases = {
            "uniform": u,
            "top 20 bits unused": u >> np.uint64(20),
            "values below 2^20 (25 copies of every key: crowded bins)": u >> np.uint64(44),
            "2^18 distinct values spread over 64 bits": (u >> np.uint64(46)) * np.uint64(0x0000400010000401),
            "constant low dword": (u & np.uint64(0xffffffff00000000)) | np.uint64(0x9abcdef0),
            "constant bits 16..47 below the digits": u & np.uint64(0xffff00000000ffff),
            "low 16 bits only below the digits": (u & np.uint64(0xffff00000000ffff)) | np.uint64(0x0000123456780000),
            "values below 2^16 (nothing left for the finish)": u >> np.uint64(48),
            "sorted": np.sort(u),
            "all equal": np.full(n, 0xdeadbeefcafef00d, dtype=np.uint64),
        }
        for nm, k in cases.items():
            k = k.astype(np.uint64)
            got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
            assert set(prof) == LARGE_U64_BIN, (nm, prof)
            assert np.array_equal(got, oracle.sort_u64(k)), nm
        for n2 in ((1 << 20) + 77, 3000001, (1 << 25) + 4321, (1 << 27) + 99):   # all three tile tiers of the finish
            k = oracle.keys_u64(n2, seed=n2 & 0xff)
            assert np.array_equal(gpu_sort_u64(dev, p, k), oracle.sort_u64(k)), n2
        dev.setParam("sort.binfinish", 1)   # the default: binning where segments hold ~384 keys and more -- from 24 Mi keys in 65536
        k = oracle.keys_u64(n, seed=6)      # segments, and below 16 Mi keys, where the second digit is narrow (8 Mi: 8192 segments)
        got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
        assert set(prof) == LARGE_U64_BIN, prof
        assert np.array_equal(got, oracle.sort_u64(k))
        k = oracle.keys_u64((5 << 22) + 9, seed=6)   # 20 Mi keys in 65536 segments of ~320: the LSD finish
        got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
        assert set(prof) == LARGE_U64_LSD, prof
        assert np.array_equal(got, oracle.sort_u64(k))
        k = oracle.keys_u64((1 << 25) + 3, seed=7)
        got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
        assert set(prof) == LARGE_U64_BIN, prof
        assert np.array_equal(got, oracle.sort_u64(k))
        k = oracle.keys_u64((3 << 24) + 3, seed=7)
        dev.setParam("sort.msd2", 2)        # forced, forms as the automatic choice takes them: 48 Mi + 3 keys -> stable passes
        got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
        assert set(prof) == LARGE_U64_BIN, prof
        assert np.array_equal(got, oracle.sort_u64(k))
        dev.setParam("sort.msd2", 5)        # hybrid: stable first pass, cursor-placed second pass over its sub-slabs
        got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
        assert {"msd2s_pass1_u64", "msd2h_pass2_u64", "segment_sort_bin_u64"} <= set(prof), prof
        assert np.array_equal(got, oracle.sort_u64(k))
        dev.setParam("sort.msd2", 2)
        k = oracle.keys_u64((1 << 24) + 3, seed=8)   # ... 16 Mi + 3 keys -> cursor passes, narrow second digit, binning finish
        got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
        assert set(prof) == {"msd2_sample", "msd2_pass1_u64", "msd2_pass2_u64", "msd2_offsets", "segment_sort_bin_u64", "segment_sort_listed_e64"}, prof
        assert np.array_equal(got, oracle.sort_u64(k))
        k = oracle.keys_u64((1 << 25) + 3, seed=7)
        dev.setParam("sort.msd2", 4)        # cursor passes
        got, prof = _profiled(dev, lambda: gpu_sort_u64(dev, p, k))
        assert {"msd2_pass1_u64", "msd2_pass2_u64", "segment_sort_bin_u64"} <= set(prof), prof
        assert np.array_equal(got, oracle.sort_u64(k))
        DeviceUtils.waitForCompletion(dev)
    finally:
        dev.setParam("sort.msd2", 1)
        dev.setParam("sort.binfinish", 1)
        p.close()


def test_small_partition_keeps_off_the_paths_with_a_grid_barrier(dev):
    """A device (or partition) that cannot keep 256 workgroups resident must not take the paths whose safety nets hold a grid-wide
    barrier over 256 workgroups, nor the one-workgroup-per-bucket finish ("debug.resident_wgs" stands in for such a device): the
    per-digit passes run instead and the result is right."""
    set_algo(dev, (-1, 8, -1))
    p = Pprims()
    real = dev.getParam("debug.resident_wgs")
    assert real >= 256, real
    try:
        dev.setParam("debug.resident_wgs", 120)
        for n in (300007, (1 << 22) + 5):
            keys = oracle.keys_u32(n, seed=3)
            got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, p, keys))
            assert not any(k.startswith(("mid_", "msd2", "segment_sort")) for k in prof), prof
            assert np.array_equal(got, oracle.sort_u32(keys)), n
    finally:
        dev.setParam("debug.resident_wgs", 0)
        assert dev.getParam("debug.resident_wgs") == real
        p.close()


def test_hybrid_form_whole_u32_keys(dev):
    """Whole u32 keys through the hybrid form ("sort.msd2" = 5; the automatic choice from 192 Mi keys): stable first pass into
    sub-slabs, cursor-placed second pass over buckets made of those sub-slabs (tiles inside one sub-slab, across two, across
    many), 16-bit second slab.  Friendly, narrow, skewed (safety net) and outlier keys; bit-exact against the oracle."""
    dev.setParam("sort.msd2", 5)
    p = Pprims()
    try:
        for n in ((1 << 20) + 77, 2500000, (1 << 24) + 5, (1 << 26) + 999999):
            k = oracle.keys_u32(n, seed=n & 0xff)
            assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), ("uniform", n)
        n = (1 << 23) + 77
        u = oracle.keys_u32(n, seed=19)
        cases = {
            "one eighth of the key range": (u >> np.uint32(3)) | np.uint32(0xa0000000),
            "below 2^20": u >> np.uint32(12),
            "below 2^16 (nothing left for the finish)": u >> np.uint32(16),
            "first half of the input in few buckets (sub-slabs of very different sizes)":
                np.where(np.arange(n) < n // 2, u & np.uint32(0x07ffffff), u).astype(np.uint32),
            "one heavy top byte (safety net)": np.where(np.arange(n) % 10 != 0, u >> np.uint32(8), u).astype(np.uint32),
            "all equal (safety net)": np.full(n, 0xdeadbeef, dtype=np.uint32),
            "sorted": np.sort(u),
        }
        for nm, k in cases.items():
            assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), nm
        k = (u >> np.uint32(8)).copy()
        k[12345] = 0xf0000001
        assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), "outlier above the sampled range"
        DeviceUtils.waitForCompletion(dev)
        dev.setParam("sort.msd2", 2)   # the automatic forms: 200 Mi + 5 keys take the hybrid form (round 4: from 192 Mi keys; the
        n = (200 << 20) + 5            # persistent cursor passes are the faster ones below), 128 Mi + 5 keys the cursor form
        k = oracle.keys_u32(n, seed=55)
        got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, p, k))
        assert {"msd2s_pass1_u32", "msd2h_pass2_u32"} <= set(prof), prof
        assert np.array_equal(got, oracle.sort_u32(k))
        n = (1 << 27) + 5
        k = oracle.keys_u32(n, seed=55)
        got, prof = _profiled(dev, lambda: gpu_sort_u32(dev, p, k))
        assert {"msd2_pass1_u32", "msd2_pass2_u32", "segment_sort_wave_u32"} <= set(prof), prof
        assert np.array_equal(got, oracle.sort_u32(k))
    finally:
        dev.setParam("sort.msd2", 1)
        p.close()


def test_scratch_levels_and_graceful_degradation(dev):
    """The work buffer decides the path, never the result: with the minimum (level 0 = the reference's own contract, a digit
    table beside the partner array, Pprims.cpp:332-337) a sort runs the per-digit three-kernel passes; with the full-speed size for
    whole keys a partial-bit sort (whose stable form needs more) still runs, on the fastest path that fits; level 1 for those bits
    gives it the large sort.  Sizes are monotone in n at both levels for partial bits too."""
    import ctypes
    lib = _lib.load()
    n = (5 << 22) + 77   # 20 Mi keys: whole keys keep their 16-bit second slab in the partner array, a 28-bit sort cannot
    keys = oracle.keys_u32(n, seed=8)

    def sizes(kind, nn, bits, level):
        tb, wb = ctypes.c_size_t(), ctypes.c_size_t()
        assert lib.adlhip_radix_sort_scratch_bytes_for(dev._h, kind, nn, bits, level, ctypes.byref(tb), ctypes.byref(wb)) == 0
        return tb.value, wb.value

    t0, w0 = sizes(0, n, 32, 0)
    t1, w1 = sizes(0, n, 32, 1)
    t2, w2 = sizes(0, n, 28, 1)
    assert t0 == t1 == t2 and w0 < (4 << 20) < w1 < w2, (w0, w1, w2)
    set_algo(dev, (-1, 8, -1))
    dev.setParam("sort.msd2", 2)
    data, tmp = Buffer(dev, n, np.uint32), Buffer(dev, n, np.uint32)
    try:
        for work_bytes, bits, want in ((w0, 32, {"count_u32_8b", "scan_table", "scatter_u32_8b"}),
                                       (w1, 32, LARGE_U32),
                                       (w1, 28, None),          # the stable form's slabs do not fit: any slower path will do
                                       (w2, 28, {"msd2s_prep", "msd2s_pass1_u32", "msd2s_pass2_u32", "msd2s_offsets", "segment_sort_wave_u32"})):
            work = Buffer(dev, work_bytes, np.uint8)
            data.write(keys)
            dev.toggleProfiling(True); dev.profile(reset=True)
            check(lib.adlhip_radix_sort_u32(dev._h, data.ptr(), tmp.ptr(), work.ptr(), work_bytes, n, bits), "sort")
            prof = dev.profile(reset=True); dev.toggleProfiling(False)
            if want is not None:
                assert set(prof) == want, (work_bytes, bits, prof)
            else:
                assert "msd2s_pass1_u32" not in prof, prof
            assert np.array_equal(data.toHost(), oracle.sort_u32_bits(keys, bits)), (work_bytes, bits)
            work.release()
        # one byte less than the minimum is refused
        work = Buffer(dev, w0, np.uint8)
        assert lib.adlhip_radix_sort_u32(dev._h, data.ptr(), tmp.ptr(), work.ptr(), w0 - 1, n, 32) != 0
        assert b"minimum" in lib.adlhip_last_error()
        work.release()
    finally:
        dev.toggleProfiling(False)
        dev.setParam("sort.msd2", 1)
        data.release(); tmp.release()
    # level 2, lean: whole keys keep the large sort with 12 % of head-room in the first slabs -- 64 Mi u32 keys within 330 MB
    n = 1 << 26
    keys = oracle.keys_u32(n, seed=9)
    _, lean = sizes(0, n, 32, 2)
    _, full = sizes(0, n, 32, 1)
    assert sizes(0, n, 32, 0)[1] < lean <= 330 * 1000 * 1000 < full, (lean, full)
    assert sizes(2, 1 << 24, 64, 0)[1] < sizes(2, 1 << 24, 64, 2)[1] < sizes(2, 1 << 24, 64, 1)[1]
    data, tmp, work = Buffer(dev, n, np.uint32), Buffer(dev, n, np.uint32), Buffer(dev, lean, np.uint8)
    try:
        dev.setParam("sort.msd2", 2)
        # uniform keys: the large sort, no safety net; keys with a 30 % denser lower half: the net (correct all the same)
        skew = np.where(np.arange(n) % 10 < 3, keys >> np.uint32(1), keys).astype(np.uint32)
        for name, k, net in (("uniform", keys, False), ("denser lower half", skew, True)):
            data.write(k)
            dev.toggleProfiling(True); dev.profile(reset=True)
            check(lib.adlhip_radix_sort_u32(dev._h, data.ptr(), tmp.ptr(), work.ptr(), lean, n, 32), "sort")
            prof = dev.profile(reset=True); dev.toggleProfiling(False)
            assert set(prof) == LARGE_U32, (name, prof)
            assert (prof["msd2_offsets"][1] > 0.3) == net, (name, prof["msd2_offsets"])
            got = data.toHost()
            assert np.array_equal(got, np.sort(k)), name
        dev.checkFault()
    finally:
        dev.toggleProfiling(False)
        dev.setParam("sort.msd2", 1)
        data.release(); tmp.release(); work.release()
    rng = np.random.default_rng(6)
    sz = sorted(set([1 << k for k in range(12, 29)] + [(16 << 20) - 1, 16 << 20, (96 << 20) + 1] + [int(x) for x in rng.integers(1 << 12, 3 << 27, 120)]))
    for kind, bits in ((0, 28), (0, 16), (1, 24), (2, 44), (3, 20), (0, 32), (2, 64)):
        for level in (0, 1, 2):
            last = 0
            for nn in sz:
                w = sizes(kind, nn, bits, level)[1]
                assert w >= last, (kind, bits, level, nn, w, last)
                last = w


def test_narrow_second_digit_on_small_inputs(dev):
    """Below 16 Mi elements (u64 keys on the cursor form: at every size) the second MSD digit is narrower than 8 bits, so that
    segments hold about a thousand elements instead of a few dozen (hybrid_kernels.hpp slot_to_segment): 256 << w slabs and
    finish waves, the finish sorts 8 - w bits more.  Every width w = 2 ... 7 for u32 keys (whole-key second slab), u64 keys
    (binning finish) and pairs (stable form), with keys that leave top bits unused, duplicates and skew (safety net)."""
    p = Pprims()
    try:
        for forced in (4, 3):   # cursor form / stable form
            dev.setParam("sort.msd2", forced)
            for n in ((1 << 20) + 5, 1500000, 2500000, (1 << 22) + 77, 6000001, (1 << 23) + 1, 12000000, (1 << 24) - 3):
                k = oracle.keys_u32(n, seed=n & 0xff)
                assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), ("u32", forced, n)
                k64 = oracle.keys_u64(n, seed=n & 0xff)
                assert np.array_equal(gpu_sort_u64(dev, p, k64), oracle.sort_u64(k64)), ("u64", forced, n)
            n = 3000001
            u = oracle.keys_u32(n, seed=77)
            for nm, k in {"below 2^24": u >> np.uint32(8), "below 2^19": u >> np.uint32(13), "below 2^16": u >> np.uint32(16),
                          "one eighth of the range": (u >> np.uint32(3)) | np.uint32(0x40000000),
                          "4096 distinct keys": (u >> np.uint32(20)) * np.uint32(0x00100801),
                          "one heavy top byte (safety net)": np.where(np.arange(n) % 10 != 0, u >> np.uint32(8), u).astype(np.uint32),
                          "sorted": np.sort(u)}.items():
                assert np.array_equal(gpu_sort_u32(dev, p, k), oracle.sort_u32(k)), (nm, forced)
        dev.setParam("sort.msd2", 2)
        for n in ((1 << 20) + 5, 1500000, (1 << 22) + 77, 6000001, 12000000):
            pr = oracle.pairs_kv32(n, seed=n & 0xff) & np.uint64(0xffffffff000fffff)   # 20-bit keys: duplicates, stability
            got, prof = _profiled(dev, lambda: gpu_sort_kv(dev, p, pr))
            assert set(prof) == LARGE_PAIRS, (n, prof)
            assert np.array_equal(got, oracle.sort_kv32(pr)), ("pairs", n)
            assert np.array_equal(gpu_sort_kv(dev, p, pr, 20), oracle.sort_e64_bits(pr, 20)), ("pairs, 20 bits", n)
        DeviceUtils.waitForCompletion(dev)
    finally:
        dev.setParam("sort.msd2", 1)
        p.close()


def _net_stats(d):
    return d.getParam("stat.net_runs"), d.getParam("stat.net_counting")


def test_counting_sort_for_keys_with_few_distinct_values():
    """Whole-key sorts of u32 / u64 keys that take at most 256 distinct values: such keys cannot fit the large sort's slabs, its
    passes give up within the first tiles and the safety net inside the offsets kernel sorts them by COUNTING (dict_kernels.hpp):
    a dictionary from 16 Ki sampled keys, every key looked up and counted, the runs written in place.  A key the sample did not
    see, more than 256 values, pairs and partial sorts take the net's LSD passes.  Nothing is remembered between sorts: the
    first sort of a handle launches what the third does.  Bit-exact against the oracle."""
    rng = np.random.RandomState(11)
    n = 3000001

    def fresh():
        d = DeviceUtils.allocate()
        set_algo(d, (-1, 8, -1))
        return d, Pprims()

    def vals32(k):
        return rng.randint(0, 2**32, k, dtype=np.uint64).astype(np.uint32)

    for name, keys in (("all equal", np.full(n, 0x12345678, dtype=np.uint32)),
                       ("two values", vals32(2)[rng.randint(0, 2, n)]),
                       ("16 values", vals32(16)[rng.randint(0, 16, n)]),
                       ("256 values, skewed (the rarest has 0.1 % of the keys)",
                        vals32(256)[rng.choice(256, n, p=(1.0 / (np.arange(256) + 8)) / (1.0 / (np.arange(256) + 8)).sum())]),
                       ("low byte only", rng.randint(0, 256, n).astype(np.uint32)),
                       ("0 and 0xffffffff", np.where(rng.rand(n) < 0.3, np.uint32(0), np.uint32(0xffffffff)).astype(np.uint32)),
                       # (advisor, round 3) the all-ones key beside ~200 other values: its hash slot must not read as free
                       ("0xffffffff and 200 categories", np.concatenate([vals32(200), np.array([0xffffffff], dtype=np.uint32)])[rng.randint(0, 201, n)])):
        d, p = fresh()
        try:
            for rep in range(3):   # the first sort of a handle is like every other
                got, prof = _profiled(d, lambda: gpu_sort_u32(d, p, keys))
                assert set(prof) == LARGE_U32, (name, rep, prof)
                assert np.array_equal(got, oracle.sort_u32(keys)), (name, rep)
                assert _net_stats(d) == (rep + 1, rep + 1), (name, rep, _net_stats(d))   # the net ran, and sorted by counting
            d.checkFault()
        finally:
            p.close(); DeviceUtils.deallocate(d)
    # u64 keys, the all-ones key among the values
    v64 = np.concatenate([rng.randint(0, 2**63, 40, dtype=np.int64).astype(np.uint64) * np.uint64(2) + np.uint64(1),
                          np.array([0xffffffffffffffff, 0], dtype=np.uint64)])
    k64 = v64[rng.randint(0, v64.size, n)]
    d, p = fresh()
    try:
        got, prof = _profiled(d, lambda: gpu_sort_u64(d, p, k64))
        assert "msd2_offsets" in prof and "msd2_pass1_u64" in prof, prof   # (3 M u64 keys: the cursor form)
        assert np.array_equal(got, oracle.sort_u64(k64))
        assert _net_stats(d) == (1, 1)
        # a value the sample cannot see (one key in three million): the count phase misses it, the net's LSD passes sort
        odd = k64.copy()
        odd[1234567] = np.uint64(0x0123456789abcdef)
        got, prof = _profiled(d, lambda: gpu_sort_u64(d, p, odd))
        assert np.array_equal(got, oracle.sort_u64(odd))
        assert _net_stats(d) == (2, 1)
        # results stay right whatever comes next, and only the few-valued inputs reach the net
        for i, k in enumerate((odd, oracle.keys_u64(n, seed=3), k64, k64, oracle.keys_u64(n, seed=4), k64)):
            assert np.array_equal(gpu_sort_u64(d, p, k), oracle.sort_u64(k)), i
        assert _net_stats(d) == (6, 4)
        d.checkFault()
    finally:
        p.close(); DeviceUtils.deallocate(d)
    # what is not sorted by counting: 257+ values, pairs, partial sorts, the knob
    d, p = fresh()
    try:
        # (300 values of u32 keys: the larger dictionary, test_u32_keys_of_up_to_4096_values_are_sorted_by_counting; of u64 keys: LSD passes)
        many = vals32(300)[rng.randint(0, 300, n)]
        m64 = (many.astype(np.uint64) << np.uint64(32)) | many.astype(np.uint64)
        assert np.array_equal(gpu_sort_u64(d, p, m64), oracle.sort_u64(m64))
        assert _net_stats(d) == (1, 0)
        few = vals32(16)[rng.randint(0, 16, n)]
        pairs = few.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
        assert np.array_equal(gpu_sort_kv(d, p, pairs), oracle.sort_kv32(pairs))
        assert _net_stats(d) == (2, 1)   # (pairs: one stable pass on the key's rank in the dictionary, test_pairs_with_few_valued_keys...)
        assert np.array_equal(gpu_sort_kv(d, p, pairs, 24), oracle.sort_e64_bits(pairs, 24))
        assert _net_stats(d) == (3, 1)
        assert np.array_equal(gpu_sort_u32(d, p, few, 24), oracle.sort_u32_bits(few, 24))
        assert _net_stats(d) == (4, 1)
        d.setParam("sort.dict", 0)
        assert np.array_equal(gpu_sort_u32(d, p, few), oracle.sort_u32(few))
        assert _net_stats(d) == (5, 1)
        assert np.array_equal(gpu_sort_kv(d, p, pairs), oracle.sort_kv32(pairs))
        assert _net_stats(d) == (6, 1)
        d.setParam("sort.dict", 1)
        assert np.array_equal(gpu_sort_u32(d, p, few), oracle.sort_u32(few))
        assert _net_stats(d) == (7, 2)
        # keys that fit never see the net
        u = oracle.keys_u32(n, seed=8)
        assert np.array_equal(gpu_sort_u32(d, p, u), oracle.sort_u32(u))
        assert _net_stats(d) == (7, 2)
    finally:
        p.close(); DeviceUtils.deallocate(d)


def test_sample_flags_heavily_repeated_keys_before_the_passes_move_them():
    """The large sort's first kernel looks at 2048 sampled keys (hybrid_kernels.hpp sample_accumulate): when they repeat
    themselves so often that some value must outgrow its slab, the overflow flag is up before pass 1 starts, both passes leave at
    their first instruction and the net starts at once.  Keys that fit never trip it (uniform, sorted, keys computed from their
    index); keys computed from their index that DO repeat are caught although evenly spaced samples of them never would be.
    u32 / u64 keys and pairs, bit-exact; the passes' time shows whether they moved keys."""
    rng = np.random.RandomState(23)
    n = 1 << 25
    d = DeviceUtils.allocate()
    p = Pprims()
    try:
        set_algo(d, (-1, 8, -1))
        d.setParam("sort.msd2", 2)
        idx = np.arange(n, dtype=np.uint32)
        uniform = oracle.keys_u32(n, seed=5)
        (_, prof) = _profiled(d, lambda: gpu_sort_u32(d, p, uniform))
        real_pass = prof["msd2_pass1_u32"][1]
        assert real_pass > 0.04, prof   # 128 MiB in, 128 MiB out
        vals = rng.randint(0, 2**32, 3000, dtype=np.uint64).astype(np.uint32)
        cases = (("3000 values", vals[rng.randint(0, 3000, n)], True),
                 ("16 values", vals[rng.randint(0, 16, n)], True),
                 ("values from the index", (idx * np.uint32(2654435761) >> np.uint32(22)) * np.uint32(0x00400801), True),
                 ("one value on 10 % of the keys", np.where(rng.rand(n) < 0.10, np.uint32(77), uniform).astype(np.uint32), True),
                 ("uniform", uniform, False),
                 ("sorted", np.sort(uniform), False),
                 ("index times a constant", idx * np.uint32(2654435761), False))
        for name, keys, repeated in cases:
            runs = _net_stats(d)[0]
            got, prof = _profiled(d, lambda: gpu_sort_u32(d, p, keys))
            assert np.array_equal(got, np.sort(keys)), name
            if repeated:
                assert prof["msd2_pass1_u32"][1] < 0.3 * real_pass and prof["msd2_pass2_u32"][1] < 0.3 * real_pass, (name, prof)
                assert _net_stats(d)[0] == runs + 1, name
            else:
                assert prof["msd2_pass1_u32"][1] > 0.6 * real_pass, (name, prof)
                assert _net_stats(d)[0] == runs, name
        # the stable form: u64 keys and pairs (the sample looks at the KEY of a pair)
        few = vals[rng.randint(0, 1000, n)]
        k64 = (few.astype(np.uint64) << np.uint64(32)) | few.astype(np.uint64)
        got, prof = _profiled(d, lambda: gpu_sort_u64(d, p, k64))
        assert np.array_equal(got, np.sort(k64))
        passes = [ms for k, (c, ms) in prof.items() if "_pass" in k]
        assert len(passes) == 2 and max(passes) < 0.04, prof
        pairs = few.astype(np.uint64) | (idx.astype(np.uint64) << np.uint64(32))
        got, prof = _profiled(d, lambda: gpu_sort_kv(d, p, pairs))
        assert np.array_equal(got, pairs[np.argsort(few, kind="stable")])
        assert prof["msd2s_pass1_kv32"][1] < 0.04 and prof["msd2s_pass2_kv32"][1] < 0.04, prof
        # pairs with distinct keys and one repeated VALUE fit: the value is not looked at
        pairs = uniform.astype(np.uint64) | (np.uint64(9) << np.uint64(32))
        runs = _net_stats(d)[0]
        got = gpu_sort_kv(d, p, pairs)
        assert np.array_equal(got, np.sort(uniform).astype(np.uint64) | (np.uint64(9) << np.uint64(32)))
        assert _net_stats(d)[0] == runs
        d.checkFault()
    finally:
        p.close(); DeviceUtils.deallocate(d)


def test_pairs_with_few_valued_keys_take_one_stable_pass_on_the_dictionary_rank():
    """{key, value} pairs whose keys take at most 256 values (group-by keys with a payload): the net builds the dictionary of the
    KEYS and sorts by ONE stable pass on the key's rank in it (hybrid_kernels.hpp coop_dict_pair_sort, dict_kernels.hpp
    DictPairIO) instead of four LSD passes.  Stability is what the values show: equal keys keep their input order, bit-exact
    against the oracle (the reference's CPU sort is stable: RadixSort.cpp:58-104).  Edge cases: one value, the all-ones key among
    the values, 256 values with a rare one, an odd element count, a key the 16 Ki-sample dictionary cannot know (falls through to
    the LSD passes with the input untouched), 257 values (no dictionary), sort.rank = 0."""
    rng = np.random.RandomState(31)
    d = DeviceUtils.allocate()
    p = Pprims()
    try:
        set_algo(d, (-1, 8, -1))

        def check(keys, counted, name):
            n = keys.size
            pairs = keys.astype(np.uint64) | (rng.randint(0, 2**32, n, dtype=np.uint64) << np.uint64(32))
            runs, cnt = _net_stats(d)
            got = gpu_sort_kv(d, p, pairs)
            assert np.array_equal(got, oracle.sort_kv32(pairs)), name
            assert _net_stats(d) == (runs + 1, cnt + (1 if counted else 0)), (name, _net_stats(d))

        def vals(k):
            return rng.randint(0, 2**32, k, dtype=np.uint64).astype(np.uint32)

        n = 3000001
        check(np.full(n, 0xdeadbeef, dtype=np.uint32), True, "one value")
        check(vals(2)[rng.randint(0, 2, n)], True, "two values")
        check(vals(40)[rng.randint(0, 40, n)], True, "40 values")
        check(np.concatenate([vals(100), np.array([0xffffffff, 0], dtype=np.uint32)])[rng.randint(0, 102, n)], True, "0 and the all-ones key among the values")
        check(np.where(rng.rand(n) < 0.5, np.uint32(0xffffffff), np.uint32(0xfffffffe)).astype(np.uint32), True, "all ones and its neighbour")
        w = 1.0 / (np.arange(256) + 8)
        check(vals(256)[rng.choice(256, n, p=w / w.sum())], True, "256 values, skewed")
        check(np.sort(vals(64)[rng.randint(0, 64, n)]), True, "64 values, keys already in order")
        check(rng.randint(0, 256, 1 << 23).astype(np.uint32), True, "8 Mi pairs, low byte only")
        odd = vals(16)[rng.randint(0, 16, n)]
        odd[2345678] = np.uint32(0x01234567)
        check(odd, False, "a key the sample cannot see")
        check(vals(300)[rng.randint(0, 300, n)], False, "300 values")
        d.setParam("sort.rank", 0)
        check(vals(40)[rng.randint(0, 40, n)], True, "40 values, ballot ranking")
        d.setParam("sort.rank", 1)
        # keys and values on separate arrays (SortAndScatterKernel's layout): the net packs them, takes the same pass, unpacks
        for nv, counted in ((1, True), (30, True), (256, True), (400, False)):
            keys = vals(nv)[rng.randint(0, nv, n)]
            values = rng.randint(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
            kb = Buffer(d, n, np.uint32); vb = Buffer(d, n, np.uint32)
            kb.write(keys); vb.write(values)
            runs, cnt = _net_stats(d)
            p.radixSortSoA(d, kb, vb, n)
            gk, gv = kb.toHost(), vb.toHost()
            kb.release(); vb.release()
            wk, wv = oracle.sort_soa(keys, values)
            assert np.array_equal(gk, wk) and np.array_equal(gv, wv), ("SoA", nv)
            assert _net_stats(d) == (runs + 1, cnt + (1 if counted else 0)), ("SoA", nv, _net_stats(d))
        # the 64-Mi row of the benchmark, once: sortedness by key, and every key's values in input order
        n = 1 << 26
        keys = vals(256)[rng.randint(0, 256, n)]
        pairs = keys.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
        got = gpu_sort_kv(d, p, pairs)
        gk = (got & np.uint64(0xffffffff)).astype(np.uint32)
        gv = (got >> np.uint64(32)).astype(np.uint32)
        assert np.all(gk[1:] >= gk[:-1])
        assert np.all((gv[1:] > gv[:-1]) | (gk[1:] != gk[:-1]))   # values = input positions: ascending inside every run of equal keys
        assert np.array_equal(keys[gv], gk)
        d.checkFault()
    finally:
        p.close(); DeviceUtils.deallocate(d)


def test_u32_keys_of_up_to_4096_values_are_sorted_by_counting():
    """u32 keys that repeat up to 4096 values -- categories, codes, dates -- end in the large sort's net like all keys that repeat,
    and are sorted there by COUNTING with the larger dictionary (dict_big_kernels.hpp: 64 Ki samples, values ordered by a bitonic sort
    in LDS, 8192 hash slots, one set of counters per workgroup) instead of four LSD passes.  Bit-exact against the oracle
    (RadixSort.cpp:58-104).  Edge cases: 257 and 4096 values, the all-ones key and 0 among them, skew, sorted input, a key the
    sample cannot know and 5000 values (both fall through to the LSD passes), element counts that are not multiples of four."""
    rng = np.random.RandomState(41)
    d = DeviceUtils.allocate()
    p = Pprims()
    try:
        set_algo(d, (-1, 8, -1))

        def vals(k):
            v = np.unique(rng.randint(0, 2**32, 2 * k, dtype=np.uint64).astype(np.uint32))
            rng.shuffle(v)
            return v[:k]

        def check(keys, counted, name):
            runs, cnt = _net_stats(d)
            got = gpu_sort_u32(d, p, keys)
            assert np.array_equal(got, oracle.sort_u32(keys)), name
            assert _net_stats(d) == (runs + 1, cnt + (1 if counted else 0)), (name, _net_stats(d))

        n = 3000003
        check(vals(257)[rng.randint(0, 257, n)], True, "257 values")
        check(vals(1000)[rng.randint(0, 1000, n)], True, "1000 values")
        check(vals(4096)[rng.randint(0, 4096, n + 2)], True, "4096 values")
        check(np.concatenate([vals(2000), np.array([0xffffffff, 0], dtype=np.uint32)])[rng.randint(0, 2002, n)], True, "0 and the all-ones key among 2002")
        w = 1.0 / (np.arange(1500) + 300)
        check(vals(1500)[rng.choice(1500, n, p=w / w.sum())], True, "1500 values, skewed (the rarest: 1 key in 3200)")
        # values so rare that 64 Ki samples miss some of them: the count stops at the first miss, the LSD passes sort
        w = 1.0 / (np.arange(3000) + 2) ** 1.3
        check(vals(3000)[rng.choice(3000, n, p=w / w.sum())], False, "3000 values, Zipf-like")
        check(np.sort(vals(1500)[rng.randint(0, 1500, n + 1)]), True, "1500 values, in order")
        check((np.arange(n, dtype=np.uint32) * np.uint32(2654435761) >> np.uint32(20)) * np.uint32(0x00100801), True, "4096 values from the index")
        odd = vals(800)[rng.randint(0, 800, n)]
        odd[1234567] = np.uint32(0x01234567)
        check(odd, False, "a key the sample cannot see")
        check(vals(5000)[rng.randint(0, 5000, 1 << 23)], False, "5000 values")
        # one sort at the benchmark's size: multiset and order
        n = 1 << 26
        keys = vals(4000)[rng.randint(0, 4000, n)]
        got = gpu_sort_u32(d, p, keys)
        assert np.all(got[1:] >= got[:-1]) and np.array_equal(np.bincount(np.searchsorted(np.unique(keys), got), minlength=4000),
                                                               np.bincount(np.searchsorted(np.unique(keys), keys), minlength=4000))
        d.checkFault()
    finally:
        p.close(); DeviceUtils.deallocate(d)


def test_safety_net_beside_a_second_handle_that_keeps_the_cus_busy():
    """The safety nets hold a grid-wide barrier over 256 workgroups.  Here they run while a second handle on another stream sorts
    64 Mi keys over and over -- its workgroups compete for the same CUs -- on skewed keys with the large sort forced (u32 keys:
    the net inside the offsets kernel; pairs: the stable form's; low-cardinality u64 keys with a value outside the dictionary:
    the net inside the fill kernel): every spin is bounded, so the outcome is a right result and no fault word, never a hang."""
    import threading
    d1, d2 = DeviceUtils.allocate(), DeviceUtils.allocate()
    p1, p2 = Pprims(), Pprims()
    n = (1 << 22) + 99
    stop = threading.Event()
    err = []

    def hammer():
        try:
            b = Buffer(d2, 1 << 26, np.uint32)
            i = 0
            while not stop.is_set():
                b.generate(1 << 26, seed=1000 + i)
                for _ in range(8):
                    p2.radixSort(d2, b, 1 << 26)
                DeviceUtils.waitForCompletion(d2)
                i += 1
            b.release()
        except Exception as e:   # pragma: no cover
            err.append(e)

    t = threading.Thread(target=hammer)
    t.start()
    try:
        d1.setParam("sort.msd2", 2)
        u = oracle.keys_u32(n, seed=5)
        skew = np.where(np.arange(n) % 10 != 0, u >> np.uint32(9), u).astype(np.uint32)
        for rep in range(3):
            assert np.array_equal(gpu_sort_u32(d1, p1, skew), oracle.sort_u32(skew)), rep
            pairs = skew.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
            assert np.array_equal(gpu_sort_kv(d1, p1, pairs), oracle.sort_kv32(pairs)), rep
        d1.setParam("sort.msd2", 1)
        k64 = (oracle.keys_u64(n, seed=6) % np.uint64(5)) * np.uint64(0x0101010101010101)
        assert np.array_equal(gpu_sort_u64(d1, p1, k64), oracle.sort_u64(k64))       # counting sort
        k64[777] = np.uint64(0x123456789)                                             # ... and its safety net
        assert np.array_equal(gpu_sort_u64(d1, p1, k64), oracle.sort_u64(k64))
        d1.checkFault()
    finally:
        stop.set()
        t.join()
        d1.setParam("sort.msd2", 1)
        p1.close(); p2.close()
        DeviceUtils.deallocate(d1); DeviceUtils.deallocate(d2)
    assert not err, err


def test_first_sort_of_a_fresh_handle_places_its_digits_from_the_sample():
    """The sample words of the large sort idle at or = 0 / and = ~0; a fresh handle must start from those values too, or its
    first sort of keys that leave their top bits constant puts the first digit on those bits, overflows and goes through the
    safety net (right result, 7 x the time).  The offsets kernel is where the net would run: it must take microseconds."""
    n = (1 << 22) + 11
    keys = (oracle.keys_u32(n, seed=4) >> np.uint32(3)) | np.uint32(0xa0000000)   # one eighth of the key range
    d = DeviceUtils.allocate()
    p = Pprims()
    try:
        d.setParam("sort.msd2", 4)
        got, prof = _profiled(d, lambda: gpu_sort_u32(d, p, keys))
        assert "msd2_offsets" in prof and prof["msd2_offsets"][1] < 0.1, prof
        assert np.array_equal(got, oracle.sort_u32(keys))
    finally:
        p.close(); DeviceUtils.deallocate(d)


def test_few_valued_and_other_keys_back_to_back():
    """Few-valued keys (the net's counting sort), then keys that are nothing like that, several sorts queued back to back with no
    synchronisation in between, then few-valued keys again: every net starts its grid-barrier counter, its dictionary and its
    counters from scratch, and no sort depends on what the one before it met.  (Round 3 steered this by a host-side hint; a net
    that started from a stale barrier counter was found by tools/stress.py then.)"""
    n = (1 << 22) + 321
    d = DeviceUtils.allocate()
    set_algo(d, (-1, 8, -1))
    p = Pprims()
    try:
        few = (oracle.keys_u32(n, seed=1) % np.uint32(5)) * np.uint32(0x01010101)
        assert np.array_equal(gpu_sort_u32(d, p, few), oracle.sort_u32(few))
        assert _net_stats(d) == (1, 1)
        nets = 1
        for kind in ("u32", "u64"):
            keys = oracle.keys_u32(n, seed=2) if kind == "u32" else oracle.keys_u64(n, seed=2)
            fk = few if kind == "u32" else few.astype(np.uint64) * np.uint64(0x100000001)
            bufs = [Buffer(d, n, keys.dtype) for _ in range(6)]
            for i, b in enumerate(bufs):
                b.write(fk if i in (1, 4) else keys)
            DeviceUtils.waitForCompletion(d)
            for b in bufs:                        # six sorts queued without a synchronisation in between
                (p.radixSort if kind == "u32" else p.radixSort64)(d, b, n)
            want = oracle.sort_u32(keys) if kind == "u32" else oracle.sort_u64(keys)
            wantf = oracle.sort_u32(fk) if kind == "u32" else oracle.sort_u64(fk)
            for i, b in enumerate(bufs):
                assert np.array_equal(b.toHost(), wantf if i in (1, 4) else want), (kind, i)
                b.release()
            nets += 2
            assert _net_stats(d) == (nets, nets)
        d.checkFault()
    finally:
        p.close(); DeviceUtils.deallocate(d)


@pytest.mark.gpu
def test_probe_sample_positions_stay_inside_the_array(dev):
    """The key probe reads 16384 samples, one somewhere inside every 16384th of the array.  The first version took the offset inside
    the cell as `hash % cell`; the compiler expands that remainder (operands known to fit 24 bits) through float and overshoots
    for some operands, so that 13 samples of a 7726351-key array were read 64 MiB past its end -- a GPU memory fault when nothing
    is mapped there (tools/stress.py, iteration 2225).  adlhip_selftest_probe_positions evaluates the device code's positions."""
    lib = _lib.load()
    rng = np.random.RandomState(5)
    sizes = [16384, 16385, 20543, 730669, 7726351, (1 << 24) - 1, 1 << 26, (1 << 28) + 12345, (1 << 30) + 7]
    sizes += [int(2 ** rng.uniform(14.1, 30)) for _ in range(300)]
    for n in sizes:
        hi, bad = ctypes.c_uint32(0), ctypes.c_uint32(1)
        check(lib.adlhip_selftest_probe_positions(dev._h, n, ctypes.byref(hi), ctypes.byref(bad)), "probe positions")
        assert bad.value == 0 and hi.value < n, (n, hi.value, bad.value)
        assert hi.value >= n - 2 * (n // 16384) - 2, (n, hi.value)   # the last cell is sampled too
    # the sizes that raised the fault, on keys that send the sort to its net -- where the samples are read since round 4
    for kind, n in (("u32", 7726351), ("u64", 730669)):
        d = DeviceUtils.allocate()
        set_algo(d, (-1, 8, -1))
        p = Pprims()
        try:
            if kind == "u32":
                k = (oracle.keys_u32(n, seed=2225) >> np.uint32(29)) * np.uint32(0x01020304)
                assert np.array_equal(gpu_sort_u32(d, p, k), oracle.sort_u32(k))
            else:
                k = (oracle.keys_u64(n, seed=77) >> np.uint64(61)) * np.uint64(0x0102030405060708)
                assert np.array_equal(gpu_sort_u64(d, p, k), oracle.sort_u64(k))
            assert _net_stats(d) == (1, 1)
            d.checkFault()
        finally:
            p.close(); DeviceUtils.deallocate(d)


def test_config4_memory_budget_on_one_device():
    """BASELINE config #4 per GPU (review, round 3, item 7c): HipBackend.reserve(2^27) -- receive slots with 25 % head-room and
    partition slots, three deep, plus both stages' level-1 scratch -- must fit one MI355X many times over.  Measured on the device:
    what torch and the library actually hold afterwards."""
    import torch
    from oclradixsort_amd.dist import HipBackend
    torch.cuda.synchronize()
    free0, total = torch.cuda.mem_get_info()
    be = HipBackend()
    be.reserve(1 << 27)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    used = free0 - free1
    assert used < 16 * (1 << 30), used            # measured: about 6 GiB of 288 GB
    assert used > 3 * (int(1.25 * (1 << 27)) * 4)  # (at least the three receive slots: the reservation really happened)
    assert total > 200 * (1 << 30)
    be.close() if hasattr(be, "close") else None


def test_partition_top_byte_one_lookback_pass_equals_three_kernels(dev):
    """The multi-GPU send side (review, round 3, item 7b): from 24 MiB of data the top-byte partition is ONE look-back pass (the
    one-sweep path's histogram + chain kernel with a one-pass plan) instead of count -> scan -> scatter.  Both are stable, so the
    output and the 256 totals must be the same bit for bit -- checked against numpy's stable argsort and against the three-kernel
    form ("partition.lookback" = 0), for keys and for pairs, uniform and skewed, with the kernel names asserted."""
    lib = _lib.load()
    for kind, n in (("u32", (1 << 23) + 4321), ("kv32", (1 << 22) + 77), ("u32", (1 << 26) + 5)):
        pairs = kind == "kv32"
        dtype = np.uint64 if pairs else np.uint32
        for skew in (False, True):
            k = oracle.keys_u32(n, seed=31 + n % 7)
            if skew:
                k = np.where(np.arange(n) % 10 != 0, k >> np.uint32(3), k).astype(np.uint32)
            host = (k.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32))) if pairs else k
            order = np.argsort(k >> np.uint32(24), kind="stable")
            want, want_tot = host[order], np.bincount((k >> np.uint32(24)).astype(np.int64), minlength=256).astype(np.uint32)
            tb, wb = ctypes.c_size_t(), ctypes.c_size_t()
            check(lib.adlhip_radix_sort_scratch_bytes(dev._h, 1 if pairs else 0, n, ctypes.byref(tb), ctypes.byref(wb)), "scratch")
            src, dst, tot, work = Buffer(dev, n, dtype), Buffer(dev, n, dtype), Buffer(dev, 256, np.uint32), Buffer(dev, wb.value, np.uint8)
            src.write(host)
            fn = lib.adlhip_partition_top_byte_kv32 if pairs else lib.adlhip_partition_top_byte_u32
            try:
                for lookback in (1, 0):
                    dev.setParam("partition.lookback", lookback)
                    dst.clear(); tot.clear()
                    dev.toggleProfiling(True); dev.profile(reset=True)
                    check(fn(dev._h, src.ptr(), dst.ptr(), tot.ptr(), work.ptr(), wb.value, n), "partition")
                    prof = dev.profile(reset=True); dev.toggleProfiling(False)
                    names = set(prof)
                    if lookback:
                        assert any(x.startswith("onesweep_") for x in names) and "os_fold_totals" in names, names
                    else:
                        assert any(x.startswith("scatter_") for x in names) and not any(x.startswith("onesweep_") for x in names), names
                    assert np.array_equal(dst.toHost(), want), (kind, n, skew, lookback)
                    assert np.array_equal(tot.toHost(), want_tot), (kind, n, skew, lookback)
                    assert np.array_equal(src.toHost(), host)      # the input is left intact
                dev.checkFault()
            finally:
                dev.toggleProfiling(False)
                dev.setParam("partition.lookback", 1)
                for b in (src, dst, tot, work):
                    b.release()


def test_lean_scratch_level_for_pairs_keeps_the_large_sort(dev):
    """Level 2 of adlhip_radix_sort_scratch_bytes_for for {key, value} pairs (review, round 3, item 8): the stable form with
    statistical head-room only (mean + 8 sd / + 7.5 sd instead of + 50 %) -- 64 Mi pairs within 1.3 GB instead of 1.7.  Evenly
    spread keys take the large sort's kernels with that buffer; keys that are 20 % denser in one half take the net; both bit-exact."""
    lib = _lib.load()
    set_algo(dev, (-1, 8, -1))

    def sizes(kind, n, bits, level):
        tb, wb = ctypes.c_size_t(), ctypes.c_size_t()
        check(lib.adlhip_radix_sort_scratch_bytes_for(dev._h, kind, n, bits, level, ctypes.byref(tb), ctypes.byref(wb)), "scratch")
        return tb.value, wb.value

    assert sizes(1, 1 << 26, 32, 0)[1] < sizes(1, 1 << 26, 32, 2)[1] <= 1300 * 1000 * 1000 < sizes(1, 1 << 26, 32, 1)[1]
    n = (1 << 23) + 77
    _, lean = sizes(1, n, 32, 2)
    _, full = sizes(1, n, 32, 1)
    assert lean < 0.85 * full
    pairs = oracle.pairs_kv32(n, seed=12) & np.uint64(0xffffffff00ffffff)          # 24-bit keys: duplicates, stability visible
    k = (pairs & np.uint64(0xffffffff)).astype(np.uint32)
    skew = np.where(np.arange(n) % 10 < 2, k >> np.uint32(1), k).astype(np.uint64) | (pairs & np.uint64(0xffffffff00000000))
    data, tmp, work = Buffer(dev, n, np.uint64), Buffer(dev, n, np.uint64), Buffer(dev, lean, np.uint8)
    try:
        runs0 = dev.getParam("stat.net_runs")
        for name, host, net in (("even", pairs, 0), ("denser lower half", skew, 1)):
            data.write(host)
            dev.toggleProfiling(True); dev.profile(reset=True)
            check(lib.adlhip_radix_sort_kv32(dev._h, data.ptr(), tmp.ptr(), work.ptr(), lean, n, 32), "sort")
            prof = dev.profile(reset=True); dev.toggleProfiling(False)
            assert set(prof) == LARGE_PAIRS, (name, prof)
            assert np.array_equal(data.toHost(), oracle.sort_kv32(host)), name
            assert dev.getParam("stat.net_runs") - runs0 == net, name
            runs0 += net
        dev.checkFault()
    finally:
        dev.toggleProfiling(False)
        for b in (data, tmp, work):
            b.release()


def test_rank0_pairs_take_the_stable_large_sort(dev):
    """"sort.rank" = 0 -- no reliance on the lane order of colliding returning DS atomics -- used to send every sort to the per-digit
    passes.  Pairs now keep the stable large sort: its look-back passes, its wave-per-segment finish and its safety net have
    ballot-ranked variants (review, round 3, item 5).  Stability-revealing inputs (few distinct keys, value = index), sizes across
    the finish's tiles, SoA arrays, a skewed input through the net; rank 1 and rank 0 must agree bit for bit with the oracle."""
    p = Pprims()
    try:
        for n in ((1 << 21) + 33, (1 << 23) + 5, (1 << 25) + 77, (1 << 26)):
            for name, mask in (("24-bit keys", 0xffffffff00ffffff), ("256 distinct keys", 0xffffffff000000ff)):
                pairs = oracle.pairs_kv32(n, seed=n & 0xff) & np.uint64(mask)
                want = oracle.sort_kv32(pairs)
                for rank in (0, 1):
                    set_algo(dev, (-1, 8, -1, rank))
                    got, prof = _profiled(dev, lambda: gpu_sort_kv(dev, p, pairs))
                    if name == "24-bit keys":
                        assert set(prof) == LARGE_PAIRS, (n, rank, prof)        # the large sort, whichever way it ranks
                    assert np.array_equal(got, want), (n, name, rank)
        # SoA arrays and a skewed input (the net) with rank 0
        set_algo(dev, (-1, 8, -1, 0))
        n = (1 << 22) + 11
        k = oracle.keys_u32(n, seed=3) & np.uint32(0x0000ffff)
        v = np.arange(n, dtype=np.uint32)
        kb, vb = Buffer(dev, n, np.uint32), Buffer(dev, n, np.uint32)
        kb.write(k); vb.write(v)
        p.radixSortSoA(dev, kb, vb, n)
        order = np.argsort(k, kind="stable")
        assert np.array_equal(kb.toHost(), k[order]) and np.array_equal(vb.toHost(), v[order])
        kb.release(); vb.release()
        runs = dev.getParam("stat.net_runs")
        skew = np.where(np.arange(n) % 10 != 0, k >> np.uint32(9), k | np.uint32(0xffff0000)).astype(np.uint64) | (v.astype(np.uint64) << np.uint64(32))
        dev.setParam("sort.msd2", 2)
        assert np.array_equal(gpu_sort_kv(dev, p, skew), oracle.sort_kv32(skew))
        assert dev.getParam("stat.net_runs") == runs + 1
        dev.checkFault()
    finally:
        dev.setParam("sort.msd2", 1)
        set_algo(dev, (-1, 8, -1))
        p.close()


def test_net_passes_by_lookback_and_by_count_scan_scatter_agree(dev):
    """The safety net's LSD passes exist twice: look-back passes (the one-sweep path's histogram, tables and tile body, taken in
    turns by the net's resident workgroups; "sort.net_lookback" = 1, the default for whole keys) and count -> scan -> scatter passes
    with per-workgroup carries (0; also what sorts on part of the key and SoA arrays get).  Skewed u32 keys, u64 keys and pairs
    through both, bit-exact against the oracle; keys with constant bytes (passes skipped, an odd number left to run) too."""
    p = Pprims()
    n = (1 << 22) + 4321
    u = oracle.keys_u32(n, seed=41)
    heavy = np.where(np.arange(n) % 10 != 0, (u >> np.uint32(8)) | np.uint32(0x37000000), u).astype(np.uint32)
    odd = (heavy & np.uint32(0x00ffffff)) | np.uint32(0x5a000000)                      # constant top byte: three passes vary
    k64 = (heavy.astype(np.uint64) << np.uint64(32)) | u.astype(np.uint64)
    k64c = (heavy.astype(np.uint64) << np.uint64(24)) | np.uint64(0x7700000000000000)   # constant bytes in both 32-bit halves
    pairs = heavy.astype(np.uint64) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
    pairs_few = (u % np.uint32(1000)).astype(np.uint64) * np.uint64(0x10001) | (np.arange(n, dtype=np.uint64) << np.uint64(32))
    try:
        dev.setParam("sort.msd2", 2)
        for lookback in (1, 0):
            dev.setParam("sort.net_lookback", lookback)
            runs = dev.getParam("stat.net_runs")
            assert np.array_equal(gpu_sort_u32(dev, p, heavy), oracle.sort_u32(heavy)), lookback
            assert np.array_equal(gpu_sort_u32(dev, p, odd), oracle.sort_u32(odd)), lookback
            assert np.array_equal(gpu_sort_u64(dev, p, k64), oracle.sort_u64(k64)), lookback
            assert np.array_equal(gpu_sort_u64(dev, p, k64c), oracle.sort_u64(k64c)), lookback
            assert np.array_equal(gpu_sort_kv(dev, p, pairs), oracle.sort_kv32(pairs)), lookback
            assert np.array_equal(gpu_sort_kv(dev, p, pairs_few), oracle.sort_kv32(pairs_few)), lookback
            assert dev.getParam("stat.net_runs") >= runs + 4, lookback   # (the keys with a constant top byte fit once the digits are placed)
        dev.checkFault()
    finally:
        dev.setParam("sort.net_lookback", 1)
        dev.setParam("sort.msd2", 1)
        p.close()
