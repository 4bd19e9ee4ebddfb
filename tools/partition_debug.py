import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oclradixsort_amd import Buffer, DeviceUtils, _lib
from oclradixsort_amd._lib import check
lib = _lib.load()
dev = DeviceUtils.allocate()
for n in ((1 << 23), (1 << 23) + 4321, (1 << 23) + 16384 * 3):
    k = oracle.keys_u32(n, seed=31 + n % 7)
    order = np.argsort(k >> np.uint32(24), kind="stable")
    want = k[order]
    tb, wb = ctypes.c_size_t(), ctypes.c_size_t()
    check(lib.adlhip_radix_sort_scratch_bytes(dev._h, 0, n, ctypes.byref(tb), ctypes.byref(wb)), "scratch")
    src, dst, tot, work = Buffer(dev, n, np.uint32), Buffer(dev, n, np.uint32), Buffer(dev, 256, np.uint32), Buffer(dev, wb.value, np.uint8)
    src.write(k); dst.clear()
    check(lib.adlhip_partition_top_byte_u32(dev._h, src.ptr(), dst.ptr(), tot.ptr(), work.ptr(), wb.value, n), "partition")
    got = dst.toHost()
    bad = np.nonzero(got != want)[0]
    print("n", n, "mismatches", bad.size, "first", bad[:5], "last", bad[-5:] if bad.size else None)
    if bad.size:
        i = bad[0]
        print(" got", [hex(x) for x in got[i - 2:i + 4]], "want", [hex(x) for x in want[i - 2:i + 4]])
        print(" multiset equal:", np.array_equal(np.sort(got), np.sort(want)), " top bytes sorted:", bool(np.all(np.diff((got >> 24).astype(np.int64)) >= 0)))
        # where does want[i] sit in the input, and got[i]?
        pos = {int(v): int(j) for j, v in enumerate(k[:0])}
        print(" input index of want[i]:", np.nonzero(k == want[i])[0][:3], "of got[i]:", np.nonzero(k == got[i])[0][:3])
    print(" totals ok:", np.array_equal(tot.toHost(), np.bincount((k >> 24).astype(np.int64), minlength=256).astype(np.uint32)))
    for b in (src, dst, tot, work): b.release()
