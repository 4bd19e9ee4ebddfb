"""Python mirror of `Tahoe::Pprims` (Tahoe/ParallelPrimitives/Pprims.h:11-48) over the HIP back-end.

    p = Pprims()
    p.radixSort(device, buffer_u32, n, sortBits=32)        # Pprims.h:41
    p.radixSort(device, buffer_uint2, n, sortBits=32)      # Pprims.h:38  (dtype uint64 = {key, value})
    p.radixSort64(device, buffer_u64, n, sortBits=64)      # 64-bit keys (no reference counterpart)
    p.scan(device, dst, src, n, sumOut=None)               # Pprims.h:35

Like the reference object it owns lazily grown device scratch (m_u32WorkBuffer[0] = ping-pong data
buffer, m_u32WorkBuffer[1] = histogram table; Pprims.h:44-45, Pprims.cpp:226-232, :332-337) and must be
destroyed (close()) before DeviceUtils.deallocate, which refuses while memory is live (Adl.inl:102).
Calls enqueue and return without synchronising, as the reference's GPU branches do.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import AdlHipError, check
from .adl import Buffer

ELEM_U32 = 0
ELEM_KV32 = 1
ELEM_U64 = 2
ELEM_SOA32 = 3


class Pprims:
    SCAN_BLOCK_SIZE = 128               # Pprims.h:24 (kept for API parity; unused by the HIP kernels)
    R32SORT_DATA_ALIGNMENT = 256        # Pprims.h:28: the reference needs n % 256 == 0; this build does not
    R32SORT_WG_SIZE = 64                # Pprims.h:29
    R32SORT_BITS_PER_PASS = 4           # Pprims.h:31: available via device.setParam("sort.digit_bits", 4)

    def __init__(self):
        self.m_tmp = None       # m_u32WorkBuffer[0]
        self.m_work = None      # m_u32WorkBuffer[1]
        self.m_cacheKernel = True
        self._sum_keep = None

    def cacheKernel(self, cache):   # Pprims.h:20 -- kernels are compiled ahead of time; nothing to cache
        self.m_cacheKernel = bool(cache)

    def close(self):
        for b in (self.m_tmp, self.m_work):
            if b is not None:
                b.release()
        self.m_tmp = self.m_work = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scratch (uArray::setSize semantics: grow-only, contents not preserved)
    def _scratch(self, device, tmp_bytes, work_bytes):
        if self.m_tmp is None or self.m_tmp.m_device is not device:
            self.close()
            self.m_tmp = Buffer(device, 0, np.uint8)
            self.m_work = Buffer(device, 0, np.uint8)
        if self.m_tmp.getSize() < tmp_bytes:
            self.m_tmp.setSize(tmp_bytes)
        if self.m_work.getSize() < work_bytes:
            self.m_work.setSize(work_bytes)

    def reserve(self, device, kind, n):
        """Size the scratch for sorts of up to n elements of `kind` now, so that later calls do not grow it
        (growing syncs the device and reallocates, uArray.h:124-132)."""
        tb = ctypes.c_size_t()
        wb = ctypes.c_size_t()
        check(_lib.load().adlhip_radix_sort_scratch_bytes(device._h, kind, int(n), ctypes.byref(tb), ctypes.byref(wb)),
              "adlhip_radix_sort_scratch_bytes")
        self._scratch(device, (2 if kind == ELEM_SOA32 else 1) * tb.value, wb.value)

    def _sort(self, device, kind, fn, inout, n, sortBits):
        if device is None:
            raise AdlHipError("radixSort needs a device (the Python mirror has no host fallback)")
        n = int(n)
        lib = _lib.load()
        tb = ctypes.c_size_t()
        wb = ctypes.c_size_t()
        # full-speed scratch for a sort on these bits (level 1; a sort on fewer bits than the key has needs more)
        check(lib.adlhip_radix_sort_scratch_bytes_for(device._h, kind, n, int(sortBits), 1, ctypes.byref(tb), ctypes.byref(wb)),
              "adlhip_radix_sort_scratch_bytes_for")
        self._scratch(device, tb.value, wb.value)
        check(fn(device._h, inout.ptr(), self.m_tmp.ptr(), self.m_work.ptr(), self.m_work.getSize(), n, int(sortBits)),
              "radixSort")

    def radixSort(self, device, inout, n, sortBits=32):
        """u32 keys (dtype uint32) or {u32 key, u32 value} pairs (dtype uint64, key in the low dword)."""
        lib = _lib.load()
        if inout.dtype == np.uint32:
            self._sort(device, ELEM_U32, lib.adlhip_radix_sort_u32, inout, n, sortBits)
        elif inout.dtype == np.uint64:
            self._sort(device, ELEM_KV32, lib.adlhip_radix_sort_kv32, inout, n, sortBits)
        else:
            raise AdlHipError("radixSort: unsupported element type %s" % inout.dtype)

    def radixSortSoA(self, device, keys, values, n, sortBits=None):
        """Key-value sort on separate key and value buffers (structure of arrays): ascending by key,
        stable, values follow their keys.  Same contract as radixSort on {key, value} pairs.  Keys: uint32 or
        uint64; values: any element type of 4, 8 or 16 bytes (u32 + u32 is the reference kernel's own layout,
        RadixSortKeyValueKernels.cl:354-509)."""
        kb, vb = keys.dtype.itemsize, values.dtype.itemsize
        if keys.dtype not in (np.uint32, np.uint64) or vb not in (4, 8, 16):
            raise AdlHipError("radixSortSoA: keys %s / values %s unsupported" % (keys.dtype, values.dtype))
        if sortBits is None:
            sortBits = 8 * kb
        if device is None:
            raise AdlHipError("radixSortSoA needs a device")
        n = int(n)
        lib = _lib.load()
        tk = ctypes.c_size_t()
        tv = ctypes.c_size_t()
        wb = ctypes.c_size_t()
        check(lib.adlhip_radix_sort_soa_scratch_bytes(device._h, kb, vb, n, int(sortBits), ctypes.byref(tk), ctypes.byref(tv),
                                                      ctypes.byref(wb)), "adlhip_radix_sort_soa_scratch_bytes")
        self._scratch(device, tk.value + tv.value, wb.value)          # tmp keys + tmp values, back to back
        tmp_k = self.m_tmp.ptr()
        tmp_v = ctypes.c_void_p(self.m_tmp.m_ptr + tk.value) if self.m_tmp.m_ptr else None
        check(lib.adlhip_radix_sort_soa(device._h, keys.ptr(), kb, values.ptr(), vb, tmp_k, tmp_v, self.m_work.ptr(),
                                        self.m_work.getSize(), n, int(sortBits)), "radixSortSoA")

    def radixSort64(self, device, inout, n, sortBits=64):
        assert inout.dtype == np.uint64
        self._sort(device, ELEM_U64, _lib.load().adlhip_radix_sort_u64, inout, n, sortBits)

    def copy(self, device, dst, src, n):
        """Pprims::copy (Pprims.cpp:31-67, commented out in the reference): first n elements of src -> dst."""
        if device is None:
            raise AdlHipError("copy needs a device")
        assert dst.dtype == src.dtype and n <= dst.getSize() and n <= src.getSize()
        if n > 0:
            check(_lib.load().adlhip_memcpy_d2d(device._h, dst.ptr(), src.ptr(), int(n) * dst.dtype.itemsize), "copy")

    def fill(self, device, dst, value, n):
        """Pprims::fill (Pprims.cpp:69-120, commented out in the reference): n copies of one element.  `value`
        is anything numpy can turn into ONE element of dst.dtype (4-, 8- or 16-byte element types)."""
        if device is None:
            raise AdlHipError("fill needs a device")
        assert n <= dst.getSize()
        pat = np.array(value, dtype=dst.dtype).reshape(1)
        if pat.dtype.itemsize not in (4, 8, 16):
            raise AdlHipError("fill: element size %d (4, 8 or 16 bytes)" % pat.dtype.itemsize)
        if n > 0:
            check(_lib.load().adlhip_fill_pattern(device._h, dst.ptr(), pat.ctypes.data_as(ctypes.c_void_p),
                                                  pat.dtype.itemsize, int(n)), "fill")

    def scan(self, device, dst, src, n, sumOut=None):
        """Exclusive prefix sum.  sumOut: optional 1-element uint32 numpy array that receives the grand
        total once the caller has synchronised (Pprims.cpp:164-167 reads it back non-blocking too)."""
        if device is None:
            raise AdlHipError("scan needs a device")   # Pprims.cpp:124-127 ADLASSERT(0)
        lib = _lib.load()
        wb = ctypes.c_size_t()
        check(lib.adlhip_scan_scratch_bytes(device._h, int(n), ctypes.byref(wb)), "adlhip_scan_scratch_bytes")
        self._scratch(device, 0, wb.value)
        hp = None
        if sumOut is not None:
            assert sumOut.dtype == np.uint32 and sumOut.size >= 1
            self._sum_keep = sumOut
            hp = sumOut.ctypes.data_as(ctypes.c_void_p)
        check(lib.adlhip_exclusive_scan_u32(device._h, dst.ptr(), src.ptr(), self.m_work.ptr(), self.m_work.getSize(),
                                            int(n), hp), "scan")
