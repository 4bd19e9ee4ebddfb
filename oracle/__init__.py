"""oracle -- ctypes access to the CPU checkers.  TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (oclradixsort_amd) never imports this module.

Two libraries:
  liboracle.so      C restatement of the reference CPU sort (oracle/radixsort_oracle.c)
  _ref/libref.so    the reference's own Tahoe::RadixSort::sort / Pprims host path, compiled
                    from /root/reference by oracle/Makefile (present where it was built and on
                    the GPU box via the snapshot; absent from git history)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libref.so")

_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(quiet=True):
    """Compile liboracle.so (and _ref/libref.so when /root/reference is present)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + out.stdout + out.stderr)
    if not quiet:
        print(out.stdout)


def _load_oracle():
    if not os.path.exists(_ORACLE_SO) or (
        os.path.getmtime(_ORACLE_SO) < os.path.getmtime(os.path.join(_HERE, "radixsort_oracle.c"))
    ):
        build()
    lib = ctypes.CDLL(_ORACLE_SO)
    lib.oracle_radix_sort_u32.argtypes = [_u32p, ctypes.c_size_t]
    lib.oracle_radix_sort_u32.restype = ctypes.c_int
    lib.oracle_radix_sort_kv32.argtypes = [_u64p, ctypes.c_size_t]
    lib.oracle_radix_sort_kv32.restype = ctypes.c_int
    lib.oracle_radix_sort_u64.argtypes = [_u64p, ctypes.c_size_t]
    lib.oracle_radix_sort_u64.restype = ctypes.c_int
    lib.oracle_radix_sort_u32_bits.argtypes = [_u32p, ctypes.c_size_t, ctypes.c_int]
    lib.oracle_radix_sort_u32_bits.restype = ctypes.c_int
    lib.oracle_radix_sort_e64_bits.argtypes = [_u64p, ctypes.c_size_t, ctypes.c_int]
    lib.oracle_radix_sort_e64_bits.restype = ctypes.c_int
    lib.oracle_radix_sort_soa.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int]
    lib.oracle_radix_sort_soa.restype = ctypes.c_int
    lib.oracle_exclusive_scan_u32.argtypes = [_u32p, _u32p, ctypes.c_size_t]
    lib.oracle_exclusive_scan_u32.restype = ctypes.c_uint32
    lib.oracle_fnv1a64.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.oracle_fnv1a64.restype = ctypes.c_uint64
    for name, ptr in (("oracle_fill_keys_u32", _u32p), ("oracle_fill_keys_u64", _u64p),
                      ("oracle_fill_pairs_kv32", _u64p)):
        fn = getattr(lib, name)
        fn.argtypes = [ptr, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint64]
        fn.restype = None
    for name, ptr in (("oracle_demo_fill_u32", _u32p), ("oracle_demo_fill_kv32", _u64p),
                      ("oracle_demo_fill_scan", _u32p)):
        fn = getattr(lib, name)
        fn.argtypes = [ptr, ctypes.c_size_t, ctypes.c_uint]
        fn.restype = None
    return lib


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load_oracle()
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    """The real reference (oracle/_ref/libref.so).  Raises if it has not been built."""
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libref.so missing: run `make -C oracle` where /root/reference exists")
        r = ctypes.CDLL(_REF_SO)
        for name, ptr in (("ref_radix_sort_u32", _u32p), ("ref_radix_sort_kv32", _u64p),
                          ("ref_host_path_sort_u32", _u32p), ("ref_host_path_sort_kv32", _u64p)):
            fn = getattr(r, name)
            fn.argtypes = [ptr, ctypes.c_int]
            fn.restype = ctypes.c_int
        _ref = r
    return _ref


def _p32(a):
    return a.ctypes.data_as(_u32p)


def _p64(a):
    return a.ctypes.data_as(_u64p)


def _own(a, dtype):
    a = np.array(a, dtype=dtype, copy=True, order="C")
    return a


# ---- restatement ------------------------------------------------------------------------------
def sort_u32(keys):
    a = _own(keys, np.uint32)
    assert lib().oracle_radix_sort_u32(_p32(a), a.size) == 0
    return a


def sort_kv32(pairs):
    """pairs: uint64 array, low dword = key, high dword = value (little-endian {key,value})."""
    a = _own(pairs, np.uint64)
    assert lib().oracle_radix_sort_kv32(_p64(a), a.size) == 0
    return a


def sort_u64(keys):
    a = _own(keys, np.uint64)
    assert lib().oracle_radix_sort_u64(_p64(a), a.size) == 0
    return a


def sort_u32_bits(keys, sort_bits):
    a = _own(keys, np.uint32)
    assert lib().oracle_radix_sort_u32_bits(_p32(a), a.size, sort_bits) == 0
    return a


def sort_e64_bits(elems, sort_bits):
    a = _own(elems, np.uint64)
    assert lib().oracle_radix_sort_e64_bits(_p64(a), a.size, sort_bits) == 0
    return a


def sort_soa(keys, values, sort_bits=None):
    """Stable key-value sort on separate arrays: keys uint32 / uint64, values of any fixed-width dtype.  Returns
    (sorted keys, values in their keys' order)."""
    k = np.array(keys, copy=True, order="C")
    v = np.array(values, copy=True, order="C")
    assert k.dtype in (np.uint32, np.uint64) and k.ndim == 1 and v.shape[0] == k.shape[0]
    vb = v.dtype.itemsize * int(np.prod(v.shape[1:], dtype=np.int64))
    bits = 8 * k.dtype.itemsize if sort_bits is None else int(sort_bits)
    assert lib().oracle_radix_sort_soa(k.ctypes.data_as(ctypes.c_void_p), k.dtype.itemsize, v.ctypes.data_as(ctypes.c_void_p), vb,
                                       k.shape[0], bits) == 0
    return k, v


def exclusive_scan_u32(src):
    s = _own(src, np.uint32)
    d = np.empty_like(s)
    total = lib().oracle_exclusive_scan_u32(_p32(d), _p32(s), s.size)
    return d, int(total)


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().oracle_fnv1a64(a.ctypes.data_as(ctypes.c_void_p), a.nbytes))


def keys_u32(n, seed=123, first_index=0):
    a = np.empty(n, dtype=np.uint32)
    lib().oracle_fill_keys_u32(_p32(a), n, seed, first_index)
    return a


def keys_u64(n, seed=123, first_index=0):
    a = np.empty(n, dtype=np.uint64)
    lib().oracle_fill_keys_u64(_p64(a), n, seed, first_index)
    return a


def pairs_kv32(n, seed=123, first_index=0):
    a = np.empty(n, dtype=np.uint64)
    lib().oracle_fill_pairs_kv32(_p64(a), n, seed, first_index)
    return a


def demo_u32(n, seed=123):
    a = np.empty(n, dtype=np.uint32)
    lib().oracle_demo_fill_u32(_p32(a), n, seed)
    return a


def demo_kv32(n, seed=123):
    a = np.empty(n, dtype=np.uint64)
    lib().oracle_demo_fill_kv32(_p64(a), n, seed)
    return a


def demo_scan(n, seed=123):
    a = np.empty(n, dtype=np.uint32)
    lib().oracle_demo_fill_scan(_p32(a), n, seed)
    return a


# ---- the real reference -----------------------------------------------------------------------
def ref_sort_u32(keys, host_path=False):
    a = _own(keys, np.uint32)
    fn = ref().ref_host_path_sort_u32 if host_path else ref().ref_radix_sort_u32
    assert fn(_p32(a), a.size) == 0
    return a


def ref_sort_kv32(pairs, host_path=False):
    a = _own(pairs, np.uint64)
    fn = ref().ref_host_path_sort_kv32 if host_path else ref().ref_radix_sort_kv32
    assert fn(_p64(a), a.size) == 0
    return a
