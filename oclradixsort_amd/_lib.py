"""ctypes binding of the C ABI in include/adlhip.h.  No fallback: if the shared library is missing or
the GPU cannot be opened, calls raise."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ADLHIP_LIB: explicit path to another build of the same ABI (e.g. the phase-stamp diagnostic build)
LIB_PATH = os.environ.get("ADLHIP_LIB") or os.path.join(_HERE, "lib", "libadlhip.so")

c_void_pp = ctypes.POINTER(ctypes.c_void_p)
c_size_p = ctypes.POINTER(ctypes.c_size_t)


class AdlHipError(RuntimeError):
    pass


class Info(ctypes.Structure):
    _fields_ = [
        ("compute_units", ctypes.c_int32),
        ("wavefront_size", ctypes.c_int32),
        ("lds_bytes_per_cu", ctypes.c_int32),
        ("clock_khz", ctypes.c_int32),
        ("total_mem_bytes", ctypes.c_uint64),
        ("max_alloc_bytes", ctypes.c_uint64),
        ("name", ctypes.c_char * 128),
        ("arch", ctypes.c_char * 64),
        ("vendor", ctypes.c_char * 32),
    ]


# name -> (restype, argtypes); the authoritative list of exported symbols (tests check it against
# include/adlhip.h)
_VP = ctypes.c_void_p
_SZ = ctypes.c_size_t
_I = ctypes.c_int
SIGNATURES = {
    "adlhip_device_count": (_I, []),
    "adlhip_device_create": (_I, [_I, c_void_pp]),
    "adlhip_device_create_on_stream": (_I, [_I, _VP, c_void_pp]),
    "adlhip_device_destroy": (_I, [_VP]),
    "adlhip_device_info": (_I, [_VP, ctypes.POINTER(Info)]),
    "adlhip_used_bytes": (ctypes.c_uint64, [_VP]),
    "adlhip_sync": (_I, [_VP]),
    "adlhip_fault_check": (_I, [_VP]),
    "adlhip_flush": (_I, [_VP]),
    "adlhip_stream": (_VP, [_VP]),
    "adlhip_last_error": (ctypes.c_char_p, []),
    "adlhip_malloc": (_I, [_VP, _SZ, c_void_pp]),
    "adlhip_free": (_I, [_VP, _VP, _SZ]),
    "adlhip_memcpy_h2d": (_I, [_VP, _VP, _VP, _SZ]),
    "adlhip_memcpy_d2h": (_I, [_VP, _VP, _VP, _SZ]),
    "adlhip_memcpy_d2d": (_I, [_VP, _VP, _VP, _SZ]),
    "adlhip_memset": (_I, [_VP, _VP, _I, _SZ]),
    "adlhip_fill_u32": (_I, [_VP, _VP, ctypes.c_uint32, _SZ]),
    "adlhip_fill_pattern": (_I, [_VP, _VP, _VP, _SZ, _SZ]),
    "adlhip_map": (_I, [_VP, _VP, _SZ, c_void_pp]),
    "adlhip_unmap": (_I, [_VP, _VP, _VP, _SZ]),
    "adlhip_radix_sort_scratch_bytes": (_I, [_VP, _I, _SZ, c_size_p, c_size_p]),
    "adlhip_radix_sort_scratch_bytes_for": (_I, [_VP, _I, _SZ, _I, _I, c_size_p, c_size_p]),
    "adlhip_radix_sort_u32": (_I, [_VP, _VP, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_radix_sort_kv32": (_I, [_VP, _VP, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_radix_sort_soa32": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_radix_sort_u64": (_I, [_VP, _VP, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_radix_sort_soa_scratch_bytes": (_I, [_VP, _I, _I, _SZ, _I, c_size_p, c_size_p, c_size_p]),
    "adlhip_radix_sort_soa": (_I, [_VP, _VP, _I, _VP, _I, _VP, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_segment_sort": (_I, [_VP, _I, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_scan_scratch_bytes": (_I, [_VP, _SZ, c_size_p]),
    "adlhip_exclusive_scan_u32": (_I, [_VP, _VP, _VP, _VP, _SZ, _SZ, _VP]),
    "adlhip_partition_msb_u32": (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_partition_msb_kv32": (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _SZ, _I]),
    "adlhip_partition_top_byte_u32": (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _SZ]),
    "adlhip_partition_top_byte_kv32": (_I, [_VP, _VP, _VP, _VP, _VP, _SZ, _SZ]),
    "adlhip_group_create": (_I, [ctypes.POINTER(_I), _I, c_void_pp]),
    "adlhip_group_destroy": (_I, [_VP]),
    "adlhip_group_size": (_I, [_VP]),
    "adlhip_group_device": (_VP, [_VP, _I]),
    "adlhip_group_last_bounds": (_I, [_VP, ctypes.POINTER(_I)]),
    "adlhip_sharded_sort_u32": (_I, [_VP, c_void_pp, c_size_p, c_void_pp, c_size_p, c_size_p]),
    "adlhip_sharded_sort_kv32": (_I, [_VP, c_void_pp, c_size_p, c_void_pp, c_size_p, c_size_p]),
    "adlhip_generate_keys": (_I, [_VP, _I, _VP, _SZ, ctypes.c_uint64, ctypes.c_uint64]),
    "adlhip_set_param": (_I, [_VP, ctypes.c_char_p, _I]),
    "adlhip_get_param": (_I, [_VP, ctypes.c_char_p, ctypes.POINTER(_I)]),
    "adlhip_event_create": (_I, [_VP, c_void_pp]),
    "adlhip_event_record": (_I, [_VP, _VP]),
    "adlhip_event_elapsed_ms": (_I, [_VP, _VP, _VP, ctypes.POINTER(ctypes.c_float)]),
    "adlhip_event_destroy": (_I, [_VP, _VP]),
    "adlhip_event_synchronize": (_I, [_VP, _VP]),
    "adlhip_event_query": (_I, [_VP, _VP, ctypes.POINTER(_I)]),
    "adlhip_profile_reset": (_I, [_VP]),
    "adlhip_profile_count": (_I, [_VP]),
    "adlhip_profile_get": (_I, [_VP, _I, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64),
                                ctypes.POINTER(ctypes.c_double)]),
    "adlhip_profile_write_csv": (_I, [_VP, ctypes.c_char_p]),
    "adlhip_probe_copy": (_I, [_VP, _VP, _VP, _SZ]),
    "adlhip_probe_read": (_I, [_VP, _VP, _SZ, _VP]),
    "adlhip_probe_copy_ex": (_I, [_VP, _VP, _VP, _SZ, _I, _I]),
    "adlhip_probe_read_ex": (_I, [_VP, _VP, _SZ, _VP, _I, _I]),
    "adlhip_selftest_lds_order": (_I, [_VP, _I, ctypes.POINTER(ctypes.c_uint32)]),
    "adlhip_selftest_probe_positions": (_I, [_VP, _SZ, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "adlhip_version": (ctypes.c_char_p, []),
}

_lib = None


def load():
    """Load libadlhip.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AdlHipError(
                "HIP back-end %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback for the device path)" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().adlhip_last_error()
        raise AdlHipError("%s failed: %s" % (what or "adlhip call", msg.decode() if msg else "?"))
