"""Python mirror of the reference's `adl::` device abstraction for the HIP back-end.

Same names and argument meaning as Adl/Adl.h (reference), so that the parity tests read like
UnitTest/main.cpp:
    DeviceUtils.allocate / deallocate / waitForCompletion / getNCUs / getNDevices   (Adl.h:71-116)
    Device.getUsedMemory / getDeviceName / getType / getProcType                    (Adl.h:123-155)
    Buffer(device, nElems, dtype): write / read / getHostPtr / returnHostPtr / setSize / getSize
                                                                                    (Adl.h:164-222)
Only the device path exists here (TYPE_CL is served by the HIP back-end; there is no OpenCL).
Everything is thin plumbing over the C ABI (include/adlhip.h); failures raise AdlHipError where the
reference would ADLASSERT.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import AdlHipError, check

TYPE_CL = 0     # Adl.h:41  -- served by the HIP back-end
TYPE_DX11 = 1   # Adl.h:42  -- not available
TYPE_HOST = 2   # Adl.h:43  -- C++ facade only (include/Adl); not mirrored in Python
TYPE_HIP = TYPE_CL


class Config:
    """DeviceUtils::Config (Adl.h:74-96)."""
    DEVICE_GPU = 0
    DEVICE_CPU = 1

    def __init__(self, deviceIdx=0):
        self.m_type = Config.DEVICE_GPU
        self.m_deviceIdx = deviceIdx


class Device:
    """adl::Device for TYPE_CL (Adl.h:123-155) backed by an adlhip_device handle."""

    def __init__(self, handle, cfg):
        self._h = ctypes.c_void_p(handle)
        self.m_type = TYPE_CL
        self.m_procType = cfg.m_type
        info = _lib.Info()
        check(_lib.load().adlhip_device_info(self._h, ctypes.byref(info)), "adlhip_device_info")
        self.info = info

    def getType(self):
        return self.m_type

    def getProcType(self):
        return self.m_procType

    def getUsedMemory(self):
        return int(_lib.load().adlhip_used_bytes(self._h))

    def getDeviceName(self):
        return self.info.name.decode()

    def getDeviceVendor(self):
        return self.info.vendor.decode()

    def toggleProfiling(self, enable):
        """Device::toggleProfiling (Adl.h:142): per-launch timing of every kernel."""
        check(_lib.load().adlhip_set_param(self._h, b"profile", 1 if enable else 0), "set profile")

    def setParam(self, name, value):
        check(_lib.load().adlhip_set_param(self._h, name.encode(), int(value)), "adlhip_set_param")

    def getParam(self, name):
        v = ctypes.c_int(0)
        check(_lib.load().adlhip_get_param(self._h, name.encode(), ctypes.byref(v)), "adlhip_get_param")
        return v.value

    def profile(self, reset=False):
        """{kernel name: (launches, total_ms)} collected while profiling was on."""
        lib = _lib.load()
        n = lib.adlhip_profile_count(self._h)
        if n < 0:
            check(1, "adlhip_profile_count")
        out = {}
        for i in range(n):
            name = ctypes.create_string_buffer(64)
            cnt = ctypes.c_uint64(0)
            ms = ctypes.c_double(0)
            check(lib.adlhip_profile_get(self._h, i, name, ctypes.byref(cnt), ctypes.byref(ms)), "profile_get")
            out[name.value.decode()] = (int(cnt.value), float(ms.value))
        if reset:
            check(lib.adlhip_profile_reset(self._h), "profile_reset")
        return out

    def writeProfileCsv(self, path):
        """Append the per-kernel timing table as CSV (the reference's ProfileCL.*.csv hook)."""
        check(_lib.load().adlhip_profile_write_csv(self._h, path.encode()), "adlhip_profile_write_csv")

    def checkFault(self):
        """Non-blocking, stream-ordered fault check (adlhip_fault_check) for handles whose stream is synchronised by
        someone else (torch): raises for a device-side fault of an EARLIER, completed batch."""
        check(_lib.load().adlhip_fault_check(self._h), "adlhip_fault_check")

    @property
    def stream(self):
        return _lib.load().adlhip_stream(self._h)


class DeviceUtils:
    Config = Config

    @staticmethod
    def getNDevices(deviceType=TYPE_CL):
        return int(_lib.load().adlhip_device_count())

    @staticmethod
    def allocate(deviceType=TYPE_CL, cfg=None, stream=None):
        """DeviceUtils::allocate (Adl.inl:73-98).  `stream`: optional raw hipStream_t to enqueue on."""
        if deviceType != TYPE_CL:
            raise AdlHipError("only TYPE_CL (HIP back-end) devices exist in the Python mirror")
        cfg = cfg or Config()
        h = ctypes.c_void_p()
        lib = _lib.load()
        if stream is None:
            check(lib.adlhip_device_create(cfg.m_deviceIdx, ctypes.byref(h)), "adlhip_device_create")
        else:
            check(lib.adlhip_device_create_on_stream(cfg.m_deviceIdx, ctypes.c_void_p(stream), ctypes.byref(h)),
                  "adlhip_device_create_on_stream")
        return Device(h.value, cfg)

    @staticmethod
    def deallocate(device):
        """DeviceUtils::deallocate (Adl.inl:100-105): refuses while buffers are alive."""
        check(_lib.load().adlhip_device_destroy(device._h), "adlhip_device_destroy")
        device._h = None

    @staticmethod
    def waitForCompletion(device):
        try:
            check(_lib.load().adlhip_sync(device._h), "adlhip_sync")
        finally:
            device.__dict__.pop("_staged", None)     # host arrays staged by Buffer.write are no longer needed

    @staticmethod
    def flush(device):
        check(_lib.load().adlhip_flush(device._h), "adlhip_flush")

    @staticmethod
    def getNCUs(device):
        return int(device.info.compute_units)


class Buffer:
    """adl::Buffer<T> (Adl.h:164-222).  dtype: np.uint32 (u32 / int), np.uint64 (uint2 pairs, u64 keys)."""

    def __init__(self, device=None, nElems=0, dtype=np.uint32):
        self.m_device = None
        self.m_size = 0
        self.m_ptr = None
        self.m_allocated = False
        self.dtype = np.dtype(dtype)
        self._maps = {}
        if device is not None:
            self.allocate(device, nElems)

    # -- allocation (Adl.inl:216-271)
    def allocate(self, device, nElems):
        assert self.m_ptr is None, "Buffer already allocated"
        self.m_device = device
        p = ctypes.c_void_p()
        check(_lib.load().adlhip_malloc(device._h, int(nElems) * self.dtype.itemsize, ctypes.byref(p)), "adlhip_malloc")
        self.m_ptr = p.value
        self.m_size = int(nElems)
        self.m_allocated = True

    def release(self):
        if self.m_allocated and self.m_ptr is not None and self.m_device is not None and self.m_device._h is not None:
            check(_lib.load().adlhip_free(self.m_device._h, ctypes.c_void_p(self.m_ptr),
                                          self.m_size * self.dtype.itemsize), "adlhip_free")
        self.m_ptr = None
        self.m_size = 0
        self.m_allocated = False

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def setRawPtr(self, device, ptr, size):
        """Buffer::setRawPtr (Adl.inl:238-253): wrap foreign device memory without owning it."""
        self.m_device = device
        self.m_ptr = int(ptr)
        self.m_size = int(size)
        self.m_allocated = False

    def setSize(self, size):
        """Buffer::setSize (Adl.inl:331-356): grow-only, contents not preserved."""
        if size > self.m_size:
            dev = self.m_device
            DeviceUtils.waitForCompletion(dev)
            self.release()
            self.allocate(dev, size)

    def getSize(self):
        return self.m_size

    def ptr(self, offsetNElems=0):
        return ctypes.c_void_p((self.m_ptr or 0) + int(offsetNElems) * self.dtype.itemsize)

    # -- copies (Adl.inl:273-303): asynchronous, caller syncs
    def write(self, hostSrc, nElems=None, dstOffsetNElems=0):
        if isinstance(hostSrc, Buffer):
            n = nElems if nElems is not None else hostSrc.m_size
            check(_lib.load().adlhip_memcpy_d2d(self.m_device._h, self.ptr(), hostSrc.ptr(), n * self.dtype.itemsize), "d2d")
            return
        a = np.ascontiguousarray(hostSrc, dtype=self.dtype)
        n = a.size if nElems is None else int(nElems)
        assert dstOffsetNElems + n <= self.m_size
        # keep every staged host array alive until the caller syncs (a second write before the sync must not drop
        # the first one's source while its copy may still be pending)
        keep = self.m_device.__dict__.setdefault("_staged", [])
        keep.append(a)
        check(_lib.load().adlhip_memcpy_h2d(self.m_device._h, self.ptr(dstOffsetNElems),
                                            a.ctypes.data_as(ctypes.c_void_p), n * self.dtype.itemsize), "h2d")

    def read(self, hostDst, nElems=None, srcOffsetNElems=0):
        assert hostDst.dtype == self.dtype and hostDst.flags["C_CONTIGUOUS"]
        n = hostDst.size if nElems is None else int(nElems)
        assert srcOffsetNElems + n <= self.m_size
        check(_lib.load().adlhip_memcpy_d2h(self.m_device._h, hostDst.ctypes.data_as(ctypes.c_void_p),
                                            self.ptr(srcOffsetNElems), n * self.dtype.itemsize), "d2h")

    def toHost(self, nElems=None):
        """read() + waitForCompletion: convenience for tests."""
        n = self.m_size if nElems is None else int(nElems)
        out = np.empty(n, dtype=self.dtype)
        if n:
            self.read(out, n)
        DeviceUtils.waitForCompletion(self.m_device)
        return out

    # -- map / unmap (Adl.inl:317-329)
    def getHostPtr(self, size=None):
        n = self.m_size if size is None else int(size)
        h = ctypes.c_void_p()
        nbytes = n * self.dtype.itemsize
        check(_lib.load().adlhip_map(self.m_device._h, self.ptr(), nbytes, ctypes.byref(h)), "adlhip_map")
        if n == 0:
            return np.empty(0, dtype=self.dtype)
        arr = np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(h.value)).view(self.dtype)
        self._maps[arr.ctypes.data] = (h.value, nbytes)
        return arr

    def returnHostPtr(self, arr):
        if arr.size == 0:
            return
        h, nbytes = self._maps.pop(arr.ctypes.data)
        check(_lib.load().adlhip_unmap(self.m_device._h, self.ptr(), ctypes.c_void_p(h), nbytes), "adlhip_unmap")

    def generate(self, nElems=None, seed=123, firstIndex=0, kind=None):
        """Fill with the index-reproducible synthetic keys (adlhip_generate_keys).  kind defaults to
        u32 keys for uint32 buffers and 64-bit keys for uint64 buffers; pass 1 for {key,index} pairs."""
        n = self.m_size if nElems is None else int(nElems)
        if kind is None:
            kind = 0 if self.dtype == np.uint32 else 2
        check(_lib.load().adlhip_generate_keys(self.m_device._h, int(kind), self.ptr(), n, int(seed), int(firstIndex)),
              "adlhip_generate_keys")

    def clear(self):
        check(_lib.load().adlhip_memset(self.m_device._h, self.ptr(), 0, self.m_size * self.dtype.itemsize), "memset")


class Stopwatch:
    """adl::Stopwatch (AdlStopwatch.h:60-83) on device time: start / split / stop / getMs."""
    CAPACITY = 64

    def __init__(self, device):
        self.m_device = device
        self._ev = []

    def _record(self):
        e = ctypes.c_void_p()
        lib = _lib.load()
        check(lib.adlhip_event_create(self.m_device._h, ctypes.byref(e)), "event_create")
        check(lib.adlhip_event_record(self.m_device._h, e), "event_record")
        self._ev.append(e)

    def start(self):
        self._free()
        self._record()

    def split(self):
        assert len(self._ev) < Stopwatch.CAPACITY
        self._record()

    def stop(self):
        self._record()

    def getMs(self, index=0):
        ms = ctypes.c_float(0)
        check(_lib.load().adlhip_event_elapsed_ms(self.m_device._h, self._ev[index], self._ev[index + 1],
                                                  ctypes.byref(ms)), "event_elapsed")
        return ms.value

    def _free(self):
        for e in self._ev:
            _lib.load().adlhip_event_destroy(self.m_device._h, e)
        self._ev = []

    def __del__(self):
        try:
            if self.m_device._h is not None:
                self._free()
        except Exception:
            pass
