// Adl/Adl.h -- the reference's `adl::` device abstraction, re-created over the MI355X HIP back-end.
//
// Call surface kept from the reference (Adl/Adl.h:39-274, Adl/Adl.inl, Adl/AdlStopwatch.h) so that code
// written against it -- Tahoe::Pprims and UnitTest/main.cpp -- compiles unchanged:
//   enum DeviceType { TYPE_CL, TYPE_DX11, TYPE_HOST }
//   DeviceUtils::{Config, getNDevices, getNCUs, allocate, deallocate, waitForCompletion, flush}
//   Device::{getType, getProcType, getUsedMemory, getDeviceName, getDeviceVendor, toggleProfiling, ...}
//   Buffer<T>::{Buffer, allocate, setRawPtr, write, read, clear, fill, getHostPtr, returnHostPtr, setSize,
//               getSize, getType} and the public members m_device / m_size / m_ptr / m_allocated
//   HostBuffer<T>, Stopwatch
// Mechanism is new: the reference dispatches on m_type with macros into OpenCL / host .inl files
// (Adl/Adl.inl:131-199); here Device is an interface with byte-level hooks and two implementations:
//   TYPE_CL   -> DeviceHip : the C ABI in include/adlhip.h (HIP on gfx950; there is no OpenCL here)
//   TYPE_HOST -> DeviceHost: plain new[]/memcpy, as Adl/Host/AdlHost.inl:8-72
// TYPE_DX11 does not exist (disabled in the reference too, Adl/AdlConfig.h:6).  The generic
// Launcher / KernelManager / KernelBuilder of Adl/AdlKernel.h are intentionally not re-created: kernels
// are compiled ahead of time into libadlhip.so and exposed as whole primitives.
// Failures surface as ADLASSERT (Tahoe/Math/Error.h), like the reference.
#ifndef ADL_H
#define ADL_H

#include <adlhip.h>
#include <limits.h>
#include <stdio.h>
#include <string.h>
#include <chrono>

#include <Tahoe/Math/Error.h>

// Run a C-ABI call and assert on its status.  The call sits OUTSIDE the assertion expression, so it runs whatever ADLASSERT
// is defined to (the reference's release build defines it to nothing, Tahoe/Math/Error.h:24-38).
#define ADLHIP_CALL(expr)                \
    do {                                 \
        const int adlhip_rc_ = (expr);   \
        ADLASSERT(adlhip_rc_ == 0);      \
    } while (0)

namespace adl {

typedef unsigned long long u64;

extern char s_cacheDirectory[128];   // defined by the application (UnitTest/main.cpp:74); unused: kernels are AOT

#define ADL_SUCCESS 0
#define ADL_FAILURE 1

template <typename T> inline T max2(const T& a, const T& b) { return (a > b) ? a : b; }
template <typename T> inline T min2(const T& a, const T& b) { return (a < b) ? a : b; }

enum DeviceType {
    TYPE_CL = 0,    // served by the HIP back-end
    TYPE_DX11 = 1,  // not available
    TYPE_HOST,
};

struct Device;
struct SyncObject;   // an event of the device's queue (AdlKernel.h:45-54); defined below Device

struct BufferBase {
    enum BufferType {
        BUFFER, BUFFER_CONST, BUFFER_STAGING, BUFFER_APPEND, BUFFER_RAW, BUFFER_W_COUNTER, BUFFER_INDEX,
        BUFFER_VERTEX, BUFFER_ZERO_COPY,
    };
};

class DeviceUtils {
public:
    struct Config {
        enum DeviceType { DEVICE_GPU, DEVICE_CPU };
        enum DeviceVendor { VD_AMD, VD_INTEL, VD_NV };
        Config() : m_type(DEVICE_GPU), m_deviceIdx(0), m_vendor(VD_AMD), m_clContextProperties(0) {}
        DeviceType m_type;
        int m_deviceIdx;
        DeviceVendor m_vendor;
        void* m_clContextProperties;
    };

    static inline int getNDevices(DeviceType type);
    static inline int getNCUs(const Device* device);
    static inline Device* allocate(DeviceType type, Config cfg = Config());
    static inline void deallocate(Device* device);
    static inline void waitForCompletion(const Device* device);
    static inline void waitForCompletion(const SyncObject* syncObj);   // AdlCL.inl:572-587
    static inline bool isComplete(const SyncObject* syncObj);         // AdlCL.inl:589-612
    static inline void flush(const Device* device);
};

// ---------------------------------------------------------------------------------------------
// Device: the reference's public queries + byte-level hooks the templated Buffer<T> forwards to.
// ---------------------------------------------------------------------------------------------
struct Device {
    typedef DeviceUtils::Config Config;

    explicit Device(DeviceType type)
        : m_type(type), m_procType(Config::DEVICE_GPU), m_memoryUsage(0), m_interopAvailable(false),
          m_enableProfiling(false), m_binaryFileVersion(0) {}
    virtual ~Device() {}

    virtual void* getContext() const { return 0; }
    virtual void initialize(const Config& cfg) = 0;
    virtual void release() = 0;
    virtual void waitForCompletion() const = 0;
    virtual void flush() const {}
    virtual void getDeviceName(char nameOut[128]) const { nameOut[0] = 0; }
    virtual void getDeviceVendor(char nameOut[128]) const { nameOut[0] = 0; }
    virtual int getNCUs() const { return 1; }
    virtual u64 getUsedMemory() const { return m_memoryUsage; }
    virtual u64 getMaxAllocationSize() const { return ULLONG_MAX; }
    virtual void toggleProfiling(bool enable) { m_enableProfiling = enable; }
    void setBinaryFileVersion(unsigned int ver) { m_binaryFileVersion = ver; }
    unsigned int getBinaryFileVersion() const { return m_binaryFileVersion; }
    DeviceType getType() const { return m_type; }
    Config::DeviceType getProcType() const { return m_procType; }

    // byte-level hooks
    virtual void* allocBytes(u64 bytes) = 0;
    virtual void freeBytes(void* p, u64 bytes) = 0;
    virtual void copyH2D(void* dst, const void* src, u64 bytes) const = 0;
    virtual void copyD2H(void* dst, const void* src, u64 bytes) const = 0;
    virtual void copyD2D(void* dst, const void* src, u64 bytes) const = 0;
    virtual void clearBytes(void* p, u64 bytes) const = 0;
    virtual void fillU32(void* p, unsigned int pattern, u64 count) const = 0;
    virtual void fillPattern(void* p, const void* pattern, int patternBytes, u64 count) const = 0;   // 4, 8 or 16 bytes
    virtual void* mapBytes(void* p, u64 bytes) const = 0;
    virtual void unmapBytes(void* p, void* host, u64 bytes) const = 0;
    // events behind SyncObject (AdlCL.inl:98-118, 572-612): TYPE_HOST copies are synchronous, its events are always complete
    virtual void* allocSyncObj() const { return 0; }
    virtual void freeSyncObj(void*) const {}
    virtual void recordSyncObj(void*) const {}
    virtual void waitSyncObj(void*) const {}
    virtual bool isSyncObjComplete(void*) const { return true; }
    // the C-ABI handle behind a TYPE_CL device (0 for TYPE_HOST): what Pprims hands to adlhip_*
    virtual adlhip_device* hip() const { return 0; }

    DeviceType m_type;
    Config::DeviceType m_procType;
    u64 m_memoryUsage;
    bool m_interopAvailable;
    bool m_enableProfiling;
    unsigned int m_binaryFileVersion;
};

// TYPE_CL: the HIP back-end through the C ABI (replaces DeviceCL, Adl/CL/AdlCL.inl:23-143)
struct DeviceHip : public Device {
    DeviceHip() : Device(TYPE_CL), m_hip(0), m_ownsHip(true) {}
    // wrap a handle that someone else owns (a rank of an adlhip_group, Tahoe::ShardedSort): release() leaves it alone
    void attach(adlhip_device* hip)
    {
        m_procType = Config::DEVICE_GPU;
        m_hip = hip;
        m_ownsHip = false;
    }
    void initialize(const Config& cfg)
    {
        m_procType = cfg.m_type;
        const int rc = adlhip_device_create(cfg.m_deviceIdx, &m_hip);
        if (rc != ADLHIP_SUCCESS) fprintf(stderr, "adl: cannot open the HIP device: %s\n", adlhip_last_error());
        ADLASSERT(rc == ADLHIP_SUCCESS);
    }
    void release()
    {
        if (m_hip && m_ownsHip) ADLHIP_CALL(adlhip_device_destroy(m_hip));
        m_hip = 0;
    }
    void waitForCompletion() const
    {
        const int rc = adlhip_sync(m_hip);
        if (rc != ADLHIP_SUCCESS) fprintf(stderr, "adl: %s\n", adlhip_last_error());
        ADLASSERT(rc == ADLHIP_SUCCESS);
    }
    void flush() const { adlhip_flush(m_hip); }
    void getDeviceName(char nameOut[128]) const
    {
        adlhip_info i;
        nameOut[0] = 0;
        if (adlhip_device_info(m_hip, &i) == ADLHIP_SUCCESS) snprintf(nameOut, 128, "%.60s (%.60s)", i.name, i.arch);
    }
    void getDeviceVendor(char nameOut[128]) const
    {
        adlhip_info i;
        nameOut[0] = 0;
        if (adlhip_device_info(m_hip, &i) == ADLHIP_SUCCESS) snprintf(nameOut, 128, "%s", i.vendor);
    }
    int getNCUs() const
    {
        adlhip_info i;
        return adlhip_device_info(m_hip, &i) == ADLHIP_SUCCESS ? i.compute_units : 0;
    }
    u64 getUsedMemory() const { return adlhip_used_bytes(m_hip); }
    void toggleProfiling(bool enable)
    {
        m_enableProfiling = enable;
        adlhip_set_param(m_hip, "profile", enable ? 1 : 0);
    }
    void* allocBytes(u64 bytes)
    {
        void* p = 0;
        if (adlhip_malloc(m_hip, (size_t)bytes, &p) != ADLHIP_SUCCESS) {   // AdlCL.inl:390-406: log, leave ptr 0
            fprintf(stderr, "adl: %s\n", adlhip_last_error());
            return 0;
        }
        return p;
    }
    void freeBytes(void* p, u64 bytes) { ADLHIP_CALL(adlhip_free(m_hip, p, (size_t)bytes)); }
    void copyH2D(void* dst, const void* src, u64 bytes) const { ADLHIP_CALL(adlhip_memcpy_h2d(m_hip, dst, src, (size_t)bytes)); }
    void copyD2H(void* dst, const void* src, u64 bytes) const { ADLHIP_CALL(adlhip_memcpy_d2h(m_hip, dst, src, (size_t)bytes)); }
    void copyD2D(void* dst, const void* src, u64 bytes) const { ADLHIP_CALL(adlhip_memcpy_d2d(m_hip, dst, src, (size_t)bytes)); }
    void clearBytes(void* p, u64 bytes) const { ADLHIP_CALL(adlhip_memset(m_hip, p, 0, (size_t)bytes)); }
    void fillU32(void* p, unsigned int pattern, u64 count) const { ADLHIP_CALL(adlhip_fill_u32(m_hip, p, pattern, (size_t)count)); }
    void fillPattern(void* p, const void* pattern, int patternBytes, u64 count) const
    {
        ADLHIP_CALL(adlhip_fill_pattern(m_hip, p, pattern, (size_t)patternBytes, (size_t)count));
    }
    void* mapBytes(void* p, u64 bytes) const
    {
        void* h = 0;
        ADLHIP_CALL(adlhip_map(m_hip, p, (size_t)bytes, &h));
        return h;
    }
    void unmapBytes(void* p, void* host, u64 bytes) const { ADLHIP_CALL(adlhip_unmap(m_hip, p, host, (size_t)bytes)); }
    void* allocSyncObj() const
    {
        adlhip_event* e = 0;
        ADLHIP_CALL(adlhip_event_create(m_hip, &e));
        return e;
    }
    void freeSyncObj(void* e) const { if (e) ADLHIP_CALL(adlhip_event_destroy(m_hip, (adlhip_event*)e)); }
    void recordSyncObj(void* e) const { if (e) ADLHIP_CALL(adlhip_event_record(m_hip, (adlhip_event*)e)); }
    void waitSyncObj(void* e) const { if (e) ADLHIP_CALL(adlhip_event_synchronize(m_hip, (adlhip_event*)e)); }
    bool isSyncObjComplete(void* e) const
    {
        int done = 1;
        if (e) ADLHIP_CALL(adlhip_event_query(m_hip, (adlhip_event*)e, &done));
        return done != 0;
    }
    adlhip_device* hip() const { return m_hip; }

    adlhip_device* m_hip;
    bool m_ownsHip;
};

// TYPE_HOST: the CPU "device" (Adl/Host/AdlHost.inl:8-72): plain memory, synchronous copies, identity map
struct DeviceHost : public Device {
    DeviceHost() : Device(TYPE_HOST) {}
    void initialize(const Config&) { m_procType = Config::DEVICE_CPU; }
    void release() {}
    void waitForCompletion() const {}
    void getDeviceName(char nameOut[128]) const { snprintf(nameOut, 128, "Host"); }
    void* allocBytes(u64 bytes)
    {
        m_memoryUsage += bytes;
        return bytes ? (void*)new char[bytes] : 0;
    }
    void freeBytes(void* p, u64 bytes)
    {
        delete[] (char*)p;
        m_memoryUsage -= bytes;
    }
    void copyH2D(void* dst, const void* src, u64 bytes) const { memcpy(dst, src, bytes); }
    void copyD2H(void* dst, const void* src, u64 bytes) const { memcpy(dst, src, bytes); }
    void copyD2D(void* dst, const void* src, u64 bytes) const { memcpy(dst, src, bytes); }
    void clearBytes(void* p, u64 bytes) const { memset(p, 0, bytes); }
    void fillU32(void* p, unsigned int pattern, u64 count) const
    {
        for (u64 i = 0; i < count; ++i) ((unsigned int*)p)[i] = pattern;
    }
    void fillPattern(void* p, const void* pattern, int patternBytes, u64 count) const
    {
        for (u64 i = 0; i < count; ++i) memcpy((char*)p + i * patternBytes, pattern, patternBytes);
    }
    void* mapBytes(void* p, u64) const { return p; }        // AdlHost.inl:45-47: getHostPtr is m_ptr
    void unmapBytes(void*, void*, u64) const {}
};

// ---------------------------------------------------------------------------------------------
// DeviceUtils
// ---------------------------------------------------------------------------------------------
int DeviceUtils::getNDevices(DeviceType type)
{
    if (type == TYPE_CL) return adlhip_device_count();
    return type == TYPE_HOST ? 1 : 0;
}

int DeviceUtils::getNCUs(const Device* device)
{
    ADLASSERT(device != 0);
    return device ? device->getNCUs() : 0;
}

Device* DeviceUtils::allocate(DeviceType type, Config cfg)
{
    Device* d = 0;
    switch (type) {
    case TYPE_CL: d = new DeviceHip(); break;
    case TYPE_HOST: d = new DeviceHost(); break;
    default: ADLASSERT(0); return 0;   // TYPE_DX11 is not available
    }
    d->initialize(cfg);
    return d;
}

void DeviceUtils::deallocate(Device* device)
{
    ADLASSERT(device->getUsedMemory() == 0);   // Adl/Adl.inl:102
    device->release();
    delete device;
}

void DeviceUtils::waitForCompletion(const Device* device) { device->waitForCompletion(); }
void DeviceUtils::flush(const Device* device) { device->flush(); }

// SyncObject (AdlKernel.h:45-54, AdlKernel.inl:228-239): an event of the device's queue.  A copy that is handed one records it
// behind itself; waitForCompletion(syncObj) waits for that point of the queue only, isComplete(syncObj) polls it.  (The reference
// also hands one to Launcher::launch1D/2D; the Launcher is out of scope here -- kernels are launched by the back-end.)
struct SyncObject {
    explicit SyncObject(const Device* device) : m_device(device), m_ptr(device->allocSyncObj()) {}
    ~SyncObject() { m_device->freeSyncObj(m_ptr); }
    const Device* m_device;
    void* m_ptr;

private:
    SyncObject(const SyncObject&);
    SyncObject& operator=(const SyncObject&);
};
void DeviceUtils::waitForCompletion(const SyncObject* syncObj) { if (syncObj) syncObj->m_device->waitSyncObj(syncObj->m_ptr); }
bool DeviceUtils::isComplete(const SyncObject* syncObj) { return syncObj ? syncObj->m_device->isSyncObjComplete(syncObj->m_ptr) : true; }

// ---------------------------------------------------------------------------------------------
// Buffer<T>
// ---------------------------------------------------------------------------------------------
template <typename T>
struct Buffer : public BufferBase {
    Buffer() : m_device(0), m_size(0), m_ptr(0), m_allocated(false) {}
    Buffer(const Device* device, u64 nElems, BufferType type = BUFFER) : m_device(0), m_size(0), m_ptr(0), m_allocated(false)
    {
        allocate(device, nElems, type);
    }
    virtual ~Buffer()
    {
        if (m_allocated && m_ptr && m_device) const_cast<Device*>(m_device)->freeBytes(m_ptr, m_size * sizeof(T));
        m_ptr = 0;
        m_size = 0;
    }

    // wrap foreign memory without owning it (Adl.inl:238-253)
    void setRawPtr(const Device* device, T* ptr, u64 size, BufferType = BUFFER)
    {
        ADLASSERT(m_device == 0 || m_device == device);
        ADLASSERT(!m_allocated);
        m_device = device;
        m_ptr = ptr;
        m_size = size;
    }
    void allocate(const Device* device, u64 nElems, BufferType = BUFFER)
    {
        ADLASSERT(m_device == 0 || m_device == device);
        ADLASSERT(m_ptr == 0);
        m_device = device;
        m_size = 0;
        m_ptr = (T*)const_cast<Device*>(device)->allocBytes(nElems * sizeof(T));
        if (m_ptr || nElems == 0) {   // on failure m_ptr = 0, m_size = 0 (AdlCL.inl:390-406)
            m_size = nElems;
            m_allocated = true;
        }
    }
    // asynchronous, queue-ordered copies (Adl.inl:273-303); the caller synchronises
    // (a SyncObject, when given, is recorded behind the copy: AdlCL.inl:441-510 pass it to clEnqueue*Buffer as the event)
    void write(const T* hostSrcPtr, u64 nElems, u64 dstOffsetNElems = 0, SyncObject* syncObj = 0)
    {
        ADLASSERT(nElems + dstOffsetNElems <= m_size);
        m_device->copyH2D(m_ptr + dstOffsetNElems, hostSrcPtr, nElems * sizeof(T));
        if (syncObj) m_device->recordSyncObj(syncObj->m_ptr);
    }
    void read(T* hostDstPtr, u64 nElems, u64 srcOffsetNElems = 0, SyncObject* syncObj = 0) const
    {
        ADLASSERT(nElems + srcOffsetNElems <= m_size);
        m_device->copyD2H(hostDstPtr, m_ptr + srcOffsetNElems, nElems * sizeof(T));
        if (syncObj) m_device->recordSyncObj(syncObj->m_ptr);
    }
    void write(const Buffer<T>& src, u64 nElems, SyncObject* syncObj = 0)
    {
        ADLASSERT(nElems <= m_size && nElems <= src.m_size);
        m_device->copyD2D(m_ptr, src.m_ptr, nElems * sizeof(T));
        if (syncObj) m_device->recordSyncObj(syncObj->m_ptr);
    }
    void read(Buffer<T>& dst, u64 nElems, u64 offsetNElems = 0, SyncObject* syncObj = 0) const
    {
        m_device->copyD2D(dst.m_ptr, m_ptr + offsetNElems, nElems * sizeof(T));
        if (syncObj) m_device->recordSyncObj(syncObj->m_ptr);
    }
    void clear() { m_device->clearBytes(m_ptr, m_size * sizeof(T)); }
    void fill(void* pattern, int patternSize)
    {
        ADLASSERT(patternSize == 4 || patternSize == 8 || patternSize == 16);
        ADLASSERT((m_size * sizeof(T)) % patternSize == 0);
        m_device->fillPattern(m_ptr, pattern, patternSize, m_size * sizeof(T) / patternSize);
    }
    // map / unmap (Adl.inl:317-329): contents valid after waitForCompletion; writes reach the device
    // after returnHostPtr + waitForCompletion
    T* getHostPtr(u64 size = (u64)-1) const
    {
        const u64 n = size == (u64)-1 ? m_size : size;
        return (T*)m_device->mapBytes(m_ptr, n * sizeof(T));
    }
    void returnHostPtr(T* ptr) const { m_device->unmapBytes(m_ptr, ptr, 0); }   // 0 = the whole mapping
    // grow-only; contents are NOT preserved (Adl.inl:331-356)
    void setSize(u64 size)
    {
        ADLASSERT(m_device != 0);
        if (size > m_size) {
            const Device* dev = m_device;
            DeviceUtils::waitForCompletion(dev);
            if (m_allocated && m_ptr) const_cast<Device*>(dev)->freeBytes(m_ptr, m_size * sizeof(T));
            m_ptr = 0;
            m_size = 0;
            m_allocated = false;
            allocate(dev, size);
        }
    }
    u64 getSize() const { return m_size; }
    DeviceType getType() const
    {
        ADLASSERT(m_device != 0);
        return m_device->m_type;
    }

    const Device* m_device;
    u64 m_size;
    T* m_ptr;
    bool m_allocated;

private:
    Buffer(const Buffer&);
    Buffer& operator=(const Buffer&);
};

template <typename T>
struct HostBuffer : public Buffer<T> {
    HostBuffer() : Buffer<T>() {}
    HostBuffer(const Device* device, int nElems, BufferBase::BufferType type = BufferBase::BUFFER) : Buffer<T>(device, nElems, type) {}
    T& operator[](int idx) { return Buffer<T>::m_ptr[idx]; }
    const T& operator[](int idx) const { return Buffer<T>::m_ptr[idx]; }
    T* begin() { return Buffer<T>::m_ptr; }
};

// ---------------------------------------------------------------------------------------------
// BufferUtils (Adl/Adl.h:224-248, Adl.inl:370-535): a buffer of one device type seen from a device of another.
//   map<TYPE, COPY>(device, in)       in already lives on a TYPE device -> `in` itself; TYPE_HOST -> a view of in's mapping
//                                     (getHostPtr); else a new buffer on `device`, filled from `in` when COPY
//   unmap<COPY>(native, orig)         the inverse: returnHostPtr / copy back when COPY, then `native` is deleted
//   mapInplace / unmapInplace         the same into / out of a buffer the caller has allocated (nothing is deleted)
// Copies between two device buffers go through a host bounce buffer, as the reference's do; every step is completed
// (waitForCompletion) where the reference completes it.
// ---------------------------------------------------------------------------------------------
class BufferUtils {
public:
    template <DeviceType TYPE, bool COPY, typename T>
    static Buffer<T>* map(const Device* device, const Buffer<T>* in, int copySize = -1)
    {
        ADLASSERT(device->m_type == TYPE);
        if (in->getType() == TYPE) return const_cast<Buffer<T>*>(in);
        ADLASSERT(copySize <= (int)in->getSize());
        const u64 n = copySize == -1 ? in->getSize() : (u64)copySize;
        Buffer<T>* native;
        if (TYPE == TYPE_HOST) {   // a host view of the device buffer's mapping (valid after waitForCompletion(in's device))
            native = new Buffer<T>;
            native->setRawPtr(device, in->getHostPtr(n), n);
        } else {
            native = new Buffer<T>(device, n);
            if (COPY) copyAcross(native, in, n);
        }
        return native;
    }
    template <bool COPY, typename T>
    static void unmap(Buffer<T>* native, const Buffer<T>* orig, int copySize = -1)
    {
        if (native == orig) return;
        if (native->getType() == TYPE_HOST) {
            orig->returnHostPtr(native->m_ptr);
        } else {
            if (COPY) {
                const u64 n = copySize == -1 ? (orig->getSize() < native->getSize() ? orig->getSize() : native->getSize()) : (u64)copySize;
                ADLASSERT(n <= orig->getSize());
                copyAcross(const_cast<Buffer<T>*>(orig), native, n);
            }
            DeviceUtils::waitForCompletion(native->m_device);
        }
        delete native;
    }
    template <DeviceType TYPE, bool COPY, typename T>
    static Buffer<T>* mapInplace(const Device* device, Buffer<T>* allocatedBuffer, const Buffer<T>* in, int copySize = -1)
    {
        ADLASSERT(device->m_type == TYPE);
        if (in->getType() == TYPE) return const_cast<Buffer<T>*>(in);
        ADLASSERT(copySize <= (int)in->getSize());
        const u64 n = copySize == -1 ? (in->getSize() < allocatedBuffer->getSize() ? in->getSize() : allocatedBuffer->getSize()) : (u64)copySize;
        if (COPY) copyAcross(allocatedBuffer, in, n);
        return allocatedBuffer;
    }
    template <bool COPY, typename T>
    static void unmapInplace(Buffer<T>* native, const Buffer<T>* orig, int copySize = -1)
    {
        if (native == orig || !COPY) return;
        const u64 n = copySize == -1 ? (orig->getSize() < native->getSize() ? orig->getSize() : native->getSize()) : (u64)copySize;
        ADLASSERT(n <= orig->getSize());
        copyAcross(const_cast<Buffer<T>*>(orig), native, n);
    }

private:
    // dst[0, n) = src[0, n) across device types, completed on return (Adl.inl:392-410, 436-452)
    template <typename T>
    static void copyAcross(Buffer<T>* dst, const Buffer<T>* src, u64 n)
    {
        if (src->getType() == TYPE_HOST) {
            dst->write(src->m_ptr, n);
            DeviceUtils::waitForCompletion(dst->m_device);
        } else if (dst->getType() == TYPE_HOST) {
            src->read(dst->m_ptr, n);
            DeviceUtils::waitForCompletion(src->m_device);
        } else {
            T* tmp = new T[n];
            src->read(tmp, n);
            DeviceUtils::waitForCompletion(src->m_device);
            dst->write(tmp, n);
            DeviceUtils::waitForCompletion(dst->m_device);
            delete[] tmp;
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Stopwatch (Adl/AdlStopwatch.h:60-83): for TYPE_CL the reference instantiates the HOST stopwatch
// (AdlStopwatch.inl:17-20): wall clock after a device sync.  Same here.
// ---------------------------------------------------------------------------------------------
class Stopwatch {
public:
    enum { CAPACITY = 64 };
    Stopwatch(const Device* device = 0) : m_device(device), m_idx(0) {}
    void init(const Device* device) { m_device = device; m_idx = 0; }
    void start()
    {
        m_idx = 0;
        split();
    }
    void split()
    {
        if (m_device) DeviceUtils::waitForCompletion(m_device);
        if (m_idx < CAPACITY) m_t[m_idx++] = std::chrono::steady_clock::now();
    }
    void stop() { split(); }
    float getMs(int index = 0)
    {
        if (index + 1 >= m_idx) return 0.f;
        return std::chrono::duration<float, std::milli>(m_t[index + 1] - m_t[index]).count();
    }
    void getMs(float* times, int capacity)
    {
        for (int i = 0; i < capacity; ++i) times[i] = getMs(i);
    }

private:
    const Device* m_device;
    int m_idx;
    std::chrono::steady_clock::time_point m_t[CAPACITY];
};

}  // namespace adl

#endif
