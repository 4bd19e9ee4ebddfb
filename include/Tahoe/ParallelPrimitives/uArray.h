// Tahoe/ParallelPrimitives/uArray.h -- host array with a lazily created, grow-only device twin.
// Reference: Tahoe/ParallelPrimitives/uArray.h:13-66, :172-212.  The reference keeps a CPU/GPU
// dirty-state machine and re-uploads on resize; Pprims only ever uses the device side as scratch, and
// the reference test only uses the host side (operator[], begin), so that is what exists here:
//   uArray(size)            host storage of `size` elements
//   operator[], begin, getSize, setSize (host side; contents preserved on growth)
//   getGpuBuffer(device)    device buffer of at least getSize() elements, created / grown on demand
//                           (contents NOT preserved -- scratch)
#pragma once
#include <Adl/Adl.h>
#include <Tahoe/Math/Array.h>

namespace Tahoe {

template <typename T>
class uArray {
public:
    explicit uArray(u64 size = 0) : m_host(size), m_gpu(0) {}
    ~uArray() { delete m_gpu; }

    T& operator[](u64 i) { return m_host[i]; }
    const T& operator[](u64 i) const { return m_host[i]; }
    T* begin() { return m_host.begin(); }
    u64 getSize() const { return m_host.getSize(); }
    void setSize(u64 size) { m_host.setSize(size); }

    adl::Buffer<T>* getGpuBuffer(const adl::Device* device)
    {
        if (m_gpu && m_gpu->m_device != device) {
            delete m_gpu;
            m_gpu = 0;
        }
        if (!m_gpu) m_gpu = new adl::Buffer<T>(device, m_host.getSize() ? m_host.getSize() : 1);
        if (m_gpu->getSize() < m_host.getSize()) m_gpu->setSize(m_host.getSize());
        return m_gpu;
    }

private:
    uArray(const uArray&);
    uArray& operator=(const uArray&);
    Array<T> m_host;
    adl::Buffer<T>* m_gpu;
};

}  // namespace Tahoe
