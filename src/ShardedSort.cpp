// Host side of Tahoe::ShardedSort over the C ABI (include/adlhip.h: adlhip_group_*, adlhip_sharded_sort_*).
// No reference counterpart (the reference is single-device, Adl/Adl.h:90-94); everything the caller can observe --
// ownership of the per-rank devices, argument checks, where results are valid -- lives here, the steps of the
// sort behind the ABI.
#include <Tahoe/ParallelPrimitives/ShardedSort.h>

#include <cstdio>

namespace Tahoe {

ShardedSort::ShardedSort(int nDevices, const int* deviceIndices) : m_group(0)
{
    const int rc = adlhip_group_create(deviceIndices, nDevices, &m_group);
    if (rc != ADLHIP_SUCCESS) fprintf(stderr, "ShardedSort: %s\n", adlhip_last_error());
    ADLASSERT(rc == ADLHIP_SUCCESS);
    if (rc != ADLHIP_SUCCESS) return;
    for (int r = 0; r < nDevices; ++r) {
        adl::DeviceHip* d = new adl::DeviceHip();
        d->attach(adlhip_group_device(m_group, r));   // owned by the group: release() leaves the handle alone
        m_devices.push_back(d);
    }
}

ShardedSort::~ShardedSort()
{
    if (m_group) {
        const int rc = adlhip_group_destroy(m_group);   // refuses while the caller still holds memory of a rank's device
        if (rc != ADLHIP_SUCCESS) fprintf(stderr, "ShardedSort: %s\n", adlhip_last_error());
        ADLASSERT(rc == ADLHIP_SUCCESS);
    }
    for (size_t r = 0; r < m_devices.size(); ++r) delete m_devices[r];
}

void ShardedSort::waitForCompletion() const
{
    for (size_t r = 0; r < m_devices.size(); ++r) m_devices[r]->waitForCompletion();
}

template <typename T, typename Fn>
static void run(adlhip_group* group, const std::vector<adl::Device*>& devices, const adl::Buffer<T>* const* shards,
                const size_t* nIn, adl::Buffer<T>* const* out, size_t* nOut, Fn fn)
{
    ADLASSERT(group != 0 && shards != 0 && nIn != 0 && out != 0 && nOut != 0);
    if (!group || !shards || !nIn || !out || !nOut) return;
    const size_t G = devices.size();
    std::vector<void*> in(G), dst(G);
    std::vector<size_t> cap(G);
    for (size_t r = 0; r < G; ++r) {
        ADLASSERT(shards[r] != 0 && out[r] != 0);
        ADLASSERT(shards[r]->m_device == devices[r] && out[r]->m_device == devices[r]);   // a rank's buffers live on its device
        ADLASSERT((adl::u64)nIn[r] <= shards[r]->getSize());
        in[r] = shards[r]->m_ptr;
        dst[r] = out[r]->m_ptr;
        cap[r] = (size_t)out[r]->getSize();
    }
    const int rc = fn(group, in.data(), nIn, dst.data(), cap.data(), nOut);
    if (rc != ADLHIP_SUCCESS) fprintf(stderr, "ShardedSort: %s\n", adlhip_last_error());
    ADLASSERT(rc == ADLHIP_SUCCESS);
}

void ShardedSort::radixSort(const adl::Buffer<u32>* const* shards, const size_t* nIn, adl::Buffer<u32>* const* out, size_t* nOut)
{
    run<u32>(m_group, m_devices, shards, nIn, out, nOut,
             [](adlhip_group* g, void* const* in, const size_t* n, void* const* dst, const size_t* cap, size_t* no) {
                 return adlhip_sharded_sort_u32(g, reinterpret_cast<uint32_t* const*>(in), n, reinterpret_cast<uint32_t* const*>(dst), cap, no);
             });
}

void ShardedSort::radixSort(const adl::Buffer<uint2>* const* shards, const size_t* nIn, adl::Buffer<uint2>* const* out, size_t* nOut)
{
    run<uint2>(m_group, m_devices, shards, nIn, out, nOut,
               [](adlhip_group* g, void* const* in, const size_t* n, void* const* dst, const size_t* cap, size_t* no) {
                   return adlhip_sharded_sort_kv32(g, in, n, dst, cap, no);
               });
}

}  // namespace Tahoe
