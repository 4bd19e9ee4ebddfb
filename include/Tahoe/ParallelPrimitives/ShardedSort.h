// Tahoe/ParallelPrimitives/ShardedSort.h -- the multi-GPU radix sort in the API language of the reference.
// No reference counterpart: the reference's Pprims drives ONE device (Tahoe/ParallelPrimitives/Pprims.h:35-41,
// Adl/Adl.h:90-94).  One ShardedSort = one adlhip_group (include/adlhip.h, "sharded sort"): G devices driven by
// one host thread, stable top-byte partition per device, balanced splitters from the global histogram, one
// grouped RCCL send/recv exchange over xGMI, local Pprims-style radix sort of what arrives.  Rank r ends up with
// a contiguous ascending slice; the slices in rank order are the sorted whole (pairs: stable).
#pragma once
#include <Adl/Adl.h>
#include <Tahoe/Math/Math.h>

#include <stddef.h>
#include <vector>

namespace Tahoe {

class ShardedSort {
public:
    // deviceIndices: nDevices distinct device indices, or 0 for 0 .. nDevices-1 (ADLASSERTs if the group cannot be made)
    explicit ShardedSort(int nDevices, const int* deviceIndices = 0);
    ~ShardedSort();   // every Buffer allocated from getDevice(r) must be gone by now (Adl.inl:102 semantics)

    int getNDevices() const { return (int)m_devices.size(); }
    // the device of rank r: allocate that rank's shard and output Buffers with it (owned by this object)
    const adl::Device* getDevice(int rank) const { return m_devices[(size_t)rank]; }

    // shards[r]: nIn[r] elements on getDevice(r), left intact.  out[r]: a Buffer on getDevice(r); nOut[r] receives the size
    // of rank r's slice (out[r] must be at least that large: 1.25 x the mean + slack unless one top byte dominates).
    // Work is enqueued; results are valid after waitForCompletion().
    void radixSort(const adl::Buffer<u32>* const* shards, const size_t* nIn, adl::Buffer<u32>* const* out, size_t* nOut);
    void radixSort(const adl::Buffer<uint2>* const* shards, const size_t* nIn, adl::Buffer<uint2>* const* out, size_t* nOut);

    void waitForCompletion() const;

private:
    ShardedSort(const ShardedSort&);
    ShardedSort& operator=(const ShardedSort&);
    adlhip_group* m_group;
    std::vector<adl::Device*> m_devices;
};

}  // namespace Tahoe
