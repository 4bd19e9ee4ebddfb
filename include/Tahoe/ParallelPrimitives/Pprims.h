// Tahoe/ParallelPrimitives/Pprims.h -- the parallel-primitives object of the reference
// (Tahoe/ParallelPrimitives/Pprims.h:11-48) over the MI355X HIP back-end:
//   scan      exclusive prefix sum                                   (reference Pprims.h:35)
//   radixSort {u32 key, u32 value} pairs, stable                     (reference Pprims.h:38)
//   radixSort u32 keys                                               (reference Pprims.h:41)
//   radixSort u64 keys                                               (new: BASELINE config #5)
//   copy / fill                                                      (reference Pprims.cpp:31-120, commented out there)
// Same argument meaning; differences, all supersets: any n >= 0 (the reference needs n % 256 == 0 for
// keys), scan has no 1,048,576-element limit, sortBits < 32 also works on 64-bit keys up to 64.
// Device work is enqueued and the call returns (no sync), as in the reference's GPU branches.
// A TYPE_HOST device takes the CPU path (Tahoe::RadixSort::sort), exactly as Pprims.cpp:202-212/306-316;
// a TYPE_CL (HIP) device never falls back to the CPU.
#pragma once
#include <Adl/Adl.h>
#include <Tahoe/Math/Math.h>
#include <Tahoe/ParallelPrimitives/uArray.h>   // the reference's Pprims.h pulls uArray / Array in for its users

namespace Tahoe {

class Pprims {
public:
    TH_DECLARE_ALLOCATOR(Pprims);

    Pprims();
    ~Pprims();

    void cacheKernel(bool cache) { m_cacheKernel = cache; }   // kernels are built ahead of time; kept for API parity

    enum {
        SCAN_BLOCK_SIZE = 128,
        RSORT_BITS_PER_PASS = 8,
        RSORT_NUM_TABLES = (1 << RSORT_BITS_PER_PASS),
        R32SORT_DATA_ALIGNMENT = 256,   // the reference's requirement on n; not needed here
        R32SORT_WG_SIZE = 64,
        R32SORT_ELEMENTS_PER_WORK_ITEM = (256 / R32SORT_WG_SIZE),
        R32SORT_BITS_PER_PASS = 4,      // the reference's digit width; select with adlhip "sort.digit_bits"
    };

    // copy / fill (Pprims.cpp:31-120 -- present but commented out in the reference, with CopyIntKernel / CopyF4Kernel /
    // FillIntKernel / FillU32Kernel / FillF4Kernel of ClKernels/PprimsKernels.cl): first n elements, enqueued
    void copy(const adl::Device* device, adl::Buffer<int>& dst, const adl::Buffer<int>& src, int n);
    void copy(const adl::Device* device, adl::Buffer<float4>& dst, const adl::Buffer<float4>& src, int n);
    void fill(const adl::Device* device, adl::Buffer<int>& dst, int src, int n);
    void fill(const adl::Device* device, adl::Buffer<u32>& dst, u32 src, int n);
    void fill(const adl::Device* device, adl::Buffer<float4>& dst, const float4& src, int n);

    void scan(const adl::Device* device, adl::Buffer<int>& dst, const adl::Buffer<int>& src, int n, u32* sumOut = 0);

    // inout.x: key, inout.y: value
    void radixSort(const adl::Device* device, const adl::Buffer<uint2>& inout, int n, int sortBits = 32);

    void radixSort(const adl::Device* device, const adl::Buffer<u32>& inout, int n, int sortBits = 32);

    void radixSort(const adl::Device* device, const adl::Buffer<u64>& inout, int n, int sortBits = 64);

    // separate key and value buffers (the layout of the reference's never-launched SoA kernel,
    // RadixSortKeyValueKernels.cl:354-509): ascending by key, stable, values follow their keys
    void radixSort(const adl::Device* device, const adl::Buffer<u32>& keys, const adl::Buffer<u32>& values, int n,
                   int sortBits = 32);
    // the same with 64-bit values and / or 64-bit keys: {32 key bits, index} pairs are sorted, keys and values
    // gathered once at the end (adlhip_radix_sort_soa)
    void radixSort(const adl::Device* device, const adl::Buffer<u32>& keys, const adl::Buffer<u64>& values, int n,
                   int sortBits = 32);
    void radixSort(const adl::Device* device, const adl::Buffer<u64>& keys, const adl::Buffer<u32>& values, int n,
                   int sortBits = 64);
    void radixSort(const adl::Device* device, const adl::Buffer<u64>& keys, const adl::Buffer<u64>& values, int n,
                   int sortBits = 64);

private:
    // device scratch owned by the object and grown lazily (reference: m_u32WorkBuffer[0] = ping-pong data,
    // m_u32WorkBuffer[1] = histogram table; Pprims.h:44-45)
    void reserve(const adl::Device* device, size_t tmpBytes, size_t workBytes);
    void sortSoaWide(const adl::Device* device, void* keys, int keyBytes, void* values, int valueBytes, int n, int sortBits);
    adl::Buffer<unsigned char>* m_tmp;
    adl::Buffer<unsigned char>* m_work;
    bool m_cacheKernel;
};

}  // namespace Tahoe
