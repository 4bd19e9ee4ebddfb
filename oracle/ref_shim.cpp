/*
 * oracle/ref_shim.cpp -- extern "C" doorway into the REAL reference code, for tests only.
 *
 * TEST INFRASTRUCTURE.  Built only by oracle/Makefile into oracle/_ref/libref.so, linking the
 * reference's own translation units where they lie under /root/reference (nothing is copied):
 *   Tahoe/Algorithm/Sort/RadixSort.cpp          (the CPU sort = the parity oracle)
 *   Tahoe/ParallelPrimitives/Pprims.cpp         (Pprims::radixSort host branch, Pprims.cpp:202-212,306-316)
 * plus the header-only Adl/Host back-end.  This file contains no algorithm: it forwards.
 *
 * ref_host_path_* drive BASELINE config #1 literally ("Demo.Sort32 on the Adl/Host CPU path"):
 * DeviceUtils::allocate(TYPE_HOST) -> Buffer<T> -> Pprims::radixSort -> RadixSort::sort.
 */
#include <Adl/Adl.h>
#include <Tahoe/ParallelPrimitives/Pprims.h>
#include <Tahoe/Algorithm/Sort/RadixSort.h>
#include <stdint.h>
#include <string.h>

char adl::s_cacheDirectory[128] = "cache";   /* the application defines it: UnitTest/main.cpp:74 */

extern "C" {

/* Tahoe::RadixSort::sort(u32*, int) -- RadixSort.cpp:58 */
int ref_radix_sort_u32(uint32_t *data, int n)
{
    Tahoe::RadixSort::sort((Tahoe::u32 *)data, n);
    return 0;
}

/* Tahoe::RadixSort::sort(SortData*, int) -- RadixSort.cpp:10 */
int ref_radix_sort_kv32(uint64_t *pairs, int n)
{
    static_assert(sizeof(Tahoe::SortData) == 8, "SortData is {u32 key; u32 value;}");
    Tahoe::RadixSort::sort((Tahoe::SortData *)pairs, n);
    return 0;
}

/* Pprims::radixSort(Buffer<u32>) on a TYPE_HOST device -- the CPU fallback branch. */
int ref_host_path_sort_u32(uint32_t *data, int n)
{
    adl::DeviceUtils::Config cfg;
    adl::Device *d = adl::DeviceUtils::allocate(adl::TYPE_HOST, cfg);
    if (!d) return 1;
    {
        Tahoe::Pprims p;
        adl::Buffer<Tahoe::u32> buf(d, (adl::u64)n);
        Tahoe::u32 *h = buf.getHostPtr(n);
        adl::DeviceUtils::waitForCompletion(d);
        memcpy(h, data, sizeof(uint32_t) * (size_t)n);
        buf.returnHostPtr(h);
        adl::DeviceUtils::waitForCompletion(d);
        p.radixSort(d, buf, n);
        h = buf.getHostPtr(n);
        adl::DeviceUtils::waitForCompletion(d);
        memcpy(data, h, sizeof(uint32_t) * (size_t)n);
        buf.returnHostPtr(h);
    }
    adl::DeviceUtils::deallocate(d);
    return 0;
}

/* Pprims::radixSort(Buffer<uint2>) on a TYPE_HOST device. */
int ref_host_path_sort_kv32(uint64_t *pairs, int n)
{
    adl::DeviceUtils::Config cfg;
    adl::Device *d = adl::DeviceUtils::allocate(adl::TYPE_HOST, cfg);
    if (!d) return 1;
    {
        Tahoe::Pprims p;
        adl::Buffer<Tahoe::uint2> buf(d, (adl::u64)n);
        Tahoe::uint2 *h = buf.getHostPtr(n);
        adl::DeviceUtils::waitForCompletion(d);
        memcpy(h, pairs, 8 * (size_t)n);
        buf.returnHostPtr(h);
        adl::DeviceUtils::waitForCompletion(d);
        p.radixSort(d, buf, n);
        h = buf.getHostPtr(n);
        adl::DeviceUtils::waitForCompletion(d);
        memcpy(pairs, h, 8 * (size_t)n);
        buf.returnHostPtr(h);
    }
    adl::DeviceUtils::deallocate(d);
    return 0;
}

} /* extern "C" */
