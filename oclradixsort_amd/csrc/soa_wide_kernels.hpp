// soa_wide_kernels.hpp -- key-value sort on separate arrays with keys of 4 or 8 bytes and values of 4, 8 or 16 bytes
// (adlhip_radix_sort_soa; SURVEY f3, "SoA key/value API ... and 64-bit values").
//
// The reference's SoA kernel carries a 4-byte value beside every key through each of its passes
// (Tahoe/ClKernels/RadixSortKeyValueKernels.cl:354-509: gSrcVal -> ldsSortVal -> gDstVal).  Wide values are not
// carried: the sort moves {32 key bits, source index} pairs -- the 8-byte element of Pprims::radixSort(Buffer<uint2>)
// (Pprims.h:38), so the stable large sort of section 4.2d does the work at its full speed -- and the keys and values are
// fetched ONCE, at the end, from where the indices point (HBM: n x (8 + K + V) read, n x (K + V) written by the gather,
// whatever V is; carrying a 16-byte value through three sweeps would move 6 x 16 bytes per element).
// 64-bit keys take two such sorts, low dword first: each is stable, so the second leaves equal high dwords in the order
// of their low dwords -- an LSD sort with 32-bit digits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adlhip {

constexpr int kSoaNT = 256;

// pairs[i] = {low dword of keys[i] (K = uint64_t: bits 0..31), i}
template <typename K>
__global__ __launch_bounds__(kSoaNT) void soa_pack_index_kernel(const K* __restrict__ keys, uint64_t* __restrict__ pairs, uint32_t n)
{
    const uint32_t stride = gridDim.x * (uint32_t)kSoaNT;
    for (uint32_t i = blockIdx.x * (uint32_t)kSoaNT + threadIdx.x; i < n; i += stride)
        __builtin_nontemporal_store((uint64_t)(uint32_t)keys[i] | ((uint64_t)i << 32), pairs + i);
}

// second round of 64-bit keys: out[j] = {high dword of keys[idx], idx}, idx = the index in[j] carries
__global__ __launch_bounds__(kSoaNT) void soa_repack_high_kernel(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ in,
                                                                 uint64_t* __restrict__ out, uint32_t n)
{
    const uint32_t stride = gridDim.x * (uint32_t)kSoaNT;
    for (uint32_t j = blockIdx.x * (uint32_t)kSoaNT + threadIdx.x; j < n; j += stride) {
        const uint32_t idx = (uint32_t)(__builtin_nontemporal_load(in + j) >> 32);
        out[j] = (keys[idx] >> 32) | ((uint64_t)idx << 32);
    }
}

// keys_out[j] = keys_in[idx_j] (K = uint32_t: the pair's own low dword, keys_in is not read), vals_out[j] = vals_in[idx_j]
template <typename K, typename V>
__global__ __launch_bounds__(kSoaNT) void soa_gather_kernel(const uint64_t* __restrict__ pairs, const K* __restrict__ keys_in,
                                                            K* __restrict__ keys_out, const V* __restrict__ vals_in,
                                                            V* __restrict__ vals_out, uint32_t n)
{
    const uint32_t stride = gridDim.x * (uint32_t)kSoaNT;
    for (uint32_t j = blockIdx.x * (uint32_t)kSoaNT + threadIdx.x; j < n; j += stride) {
        const uint64_t p = __builtin_nontemporal_load(pairs + j);
        const uint32_t idx = (uint32_t)(p >> 32);
        if constexpr (sizeof(K) == 4) keys_out[j] = (K)(uint32_t)p;
        else keys_out[j] = keys_in[idx];
        vals_out[j] = vals_in[idx];
    }
}

}  // namespace adlhip
