// dict_kernels.hpp -- keys that take FEW DISTINCT VALUES (flags, categories, quantised data: at most 256 values among millions of
// keys) are sorted by counting: look every key up in a dictionary of the values, count, and write the runs.
//
// Why it exists.  Pprims::radixSort (Tahoe/ParallelPrimitives/Pprims.cpp:304-406) sorts such keys like any others: sortBits / 4
// passes over the data.  The large sort of this back-end (hybrid_kernels.hpp) cannot take them at all -- it gives every bucket
// and segment the same room, and a value with a million copies outgrows any slab -- so they used to fall to the one-sweep path:
// 0.60-0.65 ms for 64 Mi keys that are all equal, take 16 values or differ in their low byte only.  Counting needs one read
// and one write of the array: ~0.15 ms.  Equal keys are indistinguishable, so the output of a sort of whole keys is determined
// by the counts alone; it is bit for bit what the reference's CPU sort (Tahoe/Algorithm/Sort/RadixSort.cpp:58-104) produces.
// ({key, value} pairs and sorts on part of the key keep the ordinary paths: there equal keys are not interchangeable.)
//
//   1. large_probe_kernel (hybrid_kernels.hpp's probe, extended)  16 Ki sampled keys; if they take at most 256 values it writes
//      the dictionary -- the values in ascending order and a 1024-slot hash table over them -- and says so to the host
//   2. dict_count_kernel   every key is looked up (hash, one or two LDS reads) and counted; a key that is not in the dictionary
//      raises `miss` (the sample cannot see a value that occurs once in a million).  The last workgroup scans the counts
//   3. dict_fill_kernel    writes the runs in place -- unless `miss` is set: then the input is untouched and its first 256
//      workgroups sort it with the cooperative LSD sort (hybrid_kernels.hpp coop_lsd_sort), the large sort's safety net
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hybrid_kernels.hpp"

namespace adlhip {

// index of v in the dictionary, or 0xffffffff.  s_key / s_idx: the hash table in LDS; n_values for the all-ones key.
// K = the key type of the LDS copy: the element type itself (u32 keys: 32-bit slots -- a 64-bit LDS read of ONE address by all
// 64 lanes is not broadcast but served lane after lane); free slots hold K(~0), which for u32 keys cannot be told from the key
// 0xffffffff, so `has_max` says whether that key is a value (it is then the last one: values are ascending).
template <typename K>
__device__ __forceinline__ uint32_t dict_lookup(K v, const K* __restrict__ s_key, const uint16_t* __restrict__ s_idx, uint32_t n_values,
                                                bool has_max)
{
    if (v == (K)~(K)0) return has_max ? n_values - 1u : 0xffffffffu;
    uint32_t h = dict_hash((unsigned long long)v);
    for (int step = 0; step < kDictSlots; ++step) {
        const K k = s_key[h];
        if (k == v) return s_idx[h];
        if (k == (K)~(K)0) return 0xffffffffu;
        h = (h + 1u) & (uint32_t)(kDictSlots - 1);
    }
    return 0xffffffffu;
}

constexpr int kDictNT = 256;
constexpr int kDictWGs = 2048;

template <typename E>
__global__ __launch_bounds__(kDictNT) void dict_count_kernel(const E* __restrict__ src, uint32_t n, DictBlock* __restrict__ blk)
{
    __shared__ E s_key[kDictSlots];
    __shared__ uint16_t s_idx[kDictSlots];
    __shared__ uint32_t s_cnt[kDictNT / 64][kDictMax];
    __shared__ uint32_t s_misc[2];
    const int tid = (int)threadIdx.x;
    const uint32_t nv = blk->n_values;
    if (nv == 0u) {   // no dictionary (the keys changed since the host chose this path): the fill kernel's safety net sorts
        if (blockIdx.x == 0 && tid == 0) blk->miss = 1u;
        return;
    }
    for (int i = tid; i < kDictSlots; i += kDictNT) {
        s_key[i] = (E)blk->slot_key[i];   // (free slots: all ones in either width)
        s_idx[i] = (uint16_t)blk->slot_idx[i];
    }
    for (int i = tid; i < (kDictNT / 64) * kDictMax; i += kDictNT) (&s_cnt[0][0])[i] = 0u;
    const bool has_max = blk->value[nv - 1u] == (unsigned long long)(E)~(E)0;   // the all-ones key is a value
    __syncthreads();
    uint32_t* my = s_cnt[tid >> 6];
    bool miss = false;
    constexpr int VEC = 16 / (int)sizeof(E);
    struct alignas(16) Vec { E v[VEC]; };
    const uint32_t nvec = n / VEC;
    const Vec* vsrc = reinterpret_cast<const Vec*>(src);   // sort buffers are 16-byte aligned
    // A wave whose 64 keys are the same value (constant stretches, one dominant value) looks it up and counts it ONCE, in one lane:
    // 64 lanes on one LDS address are served one after the other (all-equal keys: 0.24 ms in this kernel instead of 0.07).
    const int lane = tid & 63;
    auto one = [&](E x, bool active) {
        const E x0 = (E)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |
                     (sizeof(E) == 8 ? (E)((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((unsigned long long)x >> 32)) << 32) : E(0));
        if (__all(active && x == x0)) {
            if (lane == 0) {
                const uint32_t ix = dict_lookup<E>(x, s_key, s_idx, nv, has_max);
                if (ix == 0xffffffffu) miss = true;
                else my[ix] += 64u;
            }
        } else if (active) {
            // (peeling off the two or three values that many lanes share, one add each, costs more than the serialised atomics it
            // saves: two values 142 -> 173 us, 16 values 86 -> 193)
            const uint32_t ix = dict_lookup<E>(x, s_key, s_idx, nv, has_max);
            if (ix == 0xffffffffu) miss = true;
            else atomicAdd(&my[ix], 1u);
        }
    };
    for (uint32_t i0 = blockIdx.x * kDictNT; i0 < nvec; i0 += gridDim.x * kDictNT) {   // whole waves stay together (the vote above)
        const uint32_t i = i0 + (uint32_t)tid;
        const bool active = i < nvec;
        Vec v;
        if (active) v = vsrc[i];
#pragma unroll
        for (int k = 0; k < VEC; ++k) one(active ? v.v[k] : E(0), active);
    }
    if (blockIdx.x == 0) {
        const bool active = (uint32_t)tid < n - nvec * VEC;
        one(active ? src[nvec * VEC + (uint32_t)tid] : E(0), active);
    }
    const int any_miss = __syncthreads_or(miss);
    if (any_miss) {
        if (tid == 0) __hip_atomic_store(&blk->miss, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if ((uint32_t)tid < nv) {
        uint32_t c = 0u;
#pragma unroll
        for (int w = 0; w < kDictNT / 64; ++w) c += s_cnt[w][tid];
        if (c) __hip_atomic_fetch_add(&blk->count[tid], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the last workgroup to finish scans the counts (they are only ever touched by agent-scope atomics: no cache to flush)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_misc[0] = __hip_atomic_fetch_add(&blk->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_misc[0] != gridDim.x - 1u) return;
    uint32_t* s_wsum = &s_cnt[0][0];
    const uint32_t c = (uint32_t)tid < nv ? __hip_atomic_load(&blk->count[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan_u32<kDictNT>(c, s_wsum, &total);
    if ((uint32_t)tid < nv) blk->offset[tid] = ex;
    if (tid == 0) {
        blk->offset[nv] = total;
        if (total != n) __hip_atomic_store(&blk->miss, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a miss elsewhere: counts incomplete)
    }
}

// Writes the runs -- or, when a key missed the dictionary, sorts the untouched input with the cooperative LSD sort (its first 256
// workgroups; all resident: the grid is launched with at least that many and they are dispatched first).
template <typename E>
__global__ __launch_bounds__(kDictNT) void dict_fill_kernel(E* __restrict__ data, E* __restrict__ tmp, uint32_t n, DictBlock* __restrict__ blk,
                                                            uint32_t* __restrict__ ctable, uint32_t* fault, uint32_t* host_report,
                                                            int key_bits, uint32_t chunk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the safety net's tile (TileCfg<E, 8, 256, 16>)
    __shared__ uint32_t s_off[kDictMax + 1];
    __shared__ unsigned long long s_val[kDictMax];
    const int tid = (int)threadIdx.x;
    const uint32_t miss = __hip_atomic_load(&blk->miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && tid == 0)   // 4 = sorted by counting, 2 = the keys did not fit (as the large sort reports it)
        __hip_atomic_store(host_report, miss ? 2u : 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (miss) {
        if (blockIdx.x < 256u) {
            // (the counters of this sort are void; the next sort's probe starts from its own zeroes)
            if (blockIdx.x == 0)
                for (int i = tid; i < kDictMax; i += kDictNT) blk->count[i] = 0u;
            coop_lsd_sort<E, kDictNT, 16>(data, tmp, n, ctable, ctable + 256 * 256, &blk->bar, fault, smem, key_bits, 0u, 256u);
        }
        return;
    }
    const uint32_t nv = blk->n_values;
    for (int i = tid; i <= (int)nv; i += kDictNT) s_off[i] = blk->offset[i];
    for (int i = tid; i < (int)nv; i += kDictNT) s_val[i] = blk->value[i];
    __syncthreads();
    const uint32_t p0 = blockIdx.x * chunk;
    if (p0 >= n) {
        // (nothing to write; the first idle workgroup -- there is one unless the grid is exact -- would be a place for chores)
    } else {
        const uint32_t p1 = p0 + chunk < n ? p0 + chunk : n;
        // value of position p0: the last index whose offset is <= p0
        uint32_t lo = 0u, hi = nv;
        while (hi - lo > 1u) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_off[mid] <= p0) lo = mid; else hi = mid;
        }
        uint32_t ix = lo;
        while (ix + 1u < nv && s_off[ix + 1u] <= p0) ++ix;   // (empty runs)
        // the chunk in stretches of one value each (nearly always one stretch: a run is n / 256 keys and more)
        uint32_t p = p0;
        while (p < p1) {
            while (ix + 1u < nv && s_off[ix + 1u] <= p) ++ix;
            const uint32_t end = s_off[ix + 1u] < p1 ? s_off[ix + 1u] : p1;
            const E v = (E)s_val[ix];
            constexpr uint32_t VEC = 16u / (uint32_t)sizeof(E);
            struct alignas(16) Vec { E v[VEC]; };
            const uint32_t a0 = (p + VEC - 1u) / VEC * VEC;             // first 16-byte boundary inside
            const uint32_t a1 = end / VEC * VEC;
            if (a0 < a1) {
                Vec vv;
#pragma unroll
                for (uint32_t k = 0; k < VEC; ++k) vv.v[k] = v;
                Vec* out = reinterpret_cast<Vec*>(data);
                for (uint32_t i = a0 / VEC + (uint32_t)tid; i < a1 / VEC; i += kDictNT) out[i] = vv;
                for (uint32_t i = p + (uint32_t)tid; i < a0; i += kDictNT) data[i] = v;
                for (uint32_t i = a1 + (uint32_t)tid; i < end; i += kDictNT) data[i] = v;
            } else {
                for (uint32_t i = p + (uint32_t)tid; i < end; i += kDictNT) data[i] = v;
            }
            p = end;
        }
    }
    // the counters go back to zero for the next sort (every workgroup has read the offsets it needs; counts are not read here)
    if (blockIdx.x == gridDim.x - 1u)
        for (int i = tid; i < kDictMax; i += kDictNT) blk->count[i] = 0u;
}

}  // namespace adlhip
