// dict_big_kernels.hpp -- u32 keys that take up to 4096 DISTINCT VALUES (categories, codes, dates: thousands of values among millions
// of keys) sorted by counting, like dict_kernels.hpp does for up to 256 values.
//
// Why a second size.  Keys that repeat a few thousand values cannot fit the large sort's slabs (hybrid_kernels.hpp) and end in its
// safety net, whose ordinary work is four LSD passes (64 Mi keys: 0.95 ms).  Equal keys are indistinguishable, so the output of a
// whole-key sort is determined by the counts alone (bit for bit what Tahoe/Algorithm/Sort/RadixSort.cpp:58-104 produces): one read
// and one write do.  The small dictionary's tables (per-wave counters, 1024 slots) stay as they are -- they are the faster ones for
// few values; this one costs a longer build: 64 Ki sampled keys (every one of 4096 equally likely values is then seen: e^-16 to
// miss one), a bitonic sort of the values in LDS, and a hash table of 8192 slots over them.  The count phase keeps the table in LDS
// with one counter per SLOT: one read and one add per key.
//
// Three phases of the net, separated by its grid barriers (net_sort):
//   1. big_dict_sample         every workgroup fetches its share of the samples; big_dict_build: workgroup 0 makes the dictionary
//                              of them (n_values = 0 if they take more than 4096 values)
//   2. big_dict_count_range    every workgroup: look up and count; a key outside the dictionary raises `miss` (LSD passes then)
//   3. big_dict_fill_range     every workgroup scans the counts itself and writes its share of the runs, in place
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_kernels.hpp"
#include "dict_build.hpp"

namespace adlhip {

constexpr int kBigMax = 4096;        // values
constexpr int kBigSlots = 8192;      // hash slots over the values (load <= 1/2), in the build's set of sampled values and in the dictionary
constexpr uint32_t kBigSamples = 65536u;
constexpr int kBigCopies = 16;
// the net tries this dictionary if the sort's first kernel counted at least this many repeats among its 16 x 128 samples: 4096
// equally likely values show 16 (128 - 4096 (1 - e^(-1/32))) = 32 +- 6 (and the small dictionary did not do)
constexpr uint32_t kBigMinRepeats = 18u;

struct BigDictBlock {                // handle-owned device memory, behind the DictBlock
    uint32_t n_values;               // 0: no dictionary
    uint32_t miss;
    uint32_t pad[2];
    uint32_t value[kBigMax];         // ascending, padded with 0xffffffff (which, if it is a value itself, is the last one)
    uint32_t slot_key[kBigSlots];    // hash table over the values: key (0xffffffff = free; the all-ones key never enters) ...
    uint32_t slot_idx[kBigSlots];    // ... -> its index in value[]
    // per value, in kBigCopies copies (workgroup w adds to copy w % kBigCopies, the fill phase sums them): 512 workgroups x 4096
    // atomics on ONE copy's 16 KiB queued on a few L2 channels -- 0.3 of the 0.9 ms that 64 Mi keys of 4096 values took
    uint32_t count[kBigCopies * kBigMax];
};

__device__ __forceinline__ uint32_t big_dict_hash(uint32_t key)   // 13 bits
{
    uint32_t h = key * 0x9E3779B1u;
    h ^= h >> 15;
    return (h * 0x85EBCA6Bu) >> 19;
}

// Probing is by double hashing -- the step is odd, so every slot is reached -- not by the next slot: at load 1/2 linear probing
// builds clusters, and the 64 lanes of a wave wait for the longest of their chains (build and count of 4096 values: 0.19 and 0.39
// ms of workgroup 0's time with linear probing).
__device__ __forceinline__ uint32_t big_dict_stride(uint32_t key)
{
    return ((key * 0xC2B2AE35u) >> 19) | 1u;
}

// where sample k of 65536 is read (n >= 2^20): inside the k-th 65536th of the array, at a scrambled offset (multiply-and-shift)
__device__ __forceinline__ size_t big_sample_index(uint32_t k, uint32_t n)
{
    const unsigned long long lo = (unsigned long long)k * n >> 16, hi = (unsigned long long)(k + 1u) * n >> 16;
    uint32_t h = k * 0x85EBCA6Bu + 0x27D4EB2Fu;
    h ^= h >> 15;
    h *= 0xC2B2AE35u;
    h ^= h >> 13;
    return (size_t)(lo + (((unsigned long long)h * (hi - lo)) >> 32));
}

// Every workgroup: its share of the 64 Ki samples into `samples` (scratch: the sort's partner array, idle until the LSD passes) and of
// the counters' clearing -- one workgroup alone took 80 us to fetch 64 Ki scattered keys.  A grid barrier follows.
template <int NT>
__device__ __forceinline__ void big_dict_sample(const uint32_t* __restrict__ src, uint32_t n, uint32_t* __restrict__ samples,
                                                BigDictBlock* __restrict__ blk)
{
    const uint32_t gtid = blockIdx.x * (uint32_t)NT + threadIdx.x, gn = gridDim.x * (uint32_t)NT;
    for (uint32_t k = gtid; k < kBigSamples; k += gn) samples[k] = src[big_sample_index(k, n)];
    for (uint32_t i = gtid; i < (uint32_t)(kBigCopies * kBigMax); i += gn) blk->count[i] = 0u;
}

// ONE workgroup of NT = 512 threads: the 64 Ki sampled keys -> the dictionary, or n_values = 0.  smem: 48 KiB of the caller's dynamic LDS.
template <int NT>
__device__ __forceinline__ void big_dict_build(const uint32_t* __restrict__ samples, BigDictBlock* __restrict__ blk, unsigned char* smem)
{
    uint32_t* s_tab = reinterpret_cast<uint32_t*>(smem);                 // [kBigSlots] open addressing over the samples (hash set)
    uint32_t* s_val = s_tab + kBigSlots;                                  // [kBigMax] the values, then sorted
    __shared__ uint32_t s_cnt, s_has_max, s_n;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kBigSlots; i += NT) s_tab[i] = 0xffffffffu;
    for (int i = tid; i < kBigMax; i += NT) s_val[i] = 0xffffffffu;
    if (tid == 0) {
        s_cnt = 0u;
        s_has_max = 0u;
        s_n = 0u;
    }
    __syncthreads();
    constexpr int PER = (int)kBigSamples / NT;   // 128
    constexpr int U = 16;
#pragma unroll 1
    for (int i0 = 0; i0 < PER; i0 += U) {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = samples[(i0 + u) * NT + tid];
        if (__hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (uint32_t)kBigMax) continue;   // too many already
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t v = x[u];
            if (v == 0xffffffffu) {
                s_has_max = 1u;
                continue;
            }
            uint32_t h = big_dict_hash(v);
            const uint32_t stride = big_dict_stride(v);
            for (int step = 0; step < kBigSlots; ++step) {
                uint32_t old = __hip_atomic_load(&s_tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // look before the compare-and-swap
                if (old == v) break;
                if (old == 0xffffffffu) {
                    old = atomicCAS(&s_tab[h], 0xffffffffu, v);
                    if (old == 0xffffffffu) {
                        atomicAdd(&s_cnt, 1u);
                        break;
                    }
                    if (old == v) break;
                }
                h = (h + stride) & (uint32_t)(kBigSlots - 1);
                if (step > 64 && __hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (uint32_t)kBigMax) break;
            }
        }
    }
    __syncthreads();
    const uint32_t distinct = s_cnt + s_has_max;
    // whatever the samples say: this sort's miss flag and counters start from zero
    if (tid == 0) blk->miss = 0u;
    if (distinct == 0u || distinct > (uint32_t)kBigMax) {
        if (tid == 0) blk->n_values = 0u;
        return;
    }
    // compact the values (the all-ones key, if present, is among the 0xffffffff that pad s_val: it sorts last by itself)
    for (int i = tid; i < kBigSlots; i += NT) {
        const uint32_t v = s_tab[i];
        if (v != 0xffffffffu) s_val[atomicAdd(&s_n, 1u)] = v;
    }
    __syncthreads();
    // bitonic sort, ascending, of the first m words (m = the power of two that holds the values and one pad; the rest are pads already)
    uint32_t m = 2u;
    while (m < distinct + 1u && m < (uint32_t)kBigMax) m <<= 1;
    for (uint32_t k = 2u; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j >= 1u; j >>= 1) {
            for (uint32_t t = (uint32_t)tid; t < m / 2u; t += (uint32_t)NT) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));   // the lower index of pair t at distance j
                const uint32_t l = i | j;
                const uint32_t a = s_val[i], b = s_val[l];
                const bool up = (i & k) == 0u;
                if ((a > b) == up) {
                    s_val[i] = b;
                    s_val[l] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < kBigMax; i += NT) blk->value[i] = s_val[i];
    // the dictionary's hash table, made in LDS (the set of samples is no longer needed) and copied out
    for (int i = tid; i < kBigSlots; i += NT) s_tab[i] = 0xffffffffu;
    __syncthreads();
    for (uint32_t r = (uint32_t)tid; r < distinct; r += (uint32_t)NT) {
        const uint32_t v = s_val[r];
        if (v != 0xffffffffu) {
            uint32_t h = big_dict_hash(v);
            const uint32_t stride = big_dict_stride(v);
            for (int step = 0; step < kBigSlots; ++step) {
                if (atomicCAS(&s_tab[h], 0xffffffffu, v) == 0xffffffffu) {
                    blk->slot_idx[h] = r;   // (global: slots are owned by the thread that won them)
                    break;
                }
                h = (h + stride) & (uint32_t)(kBigSlots - 1);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < kBigSlots; i += NT) blk->slot_key[i] = s_tab[i];
    if (tid == 0) blk->n_values = distinct;
}

// Every workgroup of the grid: look up and count src[0, n).  The hash slots carry their own counters: a key costs one LDS read (its
// first probe; sixteen keys' first probes go out together) and one LDS add -- looking the rank up first (a second, dependent read)
// or searching the ascending values (thirteen reads, LDS-bound: 0.23-0.54 ms for 64 Mi keys) cost more.  The slot counters are
// added to the value's global counter at the end.  smem: 64 KiB of the caller's dynamic LDS.
template <int NT>
__device__ __forceinline__ void big_dict_count_range(const uint32_t* __restrict__ src, uint32_t n, BigDictBlock* __restrict__ blk,
                                                     uint32_t nv, unsigned char* smem, uint32_t* stamps = nullptr /* diagnostic */)
{
    uint32_t* s_key = reinterpret_cast<uint32_t*>(smem);                                   // [kBigSlots]
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + 4 * kBigSlots);                   // [kBigSlots], one set per workgroup
    __shared__ uint32_t s_max_cnt;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kBigSlots; i += NT) {
        s_key[i] = blk->slot_key[i];
        s_cnt[i] = 0u;
    }
    if (tid == 0) s_max_cnt = 0u;
    const bool has_max = blk->value[nv - 1u] == 0xffffffffu;   // the all-ones key is a value (the last one); it has no slot
    __syncthreads();
    if (stamps && blockIdx.x == 0 && tid == 0) stamps[7] = (uint32_t)wall_clock64();
    bool miss = false;
    uint32_t max_mine = 0u;
    // a lane counts runs of equal keys by itself and adds a run at its end (constant and ordered keys would otherwise queue 64 lanes
    // on one LDS counter, key after key)
    uint32_t cur = 0u, run = 0u;
    auto count = [&](uint32_t slot) {
        if (slot == cur) {
            ++run;
        } else {
            if (run) atomicAdd(&s_cnt[cur], run);
            cur = slot;
            run = 1u;
        }
    };
    auto look = [&](uint32_t key, uint32_t first /* s_key[hash(key)] */) {
        if (key == 0xffffffffu) {
            if (has_max) ++max_mine;
            else miss = true;
            return;
        }
        uint32_t h = big_dict_hash(key);
        const uint32_t stride = big_dict_stride(key);
        uint32_t k = first;
        for (int step = 0; step < kBigSlots; ++step) {
            if (k == key) {
                count(h);
                return;
            }
            if (k == 0xffffffffu) break;
            h = (h + stride) & (uint32_t)(kBigSlots - 1);
            k = s_key[h];
        }
        miss = true;
    };
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const uint32_t nvec = n / 4u;
    const v4u* vsrc = reinterpret_cast<const v4u*>(src);   // sort buffers are 16-byte aligned
    constexpr int U = 4;
    const uint32_t step = gridDim.x * (uint32_t)NT;
    // (a lane that meets a key outside the dictionary stops counting: skewed categories miss by the thousand, every lane soon has
    // its own, and the LSD passes need not wait for a count that is void.  Looking at a shared flag once per round instead -- 65 K
    // wave-loads of ONE address that no cache may hold -- did not pay.)
    for (uint32_t i0 = blockIdx.x * (uint32_t)NT + (uint32_t)tid; i0 < nvec; i0 += (uint32_t)U * step) {
        v4u v[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = (uint64_t)i0 + (uint64_t)u * step;
            act[u] = i < nvec;
            if (act[u]) v[u] = vsrc[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (act[u]) {
                const uint32_t f0 = s_key[big_dict_hash(v[u][0])], f1 = s_key[big_dict_hash(v[u][1])];
                const uint32_t f2 = s_key[big_dict_hash(v[u][2])], f3 = s_key[big_dict_hash(v[u][3])];
                look(v[u][0], f0);
                look(v[u][1], f1);
                look(v[u][2], f2);
                look(v[u][3], f3);
            }
        }
        if (miss) break;
    }
    if (blockIdx.x == 0 && (uint32_t)tid < n - nvec * 4u) {   // the last n % 4 keys
        const uint32_t key = src[nvec * 4u + (uint32_t)tid];
        look(key, s_key[big_dict_hash(key)]);
    }
    if (run) atomicAdd(&s_cnt[cur], run);
    if (max_mine) atomicAdd(&s_max_cnt, max_mine);
    if (stamps && blockIdx.x == 0 && tid == 0) stamps[8] = (uint32_t)wall_clock64();
    const int any_miss = __syncthreads_or(miss);
    if (stamps && blockIdx.x == 0 && tid == 0) stamps[9] = (uint32_t)wall_clock64();
    if (any_miss) {
        if (tid == 0) __hip_atomic_store(&blk->miss, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        uint32_t* mine = blk->count + (blockIdx.x % (uint32_t)kBigCopies) * (uint32_t)kBigMax;
        for (int i = tid; i < kBigSlots; i += NT) {
            const uint32_t c = s_cnt[i];
            if (c) __hip_atomic_fetch_add(&mine[blk->slot_idx[i]], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0 && s_max_cnt) __hip_atomic_fetch_add(&mine[nv - 1u], s_max_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Every workgroup of the grid writes its share [p0, p1) of the runs.  The counts are final (a grid barrier lies between the count
// phase and this) and only ever touched by agent-scope atomics; every workgroup scans them itself.  Returns false (nothing written)
// when they do not add up to n (cannot happen without a miss).  smem: 33 KiB of the caller's dynamic LDS.
template <int NT>
__device__ __forceinline__ bool big_dict_fill_range(uint32_t* __restrict__ data, uint32_t n, BigDictBlock* __restrict__ blk, uint32_t nv,
                                                    unsigned char* smem)
{
    static_assert(kBigMax % NT == 0, "whole stretches of counts per thread");
    constexpr int PER = kBigMax / NT;   // 8
    uint32_t* s_off = reinterpret_cast<uint32_t*>(smem);                            // [kBigMax + 1]
    uint32_t* s_val = reinterpret_cast<uint32_t*>(smem + 4 * (kBigMax + 4));        // [kBigMax]
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 4 * (2 * kBigMax + 4));
    const int tid = (int)threadIdx.x;
    uint32_t c[PER], mine = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t r = (uint32_t)(tid * PER + i);
        c[i] = 0u;
        if (r < nv) {
#pragma unroll
            for (int k = 0; k < kBigCopies; ++k) c[i] += __hip_atomic_load(&blk->count[k * kBigMax + (int)r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        mine += c[i];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan_u32<NT>(mine, s_wsum, &total);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t r = (uint32_t)(tid * PER + i);
        s_off[r] = ex;
        ex += c[i];
        s_val[r] = r < nv ? blk->value[r] : 0xffffffffu;
    }
    if (tid == 0) s_off[kBigMax] = total;
    __syncthreads();
    if (total != n) return false;
    const uint32_t chunk = ((n + gridDim.x - 1u) / gridDim.x + 3u) / 4u * 4u;
    const uint64_t p064 = (uint64_t)blockIdx.x * chunk;
    if (p064 >= n) return true;
    const uint32_t p0 = (uint32_t)p064;
    const uint32_t p1 = p064 + chunk < n ? p0 + chunk : n;
    // value of position p0: the last index whose offset is <= p0 (offsets beyond nv equal n)
    uint32_t lo = 0u, hi = (uint32_t)kBigMax;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_off[mid] <= p0) lo = mid; else hi = mid;
    }
    uint32_t ix = lo;
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    uint32_t p = p0;
    while (p < p1) {
        while (ix + 1u < (uint32_t)kBigMax && s_off[ix + 1u] <= p) ++ix;   // (empty runs)
        const uint32_t end = s_off[ix + 1u] < p1 ? s_off[ix + 1u] : p1;
        const uint32_t v = s_val[ix];
        const uint32_t a0 = (p + 3u) / 4u * 4u;             // first 16-byte boundary inside
        const uint32_t a1 = end / 4u * 4u;
        if (a0 < a1) {
            const v4u vv = {v, v, v, v};
            v4u* out = reinterpret_cast<v4u*>(data);
            for (uint32_t i = a0 / 4u + (uint32_t)tid; i < a1 / 4u; i += NT) out[i] = vv;
            for (uint32_t i = p + (uint32_t)tid; i < a0; i += NT) data[i] = v;
            for (uint32_t i = a1 + (uint32_t)tid; i < end; i += NT) data[i] = v;
        } else {
            for (uint32_t i = p + (uint32_t)tid; i < end; i += NT) data[i] = v;
        }
        p = end;
    }
    return true;
}

}  // namespace adlhip
