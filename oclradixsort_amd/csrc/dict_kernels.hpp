// dict_kernels.hpp -- keys that take FEW DISTINCT VALUES (flags, categories, quantised data: at most 256 values among millions of
// keys) are sorted by counting: look every key up in a dictionary of the values, count, and write the runs.
//
// Why it exists.  Pprims::radixSort (Tahoe/ParallelPrimitives/Pprims.cpp:304-406) sorts such keys like any others: sortBits / 4
// passes over the data.  The large sort of this back-end (hybrid_kernels.hpp) gives every bucket and segment the same room, and a
// value with a million copies outgrows any slab: such keys end in its safety net (net_sort, hybrid_kernels.hpp), whose ordinary
// work is four LSD passes (about 1 ms for 64 Mi keys).  Counting needs one read and one write of the array.  Equal keys are
// indistinguishable, so the output of a sort of whole keys is determined by the counts alone; it is bit for bit what the
// reference's CPU sort (Tahoe/Algorithm/Sort/RadixSort.cpp:58-104) produces.  (Equal keys of {key, value} pairs are not
// interchangeable: pairs with few-valued keys take ONE stable pass on the key's rank in the same dictionary -- the last part of
// this file; sorts on part of the key go straight to the LSD passes.)
//
// Round 4: the three steps are phases of the net itself, separated by its grid barriers (round 3: three launches chosen by a
// host-side hint that a probe launch had left behind):
//   1. dict_sample_build   workgroup 0 samples 16 Ki keys; if they take at most 256 values it writes the dictionary -- the values in
//      ascending order and a 1024-slot hash table over them
//   2. dict_count_range    every key is looked up (hash, one or two LDS reads) and counted; a key that is not in the dictionary
//      raises `miss` (the sample cannot see a value that occurs once in a million) and the net goes on to its LSD passes
//   3. dict_fill_range     every workgroup scans the counts itself and writes its share of the runs, in place
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_kernels.hpp"
#include "dict_build.hpp"

namespace adlhip {

__device__ __forceinline__ size_t probe_sample_index(uint32_t k, uint32_t n);   // hybrid_kernels.hpp

// index of v in the dictionary, or 0xffffffff.  s_key / s_idx: the hash table in LDS; n_values for the all-ones key.
// K = the key type of the LDS copy: the element type itself (u32 keys: 32-bit slots -- a 64-bit LDS read of ONE address by all
// 64 lanes is not broadcast but served lane after lane); free slots hold K(~0), which for u32 keys cannot be told from the key
// 0xffffffff, so that key never enters the table (dict_sample_build) and `has_max` says whether it is a value (it is then the
// last one: values are ascending).
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

// ONE workgroup (NT = 512 threads): 16 Ki sampled keys (n >= 16384) -> the dictionary, or n_values = 0.  The all-ones key of E's
// width is handed to dict_build as kDictEmpty, the one value its tables treat apart (zero-extended, a u32 key 0xffffffff used to
// enter the hash table and, narrowed to 32 bits in the count phase's LDS copy, read as a FREE slot there: every probe chain
// through it broke, a spurious miss).
template <typename E, int NT>
__device__ __forceinline__ void dict_sample_build(const E* __restrict__ src, uint32_t n, DictBlock* __restrict__ blk, unsigned char* smem)
{
    constexpr int PER = 16384 / NT;
    unsigned long long v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const E x = src[probe_sample_index((uint32_t)((int)threadIdx.x * PER + i), n)];
        v[i] = x == (E)~(E)0 ? kDictEmpty : (unsigned long long)x;
    }
    dict_build<PER, NT>(v, blk, smem);
}

// Every workgroup of the grid: look up and count src[0, n).  smem: 14 KiB + NT / 64 KiB of the caller's dynamic LDS.
template <typename E, int NT>
__device__ __forceinline__ void dict_count_range(const E* __restrict__ src, uint32_t n, DictBlock* __restrict__ blk, uint32_t nv,
                                                 unsigned char* smem)
{
    E* s_key = reinterpret_cast<E*>(smem);                                                  // [kDictSlots]
    uint16_t* s_idx = reinterpret_cast<uint16_t*>(smem + sizeof(E) * kDictSlots);           // [kDictSlots]
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(smem + (sizeof(E) + 2) * kDictSlots);     // [NT / 64][kDictMax]
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kDictSlots; i += NT) {
        s_key[i] = (E)blk->slot_key[i];   // (free slots: all ones in either width)
        s_idx[i] = (uint16_t)blk->slot_idx[i];
    }
    for (int i = tid; i < (NT / 64) * kDictMax; i += NT) s_cnt[i] = 0u;
    const bool has_max = blk->value[nv - 1u] == kDictEmpty;   // the all-ones key is a value
    __syncthreads();
    uint32_t* my = s_cnt + (tid >> 6) * kDictMax;
    bool miss = false;
    constexpr int VEC = 16 / (int)sizeof(E);
    struct alignas(16) Vec { E v[VEC]; };
    const uint32_t nvec = n / VEC;
    const Vec* vsrc = reinterpret_cast<const Vec*>(src);   // sort buffers are 16-byte aligned
    // A wave whose 64 keys are the same value (constant stretches, one dominant value) looks it up and counts it ONCE, in one lane:
    // 64 lanes on one LDS address are served one after the other (all-equal keys: 0.24 ms in this phase instead of 0.07).
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
    // whole waves stay together (the vote above); U 16-byte loads in flight per lane
    constexpr int U = 8;
    const uint32_t step = gridDim.x * (uint32_t)NT;
    for (uint32_t i0 = blockIdx.x * (uint32_t)NT; i0 < nvec; i0 += (uint32_t)U * step) {
        Vec v[U];
        bool act[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint64_t i = (uint64_t)i0 + (uint64_t)u * step + (uint32_t)tid;
            act[u] = i < nvec;
            if (act[u]) v[u] = vsrc[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if ((uint64_t)i0 + (uint64_t)u * step < nvec) {   // (uniform over the workgroup)
#pragma unroll
                for (int k = 0; k < VEC; ++k) one(act[u] ? v[u].v[k] : E(0), act[u]);
            }
        }
    }
    if (blockIdx.x == 0 && tid < 64) {   // the last n % VEC keys
        const bool active = (uint32_t)tid < n - nvec * VEC;
        one(active ? src[nvec * VEC + (uint32_t)tid] : E(0), active);
    }
    const int any_miss = __syncthreads_or(miss);
    if (any_miss) {
        if (tid == 0) __hip_atomic_store(&blk->miss, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if ((uint32_t)tid < nv) {
        uint32_t c = 0u;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) c += s_cnt[w * kDictMax + tid];
        if (c) __hip_atomic_fetch_add(&blk->count[tid], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Every workgroup of the grid writes its share [p0, p1) of the runs.  The counts are final (a grid barrier lies between the count
// phase and this) and only ever touched by agent-scope atomics; every workgroup scans them itself.  Returns false (nothing written)
// when they do not add up to n (cannot happen without a miss).
template <typename E, int NT>
__device__ __forceinline__ bool dict_fill_range(E* __restrict__ data, uint32_t n, DictBlock* __restrict__ blk, uint32_t nv, unsigned char* smem)
{
    uint32_t* s_off = reinterpret_cast<uint32_t*>(smem);                                    // [kDictMax + 1]
    unsigned long long* s_val = reinterpret_cast<unsigned long long*>(smem + 2048);          // [kDictMax]
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + 2048 + 8 * kDictMax);
    const int tid = (int)threadIdx.x;
    const uint32_t c = (uint32_t)tid < nv ? __hip_atomic_load(&blk->count[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan_u32<NT>(c, s_wsum, &total);
    if ((uint32_t)tid < nv) {
        s_off[tid] = ex;
        s_val[tid] = blk->value[tid];
    }
    if (tid == 0) s_off[nv] = total;
    __syncthreads();
    if (total != n) return false;
    constexpr uint32_t VEC = 16u / (uint32_t)sizeof(E);
    const uint32_t chunk = ((n + gridDim.x - 1u) / gridDim.x + VEC - 1u) / VEC * VEC;
    const uint64_t p064 = (uint64_t)blockIdx.x * chunk;
    if (p064 >= n) return true;
    const uint32_t p0 = (uint32_t)p064;
    const uint32_t p1 = p064 + chunk < n ? p0 + chunk : n;
    // value of position p0: the last index whose offset is <= p0
    uint32_t lo = 0u, hi = nv;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_off[mid] <= p0) lo = mid; else hi = mid;
    }
    uint32_t ix = lo;
    // the chunk in stretches of one value each (nearly always one stretch: a run is n / 256 keys and more)
    uint32_t p = p0;
    while (p < p1) {
        while (ix + 1u < nv && s_off[ix + 1u] <= p) ++ix;   // (empty runs)
        const uint32_t end = s_off[ix + 1u] < p1 ? s_off[ix + 1u] : p1;
        const E v = (E)s_val[ix];   // (kDictEmpty narrows to the all-ones key of E's width)
        struct alignas(16) Vec { E v[VEC]; };
        const uint32_t a0 = (p + VEC - 1u) / VEC * VEC;             // first 16-byte boundary inside
        const uint32_t a1 = end / VEC * VEC;
        if (a0 < a1) {
            Vec vv;
#pragma unroll
            for (uint32_t k = 0; k < VEC; ++k) vv.v[k] = v;
            Vec* out = reinterpret_cast<Vec*>(data);
            for (uint32_t i = a0 / VEC + (uint32_t)tid; i < a1 / VEC; i += NT) out[i] = vv;
            for (uint32_t i = p + (uint32_t)tid; i < a0; i += NT) data[i] = v;
            for (uint32_t i = a1 + (uint32_t)tid; i < end; i += NT) data[i] = v;
        } else {
            for (uint32_t i = p + (uint32_t)tid; i < end; i += NT) data[i] = v;
        }
        p = end;
    }
    return true;
}


// ------------------------------------------------------------------------------------------
// {key, value} pairs whose KEYS take few distinct values (round 4).  Equal keys of pairs are not interchangeable, so counting alone
// does not sort them -- but ONE stable pass on the key's rank in the dictionary does, where the LSD passes need four: the net
// (hybrid_kernels.hpp coop_dict_pair_sort) counts ranks while it copies the pairs -- as {rank, value} -- to the scratch array, then
// scatters them back with the ordinary tile body sorting on bits [0, 8); DictPairIO below puts the keys back in the stores.
// The rank comes from a 4096-slot hash table over the (at most 256) values (load 1/16: the first probe nearly always ends the
// search -- with 512 slots the 64 lanes of a wave waited for the longest of their probe chains, 64 Mi pairs of 256 values 0.9-1.2
// ms); a slot holds key and rank together (ONE 8-byte LDS read per probe).  32 KiB, in the tile's element area: only the count
// phase looks ranks up, and the tile is not in use then.  The values themselves, for the way back, take 1 KiB of static LDS.
// ------------------------------------------------------------------------------------------
constexpr int kPairSlots = 4096;
constexpr unsigned long long kPairFree = 0x00000000ffffffffull;   // slot = rank << 32 | key; the all-ones key never enters the table
__device__ __forceinline__ uint32_t pair_dict_hash(uint32_t key)   // 12 bits; two rounds: values are often multiples or masks of each other
{
    uint32_t h = key * 0x9E3779B1u;
    h ^= h >> 15;
    return (h * 0x85EBCA6Bu) >> 20;
}
// what the first probe (slot `s0` = s_slot[pair_dict_hash(key)], read by the caller: it reads the slots of all its keys first and
// looks at them afterwards, so that the LDS latencies overlap) leaves to do.  Returns the rank of `key`, or 0xffffffff.
// max_rank: rank of the all-ones key if it is a value, else 0xffffffff.
__device__ __forceinline__ uint32_t pair_dict_rank(uint32_t key, unsigned long long s0, const unsigned long long* __restrict__ s_slot,
                                                   uint32_t max_rank)
{
    if (key == 0xffffffffu) return max_rank;
    if ((uint32_t)s0 == key) return (uint32_t)(s0 >> 32);
    if ((uint32_t)s0 == 0xffffffffu) return 0xffffffffu;
    uint32_t h = (pair_dict_hash(key) + 1u) & (uint32_t)(kPairSlots - 1);
    for (int step = 1; step < kPairSlots; ++step) {
        const unsigned long long s = s_slot[h];
        if ((uint32_t)s == key) return (uint32_t)(s >> 32);
        if ((uint32_t)s == 0xffffffffu) return 0xffffffffu;
        h = (h + 1u) & (uint32_t)(kPairSlots - 1);
    }
    return 0xffffffffu;
}
// all threads of a workgroup (NT >= 256): the tables from the dictionary block (s_val [256]: the values in ascending order, padded
// with 0xffffffff); returns max_rank; barriers inside and at the end
template <int NT>
__device__ __forceinline__ uint32_t pair_dict_load(const DictBlock* __restrict__ blk, uint32_t nv, uint32_t* s_val, unsigned long long* s_slot)
{
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < kPairSlots; i += NT) s_slot[i] = kPairFree;
    const unsigned long long mine = (uint32_t)tid < nv ? blk->value[tid] : kDictEmpty;
    if (tid < 256) s_val[tid] = (uint32_t)mine;   // (kDictEmpty narrows to the all-ones key)
    __syncthreads();
    if ((uint32_t)tid < nv && mine != kDictEmpty) {
        uint32_t h = pair_dict_hash((uint32_t)mine);
        for (int step = 0; step < kPairSlots; ++step) {
            if (atomicCAS(&s_slot[h], kPairFree, ((unsigned long long)(uint32_t)tid << 32) | (uint32_t)mine) == kPairFree) break;
            h = (h + 1u) & (uint32_t)(kPairSlots - 1);
        }
    }
    const uint32_t max_rank = blk->value[nv - 1u] == kDictEmpty ? nv - 1u : 0xffffffffu;   // (ascending: the all-ones key is the last)
    __syncthreads();
    return max_rank;
}

// ONE workgroup: 16 Ki sampled KEYS of pairs (n >= 16384) -> the dictionary, or n_values = 0
template <int NT>
__device__ __forceinline__ void dict_sample_build_pair_keys(const uint64_t* __restrict__ src, uint32_t n, DictBlock* __restrict__ blk,
                                                            unsigned char* smem)
{
    constexpr int PER = 16384 / NT;
    unsigned long long v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t x = (uint32_t)src[probe_sample_index((uint32_t)((int)threadIdx.x * PER + i), n)];
        v[i] = x == 0xffffffffu ? kDictEmpty : (unsigned long long)x;
    }
    dict_build<PER, NT>(v, blk, smem);
}

// AoS pairs that hold the RANK of their key where the key was (the count phase of coop_dict_pair_sort wrote them so): loads are
// plain, stores translate rank -> key (radix_kernels.hpp AosIO is the model).  (Looking the rank up in the loads instead -- sixteen
// unrolled probe loops in the tile body -- spilled registers: 64 Mi pairs of 256 values 2.3 ms.)
struct DictPairIO {
    typedef uint64_t elem_t;
    const uint64_t* src;
    uint64_t* dst;
    const uint32_t* s_val;   // LDS [256]: the value of rank r
    struct Cursor {
        const uint64_t* p;
        __device__ __forceinline__ uint64_t at(int off) const { return load_once(p + off); }
    };
    __device__ __forceinline__ Cursor cursor(size_t base) const { return Cursor{src + base}; }
    __device__ __forceinline__ void store(size_t i, uint64_t v) const { dst[i] = (v & 0xffffffff00000000ull) | s_val[(uint32_t)v & 255u]; }
    static constexpr uint32_t kStoreScale = 8u;
    struct Dst {
        __amdgpu_buffer_rsrc_t r;
        const uint32_t* s_val;
    };
    __device__ __forceinline__ Dst make_dst(uint32_t n) const
    {
        return Dst{__builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(n * 8u), 0x00020000), s_val};
    }
    static __device__ __forceinline__ void store_at(const Dst& d, uint32_t byte_off, uint64_t v)
    {
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t t = {d.s_val[(uint32_t)v & 255u], (uint32_t)(v >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(t, d.r, (int)byte_off, 0, 0);
    }
};

}  // namespace adlhip
