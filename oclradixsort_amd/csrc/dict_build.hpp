// dict_build.hpp -- the dictionary of the counting sort for keys that take few distinct values (dict_kernels.hpp has the story):
// its layout and the routine that builds it from sampled keys.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adlhip {

constexpr int kDictMax = 256;       // values the dictionary holds
constexpr int kDictSlots = 1024;    // hash slots over them (load <= 1/4)
constexpr unsigned long long kDictEmpty = ~0ull;
// The net tries the dictionary only if the sort's first kernel (hybrid_kernels.hpp sample_accumulate) counted this many repeats among
// its 16 x 128 samples: keys of 256 equally likely values show 16 (128 - 256 (1 - e^(-1/2))) = 436 +- 20, skewed or fewer values more.
constexpr uint32_t kDictMinRepeats = 256u;

struct DictBlock {                                  // handle-owned device memory
    uint32_t n_values;                              // 0: the sampled keys take more than kDictMax values (no dictionary)
    uint32_t miss;                                  // count kernel: some key is not in the dictionary
    uint32_t pad[2];
    unsigned long long value[kDictMax];             // ascending
    unsigned long long slot_key[kDictSlots];        // hash table: key (kDictEmpty = free) ...
    uint32_t slot_idx[kDictSlots];                  // ... -> its index in value[]
    uint32_t count[kDictMax];                       // per value; cleared by the workgroup that builds the dictionary
};

__device__ __forceinline__ uint32_t dict_hash(unsigned long long v)
{
    v ^= v >> 33;
    v *= 0xff51afd7ed558ccdull;
    v ^= v >> 29;
    return (uint32_t)v & (uint32_t)(kDictSlots - 1);
}

// Built by ONE workgroup of NT threads from the keys it has sampled.  smp[i] = the thread's samples (the all-ones key of the
// sorted width arrives as kDictEmpty, see dict_sample_build); all threads must call.  smem: 18 KiB of the caller's dynamic LDS.
// Returns the number of distinct values if they fit the dictionary, else 0.
template <int PER_THREAD, int NT>
__device__ __forceinline__ uint32_t dict_build(const unsigned long long (&smp)[PER_THREAD], DictBlock* __restrict__ blk, unsigned char* smem)
{
    unsigned long long* s_tab = reinterpret_cast<unsigned long long*>(smem);            // [2048] open addressing over the samples
    unsigned long long* s_val = s_tab + 2048;                                            // [kDictMax]
    __shared__ uint32_t s_cnt, s_maxkey, s_n;
    const int tid = (int)threadIdx.x;
    for (int i = tid; i < 2048; i += NT) s_tab[i] = kDictEmpty;
    if (tid == 0) {
        s_cnt = 0u;
        s_maxkey = 0u;
        s_n = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
        const unsigned long long v = smp[i];
        if (v == kDictEmpty) {   // the one value the table cannot hold as a key
            s_maxkey = 1u;
            continue;
        }
        if (__hip_atomic_load(&s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > (uint32_t)kDictMax) continue;   // too many already
        {   // a wave whose samples agree inserts once (64 compare-and-swaps on one LDS word are served one after the other)
            const unsigned long long v0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
                                          (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
            const unsigned long long same = __ballot(v == v0);
            if (v == v0 && (tid & 63) != (int)__builtin_ctzll(same)) continue;
        }
        uint32_t h = (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> 53);   // 11 bits
        for (int step = 0; step < 2048; ++step) {
            // look before the compare-and-swap: once a value sits in the table every later sample of it ends here with a plain read
            // (few values: 64 lanes' compare-and-swaps on a handful of LDS words were served one after the other, ~50 us of the build)
            unsigned long long old = __hip_atomic_load(&s_tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old == v) break;
            if (old == kDictEmpty) {
                old = atomicCAS(&s_tab[h], kDictEmpty, v);
                if (old == kDictEmpty) {
                    atomicAdd(&s_cnt, 1u);
                    break;
                }
                if (old == v) break;
            }
            h = (h + 1u) & 2047u;
        }
    }
    __syncthreads();
    const uint32_t distinct = s_cnt + s_maxkey;
    // whatever the samples say: this sort's miss flag and counters start from zero
    if (tid == 0) blk->miss = 0u;
    for (int i = tid; i < kDictMax; i += NT) blk->count[i] = 0u;
    if (distinct == 0u || distinct > (uint32_t)kDictMax) {
        if (tid == 0) blk->n_values = 0u;
        return 0u;
    }
    // compact the values, then rank them (at most 256: every value counts the smaller ones)
    for (int i = tid; i < 2048; i += NT) {
        const unsigned long long v = s_tab[i];
        if (v != kDictEmpty) s_val[atomicAdd(&s_n, 1u)] = v;
    }
    __syncthreads();
    if (tid == 0 && s_maxkey) s_val[s_n] = kDictEmpty;   // ranks last by itself
    for (int i = tid; i < kDictSlots; i += NT) {
        blk->slot_key[i] = kDictEmpty;
        blk->slot_idx[i] = 0u;
    }
    __syncthreads();
    if ((uint32_t)tid < distinct) {
        const unsigned long long v = s_val[tid];
        uint32_t r = 0u;
        for (uint32_t j = 0; j < distinct; ++j) r += s_val[j] < v ? 1u : 0u;
        blk->value[r] = v;
        if (v != kDictEmpty) {   // (the all-ones key is looked up by comparison, see dict_lookup)
            uint32_t h = dict_hash(v);
            for (int step = 0; step < kDictSlots; ++step) {
                const unsigned long long old = atomicCAS(&blk->slot_key[h], kDictEmpty, v);   // global, agent scope
                if (old == kDictEmpty) {
                    blk->slot_idx[h] = r;
                    break;
                }
                h = (h + 1u) & (uint32_t)(kDictSlots - 1);
            }
        }
    }
    if (tid == 0) blk->n_values = distinct;
    return distinct;
}

}  // namespace adlhip
