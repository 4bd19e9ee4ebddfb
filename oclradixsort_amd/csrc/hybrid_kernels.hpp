// hybrid_kernels.hpp -- MSD passes + LDS-resident finish: the segment sort primitive (adlhip_segment_sort), the mid-size
// sort (16 Ki .. 2 Mi keys) and the large sort (2 Mi .. 1088 Mi keys, u64 keys, pairs) built on it.
//
// The LSD sort of Pprims::radixSort (Tahoe/ParallelPrimitives/Pprims.cpp:304-406) moves every element through
// global memory once per digit and costs one to three dependent kernel launches per digit.  Between 16 Ki and
// 2 Mi keys -- every size of the reference's own test, UnitTest/main.cpp:105 -- those launches, not the bytes, are
// the cost.  One MSD pass on the most significant byte that varies + one kernel that finishes every bucket in
// LDS need three launches in all.  The result is the array any stable sort produces (total order + stability =>
// unique output), so parity with the reference's CPU sort (Tahoe/Algorithm/Sort/RadixSort.cpp:10-104) is unaffected.
//
// The same idea at full size is the large sort further down: two MSD passes of 8 bits into slabs + a finish on the 16 bits
// below.  (A first attempt -- stable passes of 7 bits with histograms, finish on 18 bits, profiles/r2_segment_sort_*.txt --
// lost to the per-digit passes: 250-320 us for the finish alone.  What made it pay: no histogram and no look-back at all
// for keys (atomic cursors into slabs), segments small enough for ONE WAVE each, and a finish without per-item predicates.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_kernels.hpp"
#include "onesweep_kernels.hpp"
#include "dict_kernels.hpp"
#include "dict_big_kernels.hpp"

namespace adlhip {

// ------------------------------------------------------------------------------------------
// The mid-size sort's safety net: the whole 4 x 8-bit LSD sort with grid-wide barriers between its phases, run by the
// workgroups of pass 3 (segment_sort_kernel) INSTEAD of their segments when mid_prep_kernel found a bucket that does not
// fit the LDS tile (MidDyn::mode = 1: clustered / low-entropy keys).  (As a launch of its own it cost 4.6 us per sort
// even when it had nothing to do: 256 workgroups that load one word and leave.)
// A skewed input therefore costs about 2-3x the per-digit passes instead of one workgroup sorting a huge bucket
// alone (measured before this existed: 1.5-2.7 ms for 1 Mi clustered keys, profiles/r2_mid_size_distributions.txt),
// and a friendly one pays nothing.  Per pass: every workgroup counts the digits of its run of tiles ->
// barrier -> workgroup d scans row d of the bucket-major table (the reference's table layout,
// RadixSort32Kernels.cl:233) -> barrier -> every workgroup scatters its tiles with a per-digit carry -> barrier.
// All workgroups are resident at once (at most one per CU is needed and each takes ~1/8 of a CU's LDS); every
// spin is bounded and raises the device fault word when it gives up.
// ------------------------------------------------------------------------------------------
// data_fence = false: the barrier orders only what went through agent-scope atomics (write-through stores that are complete at the
// s_waitcnt, loads that pass every cache): the digit table and the totals of the net's count and scan phases.  The two fences --
// write the XCD's L2 back, then drop what it holds -- cost ~10 us a barrier (the mid-size net at 1 Mi keys: twelve barriers, 200 us)
// and are needed only where ordinary stores must become visible: behind a scatter phase.
__device__ __forceinline__ bool grid_barrier(uint32_t* counter, uint32_t& target, uint32_t wgs, uint32_t* fault, bool data_fence = true)
{
    __shared__ uint32_t ok;
    if (!data_fence) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every thread's write-through stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        if (data_fence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // this workgroup's stores leave the XCD's L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        target += wgs;
        uint32_t spins = 0u, good = 1u;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 24)) { good = 0u; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (!good) raise_fault(fault, 0x80000u);
        if (data_fence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop what this CU's L1 holds of other workgroups' data
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ok = good;
    }
    __syncthreads();
    return ok != 0u;
}

// The large sort's words beside its overflow flag (flag = d_msd2 + 8192 + 65536; adlhip.hip): flag[12] = samples of the sort's
// first kernel that repeat an earlier one of their wave (sample_accumulate), flag[14] = how many of them mean "these keys cannot
// fit the slabs" (the host's sample_dup_threshold, passed on by the first kernel).  Every later kernel of the sort asks this at its
// first instruction: set = the passes have nothing to do, the offsets kernel runs the net.
constexpr int kSampleRepeatsWord = 12;
constexpr int kSampleThresholdWord = 14;
__device__ __forceinline__ uint32_t large_sort_gave_up(const uint32_t* flag)
{
    const uint32_t over = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t reps = __hip_atomic_load(flag + kSampleRepeatsWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t thr = __hip_atomic_load(flag + kSampleThresholdWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return over | (reps >= thr ? 1u : 0u);
}

struct MidCoop {          // what the safety net needs beside the two arrays (table == nullptr: there is none)
    uint32_t* table;      // [256][gridDim.x] bucket-major digit table
    uint32_t* totals;     // [256]
    uint32_t* bar;        // grid-barrier counter, zero on entry (mid_prep_kernel)
    uint32_t n;
};

// The net's count phase: the digits (key >> sb) & dm of src[e0, e1) into the calling wave's 256 LDS counters.  e0 is a multiple
// of a tile, so the run starts on a 16-byte boundary: 16-byte loads, four in flight per lane while the four before them are
// counted (round 3's loop read one key per lane and iteration and waited for it: 0.4-0.5 ms of a 0.7-ms pass at 64 Mi keys).
// A round whose keys all share one digit -- constant, sorted, few-valued input -- is added by one lane (64 lanes on ONE LDS
// counter are served one after the other).
template <typename E, int NT>
__device__ __forceinline__ void coop_count_range(const E* __restrict__ src, uint32_t e0, uint32_t e1, uint32_t* my_hist, int sb, uint32_t dm)
{
    constexpr int VEC = 16 / (int)sizeof(E);
    struct alignas(16) Vec { E v[VEC]; };
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const int tid = (int)threadIdx.x;
    if (e1 <= e0) return;
    const v4u* vraw = reinterpret_cast<const v4u*>(src + e0);
    auto vld = [&](uint32_t at) -> Vec {
        const v4u r = vraw[at];
        Vec o;
        __builtin_memcpy(&o, &r, 16);
        return o;
    };
    auto dig = [&](E x) -> uint32_t { return (uint32_t)(x >> sb) & dm; };
    auto bump4 = [&](const Vec& a, const Vec& b, const Vec& c, const Vec& d4) {
        const uint32_t da = dig(a.v[0]);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)da);
        if (__all(da == d0)) {
            bool same = true;
#pragma unroll
            for (int k = 0; k < VEC; ++k) same = same && dig(a.v[k]) == d0 && dig(b.v[k]) == d0 && dig(c.v[k]) == d0 && dig(d4.v[k]) == d0;
            if (__all(same)) {
                const uint64_t act = __ballot(true);
                if (mbcnt64(act) == 0u) atomicAdd(&my_hist[d0], (uint32_t)(4 * VEC * __popcll(act)));
                return;
            }
        }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            atomicAdd(&my_hist[dig(a.v[k])], 1u);
            atomicAdd(&my_hist[dig(b.v[k])], 1u);
            atomicAdd(&my_hist[dig(c.v[k])], 1u);
            atomicAdd(&my_hist[dig(d4.v[k])], 1u);
        }
    };
    const uint32_t nvec = (e1 - e0) / VEC;
    uint32_t i = (uint32_t)tid;
    if (i + 3u * NT < nvec) {
        Vec a = vld(i), b = vld(i + NT), c = vld(i + 2 * NT), d4 = vld(i + 3 * NT);
        i += 4u * NT;
        for (; i + 3u * NT < nvec; i += 4u * NT) {
            const Vec na = vld(i), nb = vld(i + NT), nc = vld(i + 2 * NT), nd = vld(i + 3 * NT);
            bump4(a, b, c, d4);
            a = na; b = nb; c = nc; d4 = nd;
        }
        bump4(a, b, c, d4);
    }
    for (; i < nvec; i += NT) {
        const Vec a = vld(i);
#pragma unroll
        for (int k = 0; k < VEC; ++k) atomicAdd(&my_hist[dig(a.v[k])], 1u);
    }
    for (uint32_t s = e0 + nvec * VEC + (uint32_t)tid; s < e1; s += NT) atomicAdd(&my_hist[dig(src[s])], 1u);
}

template <typename E, int NT, int K, int RANK = 1>
__device__ __forceinline__ void coop_lsd_sort(E* data, E* tmp, uint32_t n, uint32_t* __restrict__ table, uint32_t* __restrict__ totals,
                                              uint32_t* bar, uint32_t* fault, unsigned char* smem, int key_bits = 32,
                                              uint32_t target0 = 0u /* what the barrier counter has reached on entry */,
                                              uint32_t wgs_arg = 0u /* workgroups that take part (the first ones of the grid); 0 = all */)
{
    using C = TileCfg<E, 8, NT, K>;
    constexpr int NW = NT / 64;
    static_assert(NT >= 256, "one thread per digit");
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);   // [NW][256]
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + C::OFF_WSUM);
    const int tid = (int)threadIdx.x;
    const int w = tid >> 6;
    const uint32_t wg = blockIdx.x, wgs = wgs_arg ? wgs_arg : gridDim.x;
    const uint32_t tiles = (n + (uint32_t)C::TILE - 1u) / (uint32_t)C::TILE;
    const uint32_t per = (tiles + wgs - 1u) / wgs;
    const uint32_t t0 = wg * per < tiles ? wg * per : tiles;
    const uint32_t t1 = t0 + per < tiles ? t0 + per : tiles;
    const uint32_t e0 = t0 * (uint32_t)C::TILE;
    const uint32_t e1 = (uint64_t)t1 * C::TILE < n ? t1 * (uint32_t)C::TILE : n;
    uint32_t target = target0;   // the barrier counter is zero on entry unless the caller has used it already
    E* src = data;
    E* dst = tmp;
    // the digit table and the totals cross workgroups (and XCDs) through agent-scope atomics: no cache holds them, so the barriers of
    // the count and scan phases need no fence (grid_barrier, data_fence = false)
    auto tab_ld = [](const uint32_t* p) -> uint32_t { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto tab_st = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // key_bits: 32 (u32 keys, pairs), 64 (u64 keys), or the sortBits of a partial sort (a multiple of 4: the last digit is then 4
    // bits wide, and an odd number of passes leaves the result in tmp -- copied back, Pprims.cpp:400-403)
    for (int sb = 0; sb < key_bits; sb += 8) {
        const bool narrow = key_bits - sb < 8;
        const uint32_t dm = narrow ? 15u : 255u;
        // ---- count the digits of this workgroup's run of tiles --------------------------------------------------------
        for (int i = tid; i < NW * 256; i += NT) hist[i] = 0u;
        __syncthreads();
        coop_count_range<E, NT>(src, e0, e1, hist + w * 256, sb, dm);
        __syncthreads();
        if (tid < 256) {
            uint32_t c = 0u;
            for (int i = 0; i < NW; ++i) c += hist[i * 256 + tid];
            tab_st(&table[(size_t)tid * wgs + wg], c);
        }
        if (!grid_barrier(bar, target, wgs, fault, false)) return;
        // ---- workgroup d scans row d (rows d, d + wgs, ... when there are fewer workgroups than digits) ---------------
        for (uint32_t d = wg; d < 256u; d += wgs) {
            uint32_t carry = 0u;
            for (uint32_t base = 0; base < wgs; base += (uint32_t)NT) {
                const uint32_t j = base + (uint32_t)tid;
                const uint32_t v = j < wgs ? tab_ld(&table[(size_t)d * wgs + j]) : 0u;
                uint32_t tot;
                const uint32_t ex = block_excl_scan_u32<NT>(v, s_wsum, &tot);
                if (j < wgs) tab_st(&table[(size_t)d * wgs + j], carry + ex);
                carry += tot;
            }
            if (tid == 0) tab_st(&totals[d], carry);
        }
        if (!grid_barrier(bar, target, wgs, fault, false)) return;
        // ---- scatter this workgroup's tiles, carrying per-digit offsets from tile to tile ------------------------------
        {
            const uint32_t tot_d = tid < 256 ? tab_ld(&totals[tid]) : 0u;
            // every key has the same digit here (a constant byte: small ranges, few-valued keys): the pass would move nothing
            // (the next pass's count phase writes this workgroup's own table column only, and nobody rewrites the totals before
            // the next barrier, so the barrier that ends a pass is not needed either)
            if (__syncthreads_or(tid < 256 && tot_d == n)) continue;
            const uint32_t base_d = block_excl_scan_u32<NT>(tot_d, s_wsum, nullptr);
            uint32_t carry = tid < 256 ? base_d + tab_ld(&table[(size_t)tid * wgs + wg]) : 0u;
            const AosIO<E> io{src, dst};
            for (uint32_t t = t0; t < t1; ++t) {
                const uint32_t tb = t * (uint32_t)C::TILE;
                const uint32_t left = n - tb;
                const uint32_t valid = left < (uint32_t)C::TILE ? left : (uint32_t)C::TILE;
                if (narrow)
                    sort_scatter_tile<AosIO<E>, 4, NT, K, RANK>(io, tb, valid, n, sb, smem,
                                                             [&](int, uint32_t c) { const uint32_t g = carry; carry += c; return g; });
                else
                    sort_scatter_tile<AosIO<E>, 8, NT, K, RANK>(io, tb, valid, n, sb, smem,
                                                             [&](int, uint32_t c) { const uint32_t g = carry; carry += c; return g; });
            }
        }
        if (!grid_barrier(bar, target, wgs, fault)) return;
        E* t = src; src = dst; dst = t;
    }
    if (src != data) {   // odd number of passes: the result sits in tmp (the barrier above made it visible)
        for (size_t i = (size_t)wg * NT + (size_t)tid; i < n; i += (size_t)wgs * NT) data[i] = src[i];
    }
}

// Scratch of the net's look-back passes (coop_onesweep_sort), carved by the host out of the first slab area -- void by the time the
// net runs.  tables == nullptr: none (sorts on part of the key, tiny or huge inputs): the net then runs coop_lsd_sort.
struct OsNet {
    u32x4* ctrl;             // chain tickets of all passes (kTicketVecs vectors)
    PassTable* tables;       // [passes]
    uint32_t* joint;         // joint histograms of all passes
    uint32_t* part;          // [hist_wgs][total_bins] partial histograms
    uint32_t* status;        // [passes][rows][256] tile status words
    uint32_t rows;           // status rows of one pass
    uint32_t hist_wgs, per_wg, slice0;
};

// The net's LSD passes as the one-sweep path runs them (onesweep_kernels.hpp), inside ONE kernel: the resident workgroups take the
// histogram workgroups, then the tiles of every pass, in turns; grid barriers stand where the kernel boundaries were.  Every key
// is read once for all histograms and once per pass -- coop_lsd_sort reads it twice per pass and pays three barriers per pass
// (64 Mi u32 keys: ~0.7 ms against 1.1).  P whole 8-bit passes (P even: the result ends in `data`).
// base_bit: the passes sort bits [base_bit, base_bit + 8 P) -- u64 keys take two rounds of four passes (an LSD sort with two
// 32-bit digits, each round stable), because the joint histograms of eight passes do not fit the tile's LDS.
template <typename E, int NT, int K, int P, int RANK>
__device__ __forceinline__ bool coop_onesweep_sort(E* data, E* tmp, uint32_t n, const OsNet& os, uint32_t* bar, uint32_t& target,
                                                   uint32_t* fault, unsigned char* smem, int base_bit = 0)
{
    static_assert(P % 2 == 0, "an even number of passes ends in the caller's array");
    using C = TileCfg<E, 8, NT, K>;
    const uint32_t wgs = gridDim.x;
    PassDesc desc;
    desc.num_passes = P;
#pragma unroll
    for (int p = 0; p < kMaxPasses; ++p) {
        desc.start_bit[p] = (uint8_t)(p < P ? base_bit + 8 * p : 0);
        desc.nbits[p] = (uint8_t)(p < P ? 8 : 0);
    }
    constexpr uint32_t total_bins = (uint32_t)P * ((uint32_t)kChains << 8);
    const uint32_t status_bytes = os.rows * 256u * 4u;
    const size_t status_vecs = (size_t)P * os.rows * 256u * 4u / 16u;
    // ---- joint histograms of all passes (and: tickets and status rows zeroed) --------------------------------------------
    for (uint32_t v = blockIdx.x; v < os.hist_wgs; v += wgs) {
        onesweep_hist_body<E, P, NT>(data, os.part, n, os.per_wg, os.slice0, desc, total_bins, os.ctrl, reinterpret_cast<u32x4*>(os.status),
                                     status_vecs, fault, v, os.hist_wgs, smem);
        __syncthreads();
    }
    if (!grid_barrier(bar, target, wgs, fault)) return false;
    for (uint32_t v = blockIdx.x; v < (total_bins + 255u) / 256u; v += wgs)
        onesweep_hist_reduce_body<NT>(os.part, os.joint, os.hist_wgs, total_bins, v);
    if (!grid_barrier(bar, target, wgs, fault)) return false;
    for (uint32_t v = blockIdx.x; v < (uint32_t)P; v += wgs) onesweep_tables_body<NT>(os.joint, os.tables, desc, (uint32_t)C::TILE, (int)v);
    if (!grid_barrier(bar, target, wgs, fault)) return false;
    // a pass whose digit is the same for every key moves nothing: skipped -- except that the number of passes run must stay even
    // (the result belongs in `data`): one constant pass, a plain stable copy, runs then
    uint32_t run_mask = 0u, constant_mask = 0u;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (os.tables[p].pad[0]) constant_mask |= 1u << p;
        else run_mask |= 1u << p;
    }
    if (__builtin_popcount(run_mask) & 1) run_mask |= constant_mask & (0u - constant_mask);   // (odd => some pass is constant: P is even)
    // ---- the passes ---------------------------------------------------------------------------------------------------------
    E* src = data;
    E* dst = tmp;
#pragma unroll 1
    for (int p = 0; p < P; ++p) {
        if (!((run_mask >> p) & 1u)) continue;
        const AosIO<E> io{src, dst};
        uint32_t* st = os.status + (size_t)p * os.rows * 256u;
        uint32_t* tk = reinterpret_cast<uint32_t*>(os.ctrl) + (size_t)p * kChains * kTicketStride;
        while (onesweep_tile<AosIO<E>, 8, NT, K, RANK>(io, os.tables + p, st, status_bytes, tk, fault, n, base_bit + 8 * p, blockIdx.x, smem)) {
        }
        if (!grid_barrier(bar, target, wgs, fault)) return false;
        E* t = src; src = dst; dst = t;
    }
    return true;
}

// {key, value} pairs whose keys take at most 256 values (the dictionary `blk` holds them, nv of them; dict_kernels.hpp): ONE stable
// pass on the key's rank where the LSD passes need four.  The shape is one pass of coop_lsd_sort with two differences: the count
// phase looks ranks up and copies the workgroup's run of tiles to `tmp` as it goes, each key replaced by its rank (the scatter then
// goes tmp -> data: the result belongs in `data`, and a copy back afterwards would read and write everything once more), and the
// scatter phase sorts on bits [0, 8) and puts the keys back as it stores (DictPairIO).  A workgroup scatters the tiles it copied itself, so the copy needs no fence.
// Returns 1: sorted; 0: some key is not in the dictionary -- `data` is untouched, the caller's LSD passes sort; -1: a barrier gave up.
template <int NT, int K, int RANK>
__device__ __forceinline__ int coop_dict_pair_sort(uint64_t* data, uint64_t* tmp, uint32_t n, uint32_t* __restrict__ table,
                                                   uint32_t* __restrict__ totals, uint32_t* bar, uint32_t& target, uint32_t* fault,
                                                   unsigned char* smem, DictBlock* __restrict__ blk, uint32_t nv, uint32_t* s_val /* static LDS [256] */)
{
    using C = TileCfg<uint64_t, 8, NT, K>;
    constexpr int NW = NT / 64;
    static_assert(NT >= 256, "one thread per value");
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);   // [NW][256]
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + C::OFF_WSUM);
    const int tid = (int)threadIdx.x;
    const int w = tid >> 6;
    const uint32_t wg = blockIdx.x, wgs = gridDim.x;
    const uint32_t tiles = (n + (uint32_t)C::TILE - 1u) / (uint32_t)C::TILE;
    const uint32_t per = (tiles + wgs - 1u) / wgs;
    const uint32_t t0 = wg * per < tiles ? wg * per : tiles;
    const uint32_t t1 = t0 + per < tiles ? t0 + per : tiles;
    const uint32_t e0 = t0 * (uint32_t)C::TILE;   // (workgroups beyond the last tile: t0 = t1 = tiles, e0 >= n)
    const uint32_t e1 = (uint64_t)t1 * C::TILE < n ? t1 * (uint32_t)C::TILE : n;
    const uint32_t len = e1 > e0 ? e1 - e0 : 0u;
    auto tab_ld = [](const uint32_t* p) -> uint32_t { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto tab_st = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // the hash table lies in the tile's element area, which the count phase does not use (32 KiB: 4096 slots of {key, rank})
    unsigned long long* s_slot = reinterpret_cast<unsigned long long*>(smem + C::OFF_ELEMS);
    static_assert(8 * kPairSlots <= (int)(C::OFF_WCNT - C::OFF_ELEMS), "the hash table fits the element area");
    const uint32_t max_rank = pair_dict_load<NT>(blk, nv, s_val, s_slot);
    const bool one_value = nv == 1u;   // every key the same (if none misses): nothing to move
    for (int i = tid; i < NW * 256; i += NT) hist[i] = 0u;
    __syncthreads();
    // ---- count the ranks of this workgroup's run of tiles, copying it to tmp ---------------------------------------------
    {
        uint32_t* my = hist + w * 256;
        bool miss = false;
        // a lane counts runs of equal ranks by itself and adds a run at its end: constant and ordered keys would otherwise queue 64
        // lanes on one LDS counter, element after element
        uint32_t cur = 0u, run = 0u;
        auto count = [&](uint32_t r) {
            if (r == cur) {
                ++run;
            } else {
                if (run) atomicAdd(&my[cur], run);
                cur = r;
                run = 1u;
            }
        };
        struct alignas(16) V2 { uint64_t v[2]; };
        const V2* vs = reinterpret_cast<const V2*>(data + e0);   // e0 is a multiple of the tile: 16-byte aligned
        V2* vd = reinterpret_cast<V2*>(tmp + e0);
        const uint32_t nvec = len / 2u;
        constexpr int U = 4;
        for (uint32_t i0 = (uint32_t)tid; i0 < nvec; i0 += (uint32_t)(U * NT)) {
            V2 v[U];
            bool act[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                act[u] = i0 + (uint32_t)(u * NT) < nvec;
                if (act[u]) v[u] = vs[i0 + (uint32_t)(u * NT)];
                else v[u].v[0] = v[u].v[1] = 0ull;
            }
            unsigned long long s0[2 * U];   // first probes of all eight keys, then the looks at them
#pragma unroll
            for (int j = 0; j < 2 * U; ++j) s0[j] = s_slot[pair_dict_hash((uint32_t)v[j >> 1].v[j & 1])];
#pragma unroll
            for (int j = 0; j < 2 * U; ++j) {
                uint64_t& x = v[j >> 1].v[j & 1];
                const uint32_t r = pair_dict_rank((uint32_t)x, s0[j], s_slot, max_rank);
                if (act[j >> 1]) {
                    if (r == 0xffffffffu) miss = true;
                    else count(r);
                }
                x = (x & 0xffffffff00000000ull) | (r & 255u);   // the rank where the key was
            }
            if (!one_value) {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (act[u]) vd[i0 + (uint32_t)(u * NT)] = v[u];
            }
        }
        if (tid == 0 && (len & 1u)) {   // an odd element ends the array
            uint64_t x = data[e1 - 1u];
            const uint32_t r = pair_dict_rank((uint32_t)x, s_slot[pair_dict_hash((uint32_t)x)], s_slot, max_rank);
            if (r == 0xffffffffu) miss = true;
            else count(r);
            if (!one_value) tmp[e1 - 1u] = (x & 0xffffffff00000000ull) | (r & 255u);
        }
        if (run) atomicAdd(&my[cur], run);
        const int any_miss = __syncthreads_or(miss);
        if (any_miss && tid == 0) __hip_atomic_store(&blk->miss, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < 256) {
            uint32_t c = 0u;
            for (int i = 0; i < NW; ++i) c += hist[i * 256 + tid];
            tab_st(&table[(size_t)tid * wgs + wg], c);
        }
    }
    if (!grid_barrier(bar, target, wgs, fault, false)) return -1;
    if (__hip_atomic_load(&blk->miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;   // (the same word for every workgroup)
    if (one_value) return 1;
    // ---- workgroup d scans row d ------------------------------------------------------------------------------------------
    for (uint32_t d = wg; d < 256u; d += wgs) {
        uint32_t carry = 0u;
        for (uint32_t base = 0; base < wgs; base += (uint32_t)NT) {
            const uint32_t j = base + (uint32_t)tid;
            const uint32_t v = j < wgs ? tab_ld(&table[(size_t)d * wgs + j]) : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan_u32<NT>(v, s_wsum, &tot);
            if (j < wgs) tab_st(&table[(size_t)d * wgs + j], carry + ex);
            carry += tot;
        }
        if (tid == 0) tab_st(&totals[d], carry);
    }
    if (!grid_barrier(bar, target, wgs, fault, false)) return -1;
    // ---- scatter this workgroup's tiles tmp -> data, carrying per-rank offsets from tile to tile ---------------------------
    {
        const uint32_t tot_d = tid < 256 ? tab_ld(&totals[tid]) : 0u;
        const uint32_t base_d = block_excl_scan_u32<NT>(tot_d, s_wsum, nullptr);
        uint32_t carry = tid < 256 ? base_d + tab_ld(&table[(size_t)tid * wgs + wg]) : 0u;
        const DictPairIO io{tmp, data, s_val};
        for (uint32_t t = t0; t < t1; ++t) {
            const uint32_t tb = t * (uint32_t)C::TILE;
            const uint32_t left = n - tb;
            const uint32_t valid = left < (uint32_t)C::TILE ? left : (uint32_t)C::TILE;
            sort_scatter_tile<DictPairIO, 8, NT, K, RANK>(io, tb, valid, n, 0, smem,
                                                          [&](int, uint32_t c) { const uint32_t g = carry; carry += c; return g; });
        }
    }
    return 1;
}

// diagnostic: workgroup 0 notes when it reaches the k-th phase boundary of the net (10-ns ticks; "debug.net_stamp<k>")
__device__ __forceinline__ void net_stamp(uint32_t* stats, int k)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[4 + k] = (uint32_t)wall_clock64();
}

// The large sort's safety net, run by the workgroups of its offsets kernel when a run did not fit its slab (the input is then
// untouched: the passes write only slabs).  dict != nullptr (whole-key sorts of keys): first the counting sort of
// dict_kernels.hpp -- sample, look up and count, fill -- and only if the keys take more than 256 values, or one of them missed the
// dictionary, the LSD passes.  `bar` is zero on entry.  No launch of its own, no word for the host to read: the same input takes the
// same time whether it is the handle's first sort or its hundredth.
template <typename E, int NT, int K, int RANK = 1, int P = 0>
__device__ __forceinline__ void net_sort(E* data, E* tmp, uint32_t n, uint32_t* __restrict__ table, uint32_t* bar, uint32_t* fault,
                                         unsigned char* smem, int key_bits, DictBlock* dict, uint32_t* stats, const OsNet& os,
                                         uint32_t sample_repeats /* of the sort's first kernel: keys of <= 256 values show ~436 */,
                                         uint32_t target0 = 0u, bool barrier_at_end = false /* the caller reads `data` afterwards */)
{
    uint32_t target = target0;
    const uint32_t wgs = gridDim.x;
    // stats[0] = nets run, stats[1] = of those, sorted by counting ("stat.net_runs" / "stat.net_counting": tests, bench)
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(stats + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    net_stamp(stats, 0);
    constexpr bool PAIRS = sizeof(E) == 8 && P == 4;   // {key, value} pairs (u64 keys: P == 8)
    if (dict && n >= 16384u && sample_repeats >= kDictMinRepeats) {
        if (blockIdx.x == 0) {
            if constexpr (PAIRS) dict_sample_build_pair_keys<NT>(data, n, dict, smem);
            else dict_sample_build<E, NT>(data, n, dict, smem);
        }
        if (!grid_barrier(bar, target, wgs, fault)) return;
        const uint32_t nv = __hip_atomic_load(&dict->n_values, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if constexpr (PAIRS) {
            if (nv) {   // one stable pass on the key's rank among the values
                __shared__ uint32_t s_val[256];
                const int done = coop_dict_pair_sort<NT, K, RANK>(data, tmp, n, table, table + 256 * wgs, bar, target, fault, smem, dict, nv, s_val);
                if (done == 1 && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (done == 1 && barrier_at_end) grid_barrier(bar, target, wgs, fault);   // (the LSD passes end with one of their own)
                if (done != 0) return;
                __syncthreads();   // (a key missed the dictionary: the LSD passes sort the untouched input)
            }
        } else if (nv) {
            dict_count_range<E, NT>(data, n, dict, nv, smem);
            if (!grid_barrier(bar, target, wgs, fault)) return;
            if (!__hip_atomic_load(&dict->miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && dict_fill_range<E, NT>(data, n, dict, nv, smem)) {
                if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            __syncthreads();   // (the fill's LDS is the LSD sort's)
        }
    }
    if constexpr (sizeof(E) == 4) {   // u32 keys of up to 4096 values: the same three phases with the larger dictionary (dict_big_kernels.hpp)
        if (dict && n >= (1u << 20) && sample_repeats >= kBigMinRepeats) {
            BigDictBlock* big = reinterpret_cast<BigDictBlock*>(dict + 1);   // (one allocation: adlhip.hip)
            net_stamp(stats, 1);
            uint32_t* samples = reinterpret_cast<uint32_t*>(tmp);   // (64 Ki words of the partner array: n >= 2^20)
            big_dict_sample<NT>(data, n, samples, big);
            if (!grid_barrier(bar, target, wgs, fault)) return;
            if (blockIdx.x == 0) big_dict_build<NT>(samples, big, smem);
            net_stamp(stats, 2);
            if (!grid_barrier(bar, target, wgs, fault)) return;
            net_stamp(stats, 3);
            const uint32_t bnv = __hip_atomic_load(&big->n_values, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (bnv) {
                big_dict_count_range<NT>(data, n, big, bnv, smem, stats + 4);
                net_stamp(stats, 4);
                if (!grid_barrier(bar, target, wgs, fault)) return;
                net_stamp(stats, 5);
                if (!__hip_atomic_load(&big->miss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) && big_dict_fill_range<NT>(data, n, big, bnv, smem)) {
                    net_stamp(stats, 6);
                    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(stats + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return;
                }
                __syncthreads();
            }
        }
    }
    if constexpr (P > 0) {
        if (os.tables && key_bits == 8 * P) {   // whole keys: the look-back passes, four at a time
#pragma unroll 1
            for (int base = 0; base < 8 * P; base += 32)
                if (!coop_onesweep_sort<E, NT, K, 4, RANK>(data, tmp, n, os, bar, target, fault, smem, base)) return;
            return;
        }
    }
    coop_lsd_sort<E, NT, K, RANK>(data, tmp, n, table, table + 256 * wgs, bar, fault, smem, key_bits, target);
}

// ------------------------------------------------------------------------------------------
// C: finish segments in LDS.  A workgroup takes segments blockIdx.x, blockIdx.x + gridDim.x, ...  The segment
// [seg_start[s], seg_start[s+1]) of `in` is loaded once, sorted on its low `low_bits` bits with up to three stable
// local passes of at most LBITS bits, and written to the same range of `out` (in == out: in place; everything of a
// segment is read before its first store).  Per local pass and wave: returning DS atomics on the wave's own counters
// give the in-wave rank (radix_kernels.hpp rank_in_wave has the argument for why that is the stable rank); after ONE
// barrier every wave folds the counts of all waves for its lanes' bins, DPP-scans the bin totals and writes the tile
// positions of its own (wave, bin) runs -- 16-bit, read by that wave only, so no second barrier -- and scatters its
// elements to LDS; after a second barrier the tile is read back in order.  Two barriers per pass (the first version
// had seven and spent most of a segment's ~20K cycles in them).  Slots beyond the segment's size take no part (no
// pad keys: hundreds of pads on one LDS counter serialise).
// A segment of more than NT*K elements cannot live in the tile.  With a partner array (`in` != `out`) the workgroup
// sorts it through global memory instead -- count, scan, tile-by-tile scatter with a per-digit carry, 8 bits at a
// time, ping-ponging between the segment's ranges of the two arrays (slow: one workgroup; it exists so that skewed
// inputs stay correct, and the host keeps such inputs rare).  In place there is no partner: the segment is left as
// it is and the device fault word is raised.
// dyn (may be null): dyn[1] overrides low_bits -- the mid-size sort chooses its digit positions on the device -- and a
// non-zero dyn[2] (MidDyn::mode) sends the workgroups into the cooperative LSD sort below instead.
// ------------------------------------------------------------------------------------------
// Keys-only mid-size sort (two launches): the segments are the 256 bucket SLABS msd_bucket_scatter_kernel filled --
// bucket b occupies slab[b * stride, b * stride + state[b]) -- and go to out[sum of the counts before b ...).
// state: bucket cursors = counts at [32 * b] (one 128-byte line each), [8192] overflow word, [8193] readers-done counter,
// [8194] grid-barrier counter; handle-owned, zero between sorts: the last workgroup to have read it clears it.
struct SegSlab {
    uint32_t* state;      // nullptr: ordinary segment list
    uint32_t stride;
    uint32_t* host_mode;  // pinned: 1 + overflow flag of this sort, for the host's hint
    void* partner;        // the n-element scratch array (ping-pong partner of `out` if the cooperative LSD sort has to run)
};

template <typename E, int NT, int K, int LBITS>
__global__ __launch_bounds__(NT) void segment_sort_kernel(const E* in, E* out, const uint32_t* __restrict__ seg_start,
                                                          uint32_t num_segments, uint32_t low_bits_arg,
                                                          const uint32_t* __restrict__ dyn, uint32_t* fault, MidCoop coop,
                                                          SegSlab slab)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_slab[4];
    uint32_t slab_m = 0u, slab_out = 0u;
    if (slab.state) {   // one workgroup per bucket: its count, where its output starts, and whether any bucket overflowed
        static_assert(NT >= 256, "one thread per bucket");
        uint32_t* s_wsum0 = reinterpret_cast<uint32_t*>(smem);
        const int t = (int)threadIdx.x;
        const uint32_t c = t < 256 ? __hip_atomic_load(slab.state + 32 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t ovf = t == 0 ? __hip_atomic_load(slab.state + 8192, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t ex = block_excl_scan_u32<NT>(c, s_wsum0, nullptr);
        const int used = __syncthreads_count(c != 0u);   // buckets in use: 1 = the top byte is constant
        if (t == (int)blockIdx.x) { s_slab[0] = c; s_slab[1] = ex; }
        if (t == 0) s_slab[2] = ovf;
        __syncthreads();
        slab_m = s_slab[0];
        slab_out = s_slab[1];
        const uint32_t overflow = s_slab[2];
        // everyone has read the state: the last reader clears it for the next sort on this handle
        if (t == 0) {
            const uint32_t done = __hip_atomic_fetch_add(slab.state + 8193, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_slab[3] = done;
        }
        __syncthreads();
        if (s_slab[3] == gridDim.x - 1u) {
            for (int i = t; i < 256; i += NT) __hip_atomic_store(slab.state + 32 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < 2) __hip_atomic_store(slab.state + 8192 + t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // for the host's hints: 1 = fine, 2 = overflow with several buckets in use (skewed keys: the three-launch form would
            // not fit either), 3 = overflow because the top byte is constant (the three-launch form picks a lower byte)
            if (t == 0)
                __hip_atomic_store(slab.host_mode, !overflow ? 1u : (used > 1 ? 2u : 3u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (overflow) {   // a bucket did not fit its slab: `out` still holds the unsorted input, `in` is its partner
            coop_lsd_sort<E, NT, K>(out, static_cast<E*>(slab.partner), coop.n, coop.table, coop.totals, coop.bar, fault, smem);
            return;
        }
        __syncthreads();
    }
    if (dyn && dyn[2] != 0u) {   // mid-size sort, skewed keys: `out` still holds the unsorted input, `in` is its partner
        if (coop.table) coop_lsd_sort<E, NT, K>(out, const_cast<E*>(in), coop.n, coop.table, coop.totals, coop.bar, fault, smem);
        return;
    }
    constexpr int NW = NT / 64;
    constexpr int CAP = NT * K;
    constexpr int BINS = 1 << LBITS;
    constexpr int BPL = BINS / 64;                     // bins per lane when a wave folds and scans the counts
    static_assert(BPL == 4 || BPL == 8, "256 or 512 bins");
    using F = TileCfg<E, 8, NT, K>;                    // LDS carve of the through-memory path (sort_scatter_tile)
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + sizeof(E) * CAP);            // [NW][BINS]
    uint16_t* __restrict__ s_wpos = reinterpret_cast<uint16_t*>(s_wcnt + NW * BINS);                 // [NW][BINS]

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    uint16_t* my_wpos = s_wpos + w * BINS;
    const uint32_t low_bits = dyn ? dyn[1] : low_bits_arg;
    const int npass = ((int)low_bits + LBITS - 1) / LBITS;   // digits as even as possible: 18 bits, LBITS 9 -> 9 + 9

    for (uint32_t seg = blockIdx.x; seg < num_segments; seg += gridDim.x) {
        const uint32_t begin = slab.state ? seg * slab.stride : seg_start[seg];
        const uint32_t m = slab.state ? slab_m : seg_start[seg + 1] - begin;
        if (m == 0u) continue;
        const E* src = in + begin;
        E* dst = out + (slab.state ? slab_out : begin);
        if (m > (uint32_t)CAP) {
            if (in == out || slab.state) {   // no partner array: never sort wrongly in silence
                if (tid == 0) atomicOr(fault + 1, 0x40000u);   // sticky word
                continue;
            }
            // ---- through global memory: 8-bit LSD passes by this one workgroup ------------------------------
            if (low_bits & 7u) {   // the through-memory path sorts whole bytes (the mid-size sort's digits are)
                if (tid == 0) atomicOr(fault + 1, 0x40000u);
                continue;
            }
            const int nfp = (int)low_bits / 8;
            const E* a = src;                      // pass input
            E* b = dst;                            // pass output; the partner of `out`'s range is `in`'s
            E* other = const_cast<E*>(src);
            uint32_t* hist = reinterpret_cast<uint32_t*>(smem + F::OFF_WCNT);   // [NW][256]
            uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + F::OFF_WSUM);
            for (int p = 0; p < nfp; ++p) {
                const int sb = 8 * p;
                for (int i = tid; i < NW * 256; i += NT) hist[i] = 0u;
                __syncthreads();
                for (uint32_t i = (uint32_t)tid; i < m; i += (uint32_t)NT)
                    atomicAdd(&hist[w * 256 + (((uint32_t)a[i] >> sb) & 255u)], 1u);
                __syncthreads();
                uint32_t cnt_b = 0u;
                if (tid < 256)
                    for (int i = 0; i < NW; ++i) cnt_b += hist[i * 256 + tid];
                uint32_t carry = block_excl_scan_u32<NT>(cnt_b, s_wsum, nullptr);   // thread d: where digit d's run starts
                const AosIO<E> io{a, b};
                for (uint32_t base = 0; base < m; base += (uint32_t)CAP) {
                    const uint32_t left = m - base;
                    sort_scatter_tile<AosIO<E>, 8, NT, K, 1>(io, base, left < (uint32_t)CAP ? left : (uint32_t)CAP, m, sb, smem,
                                                             [&](int, uint32_t c) { const uint32_t g = carry; carry += c; return g; });
                }
                // this pass's stores must be visible to the next pass's loads (same workgroup, other waves)
                __threadfence_block();
                __syncthreads();
                a = b;
                b = (b == dst) ? other : dst;
            }
            if (a != dst) {   // even number of passes: the result sits in the partner's range
                for (uint32_t i = (uint32_t)tid; i < m; i += (uint32_t)NT) dst[i] = a[i];
            }
            __syncthreads();
            continue;
        }
        // ---- in LDS -------------------------------------------------------------------------------------------
        // wave-striped: with keff = ceil(m / NT) items per thread in use, wave w owns elements [w*64*keff, (w+1)*64*keff)
        const int keff = (int)((m + (uint32_t)NT - 1u) / (uint32_t)NT);
        const uint32_t wbase = (uint32_t)(w * 64 * keff + lane);
        const int rem = (int)m - (int)wbase;          // item j of this lane exists iff j*64 < rem
        int wave_valid = (int)m - w * 64 * keff;      // elements of this wave: a prefix in (item, lane) order
        wave_valid = wave_valid < 0 ? 0 : (wave_valid > 64 * keff ? 64 * keff : wave_valid);
        E e[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < keff) e[j] = (j * 64 < rem) ? src[wbase + (uint32_t)(j * 64)] : E(0);
        int sb = 0;
        for (int p = 0; p < npass; ++p) {
            const int nb = ((int)low_bits - sb + (npass - p - 1)) / (npass - p);
            const uint32_t mask = (1u << nb) - 1u;
            auto digit = [&](E x) -> uint32_t { return (uint32_t)(x >> sb) & mask; };   // u64 keys: bits beyond 31 too
#pragma unroll
            for (int q = 0; q < BPL; ++q) my_wcnt[q * 64 + lane] = 0u;
            uint32_t rnk[K];
            {
                // all elements of the wave in one bin (constant low bits): ranks are item*64 + lane
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)digit(e[0]));
                bool same = true;
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (j < keff) same &= (j * 64 >= rem) | (digit(e[j]) == d0);
                if (__all(same)) {
#pragma unroll
                    for (int j = 0; j < K; ++j) rnk[j] = (uint32_t)(j * 64 + lane);
                    if (lane == 0 && wave_valid > 0)
                        __hip_atomic_store(&my_wcnt[d0], (uint32_t)wave_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                } else {
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        if (j < keff && j * 64 < rem)
                            rnk[j] = __hip_atomic_fetch_add(&my_wcnt[digit(e[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __syncthreads();   // every wave's counts are final; the previous pass's read-back is done
            {   // lane l owns bins [l*BPL, (l+1)*BPL): totals over all waves, the part of the waves before mine, tile positions
                uint32_t tot[BPL], pre[BPL];
#pragma unroll
                for (int q = 0; q < BPL; ++q) tot[q] = pre[q] = 0u;
#pragma unroll
                for (int i = 0; i < NW; ++i) {
#pragma unroll
                    for (int q = 0; q < BPL; q += 4) {
                        const u32x4 r = *reinterpret_cast<const u32x4*>(s_wcnt + i * BINS + lane * BPL + q);
                        tot[q] += r.x; tot[q + 1] += r.y; tot[q + 2] += r.z; tot[q + 3] += r.w;
                        if (i < w) { pre[q] += r.x; pre[q + 1] += r.y; pre[q + 2] += r.z; pre[q + 3] += r.w; }
                    }
                }
                uint32_t s = 0u;
#pragma unroll
                for (int q = 0; q < BPL; ++q) s += tot[q];
                uint32_t run = wave_incl_scan_u32(s) - s;
#pragma unroll
                for (int q = 0; q < BPL; q += 2) {
                    const uint32_t p0 = run + pre[q];
                    run += tot[q];
                    const uint32_t p1 = run + pre[q + 1];
                    run += tot[q + 1];
                    *reinterpret_cast<uint32_t*>(my_wpos + lane * BPL + q) = p0 | (p1 << 16);   // positions < CAP <= 65536
                }
            }
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < keff && j * 64 < rem) s_elems[(uint32_t)my_wpos[digit(e[j])] + rnk[j]] = e[j];
            __syncthreads();   // the tile is in sorted order
            if (p + 1 < npass) {
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (j < keff && j * 64 < rem) e[j] = s_elems[wbase + (uint32_t)(j * 64)];
            }
            sb += nb;
        }
        if (npass == 0) {   // nothing left to sort: the segment only moves
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < keff && j * 64 < rem) dst[wbase + (uint32_t)(j * 64)] = e[j];
        } else {
            for (uint32_t i = (uint32_t)tid; i < m; i += (uint32_t)NT) dst[i] = s_elems[i];
        }
        __syncthreads();   // the tile is free for the next segment
    }
}

// ------------------------------------------------------------------------------------------
// Small segments (at most 64*K elements, at most 16 low bits): ONE WAVE per segment, no workgroup barrier anywhere.
// The segment sits in the wave's registers (lane l, item j <-> element j*64 + l) and in the wave's private slice of
// LDS.  Per 8-bit local pass: every element bumps its bin (DS add), the lanes turn the 256 counts into bin starts
// (4 bins per lane + one DPP scan), every element fetches its slot with a RETURNING DS add on its bin -- issue order =
// item order and colliding lanes in lane order, so the slots are handed out stably (radix_kernels.hpp rank_in_wave) --
// and goes to LDS there; the tile comes back in order.  DS operations of one wave execute in issue order, so the
// phases need no barrier, and nothing but the elements lives in registers across them (the first wave-per-segment
// kernel kept ranks and predicates too, 165-175 VGPRs, and ran at two or three waves per SIMD).
// ------------------------------------------------------------------------------------------
// Slab form (seg_cnt != nullptr; the large keys-only sort): segment s is in[s * in_stride, + seg_cnt[s]) and goes to
// out[seg_start[s] ...); the kernel returns at once when *gate is non-zero.
// The body for segments of F+1 .. R rows of 64 elements, R and F compile-time constants: rows 0 .. F-1 are full and carry no
// predicate, only the last R - F are tested per lane.  (With one body for all sizes and `if (row < rows && lane has an item)`
// around every item the kernel spent ~1200 wave instructions per segment.  Filling up with all-ones pads instead of testing
// is worse: up to 127 pads bump ONE counter, and same-address LDS atomics serialise.)
// S = type of the stored elements: E, or uint16_t -- the large sort keeps only the low 16 bits of u32 keys in its second slab
// (the bits above are the segment's number) and `hi` puts them back at the final store.
// SOA: the destination is two u32 arrays (keys = dst reinterpreted, values = dst_vals), E = {key, value} as one u64.
// RANK == 0: the slots are handed out WITHOUT relying on the lane order of colliding returning DS atomics ("sort.rank" = 0): the
// lanes of one instruction that share a digit find each other by ballots (8 per element), take the bin's running start by a plain
// LDS read and their place by counting the peers below them; the lowest peer moves the start on.  DS operations of one wave
// execute in issue order, so the next row's read sees this row's store.  Documented wave intrinsics only; ~2 x the ALU work.
template <typename E, int R, int F, typename S, bool SOA, int RANK = 1>
__device__ __forceinline__ void wave_sort_rows(const S* __restrict__ src, E* __restrict__ dst, uint32_t* __restrict__ dst_vals, uint32_t m,
                                               int lane, E* __restrict__ buf, uint32_t* __restrict__ cnt, uint32_t low_bits, E hi)
{
    auto put = [&](int idx, E x) {
        if constexpr (SOA) {
            reinterpret_cast<uint32_t*>(dst)[idx] = (uint32_t)x;
            dst_vals[idx] = (uint32_t)((unsigned long long)x >> 32);
        } else {
            dst[idx] = x;
        }
    };
    const int rem = (int)m - lane;   // item j of this lane exists iff j*64 < rem
    E e[R];
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (j < F || j * 64 < rem) e[j] = (E)load_once(src + j * 64 + lane);
    const int npass = ((int)low_bits + 7) / 8;   // 8-bit digits at most, as even as possible (u64 keys: up to six passes)
    if (npass == 0) {   // nothing left to sort (the digits above covered every bit that varies): the segment only moves
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j < F || j * 64 < rem) put(j * 64 + lane, e[j] | hi);
        return;
    }
    int sb = 0;
    bool in_regs = true;   // the current order is in e[] (always, except after a final pass that ran)
    for (int p = 0; p < npass; ++p) {
        const int nb = ((int)low_bits - sb + (npass - p - 1)) / (npass - p);
        const uint32_t mask = (1u << nb) - 1u;
        auto digit = [&](E x) -> uint32_t { return (uint32_t)(x >> sb) & mask; };
        {   // a pass in which every element has the same digit changes nothing -- and would queue all 64 lanes of every
            // atomic on ONE counter (keys that are multiples of 65536: the finish took 0.91 ms instead of 0.12 at 64 Mi keys).
            // Screen on one element per lane, exact test only when it passes.
            const uint32_t dg0 = digit(e[0]);   // row 0 is full, or the lane's item is missing: then any value will do
            const uint32_t df = (uint32_t)__builtin_amdgcn_readfirstlane((int)dg0);
            if (__all(dg0 == df || rem <= 0)) {
                bool same = true;
#pragma unroll
                for (int j = 1; j < R; ++j)
                    if (j < F || j * 64 < rem) same &= digit(e[j]) == df;
                if (__all(same)) {
                    sb += nb;
                    continue;
                }
            }
        }
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(cnt + 4 * lane) = z;
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j < F || j * 64 < rem) __hip_atomic_fetch_add(&cnt[digit(e[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        {   // counts -> bin starts
            const u32x4 c = *reinterpret_cast<const u32x4*>(cnt + 4 * lane);
            const uint32_t s4 = c.x + c.y + c.z + c.w;
            const uint32_t ex = wave_incl_scan_u32(s4) - s4;
            u32x4 o;
            o.x = ex;
            o.y = ex + c.x;
            o.z = o.y + c.y;
            o.w = o.z + c.z;
            *reinterpret_cast<u32x4*>(cnt + 4 * lane) = o;
        }
        // slots: one returning atomic per element, then its store.  Small tiles (many waves per CU) go element by element; the
        // large tiles of the big segments (K = 40, 80: one to four waves per CU) would sit out one LDS round trip per element,
        // so they take their slots eight at a time
        if constexpr (RANK == 0) {
            // two sweeps over the rows so that nothing waits for an LDS round trip per row: (1) the peers of every row by ballots; the
            // LOWEST peer reserves the group's slots with one returning add -- within one instruction only one lane per digit adds,
            // so no two lanes collide and the order of colliding lanes never matters; rows follow each other in issue order --
            // (2) every lane fetches its group's start from that lane (ds_bpermute) and stores its element
            // (eight rows at a time: with all R rows' values live the kernel loses its occupancy)
            constexpr int RB = 8;
#pragma unroll
            for (int j0 = 0; j0 < R; j0 += RB) {
                uint32_t base[RB], info[RB];   // info = leader lane << 8 | peers below
#pragma unroll
                for (int jj = 0; jj < RB; ++jj) {
                    const int j = j0 + jj;
                    base[jj] = 0u;
                    info[jj] = 0u;
                    if (j < R && (j < F || j * 64 < rem)) {
                        const uint32_t dg = digit(e[j < R ? j : R - 1]);
                        uint64_t peers = __ballot(true);   // the lanes that hold an item of this row
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            const bool bit = (dg >> b) & 1u;
                            const uint64_t bal = __ballot(bit);
                            peers &= bit ? bal : ~bal;
                        }
                        const uint32_t below = mbcnt64(peers);
                        info[jj] = ((uint32_t)__builtin_ctzll(peers) << 8) | below;
                        if (below == 0u)
                            base[jj] = __hip_atomic_fetch_add(&cnt[dg], (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                }
#pragma unroll
                for (int jj = 0; jj < RB; ++jj) {
                    const int j = j0 + jj;
                    if (j < R && (j < F || j * 64 < rem)) {
                        const uint32_t start = (uint32_t)__shfl((int)base[jj], (int)(info[jj] >> 8));
                        buf[start + (info[jj] & 0xffu)] = e[j < R ? j : R - 1];
                    }
                }
            }
        }
        constexpr int B = R > 20 ? 8 : 1;   // 16 for the largest tile measured the same
#pragma unroll
        for (int j0 = 0; RANK != 0 && j0 < R; j0 += B) {
            uint32_t pos[B];
#pragma unroll
            for (int jj = 0; jj < B; ++jj) {
                const int j = j0 + jj;
                if (j < R && (j < F || j * 64 < rem))
                    pos[jj] = __hip_atomic_fetch_add(&cnt[digit(e[j < R ? j : R - 1])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (int jj = 0; jj < B; ++jj) {
                const int j = j0 + jj;
                if (j < R && (j < F || j * 64 < rem)) buf[pos[jj]] = e[j < R ? j : R - 1];
            }
        }
        if (p + 1 < npass) {
#pragma unroll
            for (int j = 0; j < R; ++j)
                if (j < F || j * 64 < rem) e[j] = buf[j * 64 + lane];
        } else {
            in_regs = false;
        }
        sb += nb;
    }
    if (in_regs) {   // the last pass (or every pass) had nothing to do
#pragma unroll
        for (int j = 0; j < R; ++j)
            if (j < F || j * 64 < rem) put(j * 64 + lane, e[j] | hi);
        return;
    }
#pragma unroll
    for (int j = 0; j < R; ++j)
        if (j < F || j * 64 < rem) put(j * 64 + lane, buf[j * 64 + lane] | hi);
}

// rows -> the smallest body that holds them: bodies for RMIN (any number of rows up to RMIN: every row tested), RMIN + STEP,
// ..., K rows.  A kernel for large tiles (K = 40, 80: segments of the large sort beyond 64 Mi keys) starts at K / 2 -- its
// segments are that large -- and steps by 4 to keep the code size in bounds.
template <typename E, int R, int K, int STEP, bool FIRST, typename S, bool SOA, int RANK = 1>
__device__ __forceinline__ void wave_sort_dispatch(int rows, const S* __restrict__ src, E* __restrict__ dst, uint32_t* __restrict__ dst_vals,
                                                   uint32_t m, int lane, E* __restrict__ buf, uint32_t* __restrict__ cnt, uint32_t low_bits,
                                                   E hi)
{
    constexpr int RR = R < K ? R : K;
    constexpr int F = FIRST ? 0 : RR - STEP;
    if constexpr (R >= K) {
        wave_sort_rows<E, RR, F, S, SOA, RANK>(src, dst, dst_vals, m, lane, buf, cnt, low_bits, hi);
    } else {
        if (rows <= R) wave_sort_rows<E, RR, F, S, SOA, RANK>(src, dst, dst_vals, m, lane, buf, cnt, low_bits, hi);
        else wave_sort_dispatch<E, R + STEP, K, STEP, false, S, SOA, RANK>(rows, src, dst, dst_vals, m, lane, buf, cnt, low_bits, hi);
    }
}

// Narrow second digit (small inputs): with fewer than ~48 Mi elements 65536 segments are too many -- the finish then spends its
// time launching waves that hold a few dozen keys each (4 Mi keys: 44 us for the finish alone) -- so the second MSD digit is
// read `8 - w` bits higher: the field [top - 8 - w, top - w) overlaps the first digit in its upper 8 - w bits, which are the same
// for every key of a bucket, so only 2^w of its 256 values occur per bucket, cursors, counts and offsets stay indexed
// [bucket][8-bit digit] as ever, and the finish sorts the top - 8 - w bits below.  Only the slabs and the finish's grid are
// compact: slot c = (bucket << w) | (digit & (2^w - 1)), 256 << w slots in all.  w = 8: slot = segment.
__device__ __forceinline__ uint32_t slot_to_segment(uint32_t slot, uint32_t w)
{
    const uint32_t b = slot >> w;
    return (b << 8) | ((b & ((1u << (8u - w)) - 1u)) << w) | (slot & ((1u << w) - 1u));
}

// LIST: the segments to do are list[0 .. *list_cnt) -- what bin_segment_sort_kernel handed over, usually nothing -- taken in turns
// by a small grid; otherwise wave i of the grid does segment i.  (One call site of the body per instantiation: with two, the
// compiler stops inlining the 80-row bodies and the kernel runs three times slower.)
template <typename E, int K, int WAVES, int STEP, int RMIN, typename S, bool SOA, bool LIST, int RANK = 1>
__global__ __launch_bounds__(64 * WAVES) void wave_segment_sort_kernel(const E* in, E* out, const uint32_t* __restrict__ seg_start,
                                                                        uint32_t num_segments, uint32_t low_bits, uint32_t* fault,
                                                                        const uint32_t* __restrict__ seg_cnt, uint32_t in_stride,
                                                                        const uint32_t* __restrict__ gate,
                                                                        const uint32_t* __restrict__ dyn_low_bits,
                                                                        uint32_t* out_vals /* SOA: out = the key array */,
                                                                        const uint32_t* __restrict__ list,
                                                                        const uint32_t* __restrict__ list_cnt, uint32_t seg_shift)
{
    if (gate && *gate != 0u) return;
    if (dyn_low_bits) low_bits = *dyn_low_bits;
    constexpr int CAP = 64 * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)threadIdx.x & 63;
    const int w = (int)threadIdx.x >> 6;
    unsigned char* mine = smem + (size_t)w * (sizeof(E) * CAP + 256 * 4);
    E* __restrict__ buf = reinterpret_cast<E*>(mine);
    uint32_t* __restrict__ cnt = reinterpret_cast<uint32_t*>(mine + sizeof(E) * CAP);
    // wave-uniform values, said so (derived from threadIdx they count as divergent)
    uint32_t slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (uint32_t)WAVES + (uint32_t)w));
    uint32_t limit = num_segments;
    if constexpr (LIST) {
        const uint32_t nl = (uint32_t)__builtin_amdgcn_readfirstlane((int)*list_cnt);
        limit = nl < num_segments ? nl : num_segments;
    }
    for (; slot < limit; slot = LIST ? slot + gridDim.x * (uint32_t)WAVES : limit) {
        uint32_t sl = slot;   // slab number (slab forms); num_segments counts slots
        if constexpr (LIST) sl = (uint32_t)__builtin_amdgcn_readfirstlane((int)list[slot]);
        const uint32_t seg = seg_shift < 8u ? slot_to_segment(sl, seg_shift) : sl;
        const uint32_t begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_start[seg]);
        const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)(seg_cnt ? seg_cnt[seg] : seg_start[seg + 1] - begin));
        if (m == 0u) continue;
        if (m > (uint32_t)CAP || low_bits > 8u * (uint32_t)sizeof(E)) {   // never sort wrongly in silence
            if (lane == 0) atomicOr(fault + 1, 0x40000u);
            continue;
        }
        if constexpr (SOA) {
            const E* src = in + (size_t)sl * in_stride;
            wave_sort_dispatch<E, RMIN, K, STEP, true, E, true, RANK>((int)((m + 63u) >> 6), src,
                                                                reinterpret_cast<E*>(reinterpret_cast<uint32_t*>(out) + begin),
                                                                out_vals + begin, m, lane, buf, cnt, low_bits, E(0));
        } else if constexpr (sizeof(S) == sizeof(E)) {
            const E* src = in + (seg_cnt ? (size_t)sl * in_stride : (size_t)begin);
            wave_sort_dispatch<E, RMIN, K, STEP, true, E, false, RANK>((int)((m + 63u) >> 6), src, out + begin, nullptr, m, lane, buf, cnt,
                                                                       low_bits, E(0));
        } else {
            // slab form with 16-bit elements: the key's bits above low_bits are (sampled prefix, segment number); seg_shift = 8
            const S* src = reinterpret_cast<const S*>(in) + (size_t)sl * in_stride;
            const E hi = (E)(((dyn_low_bits[1] << 16) | seg) << low_bits);
            wave_sort_dispatch<E, RMIN, K, STEP, true, S, false>((int)((m + 63u) >> 6), src, out + begin, nullptr, m, lane, buf, cnt,
                                                                 low_bits, hi);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Finish for segments beyond a wave's tile: the large sort of more than 280 Mi u32 keys leaves 65536 segments of 4.5 K ... 17 K
// 16-bit keys each -- too many for the 80 rows one wave holds, so ONE WORKGROUP of NT threads takes a segment slab: load (S =
// what the second slab holds) | per 8-bit local pass: returning DS atomics on the wave's own counters give the in-wave rank,
// barrier, every wave folds all waves' counts for its lanes' bins and writes its own (wave, bin) positions, scatter into LDS,
// barrier, read back | store with the bits above put back (`hi`, as in wave_segment_sort_kernel).  The body is
// segment_sort_kernel's in-LDS branch; slots beyond the segment's size take no part.  Gated by the mode word like every finish.
// ------------------------------------------------------------------------------------------
template <typename E, typename S, int NT, int K>
__global__ __launch_bounds__(NT) void wg_segment_sort_kernel(const S* __restrict__ in, E* __restrict__ out,
                                                             const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_cnt,
                                                             uint32_t in_stride, const uint32_t* __restrict__ gate,
                                                             const uint32_t* __restrict__ dyn_low_bits, uint32_t* fault)
{
    if (*gate != 0u) return;
    constexpr int NW = NT / 64;
    constexpr int CAP = NT * K;
    constexpr int BINS = 256;
    constexpr int BPL = 4;
    static_assert(CAP <= 65536, "16-bit tile positions");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    S* __restrict__ s_elems = reinterpret_cast<S*>(smem);   // the tile holds what the slab holds (16-bit keys: half the LDS, twice the workgroups)
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + sizeof(S) * CAP);   // [NW][BINS]
    uint16_t* __restrict__ s_wpos = reinterpret_cast<uint16_t*>(s_wcnt + NW * BINS);        // [NW][BINS]
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    uint16_t* my_wpos = s_wpos + w * BINS;
    const uint32_t seg = blockIdx.x;
    const uint32_t m = seg_cnt[seg];
    if (m == 0u) return;
    const uint32_t low_bits = dyn_low_bits[0];
    if (m > (uint32_t)CAP || low_bits > 8u * (uint32_t)sizeof(S)) {   // never sort wrongly in silence
        if (tid == 0) atomicOr(fault + 1, 0x40000u);
        return;
    }
    const S* __restrict__ src = in + (size_t)seg * in_stride;
    E* __restrict__ dst = out + seg_off[seg];
    const E hi = sizeof(S) < sizeof(E) ? (E)(((dyn_low_bits[1] << 16) | seg) << low_bits) : E(0);
    const int npass = ((int)low_bits + 7) / 8;
    // wave-striped: with keff = ceil(m / NT) items per thread in use, wave w owns elements [w*64*keff, (w+1)*64*keff)
    const int keff = (int)((m + (uint32_t)NT - 1u) / (uint32_t)NT);
    const uint32_t wbase = (uint32_t)(w * 64 * keff + lane);
    const int rem = (int)m - (int)wbase;   // item j of this lane exists iff j*64 < rem
    E e[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < keff) e[j] = (j * 64 < rem) ? (E)load_once(src + wbase + (uint32_t)(j * 64)) : E(0);
    int sb = 0;
    for (int p = 0; p < npass; ++p) {
        const int nb = ((int)low_bits - sb + (npass - p - 1)) / (npass - p);
        const uint32_t mask = (1u << nb) - 1u;
        auto digit = [&](E x) -> uint32_t { return (uint32_t)(x >> sb) & mask; };
#pragma unroll
        for (int q = 0; q < BPL; ++q) my_wcnt[q * 64 + lane] = 0u;
        // the in-wave rank (< 64 K <= 4096) rides in the upper half of the element's register: the keys are 16 bits wide, and a
        // second array of K registers costs a workgroup per CU at the large tiles (157 VGPRs at 40 rows)
        static_assert(sizeof(S) == 2 && sizeof(E) == 4 && 64 * K <= 65536, "16-bit keys in 32-bit registers");
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < keff && j * 64 < rem)
                e[j] |= __hip_atomic_fetch_add(&my_wcnt[digit(e[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << 16;
        __syncthreads();   // every wave's counts are final; the previous pass's read-back is done
        {   // lane l owns bins [4l, 4l + 4): totals over all waves, the part of the waves before mine, tile positions
            u32x4 tot = {0u, 0u, 0u, 0u}, pre = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(s_wcnt + i * BINS + lane * BPL);
                tot += r;
                if (i < w) pre += r;
            }
            const uint32_t s4 = tot.x + tot.y + tot.z + tot.w;
            const uint32_t run = wave_incl_scan_u32(s4) - s4;
            const uint32_t p0 = run + pre.x, p1 = run + tot.x + pre.y, p2 = run + tot.x + tot.y + pre.z,
                           p3 = run + tot.x + tot.y + tot.z + pre.w;
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 packed = {p0 | (p1 << 16), p2 | (p3 << 16)};   // positions < CAP <= 65536
            *reinterpret_cast<u32x2*>(my_wpos + lane * BPL) = packed;
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < keff && j * 64 < rem) s_elems[(uint32_t)my_wpos[digit(e[j])] + (e[j] >> 16)] = (S)e[j];
        __syncthreads();   // the tile is in sorted order
        if (p + 1 < npass) {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < keff && j * 64 < rem) e[j] = (E)s_elems[wbase + (uint32_t)(j * 64)];
        }
        sb += nb;
    }
    if (npass == 0) {   // nothing left to sort: the segment only moves
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < keff && j * 64 < rem) dst[wbase + (uint32_t)(j * 64)] = e[j] | hi;
    } else {
        for (uint32_t i = (uint32_t)tid; i < m; i += (uint32_t)NT) dst[i] = (E)s_elems[i] | hi;
    }
}

// ------------------------------------------------------------------------------------------
// Finish of the large KEYS-ONLY sort, one workgroup per segment, ONE counting pass whatever the number of low bits:
// the keys of a segment share everything above their low `low_bits` bits and are otherwise as good as random (that is what
// the two MSD digits above them leave), so binning them on the TOP `BITS` of the low bits -- about one bin per key --
// almost sorts the segment: what remains are the few keys that share a bin, and those are put in order by comparing whole
// keys.  Equal keys are indistinguishable, so nothing has to be stable: all waves share one set of counters and a key's
// rank inside its bin is the value its (returning) counting atomic handed back.
//   load -> returning DS add on the key's bin (two 16-bit counters per word) | barrier | counters -> bin starts (block scan,
//   in place) | barrier | key -> LDS at start[bin] + rank | barrier | every position p: the key there goes to
//   out[start[bin] + number of keys of its bin that are smaller (or equal and earlier)] -- bins of one key: out[p].
// LDS operations per key: ~7 (the wave-per-segment finish: 4 per 8-bit pass, 24 for the 48 low bits of u64 keys), and a
// segment of 4096 u64 keys is worked on by 8 waves instead of one wave holding 80 rows (three WAVES per CU).
// A segment whose bins fill unevenly (more than kBinLimit keys in a bin: keys that are not random below the digits --
// duplicates, constant bit fields) is not finished here: its number goes onto a list and wave_segment_sort_kernel in its
// list form (stable LSD passes, any keys) does it right behind this kernel.
// S = stored element: E, or uint16_t for the 16-bit second slab of u32 keys (hi restores the bits above).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kBinLimit = 16;
// barrier over LDS traffic only: __syncthreads() also waits for every global load and store of the wave (vmcnt(0)), which in a
// workgroup that loops would serialise the next segment's loads and this segment's stores with the LDS phases
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// words of the sort's mode block (work buffer): [0] = 1: the safety net has sorted, the finish returns at once; [4] = bits the
// finish sorts, [5] = the keys' sampled prefix; [8] = number of listed segments (cleared by the offsets kernel)
enum { kDynMode = 0, kDynLowBits = 4, kDynHardCnt = 8 };

template <typename E, typename S, int NT, int K, int BITS>
__global__ __launch_bounds__(NT, (NT >= 512 ? 6 : 4)) void bin_segment_sort_kernel(const S* __restrict__ in, E* __restrict__ out,
                                                              const uint32_t* __restrict__ seg_off, const uint32_t* __restrict__ seg_cnt,
                                                              uint32_t in_stride, uint32_t num_segments, const uint32_t* __restrict__ mode,
                                                              uint32_t* __restrict__ hard_cnt, uint32_t* __restrict__ hard_list,
                                                              uint32_t* fault, uint32_t seg_shift)
{
    if (mode[kDynMode] != 0u) return;   // the safety net has sorted instead
    constexpr int CAP = NT * K;
    constexpr int BINS = 1 << BITS;
    constexpr int WORDS = BINS / 2;     // two 16-bit counters per word: bin b = half (b & 1) of word b >> 1
    constexpr int WPT = WORDS / NT;     // words per thread in the scan: thread t owns bins [2 * WPT * t, 2 * WPT * (t + 1))
    constexpr int NW = NT / 64;
    static_assert(WORDS % NT == 0 && WPT >= 1, "whole words per thread");
    static_assert(CAP < 65536, "16-bit positions");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    S* __restrict__ s_buf = reinterpret_cast<S*>(smem);
    uint32_t* __restrict__ s_cnt = reinterpret_cast<uint32_t*>(smem + ((sizeof(S) * CAP + 15) & ~(size_t)15));   // [WORDS]
    const uint16_t* __restrict__ s_start = reinterpret_cast<const uint16_t*>(s_cnt);                               // [BINS]
    uint32_t* __restrict__ s_wsum = s_cnt + WORDS;                                                                 // [NW]
    uint32_t* __restrict__ s_hard = s_wsum + NW;                                                                   // [NW]
    const int tid0 = (int)threadIdx.x;
    const int lane = tid0 & 63;
    const int w = tid0 >> 6;
    const uint32_t low_bits = mode[kDynLowBits];
    const int sh = low_bits > (uint32_t)BITS ? (int)low_bits - BITS : 0;
    const uint32_t dmask = low_bits >= (uint32_t)BITS ? (uint32_t)BINS - 1u : (1u << low_bits) - 1u;
    auto bin_of = [&](S x) -> uint32_t { return (uint32_t)(x >> sh) & dmask; };
    uint32_t prefix = 0u;
    if constexpr (sizeof(S) != sizeof(E)) prefix = mode[kDynLowBits + 1] << 16;

    // persistent: workgroup g takes segments g, g + G, ...; the keys of the next one are requested while this one is sorted
    uint32_t seg = blockIdx.x;   // slot = slab number; sg(slot) = segment number (slot_to_segment)
    if (seg >= num_segments) return;
    auto sg = [&](uint32_t slot) -> uint32_t { return seg_shift < 8u ? slot_to_segment(slot, seg_shift) : slot; };
    uint32_t m = seg_cnt[sg(seg)];
    S e[K];
    {
        const S* __restrict__ src = in + (size_t)seg * in_stride;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((uint32_t)(j * NT + tid0) < m && m <= (uint32_t)CAP) e[j] = load_once(src + j * NT + tid0);
    }
    for (;;) {
        // the thread index, hidden from loop-invariant code motion: hoisted out of the loop, the addresses derived from it took a
        // hundred registers (167 VGPRs, one workgroup per CU instead of three)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const uint32_t next = seg + gridDim.x;
        const bool more = next < num_segments;
        const uint32_t m_next = more ? seg_cnt[sg(next)] : 0u;
        E hi_bits = E(0);
        if constexpr (sizeof(S) != sizeof(E)) hi_bits = (E)((prefix | seg) << low_bits);   // 16-bit slabs: seg_shift = 8
        E* __restrict__ dst = out + seg_off[sg(seg)];
        auto prefetch_next = [&]() {
            if (more) {
                const S* __restrict__ src = in + (size_t)next * in_stride;
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if ((uint32_t)(j * NT + tid) < m_next && m_next <= (uint32_t)CAP) e[j] = load_once(src + j * NT + tid);
            }
        };
        const bool sortable = m != 0u && m <= (uint32_t)CAP && low_bits != 0u;
        if (m > (uint32_t)CAP && tid == 0) atomicOr(fault + 1, 0x40000u);   // never sort wrongly in silence
#pragma unroll
        for (int i = 0; i < WPT; ++i) s_cnt[tid + i * NT] = 0u;
        lds_barrier();   // counters are zero; the previous segment's last reads of the tile are done
        uint32_t rk2[(K + 1) / 2];   // ranks inside the bins, two to a register
#pragma unroll
        for (int j = 0; j < (K + 1) / 2; ++j) rk2[j] = 0u;
        bool hard = false;
        if (sortable) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if ((uint32_t)(j * NT + tid) < m) {
                    const uint32_t b = bin_of(e[j]);
                    const uint32_t hs = (b & 1u) << 4;
                    const uint32_t old = __hip_atomic_fetch_add(&s_cnt[b >> 1], 1u << hs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t r = (old >> hs) & 0xffffu;
                    rk2[j >> 1] |= r << (16 * (j & 1));
                    hard |= r >= kBinLimit;
                }
            }
        }
        if (lane == 0) s_hard[w] = 0u;
        if (__any(hard) && lane == 0) s_hard[w] = 1u;
        lds_barrier();   // every count is final
        bool is_hard = false;
#pragma unroll
        for (int i = 0; i < NW; ++i) is_hard |= s_hard[i] != 0u;
        if (is_hard && tid == 0) hard_list[__hip_atomic_fetch_add(hard_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = seg;
        if (!is_hard && sortable) {
            // counts -> bin starts, in place
            {
                uint32_t wv[WPT];
                uint32_t sum = 0u;
#pragma unroll
                for (int i = 0; i < WPT; ++i) {
                    wv[i] = s_cnt[tid * WPT + i];
                    sum += (wv[i] & 0xffffu) + (wv[i] >> 16);
                }
                const uint32_t inc = wave_incl_scan_u32(sum);
                if (lane == 63) s_wsum[w] = inc;
                lds_barrier();
                uint32_t run = inc - sum;
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    if (i < w) run += s_wsum[i];
#pragma unroll
                for (int i = 0; i < WPT; ++i) {
                    const uint32_t c0 = wv[i] & 0xffffu, c1 = wv[i] >> 16;
                    const uint32_t lo = run;
                    run += c0;
                    const uint32_t hi = run;
                    run += c1;
                    s_cnt[tid * WPT + i] = lo | (hi << 16);
                }
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < K; ++j)
                if ((uint32_t)(j * NT + tid) < m) s_buf[(uint32_t)s_start[bin_of(e[j])] + ((rk2[j >> 1] >> (16 * (j & 1))) & 0xffffu)] = e[j];
            prefetch_next();   // the registers are free: the next segment's keys travel while this one is put in order and stored
            lds_barrier();
            // every key counts the keys of its bin that come before it (bins of one key: nothing to do) and goes there
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t p = (uint32_t)(j * NT + tid);
                if (p < m) {
                    const S x = s_buf[p];
                    const uint32_t b = bin_of(x);
                    const uint32_t lo = s_start[b];
                    const uint32_t hi = b == dmask ? m : (uint32_t)s_start[b + 1u];
                    uint32_t f = p;
                    if (hi - lo > 1u) {
                        uint32_t c = 0u;
                        for (uint32_t q = lo; q < hi; ++q) {
                            const S y = s_buf[q];
                            c += (y < x || (y == x && q < p)) ? 1u : 0u;
                        }
                        f = lo + c;
                    }
                    dst[f] = (E)x | hi_bits;
                }
            }
            lds_barrier();   // the tile and the bin starts are free for the next segment
        } else {
            if (!is_hard && m != 0u && m <= (uint32_t)CAP) {   // nothing left to sort: the segment only moves
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const uint32_t p = (uint32_t)(j * NT + tid);
                    if (p < m) dst[p] = (E)e[j] | hi_bits;
                }
            }
            prefetch_next();
        }
        if (!more) break;
        seg = next;
        m = m_next;
    }
}

// ------------------------------------------------------------------------------------------
// Keys-only mid-size sort, pass 1 of 2: MSD scatter on the top byte WITHOUT an up-front histogram and WITHOUT look-back.
// Equal u32 keys are indistinguishable, so this pass need not be stable across tiles: a tile reserves room for its run
// of every digit with ONE returning atomic on that bucket's cursor and writes the run into the bucket's slab (bucket b
// owns slab[b * stride, (b+1) * stride)); in what order tiles arrive does not matter, pass 2 sorts each bucket
// completely and the result is the one sorted array.  (Pairs need the stable three-launch form above: equal keys must
// keep their order.)  A bucket that outgrows its slab -- skewed keys, or keys whose top byte is constant -- sets the
// overflow word; pass 2 then sorts the untouched input with the cooperative LSD sort instead.
// ------------------------------------------------------------------------------------------
// One kernel serves the single MSD pass of the mid-size sort (linear input) and both MSD passes of the large keys-only sort
// (the second pass reads the buckets of the first: workgroup -> (source bucket, tile inside it), tiles beyond the bucket's
// count return at once; its cursors are indexed (source bucket, digit)).
struct StablePlace {   // written by msd2s_prep_kernel
    uint32_t top;      // one past the highest key bit (below sort_bits) in which two sampled keys differ (>= 16)
    uint32_t low_bits; // top - 16: what the finish sorts
    uint32_t sort_bits;
    uint32_t pad;
    unsigned long long prefix;   // (key & mask(sort_bits)) >> top of every key (top < key bits)
    unsigned long long kmask;    // mask(sort_bits)
};

// The key of an element: the whole element (u32 keys; u64 keys: KEY64), or the low dword of a {key, value} pair.
template <bool KEY64, typename E>
__device__ __forceinline__ unsigned long long key_of(E e)
{
    if constexpr (KEY64) return (unsigned long long)e;
    else return (unsigned long long)(uint32_t)e;
}

template <typename E>
struct BucketPass {
    const E* src;
    E* dst;
    uint32_t* cursors;            // [source buckets][256] << cursor_shift, zero on entry
    uint32_t cursor_shift;        // log2 of the words between two cursors: 5 = one 128-byte line each.  The 256 cursors of a
                                  // pass over ONE array take an atomic from every tile; packed into 8 lines they queue on 8
                                  // atomic units (pass 1 of the large sort: 0.33 ms instead of 0.13)
    uint32_t src_count_shift;     // the same for src_counts
    uint32_t* flag;               // set when a run does not fit its destination slab
    const uint32_t* src_counts;   // nullptr: the source is ONE array of n elements; else element counts of the source buckets
    uint32_t n;
    uint32_t src_stride;          // elements between two source buckets
    uint32_t tiles_per_bucket;    // tiles a source bucket can hold
    uint32_t dst_stride;          // elements per destination slab
    uint32_t dst_total;           // elements of the whole destination array
    int start_bit;
    uint32_t* zero_me;            // one word the first workgroup clears (the safety net's barrier counter), or nullptr
    const uint32_t* sample;       // nullptr, or the four sample words (msd2_placement): start_bit = top - 8 * which_digit
    int which_digit;              // 1 = first digit, 2 = second
    int dst16;                    // the destination slabs hold uint16_t: only the key's low 16 bits are written (the second pass of
                                  // u32 keys: the bits above are the segment's number -- 128 MiB less to write and to read back
                                  // at 64 Mi keys)
    // PASS == 3, the second pass of the hybrid form: the first pass was the STABLE one (msd_lookback_scatter_kernel, pass A), so
    // a source bucket is made of `pieces` sub-slabs (one per chain of pass A) whose sizes are the last status rows of pass A's
    // chains, and the digits' place comes from msd2s_prep_kernel
    const StablePlace* place;
    const uint32_t* status_a;
    uint32_t pieces, rows_per_chain_a, slice;
    uint32_t seg_shift;           // second pass: width w of the second digit (see slot_to_segment); 8 otherwise
};

// Digit placement of the large keys-only sort, chosen on the device from a sample of the keys: keys that do not use their top
// bits -- the local sort of a multi-GPU sort sees 1/8 of the key range, indices stay below 2^28 -- would put everything into a
// few buckets if the first digit were always the top byte.  msd2_sample_kernel ORs and ANDs 1024 keys into four words that
// belong to the device handle (or = 0 / and = ~0 between sorts; the offsets kernel resets them); every consumer derives
//   top = one past the highest bit in which two sampled keys differ (at least 16)
// first digit = bits [top-8, top), second = [top-16, top-8), the LDS finish sorts the top-16 bits below.  The bits from `top`
// up are the same in every sampled key; the first pass checks that for EVERY key and sends the sort to its safety net
// otherwise (a sample can miss an outlier).
struct Msd2Placement {
    int top;
    unsigned long long prefix;   // key >> top of every key (top < key bits)
};
__device__ __forceinline__ Msd2Placement msd2_placement(const uint32_t* __restrict__ sample /* or lo, or hi, and lo, and hi */)
{
    const unsigned long long o = ((unsigned long long)sample[1] << 32) | sample[0];
    const unsigned long long a = ((unsigned long long)sample[3] << 32) | sample[2];
    const unsigned long long diff = o ^ a;
    Msd2Placement p;
    p.top = diff ? 64 - __builtin_clzll(diff) : 0;
    if (p.top < 16) p.top = 16;
    p.prefix = p.top < 64 ? (o >> p.top) : 0ull;
    return p;
}
// Where the net's dictionary sampling (dict_sample_build; rounds 2-3: the key probe) reads sample k of 16384 (n >= 16384): somewhere inside the k-th 16384th of the array (a fixed stride would see
// one phase of periodic keys only).  The offset inside the cell is a 24-bit hash scaled by multiply-and-shift, NOT `hash % cell`:
// the compiler expands a remainder of operands it knows to fit 24 bits through float (v_rcp_iflag_f32, one upward correction),
// which overshoots the quotient for some operands (n = 7726351: 13 of the 16384 samples), the "remainder" wraps to ~2^24 and
// the load lands 64 MiB past the array -- a memory fault whenever nothing is mapped there (found by tools/stress.py;
// adlhip_selftest_probe_positions checks the positions on the device).
__device__ __forceinline__ size_t probe_sample_index(uint32_t k, uint32_t n)
{
    const uint32_t cell = n >> 14;                                                           // < 2^18
    const uint32_t hash24 = (uint32_t)(((unsigned long long)k * 0x9E3779B97F4A7C15ull) >> 40);
    const uint32_t jit = (uint32_t)(((unsigned long long)hash24 * cell) >> 24);              // in [0, cell)
    unsigned long long at = (unsigned long long)k * (unsigned long long)n / 16384ull + jit;
    if (at >= n) at = n - 1u;   // cannot happen (k n / 16384 + cell - 1 <= n - 1); the load stays inside the array whatever
    return (size_t)at;
}
// self-test: out[0] = the largest position any of the 16384 samples reads for this n, out[1] = positions outside their cell
ADLHIP_KERNEL __global__ __launch_bounds__(1024) void probe_positions_selftest_kernel(uint32_t n, uint32_t* __restrict__ out)
{
    uint32_t hi = 0u, bad = 0u;
    for (int i = 0; i < 16; ++i) {
        const uint32_t k = threadIdx.x * 16u + (uint32_t)i;
        const unsigned long long at = probe_sample_index(k, n);
        const unsigned long long lo = (unsigned long long)k * n / 16384ull, up = (unsigned long long)(k + 1u) * n / 16384ull;
        if (at < lo || at >= (up > lo ? up : lo + 1ull) || at >= n) ++bad;
        hi = at > hi ? (uint32_t)at : hi;
    }
    atomicMax(out + 0, hi);
    if (bad) atomicAdd(out + 1, bad);
}

// ------------------------------------------------------------------------------------------
// The first kernel of a large sort: 2048 sampled keys, two per lane of 16 one-wave workgroups.  They place the digits (OR / AND into
// the handle's four sample words, below) and they show what the passes would otherwise find out by moving keys (64 Mi keys of 256
// values: 0.2 ms before the net starts): keys that repeat a few thousand values cannot fit the slabs (a value with more copies than
// a segment slab holds).  Every wave counts the samples that repeat an earlier one of ITS 128 (D distinct values: 128 - D (1 -
// e^(-128/D)) of them, ~8000 / D; 32-bit fingerprints in an LDS table: two different keys share one with probability 2^-31)
// and adds the count -- if it is not zero -- to flag[kSampleRepeatsWord]; the kernels behind compare the sum with the threshold the
// host computed from n (adlhip.hip sample_dup_threshold; never below 4 where keys without repeats show 0; large_sort_gave_up): the
// passes leave at their first instruction and the net starts at once.  No state, no report: the same input is treated the same way
// every time.  Fewer than kDictMinRepeats and the net does not try its dictionary.  The offsets kernel puts the word back to zero.
// Cost: the second key per lane and two LDS compare-and-swaps, ~1-2 us of 8.  Tried and dropped: 1024 samples in ONE
// workgroup (LDS table of fingerprints + first-digit histogram for "one bucket holds 1/16 of the keys"): 1024 scattered loads from
// one compute unit queue up behind its address translation -- 10-11 us against 6, 1.6 % of a 64-Mi-key sort on every input
// (same-box A/B); sixteen workgroups handing their samples to the last one for that digest: 19 us; the sixteen waves' counts summed
// by the last to arrive (one returning atomic behind an s_waitcnt): 9 us.
constexpr int kSampleWGs = 16;
constexpr int kSamples = 2048;
constexpr int kSampleArriveWord = 13;    // flag[13]: workgroups of msd2s_prep_kernel that have arrived (zero when idle)

// where sample `k` of 2048 is read: its 1/2048th of the input, at a scrambled offset inside (evenly spaced positions meet keys
// generated from their index -- i * c >> s and the like -- at values that never repeat, whatever the input holds)
__device__ __forceinline__ size_t sample_position(int k, uint32_t n)
{
    const unsigned long long lo = (unsigned long long)k * n / (unsigned)kSamples, hi = (unsigned long long)(k + 1) * n / (unsigned)kSamples;
    uint32_t h = (uint32_t)k * 0x85EBCA6Bu + 0x27D4EB2Fu;
    h ^= h >> 15;
    h *= 0xC2B2AE35u;
    h ^= h >> 13;
    return (size_t)(lo + (((unsigned long long)h * (hi - lo)) >> 32));   // in [lo, hi) (or lo); multiply-and-shift, no remainder
}

// Every lane of the 16 one-wave workgroups, with its two (masked) samples: OR / AND into sample[0..3] (or lo, or hi, and lo, and
// hi: idle values 0 / ~0), the wave's repeats into flag[kSampleRepeatsWord].
__device__ __forceinline__ void sample_accumulate(unsigned long long v0, unsigned long long v1, uint32_t* sample, uint32_t* flag)
{
    const int lane = (int)threadIdx.x;
    unsigned long long o = v0 | v1, a = v0 & v1;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        o |= __shfl_xor(o, sh);
        a &= __shfl_xor(a, sh);
    }
    if (lane == 0) {
        __hip_atomic_fetch_or(sample + 0, (uint32_t)o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(sample + 1, (uint32_t)(o >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_and(sample + 2, (uint32_t)a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_and(sample + 3, (uint32_t)(a >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // repeats among the wave's 128 samples: fingerprints (never 0) into a 256-slot LDS table; an insert that finds its own fingerprint
    // there is a repeat.  One wave: its LDS operations execute in order, no barrier.  (Lane-by-lane compares through v_readlane: 126
    // of them and 250 compares, ~1 us of a 7-us kernel.)
    __shared__ uint32_t s_seen[256];
#pragma unroll
    for (int i = 0; i < 4; ++i) s_seen[i * 64 + lane] = 0u;
    uint32_t mine = 0u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned long long hv = ((j ? v1 : v0) + 0x632BE59BD9B4E019ull) * 0x9E3779B97F4A7C15ull;
        const uint32_t fp = (uint32_t)(hv >> 32) | 1u;
        uint32_t h = (uint32_t)(hv >> 24) & 255u;
        for (int step = 0; step < 256; ++step) {
            const uint32_t old = atomicCAS(&s_seen[h], 0u, fp);
            if (old == 0u) break;
            if (old == fp) {
                ++mine;
                break;
            }
            h = (h + 1u) & 255u;
        }
    }
    uint32_t reps = mine;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) reps += __shfl_xor(reps, sh);
    if (lane == 0 && reps) __hip_atomic_fetch_add(flag + kSampleRepeatsWord, reps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename E>
__global__ __launch_bounds__(64) void msd2_sample_kernel(const E* __restrict__ src, uint32_t n, uint32_t* sample, uint32_t* bar,
                                                         uint32_t* fault, uint32_t* flag, uint32_t dup_thr)
{
    const int tid = (int)(blockIdx.x * 64u + threadIdx.x);
    if (tid == 0) {   // the safety net's grid-barrier counter (used, if at all, in the offsets kernel)
        __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fault[0] = 0u;   // the live fault word: the first kernel of every sort clears it (include/adlhip.h)
        __hip_atomic_store(flag + kSampleThresholdWord, dup_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (this form sorts whole keys; its offsets kernel puts the sample words back to their idle values)
    const E x0 = src[sample_position(2 * tid, n)], x1 = src[sample_position(2 * tid + 1, n)];
    sample_accumulate((unsigned long long)x0, (unsigned long long)x1, sample, flag);
}

// The tile body is the one-sweep pass's (onesweep_kernels.hpp onesweep_chain_kernel) without its ticket, status rows and
// look-back: load (wave-striped) -> rank (returning DS atomics) | barrier | every wave folds the counts and writes its own
// 16-bit positions; wave 0 reserves the tile's 256 runs with returning atomics on the bucket cursors while all waves
// scatter into LDS | barrier | write-out, consecutive lanes to consecutive addresses of a run.
// PASS: 0 = the mid-size sort's single pass, 1 / 2 = first / second pass of the large sort -- the same code (everything it
// switches on is in BucketPass); the parameter only gives the launches of a sort kernel names of their own, so that rocprofv3's
// per-kernel statistics and counters tell pass 1 from pass 2.
template <typename E, int NT, int K, int PASS>
__global__ __launch_bounds__(NT) void msd_bucket_scatter_kernel(BucketPass<E> a)
{
    using C = TileCfg<E, 8, NT, K>;
    typedef AosIO<E> IO;
    constexpr int BINS = 256;
    constexpr int NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem + C::OFF_ELEMS);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);   // [NW][BINS]
    uint32_t* __restrict__ s_goff = reinterpret_cast<uint32_t*>(smem + C::OFF_GOFF);   // [BINS]
    if (a.zero_me && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.zero_me, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a run has outgrown its slab: the net will sort (net_sort), nothing written from here on is read.  The flag is requested here and
    // looked at behind the tile's key loads (below), so that its round trip hides behind theirs
    // (ONE thread's view, handed to all through LDS behind the ranking's barrier: the flag can rise between two waves' loads, and a
    // workgroup of which some waves have left would go on with stale counters)
    uint32_t give_up = 0u;
    if constexpr (PASS >= 1) {
        if (threadIdx.x == 0) give_up = large_sort_gave_up(a.flag);
    }
    uint32_t base, valid, cursor_base = 0u;
    uint32_t lin = 0u;   // PASS == 3: index of the tile's first element if the tile lies inside one sub-slab, else ~0
    if constexpr (PASS == 3) {
        // workgroup -> (bucket, tile of the bucket); the bucket = its sub-slabs one after the other (see LookbackPass, pass B)
        uint32_t* __restrict__ s_misc = reinterpret_cast<uint32_t*>(smem + C::OFF_MISC);
        const uint32_t b = blockIdx.x / a.tiles_per_bucket, t = blockIdx.x % a.tiles_per_bucket;
        if (threadIdx.x < 64u) {
            const int l = (int)threadIdx.x;
            uint32_t cnt = 0u;
            if ((uint32_t)l < a.pieces) {
                const uint32_t c0 = (uint32_t)l * a.slice;
                if (c0 < a.n) {
                    const uint32_t len = (c0 + a.slice < a.n ? c0 + a.slice : a.n) - c0;
                    const uint32_t rows = (len + (uint32_t)C::TILE - 1u) / (uint32_t)C::TILE;
                    cnt = a.status_a[((size_t)l * a.rows_per_chain_a + rows - 1u) * BINS + b] & kValMask;
                    if (cnt > a.src_stride) cnt = a.src_stride;   // overflowed in pass A: the flag is set, only stay in bounds
                }
            }
            const uint32_t incl = wave_incl_scan_u32(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (l < 32) {
                const bool real = (uint32_t)l < a.pieces;
                s_goff[l] = real ? incl : 0xffffffffu;
                s_goff[32 + l] = real ? (b * a.pieces + (uint32_t)l) * a.src_stride - (incl - cnt) : 0u;
            }
            const uint32_t pos = t * (uint32_t)C::TILE;   // position in the bucket
            uint32_t v = 0u;
            if (pos < total) v = total - pos < (uint32_t)C::TILE ? total - pos : (uint32_t)C::TILE;
            // the sub-slab the tile starts in, and the next one that holds anything: a tile that lies inside those two (with
            // sub-slabs of about a tile's size: nearly every tile) is loaded as two stretches -- one compare per element
            const unsigned long long after = __ballot((uint32_t)l < a.pieces && incl > pos && cnt != 0u);
            uint32_t ln = 0xffffffffu, split = 0u, ln2 = 0u;
            if (after && v != 0u) {
                const int c0 = __builtin_ctzll(after);
                const uint32_t end0 = (uint32_t)__builtin_amdgcn_readlane((int)incl, c0);
                const uint32_t first0 = end0 - (uint32_t)__builtin_amdgcn_readlane((int)cnt, c0);
                const unsigned long long rest = after & ~(1ull << c0);
                if (pos + v <= end0) {
                    ln = (b * a.pieces + (uint32_t)c0) * a.src_stride + (pos - first0);
                    split = v;
                } else if (rest) {
                    const int c1 = __builtin_ctzll(rest);
                    const uint32_t end1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, c1);
                    if (pos + v <= end1) {
                        ln = (b * a.pieces + (uint32_t)c0) * a.src_stride + (pos - first0);
                        split = end0 - pos;                                             // tile positions below it: first stretch
                        ln2 = (b * a.pieces + (uint32_t)c1) * a.src_stride - split;     // + tile position = index of the second
                    }
                }
            }
            if (l == 0) {
                s_misc[1] = pos;
                s_misc[2] = v;
                s_misc[3] = ln;
                s_misc[4] = split;
                s_misc[5] = ln2;
            }
        }
        __syncthreads();
        base = s_misc[1];
        valid = s_misc[2];
        lin = s_misc[3];
        if (valid == 0u) return;   // a tile beyond the bucket's keys
        cursor_base = b * 256u;
    } else if (a.src_counts == nullptr) {
        base = blockIdx.x * (uint32_t)C::TILE;
        const uint32_t left = a.n - base;
        valid = left < (uint32_t)C::TILE ? left : (uint32_t)C::TILE;
    } else {
        const uint32_t b = blockIdx.x / a.tiles_per_bucket, t = blockIdx.x % a.tiles_per_bucket;
        const uint32_t cnt = a.src_counts[b << a.src_count_shift];
        const uint32_t off = t * (uint32_t)C::TILE;
        if (off >= cnt || off >= a.src_stride) return;      // nothing of this bucket in this tile (or the bucket overflowed)
        const uint32_t room = (cnt < a.src_stride ? cnt : a.src_stride) - off;
        base = b * a.src_stride + off;
        valid = room < (uint32_t)C::TILE ? room : (uint32_t)C::TILE;
        cursor_base = b * 256u;
    }
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    int start_bit = a.start_bit;
    Msd2Placement place{0, 0ull};
    if (a.sample) {
        place = msd2_placement(a.sample);
        start_bit = place.top - 8 * a.which_digit;
    }
    if constexpr (PASS == 3) start_bit = (int)a.place->top - 16;
    if constexpr (PASS >= 2) start_bit += 8 - (int)a.seg_shift;   // narrow second digit: the field sits 8 - w bits higher
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    const IO io{a.src, a.dst};
    const bool scaled = dst_fits32<IO>(a.dst_total);

    // ---- load, wave-striped; slots beyond `valid` are all-ones pads (digit 255, highest tile positions, never stored) ----
    const uint32_t wbase = (uint32_t)(w * 64 * K + lane);
    E e[K];
    if (PASS == 3 && lin == 0xffffffffu) {
        // the tile runs across sub-slabs: position in the bucket -> sub-slab, found once for the lane's first element, then
        // carried along (positions rise).  The tables sit in the s_goff area (written again only after the ranking's barrier).
        const uint32_t* __restrict__ s_end = s_goff;
        const uint32_t* __restrict__ s_adj = s_goff + 32;
        const uint32_t q0 = base + wbase;
        uint32_t c = 0u;
        for (uint32_t i = 0; i < a.pieces; ++i) c += (q0 >= s_end[i]) ? 1u : 0u;
        const int rem = (int)valid - (int)wbase;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t q = q0 + (uint32_t)(j * 64);
            while (c < 31u && q >= s_end[c]) ++c;
            e[j] = (j * 64 < rem) ? load_once(a.src + (size_t)(q + s_adj[c])) : ~E(0);
        }
    } else if (PASS == 3 && reinterpret_cast<const uint32_t*>(smem + C::OFF_MISC)[4] < valid) {
        // two stretches: tile positions below `split` come from the first sub-slab, the others from the next one
        const uint32_t split = reinterpret_cast<const uint32_t*>(smem + C::OFF_MISC)[4];
        const uint32_t ln2 = reinterpret_cast<const uint32_t*>(smem + C::OFF_MISC)[5];
        const int rem = (int)valid - (int)wbase;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t i = wbase + (uint32_t)(j * 64);
            e[j] = (j * 64 < rem) ? load_once(a.src + (size_t)((i < split ? lin : ln2) + i)) : ~E(0);
        }
    } else {
        const typename IO::Cursor p = io.cursor((size_t)(PASS == 3 ? lin : base) + wbase);
        if (valid == (uint32_t)C::TILE) {
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = p.at(j * 64);
        } else {
            const int rem = (int)valid - (int)wbase;
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = (j * 64 < rem) ? p.at(j * 64) : ~E(0);
        }
    }
    if (a.sample && a.which_digit == 1 && place.top < (int)(8 * sizeof(E))) {
        // a key outside the sampled range would land in a wrong bucket: let the safety net sort instead
        const E pre = (E)place.prefix;
        const int rem = (int)valid - (int)wbase;
        E bad = E(0);
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j * 64 < rem) bad |= (e[j] >> place.top) ^ pre;
        if (bad != E(0)) __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- rank ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int b = lane; b < BINS; b += 64) my_wcnt[b] = 0u;
    uint32_t rnk2[(K + 1) / 2];
    {
        uint32_t rnk[K];
        rank_in_wave<E, 8, K, 1>(e, rnk, my_wcnt, start_bit);
#pragma unroll
        for (int j = 0; j < K; j += 2) rnk2[j >> 1] = rnk[j] | ((j + 1 < K ? rnk[j + 1] : 0u) << 16);
    }
#pragma unroll
    for (int j = 0; j < (K + 1) / 2; ++j) asm volatile("" : "+v"(rnk2[j]));
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(e[j]));
    if constexpr (PASS >= 1) {
        if (threadIdx.x == 0) reinterpret_cast<uint32_t*>(smem + C::OFF_MISC)[8] = give_up;
    }
    __syncthreads();
    if constexpr (PASS >= 1) {
        if (reinterpret_cast<const uint32_t*>(smem + C::OFF_MISC)[8]) return;   // every wave leaves, or none
    }
    // ---- every wave: counts of all waves for its lanes' digits -> tile offsets -> its own (wave, digit) positions -------------
    u32x4 cnt4 = {0u, 0u, 0u, 0u};
    u32x4 toff4 = {0u, 0u, 0u, 0u};
    uint16_t* __restrict__ my_wpos = reinterpret_cast<uint16_t*>(smem + C::OFF_WPOS) + w * BINS;
    {
        u32x4 pre4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(s_wcnt + i * BINS + 4 * lane);
            cnt4 += r;
            if (i < w) pre4 += r;
        }
        const uint32_t s4 = cnt4.x + cnt4.y + cnt4.z + cnt4.w;
        const uint32_t ex = wave_incl_scan_u32(s4) - s4;
        toff4.x = ex;
        toff4.y = ex + cnt4.x;
        toff4.z = toff4.y + cnt4.y;
        toff4.w = toff4.z + cnt4.z;
        const u32x4 p4 = toff4 + pre4;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 packed = {p4.x | (p4.y << 16), p4.z | (p4.w << 16)};
        *reinterpret_cast<u32x2*>(my_wpos + 4 * lane) = packed;
    }
    // ---- wave 0: reserve room for the tile's runs (requested now, consumed after the scatter) -----------------------------------
    u32x4 at4 = {0u, 0u, 0u, 0u};
    u32x4 real4 = cnt4;
    if (w == 0) {
        if (lane == 63) real4.w -= (uint32_t)C::TILE - valid;   // the pads sit under digit 255
        uint32_t* cur = a.cursors + ((size_t)(cursor_base + 4u * (uint32_t)lane) << a.cursor_shift);
        const size_t step = (size_t)1 << a.cursor_shift;
        if (real4.x) at4.x = __hip_atomic_fetch_add(cur + 0 * step, real4.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (real4.y) at4.y = __hip_atomic_fetch_add(cur + 1 * step, real4.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (real4.z) at4.z = __hip_atomic_fetch_add(cur + 2 * step, real4.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (real4.w) at4.w = __hip_atomic_fetch_add(cur + 3 * step, real4.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- scatter into tile-sorted order ---------------------------------------------------------------------------------------------
    {
        constexpr int CH = K < 8 ? K : 8;
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += CH) {
            uint32_t pos[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) pos[j] = my_wpos[digit_of<8>(e[(j0 + j < K) ? j0 + j : K - 1], start_bit)];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j0 + j < K) {
                    const uint32_t r = (rnk2[(j0 + j) >> 1] >> (16 * ((j0 + j) & 1))) & 0xffffu;
                    s_elems[pos[j] + r] = e[j0 + j];
                }
            }
        }
    }
    if (w == 0) {
        const bool over = (at4.x + real4.x > a.dst_stride) | (at4.y + real4.y > a.dst_stride) | (at4.z + real4.z > a.dst_stride) |
                          (at4.w + real4.w > a.dst_stride);
        if (over) __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t d0 = cursor_base + 4u * (uint32_t)lane;
        if constexpr (PASS >= 2)   // slab of (bucket, digit): slot (bucket << w) | (digit & (2^w - 1)); w = 8: the same number
            d0 = ((cursor_base >> 8) << a.seg_shift) + ((4u * (uint32_t)lane) & ((1u << a.seg_shift) - 1u));
        u32x4 go;   // destination index = goff[digit] + tile position
        go.x = (d0 + 0u) * a.dst_stride + at4.x - toff4.x;
        go.y = (d0 + 1u) * a.dst_stride + at4.y - toff4.y;
        go.z = (d0 + 2u) * a.dst_stride + at4.z - toff4.z;
        go.w = (d0 + 3u) * a.dst_stride + at4.w - toff4.w;
        *reinterpret_cast<u32x4*>(s_goff + 4 * lane) = (scaled && !a.dst16) ? go * (uint32_t)IO::kStoreScale : go;
    }
    __syncthreads();
    if (a.dst16) {
        // one 2-byte store per key (two tile positions per lane packed into one 4-byte store where they share a run was no
        // faster: 119.9 vs 120.8 us at 64 Mi keys)
        uint16_t* __restrict__ d16 = reinterpret_cast<uint16_t*>(a.dst);
#pragma unroll 8
        for (int i = 0; i < K; ++i) {
            const uint32_t pos = (uint32_t)(tid + i * NT);
            if (pos < valid) {
                const E v = s_elems[pos];
                const uint32_t g = s_goff[digit_of<8>(v, start_bit)] + pos;
                if (g < a.dst_total) d16[g] = (uint16_t)v;
            }
        }
        return;
    }
    write_out_tile<IO, 8, NT, K, ADLHIP_WRITE_UNROLL>(io, s_elems, s_goff, valid, a.dst_total, start_bit, scaled);
}

// ------------------------------------------------------------------------------------------
// Large keys-only sort ("sort.msd2", 2 Mi < n <= 1088 Mi u32 keys, u64 keys up to 260 Mi): TWO unstable MSD passes with bucket
// cursors (first digit, then second digit inside every bucket: 65536 segments of n / 65536 keys; where the digits sit is chosen
// from a sample of the keys, msd2_placement) and ONE LDS finish (wave_segment_sort_kernel on the bits below).  No histogram
// kernel, no look-back, every key moved 6 times instead of 9 (u32 keys: the second slab holds their low 16 bits only).  msd2_offsets_kernel sits between the
// second pass and the finish: workgroup b turns bucket b's 256 cursors into output offsets (bucket base = scan of the first
// pass's cursors), saves the counts for the finish, clears the cursors for the next sort (they belong to the device handle),
// and the last workgroup publishes the mode word: 0 = every run fitted its slab; else 1 -- the finish returns at once, and the
// workgroups of THIS kernel go on to sort the untouched input with net_sort (counting sort for few distinct values, else the
// cooperative LSD passes).  Nothing is told to the host: the same input takes the same path and time at every call.
// ------------------------------------------------------------------------------------------
// The safety net lives in this kernel too: its workgroups (512 of 512 threads, two per CU) are all resident, so when the overflow flag is set they go on to
// sort the untouched input with the cooperative LSD sort (a launch of its own that returns at once cost 5-6 us per sort).
template <typename E, int NT, int K>
__global__ __launch_bounds__(NT) void msd2_offsets_kernel(uint32_t* cursors_a, uint32_t* cursors_b, uint32_t* flag, uint32_t* done,
                                                          uint32_t* bar, uint32_t* __restrict__ seg_cnt, uint32_t* __restrict__ seg_off,
                                                          uint32_t* __restrict__ mode, uint32_t n,
                                                          uint32_t* sample, E* data, E* tmp, uint32_t* __restrict__ ctable,
                                                          uint32_t* fault, int key_bits, uint32_t d2_shift /* 8 - w, slot_to_segment */,
                                                          DictBlock* dict /* the net's counting sort, or nullptr */, uint32_t* stats,
                                                          OsNet os)
{
    static_assert(NT >= 256, "one thread per digit");
    __shared__ uint32_t s_wsum[NT / 64 + 1];
    __shared__ uint32_t s_misc[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the safety net's tile (TileCfg<E, 8, NT, K>)
    const int t = (int)threadIdx.x;
    const uint32_t b = blockIdx.x;   // the grid has at least 256 workgroups: workgroup b < 256 serves bucket b
    // final since pass 2 has completed; every workgroup reads it BEFORE it counts itself done, the last one done clears it
    const uint32_t overflow = large_sort_gave_up(flag);
    const uint32_t sample_repeats = __hip_atomic_load(flag + kSampleRepeatsWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b < 256u) {
        const uint32_t ca = t < 256 ? __hip_atomic_load(cursors_a + 32 * t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;   // one line per cursor
        const uint32_t exa = block_excl_scan_u32<NT>(ca, s_wsum, nullptr);
        if (t == (int)b) s_misc[0] = exa;
        const uint32_t cb = t < 256 ? __hip_atomic_load(cursors_b + b * 256u + (uint32_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t exb = block_excl_scan_u32<NT>(cb, s_wsum, nullptr);
        __syncthreads();
        if (t < 256) {
            seg_cnt[b * 256u + (uint32_t)t] = cb;
            seg_off[b * 256u + (uint32_t)t] = s_misc[0] + exb;
            __hip_atomic_store(cursors_b + b * 256u + (uint32_t)t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the first pass's cursors are read by the first 256 workgroups: the last workgroup to be done clears them and publishes the mode
    if (t == 0) s_misc[1] = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_misc[1] == gridDim.x - 1u) {
        if (t < 256) __hip_atomic_store(cursors_a + 32 * t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0) {
            *mode = overflow ? 1u : 0u;   // the finish returns at once when it is set
            mode[kDynHardCnt] = 0u;
            seg_off[65536] = n;
            if (sample) {   // the finish sorts the bits below the second digit; the sample words go back to or = 0 / and = ~0
                mode[kDynLowBits] = (uint32_t)(msd2_placement(sample).top - 16) + d2_shift;
                mode[kDynLowBits + 1] = (uint32_t)msd2_placement(sample).prefix;   // for the finish of a 16-bit second slab
                __hip_atomic_store(sample + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sample + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sample + 2, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(sample + 3, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(flag + kSampleRepeatsWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (overflow) {   // `bar` is zero here: msd2_sample_kernel, the first launch of every sort, clears it
        __syncthreads();
        net_sort<E, NT, K, 1, (int)sizeof(E)>(data, tmp, n, ctable, bar, fault, smem, key_bits, dict, stats, os, sample_repeats);
    }
}

// ------------------------------------------------------------------------------------------
// The STABLE form of the large sort, for {key, value} pairs ("sort.msd2" as well): equal keys must keep their order, so a
// tile cannot take whatever room an atomic cursor hands it.  Same two MSD passes into slabs and the same finish, but a tile
// learns where its run of a digit goes by decoupled LOOK-BACK over the tiles before it in its chain (the one-sweep pass's
// status rows and lookback_exclusive4, onesweep_kernels.hpp) -- and since every (digit, chain) pair has a slab of its own,
// no histogram and no offset table is needed:
//   pass A  P = 16 chains = 16 equal slices of the input (in order; workgroup i takes chain i % 16 and runs on XCD i % 8, so a
//           chain stays on one XCD); the run of digit d of a tile of chain c goes to sub-slab
//           (d, c) = slab_a[(d * P + c) * stride_a ...) behind the runs of the chain's earlier tiles.  Bucket d = its P
//           sub-slabs in order: input order is kept.
//   pass B  256 chains = the buckets; chain b's tiles are consecutive stretches of its P sub-slabs taken one after the other
//           (the sub-slabs' sizes are the last status rows of pass A's chains; a tile may start in one and end in the next); the run of second digit d2 goes to slab_b[(b * 256 + d2) * stride_b ...) behind the chain's
//           earlier tiles.
//   finish  wave_segment_sort_kernel on the bits below (its LDS passes are stable), gated by the mode word like the
//           keys-only form; the safety net is the same cooperative LSD sort.
// Tickets give tile indices in arrival order, so a tile only ever waits for tiles that already run.
// ------------------------------------------------------------------------------------------
// 16 one-wave workgroups (see msd2_sample_kernel): sample 2048 keys -> digit placement; clear the tickets of both passes.  Only the
// low sort_bits bits of a key take part in the sort (Pprims.cpp:357: the passes cover bits [0, sortBits)); the digits are placed
// inside them, and the sample is looked at as it counts for this sort: the KEY of a pair, masked to the sorted bits.  The last
// workgroup to arrive turns the four sample words into `place` and puts them back to their idle values.
template <typename E, bool KEY64>
__global__ __launch_bounds__(64) void msd2s_prep_kernel(const E* __restrict__ src, uint32_t n, StablePlace* __restrict__ place,
                                                         uint32_t* __restrict__ tickets, uint32_t ticket_words, uint32_t* bar,
                                                         uint32_t* fault, uint32_t sort_bits, uint32_t* flag, uint32_t dup_thr)
{
    const int tid = (int)(blockIdx.x * 64u + threadIdx.x);
    if (tid == 0) {
        __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the safety net's grid-barrier counter
        fault[0] = 0u;   // the live fault word: the first kernel of every sort clears it (the look-back's waiters poll it)
        __hip_atomic_store(flag + kSampleThresholdWord, dup_thr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long kmask = sort_bits >= 64u ? ~0ull : ((1ull << sort_bits) - 1ull);
    const E x0 = src[sample_position(2 * tid, n)], x1 = src[sample_position(2 * tid + 1, n)];
    for (uint32_t i = (uint32_t)tid; i < ticket_words; i += 64u * (uint32_t)kSampleWGs) tickets[i] = 0u;
    uint32_t* sample = flag + 8;
    sample_accumulate(key_of<KEY64>(x0) & kmask, key_of<KEY64>(x1) & kmask, sample, flag);
    // the workgroup that arrives last finds all sixteen contributions in the sample words (agent-scope atomics, each wave's complete
    // at its s_waitcnt before it arrives)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t arrived = 0u;
    if (threadIdx.x == 0) arrived = __hip_atomic_fetch_add(flag + kSampleArriveWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)arrived) != (uint32_t)kSampleWGs - 1u) return;
    if (threadIdx.x == 0) {
        __hip_atomic_store(flag + kSampleArriveWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long o = ((unsigned long long)__hip_atomic_load(sample + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) |
                                     __hip_atomic_load(sample + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long a = ((unsigned long long)__hip_atomic_load(sample + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) |
                                     __hip_atomic_load(sample + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sample + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sample + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sample + 2, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sample + 3, ~0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long diff = o ^ a;
        uint32_t top = diff ? 64u - (uint32_t)__builtin_clzll(diff) : 0u;
        if (top < 16u) top = 16u;
        place->top = top;
        place->prefix = top < 64u ? (o >> top) : 0ull;
        place->low_bits = top - 16u;
        place->sort_bits = sort_bits;
        place->kmask = kmask;
    }
}

template <typename E>
struct LookbackPass {
    const E* src;
    E* dst;
    uint32_t* status;             // [chains * rows_per_chain][256] status rows, zero on entry
    uint32_t status_bytes;
    uint32_t* tickets;            // one counter per chain, kTicketStride words apart, zero on entry
    uint32_t* flag;               // set when a run does not fit its slab, or a key lies outside the sampled range
    uint32_t* fault;
    const StablePlace* place;
    int which_digit;              // 1: pass A, 2: pass B
    uint32_t n;
    uint32_t chains;              // pass A: pieces | pass B: 256
    uint32_t pieces;              // chains of pass A = sub-slabs per bucket (16; up to 32 would work)
    uint32_t rows_per_chain;      // status rows of a chain = the most tiles it can have
    uint32_t slice;               // pass A: elements per chain (a multiple of the tile)
    uint32_t src_stride;          // pass B: elements between two sub-slabs of pass A
    const uint32_t* status_a;     // pass B: pass A's status rows (sub-slab sizes = last row of each of its chains) ...
    uint32_t rows_per_chain_a;    // ... rows_per_chain of pass A
    uint32_t dst_stride;          // elements per destination slab
    uint32_t dst_total;
    const uint32_t* soa_keys;     // pass A of an SoA sort: the input is two u32 arrays (src unused), packed to {key, value} on load
    const uint32_t* soa_vals;
    int dst16;                    // pass B of u32 keys: the destination slabs hold uint16_t -- a segment's keys share everything above
                                  // their low 16 bits (BucketPass::dst16)
    uint32_t seg_shift;           // pass B: width w of the second digit (slot_to_segment); 8 otherwise
};

template <typename E, int NT, int K, bool KEY64, int RANK = 1>
__global__ __launch_bounds__(NT) void msd_lookback_scatter_kernel(LookbackPass<E> a)
{
    using C = TileCfg<E, 8, NT, K>;
    typedef AosIO<E> IO;
    constexpr int BINS = 256;
    constexpr int NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem + C::OFF_ELEMS);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);   // [NW][BINS]
    uint32_t* __restrict__ s_goff = reinterpret_cast<uint32_t*>(smem + C::OFF_GOFF);   // [BINS]
    uint32_t* __restrict__ s_misc = reinterpret_cast<uint32_t*>(smem + C::OFF_MISC);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    // pass A: workgroup i takes chain i % 16 (and runs on XCD i % 8: a chain stays on one XCD).  Pass B: bucket-major, the
    // workgroups that run at the same time work on the same few buckets and write into the same few hundred segment slabs
    // (chain = i % 256, which spreads every moment's writes over all 65536 slabs, measured 0.299 vs 0.272 ms at 64 Mi pairs and
    // 0.637 vs 0.527 ms at 128 Mi)
    const uint32_t chain = a.which_digit == 2 ? blockIdx.x / a.rows_per_chain : blockIdx.x % a.chains;
    // (pass B with a narrow second digit: the field sits 8 - w bits higher, see slot_to_segment)
    const int start_bit = (int)a.place->top - 8 * a.which_digit + (a.which_digit == 2 ? 8 - (int)a.seg_shift : 0);

    // ---- ticket -> tile index in the chain -> where the tile's elements are ------------------------------------------------
    if (w == 0) {
        // once the overflow flag is up (the sample's, pass A's, or an earlier tile's of this pass): the net will sort (net_sort); wave 0
        // decides for the workgroup -- no ticket, valid = 0, everybody leaves behind the barrier below
        // (pass A too: the sample may have raised the flag before the first tile, and a tile that takes its ticket after the flag rose
        // has only successors that do the same -- nobody waits for its status row)
        const bool give_up = __builtin_amdgcn_readfirstlane((int)large_sort_gave_up(a.flag)) != 0;
        uint32_t index = 0u;
        if (lane == 0 && !give_up) index = atomicAdd(&a.tickets[chain * (uint32_t)kTicketStride], 1u);
        index = (uint32_t)__builtin_amdgcn_readfirstlane((int)index);
        uint32_t base = 0u, valid = 0u;
        if (give_up) {
            // (nothing: valid stays 0)
        } else if (a.which_digit == 1) {
            const uint32_t c0 = chain * a.slice;
            const uint32_t c1 = c0 + a.slice < a.n ? c0 + a.slice : a.n;
            const uint32_t off = index * (uint32_t)C::TILE;
            if (c0 < a.n && off < c1 - c0) {
                base = c0 + off;
                valid = c1 - base < (uint32_t)C::TILE ? c1 - base : (uint32_t)C::TILE;
            }
        } else {
            // sub-slab (chain, c), c = lane < pieces (<= 32): its size is the inclusive prefix of digit `chain` in the last row of pass A's
            // chain c; the tile index counts tiles over the 16 sub-slabs in order
            uint32_t cnt = 0u;
            if ((uint32_t)lane < a.pieces) {
                const uint32_t c0 = (uint32_t)lane * a.slice;
                if (c0 < a.n) {
                    const uint32_t len = (c0 + a.slice < a.n ? c0 + a.slice : a.n) - c0;
                    const uint32_t rows = (len + (uint32_t)C::TILE - 1u) / (uint32_t)C::TILE;
                    cnt = a.status_a[((size_t)lane * a.rows_per_chain_a + rows - 1u) * BINS + chain] & kValMask;
                    if (cnt > a.src_stride) cnt = a.src_stride;   // overflowed in pass A: the flag is set, only stay in bounds
                }
            }
            // a tile is C::TILE consecutive elements of the bucket = of its sub-slabs one after the other: it may start in one
            // sub-slab and end in another (tiles cut at sub-slab ends left every second sub-slab of 64 Mi pairs with a third,
            // nearly empty tile, and small inputs with nothing but part-filled tiles).  Tables for the load below, in the
            // s_goff area (written again only after the barrier behind the ranking): where sub-slab c ends in the bucket,
            // and what turns a position in the bucket into an index into slab_a.
            const uint32_t incl = wave_incl_scan_u32(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (lane < 32) {
                const bool real = (uint32_t)lane < a.pieces;
                s_goff[lane] = real ? incl : 0xffffffffu;
                s_goff[32 + lane] = real ? (chain * a.pieces + (uint32_t)lane) * a.src_stride - (incl - cnt) : 0u;
            }
            base = index * (uint32_t)C::TILE;   // position in the bucket
            if (base < total) valid = total - base < (uint32_t)C::TILE ? total - base : (uint32_t)C::TILE;
            // a tile that lies inside ONE sub-slab (most tiles of a large input) is loaded like a tile of pass A
            const bool holds = (uint32_t)lane < a.pieces && base >= incl - cnt && base + valid <= incl && valid != 0u;
            const unsigned long long hm = __ballot(holds);
            uint32_t lin = 0xffffffffu;
            if (hm) {
                const int c = __builtin_ctzll(hm);
                const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)(incl - cnt), c);
                lin = (chain * a.pieces + (uint32_t)c) * a.src_stride + (base - first);
            }
            if (lane == 0) s_misc[3] = lin;
        }
        if (lane == 0) {
            s_misc[0] = index;
            s_misc[1] = base;
            s_misc[2] = valid;
        }
    }
    __syncthreads();
    const uint32_t index = s_misc[0];
    const uint32_t base = s_misc[1];
    const uint32_t valid = s_misc[2];
    if (valid == 0u) return;   // a ticket beyond the chain's tiles
    const uint32_t first_row = chain * a.rows_per_chain;
    const uint32_t row = first_row + index;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.status, 0, (int)a.status_bytes, 0x00020000);
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    const IO io{a.src, a.dst};
    const bool scaled = dst_fits32<IO>(a.dst_total);

    // ---- load, wave-striped; slots beyond `valid` are all-ones pads (digit 255, highest tile positions, never stored) ----
    const uint32_t wbase = (uint32_t)(w * 64 * K + lane);
    E e[K];
    const uint32_t lin = a.which_digit == 1 ? base : s_misc[3];   // index of the tile's first element if it is one stretch
    if (sizeof(E) == 8 && a.soa_keys) {
        if constexpr (sizeof(E) == 8) {
            const int rem = (int)valid - (int)wbase;
            const uint32_t* __restrict__ kp = a.soa_keys + (size_t)lin + wbase;
            const uint32_t* __restrict__ vp = a.soa_vals + (size_t)lin + wbase;
#pragma unroll
            for (int j = 0; j < K; ++j)
                e[j] = (j * 64 < rem) ? ((E)kp[j * 64] | ((E)vp[j * 64] << 32)) : ~E(0);
        }
    } else if (lin != 0xffffffffu) {
        const typename IO::Cursor p = io.cursor((size_t)lin + wbase);
        if (valid == (uint32_t)C::TILE) {
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = p.at(j * 64);
        } else {
            const int rem = (int)valid - (int)wbase;
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = (j * 64 < rem) ? p.at(j * 64) : ~E(0);
        }
    } else {
        // position in the bucket -> sub-slab: found once for the lane's first element, then carried along (positions rise)
        const uint32_t* __restrict__ s_end = s_goff;
        const uint32_t* __restrict__ s_adj = s_goff + 32;
        const uint32_t q0 = base + wbase;
        uint32_t c = 0u;
        for (uint32_t i = 0; i < a.pieces; ++i) c += (q0 >= s_end[i]) ? 1u : 0u;
        const int rem = (int)valid - (int)wbase;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const uint32_t q = q0 + (uint32_t)(j * 64);
            while (c < 31u && q >= s_end[c]) ++c;
            e[j] = (j * 64 < rem) ? load_once(a.src + (size_t)(q + s_adj[c])) : ~E(0);
        }
    }
    if (a.which_digit == 1 && a.place->top < (KEY64 ? 64u : 32u)) {   // a key outside the sampled range would land in a wrong bucket
        const uint32_t top = a.place->top;
        const int rem = (int)valid - (int)wbase;
        if constexpr (KEY64) {
            const unsigned long long pre = a.place->prefix, km = a.place->kmask;
            unsigned long long bad = 0ull;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j * 64 < rem) bad |= (((unsigned long long)e[j] & km) >> top) ^ pre;
            if (bad) __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const uint32_t pre = (uint32_t)a.place->prefix, km = (uint32_t)a.place->kmask;
            uint32_t bad = 0u;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j * 64 < rem) bad |= (((uint32_t)e[j] & km) >> top) ^ pre;
            if (bad) __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- rank ---------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int b = lane; b < BINS; b += 64) my_wcnt[b] = 0u;
    uint32_t rnk2[(K + 1) / 2];
    {
        uint32_t rnk[K];
        rank_in_wave<E, 8, K, RANK>(e, rnk, my_wcnt, start_bit);
#pragma unroll
        for (int j = 0; j < K; j += 2) rnk2[j >> 1] = rnk[j] | ((j + 1 < K ? rnk[j + 1] : 0u) << 16);
    }
#pragma unroll
    for (int j = 0; j < (K + 1) / 2; ++j) asm volatile("" : "+v"(rnk2[j]));
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(e[j]));
    __syncthreads();
    // ---- every wave: counts of all waves for its lanes' digits -> tile offsets -> its own (wave, digit) positions; wave 0
    // publishes the tile's counts at once (the chain's first tile: they are its prefix) ---------------------------------------
    u32x4 cnt4 = {0u, 0u, 0u, 0u};
    u32x4 toff4 = {0u, 0u, 0u, 0u};
    uint16_t* __restrict__ my_wpos = reinterpret_cast<uint16_t*>(smem + C::OFF_WPOS) + w * BINS;
    u32x4 real4;
    {
        u32x4 pre4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const u32x4 r = *reinterpret_cast<const u32x4*>(s_wcnt + i * BINS + 4 * lane);
            cnt4 += r;
            if (i < w) pre4 += r;
        }
        real4 = cnt4;
        if (lane == 63) real4.w -= (uint32_t)C::TILE - valid;   // the pads sit under digit 255
        if (w == 0)
            __builtin_amdgcn_raw_buffer_store_b128(real4 | (index == 0u ? kFlagPfx : kFlagAgg), rsrc,
                                                   (row * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u, 0, 16 /* sc1 */);
        const uint32_t s4 = cnt4.x + cnt4.y + cnt4.z + cnt4.w;
        const uint32_t ex = wave_incl_scan_u32(s4) - s4;
        toff4.x = ex;
        toff4.y = ex + cnt4.x;
        toff4.z = toff4.y + cnt4.y;
        toff4.w = toff4.z + cnt4.z;
        const u32x4 p4 = toff4 + pre4;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 packed = {p4.x | (p4.y << 16), p4.z | (p4.w << 16)};
        *reinterpret_cast<u32x2*>(my_wpos + 4 * lane) = packed;
    }
    // ---- scatter into tile-sorted order ---------------------------------------------------------------------------------------------
    {
        constexpr int CH = K < 8 ? K : 8;
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += CH) {
            uint32_t pos[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) pos[j] = my_wpos[digit_of<8>(e[(j0 + j < K) ? j0 + j : K - 1], start_bit)];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j0 + j < K) {
                    const uint32_t r = (rnk2[(j0 + j) >> 1] >> (16 * ((j0 + j) & 1))) & 0xffffu;
                    s_elems[pos[j] + r] = e[j0 + j];
                }
            }
        }
    }
    // ---- look-back (as late as possible), room check, destinations ----------------------------------------------------------
    if (w == 0) {
        u32x4 excl = {0u, 0u, 0u, 0u};
        if (index != 0u) {
            excl = lookback_exclusive4<BINS, kLookbackWindow>(rsrc, row, first_row, lane, a.fault, start_bit);
            __builtin_amdgcn_raw_buffer_store_b128(((excl + real4) & kValMask) | kFlagPfx, rsrc,
                                                   (row * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u, 0, 16);
        }
        const u32x4 end4 = excl + real4;
        const bool over = (end4.x > a.dst_stride) | (end4.y > a.dst_stride) | (end4.z > a.dst_stride) | (end4.w > a.dst_stride);
        if (over) __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // slab of digit d: pass A (d * pieces + chain), pass B (chain * 256 + d)
        const uint32_t d0 = 4u * (uint32_t)lane;
        u32x4 sl;
        if (a.which_digit == 1) {
            sl.x = (d0 + 0u) * a.pieces + chain; sl.y = (d0 + 1u) * a.pieces + chain; sl.z = (d0 + 2u) * a.pieces + chain;
            sl.w = (d0 + 3u) * a.pieces + chain;
        } else {   // slot (bucket << w) | (digit & (2^w - 1)); w = 8: bucket * 256 + digit
            sl.x = (chain << a.seg_shift) + (d0 & ((1u << a.seg_shift) - 1u)); sl.y = sl.x + 1u; sl.z = sl.x + 2u; sl.w = sl.x + 3u;
        }
        // an overflowing run would be written beyond its slab: park it at the slab's start instead (the result is discarded)
        u32x4 in4 = excl;
        if (end4.x > a.dst_stride) in4.x = 0u;
        if (end4.y > a.dst_stride) in4.y = 0u;
        if (end4.z > a.dst_stride) in4.z = 0u;
        if (end4.w > a.dst_stride) in4.w = 0u;
        const u32x4 go = sl * a.dst_stride + in4 - toff4;
        *reinterpret_cast<u32x4*>(s_goff + 4 * lane) = (scaled && !a.dst16) ? go * (uint32_t)IO::kStoreScale : go;
    }
    __syncthreads();
    if constexpr (sizeof(E) == 4) {
        if (a.dst16) {   // one 2-byte store per key
            uint16_t* __restrict__ d16 = reinterpret_cast<uint16_t*>(a.dst);
#pragma unroll 8
            for (int i = 0; i < K; ++i) {
                const uint32_t pos = (uint32_t)(tid + i * NT);
                if (pos < valid) {
                    const E v = s_elems[pos];
                    const uint32_t g = s_goff[digit_of<8>(v, start_bit)] + pos;
                    if (g < a.dst_total) d16[g] = (uint16_t)v;
                }
            }
            return;
        }
    }
    write_out_tile<IO, 8, NT, K, ADLHIP_WRITE_UNROLL>(io, s_elems, s_goff, valid, a.dst_total, start_bit, scaled);
}

// Between pass B and the finish: workgroup b turns bucket b's final counts (the last status row of chain b) into the
// segments' sizes and output offsets; the last workgroup publishes the mode word.
// As in the keys-only form, the safety net runs in this kernel when the overflow flag is set.
template <typename E, int TILE, int NT, int K, int RANK = 1, int P = 0>
__global__ __launch_bounds__(NT) void msd2s_offsets_kernel(const uint32_t* __restrict__ status_a, uint32_t rows_per_chain_a, uint32_t slice,
                                                            uint32_t pieces,
                                                            const uint32_t* __restrict__ status_b, uint32_t rows_per_chain_b,
                                                            uint32_t src_stride, uint32_t* flag, uint32_t* done, uint32_t* bar,
                                                            uint32_t* __restrict__ seg_cnt, uint32_t* __restrict__ seg_off,
                                                            uint32_t* __restrict__ mode, uint32_t n,
                                                            const StablePlace* __restrict__ place, E* data, E* tmp,
                                                            uint32_t* __restrict__ ctable, uint32_t* fault, uint32_t* soa_keys,
                                                            uint32_t* soa_vals, uint32_t* cursors_b /* hybrid form, else nullptr */,
                                                            uint32_t d2_shift /* 8 - w (slot_to_segment); stable second pass: 0 */,
                                                            DictBlock* dict /* whole-key sorts of keys: the net's counting sort, else nullptr */,
                                                            uint32_t* stats, OsNet os)
{
    static_assert(NT >= 256, "one thread per digit");
    __shared__ uint32_t s_wsum[NT / 64 + 1];
    __shared__ uint32_t s_misc[4];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // the safety net's tile (TileCfg<E, 8, NT, K>)
    const int t = (int)threadIdx.x;
    const uint32_t b = blockIdx.x;   // at least 256 workgroups: workgroup b < 256 serves bucket b
    // final since pass B has completed; read by every workgroup BEFORE it counts itself done, cleared by the last one done
    const uint32_t overflow = large_sort_gave_up(flag);
    const uint32_t sample_repeats = __hip_atomic_load(flag + kSampleRepeatsWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (b < 256u) {
    // thread t: size of bucket t = sum of its 16 sub-slabs; and the tiles pass B made of it
    uint32_t size_t_ = 0u, tiles_t = 0u;
    for (uint32_t c = 0; c < pieces && t < 256; ++c) {
        const uint32_t c0 = c * slice;
        if (c0 >= n) break;
        const uint32_t len = (c0 + slice < n ? c0 + slice : n) - c0;
        const uint32_t rows = (len + (uint32_t)TILE - 1u) / (uint32_t)TILE;
        uint32_t cnt = status_a[((size_t)c * rows_per_chain_a + rows - 1u) * 256u + (uint32_t)t] & kValMask;
        if (cnt > src_stride) cnt = src_stride;
        size_t_ += cnt;
    }
    tiles_t = (size_t_ + (uint32_t)TILE - 1u) / (uint32_t)TILE;   // pass B's tiles run across the sub-slabs
    const uint32_t exa = block_excl_scan_u32<NT>(size_t_, s_wsum, nullptr);
    if (t == (int)b) {
        s_misc[0] = exa;
        s_misc[1] = tiles_t;
    }
    __syncthreads();
    const uint32_t tiles_b = s_misc[1];
    uint32_t cb = 0u;
    if (t < 256) {
        if (cursors_b) {   // hybrid form: the second pass placed its runs with cursors [bucket][digit]; they go back to zero here
            cb = __hip_atomic_load(cursors_b + b * 256u + (uint32_t)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(cursors_b + b * 256u + (uint32_t)t, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            cb = tiles_b ? (status_b[((size_t)b * rows_per_chain_b + tiles_b - 1u) * 256u + (uint32_t)t] & kValMask) : 0u;
        }
    }
    const uint32_t exb = block_excl_scan_u32<NT>(cb, s_wsum, nullptr);
    if (t < 256) {
        seg_cnt[b * 256u + (uint32_t)t] = cb;
        seg_off[b * 256u + (uint32_t)t] = s_misc[0] + exb;
    }
    }
    if (t == 0) s_misc[2] = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_misc[2] == gridDim.x - 1u && t == 0) {
        *mode = overflow ? 1u : 0u;
        mode[kDynLowBits] = place->low_bits + d2_shift;
        mode[kDynLowBits + 1] = (uint32_t)place->prefix;   // for the finish of a 16-bit second slab
        mode[kDynHardCnt] = 0u;
        seg_off[65536] = n;
        __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(flag + kSampleRepeatsWord, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (overflow) {   // `bar` is zero here: msd2s_prep_kernel, the first launch of every sort, clears it
        __syncthreads();
        uint32_t target = 0u;
        if constexpr (sizeof(E) == 8) {
            if (soa_keys) {   // SoA input: pack it into `data` (here: the first slab area), sort that, unpack
                for (size_t i = (size_t)blockIdx.x * NT + (size_t)t; i < n; i += (size_t)gridDim.x * NT)
                    data[i] = (E)soa_keys[i] | ((E)soa_vals[i] << 32);
                if (!grid_barrier(bar, target, gridDim.x, fault)) return;
            }
        }
        net_sort<E, NT, K, RANK, P>(data, tmp, n, ctable, bar, fault, smem, (int)place->sort_bits, dict, stats, os, sample_repeats, target,
                                    soa_keys != nullptr);
        if constexpr (sizeof(E) == 8) {
            if (soa_keys) {   // the sort's last phase ends with a grid barrier: `data` is complete
                for (size_t i = (size_t)blockIdx.x * NT + (size_t)t; i < n; i += (size_t)gridDim.x * NT) {
                    const E x = data[i];
                    soa_keys[i] = (uint32_t)x;
                    soa_vals[i] = (uint32_t)(x >> 32);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Mid-size sort (16 Ki < n <= 2 Mi keys): THREE launches instead of the 8-12 dependent launches of the per-digit
// passes, which is what these sizes -- every size of the reference's own test, UnitTest/main.cpp:105 -- pay for
// (~3 us per kernel boundary; profiles/r1_ncurve.txt).
//   1. mid_prep_kernel      histograms of all four key bytes; the last workgroup to finish picks the MOST SIGNIFICANT
//                           BYTE THAT VARIES (keys below 2^24, small indices, ... would otherwise all share one
//                           bucket), and builds the offset tables of pass 2 and the 257 bucket bounds of pass 3
//   2. onesweep_chain_kernel one MSD pass on that byte (16 look-back chains = 16 slices of the input), data -> tmp
//   3. segment_sort_kernel  every bucket finished in LDS on the bytes below, tmp -> data
// Skewed keys can leave a bucket larger than the LDS tile: mid_prep_kernel then sets MidDyn::mode, pass 2 returns at
// once and the workgroups of pass 3 run a cooperative 4 x 8-bit LSD sort instead ("sort.mid" = 0 turns the path off).
// ------------------------------------------------------------------------------------------
constexpr int kMidChunk = 8192;       // elements per histogram workgroup (two 4 Ki-element tiles of pass 2)
constexpr int kMidPrepNT = 256;

struct MidDyn {          // written by mid_prep_kernel, read by the kernels after it
    uint32_t start_bit;  // first bit of the MSD byte
    uint32_t low_bits;   // bits below it, finished in LDS
    uint32_t mode;       // 0: passes 2 and 3 run (every bucket fits the LDS tile); 1: they return at once and the
                         // cooperative LSD kernel sorts instead (skewed keys: a bucket would not fit)
    uint32_t max_bucket; // largest bucket (diagnostic)
};

template <typename E>
__global__ __launch_bounds__(kMidPrepNT) void mid_prep_kernel(const E* __restrict__ src, uint32_t n, uint32_t tile,
                                                              uint32_t* __restrict__ slice_hist /*[16][4][256], zero on entry*/,
                                                              PassTable* __restrict__ table, uint32_t* __restrict__ seg_start,
                                                              MidDyn* __restrict__ dyn, u32x4* __restrict__ tickets,
                                                              u32x4* __restrict__ status, uint32_t status_vecs,
                                                              uint32_t* __restrict__ done, uint32_t* __restrict__ fault,
                                                              uint32_t bucket_cap, uint32_t* __restrict__ host_mode)
{
    __shared__ uint32_t hist[4][256];
    __shared__ uint32_t s_wsum[kMidPrepNT / 64 + 1];
    __shared__ uint32_t s_misc[8];
    const int tid = (int)threadIdx.x;
    const uint32_t wgs = gridDim.x;
    const uint32_t wps = (wgs + (uint32_t)kChains - 1u) / (uint32_t)kChains;   // workgroups per slice (= chain of pass 2)
    {   // tickets and status rows of pass 2, the live fault word
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (blockIdx.x == 0) {
            for (int i = tid; i < kChains * kTicketStride * 4 / 16; i += kMidPrepNT) tickets[i] = z;
            if (tid == 0) fault[0] = 0u;
        }
        for (uint32_t i = blockIdx.x * kMidPrepNT + (uint32_t)tid; i < status_vecs; i += wgs * kMidPrepNT) status[i] = z;
    }
    for (int i = tid; i < 4 * 256; i += kMidPrepNT) (&hist[0][0])[i] = 0u;
    __syncthreads();
    const uint32_t begin = blockIdx.x * (uint32_t)kMidChunk;
    const uint32_t end = (begin + (uint32_t)kMidChunk < n) ? begin + (uint32_t)kMidChunk : n;
    auto bump = [&](uint32_t k) {
        atomicAdd(&hist[0][k & 255u], 1u);
        atomicAdd(&hist[1][(k >> 8) & 255u], 1u);
        atomicAdd(&hist[2][(k >> 16) & 255u], 1u);
        atomicAdd(&hist[3][k >> 24], 1u);
    };
    {
        constexpr int VEC = 16 / (int)sizeof(E);
        struct alignas(16) Vec { E v[VEC]; };
        const uint32_t nvec = (end - begin) / VEC;
        const Vec* vsrc = reinterpret_cast<const Vec*>(src + begin);   // begin is a multiple of the chunk: 16-byte aligned
        for (uint32_t i = (uint32_t)tid; i < nvec; i += kMidPrepNT) {
            const Vec v = vsrc[i];
#pragma unroll
            for (int k = 0; k < VEC; ++k) bump((uint32_t)v.v[k]);
        }
        for (uint32_t i = begin + nvec * VEC + (uint32_t)tid; i < end; i += kMidPrepNT) bump((uint32_t)src[i]);
    }
    __syncthreads();
    // add this workgroup's counts to its slice's row (a slice = `wps` consecutive workgroups = one chain of pass 2)
    uint32_t* row = slice_hist + (size_t)(blockIdx.x / wps) * 1024;
    for (int i = tid; i < 1024; i += kMidPrepNT) {
        const uint32_t c = (&hist[0][0])[i];
        if (c) atomicAdd(row + i, c);
    }
    // ---- the last workgroup to get here builds the tables (every other one is done) ------------------------------
    // The rows are only ever touched by agent-scope atomics (performed at the memory side, never cached in an L1 or a
    // non-coherent L2), so all the hand-off needs is that this workgroup's adds have been acknowledged before it counts
    // itself done: no cache write-back or invalidate (an agent-scope fence pair costs ~3.5 us, MI355X_MICROARCH.md).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_misc[0] = atomicAdd(done, 1u);
    __syncthreads();
    if (s_misc[0] != wgs - 1u) return;
    // thread t owns value t of every byte: the sixteen slice rows in ONE round trip, then the rows are cleared for the
    // next sort on this handle (the area belongs to the device handle and is zero between sorts)
    uint32_t sl[kChains][4];
#pragma unroll
    for (int c = 0; c < kChains; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            sl[c][k] = __hip_atomic_load(slice_hist + (size_t)c * 1024 + k * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int c = 0; c < kChains; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            __hip_atomic_store(slice_hist + (size_t)c * 1024 + k * 256 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0) {
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // grid-barrier counter of the cooperative LSD kernel
    }
    uint32_t tot[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < kChains; ++c)
#pragma unroll
        for (int k = 0; k < 4; ++k) tot[k] += sl[c][k];
    int b = 0;   // most significant byte with more than one value present (all keys equal: byte 0)
#pragma unroll
    for (int k = 3; k >= 1; --k) {
        const int present = __syncthreads_count(tot[k] != 0u);
        if (b == 0 && present > 1) b = k;
    }
    const uint32_t total_b = b == 3 ? tot[3] : (b == 2 ? tot[2] : (b == 1 ? tot[1] : tot[0]));
    const uint32_t gbase = block_excl_scan_u32<kMidPrepNT>(total_b, s_wsum, nullptr);
    seg_start[tid] = gbase;
    if (tid == 0) seg_start[256] = n;
    {   // largest bucket (diagnostic)
        uint32_t mx = total_b;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)mx, o);
            mx = other > mx ? other : mx;
        }
        if ((tid & 63) == 0) s_misc[1 + (tid >> 6)] = mx;
        __syncthreads();
        if (tid == 0) {
            uint32_t m4 = s_misc[1];
            for (int i = 1; i < kMidPrepNT / 64; ++i) m4 = s_misc[1 + i] > m4 ? s_misc[1 + i] : m4;
            dyn->start_bit = 8u * (uint32_t)b;
            dyn->low_bits = 8u * (uint32_t)b;
            dyn->mode = m4 > bucket_cap ? 1u : 0u;
            dyn->max_bucket = m4;
            // tell the host (pinned memory, read without synchronising at its next call): 1 + mode of this sort
            __hip_atomic_store(host_mode, 1u + (m4 > bucket_cap ? 1u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    uint32_t run = gbase;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
        table->cbase[c][tid] = run;
        run += b == 3 ? sl[c][3] : (b == 2 ? sl[c][2] : (b == 1 ? sl[c][1] : sl[c][0]));
    }
    if (tid <= kChains) {
        const uint64_t e0 = (uint64_t)tid * wps * (uint64_t)kMidChunk;
        const uint32_t es = e0 < n ? (uint32_t)e0 : n;
        table->chunk_start[tid] = es;
        table->tile_start[tid] = (es + tile - 1u) / tile;   // slices are whole tiles except the last one
    }
}

}  // namespace adlhip
