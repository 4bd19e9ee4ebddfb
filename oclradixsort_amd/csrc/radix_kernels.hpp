// radix_kernels.hpp -- gfx950 (CDNA4, wave64) device code for the LSD radix-sort hot path.
//
// Written for MI355X only: 64-lane wavefronts are assumed everywhere (ballot masks are 64-bit,
// mbcnt prefix counts, DS ops of one wave execute in issue order).  No MFMA: this is integer,
// HBM-bandwidth-bound work.
//
// Reference behaviour reproduced (all paths relative to the reference repository):
//   Tahoe/ClKernels/RadixSort32Kernels.cl:176-236   StreamCountKernel      -> radix_count_kernel
//   Tahoe/ClKernels/RadixSort32Kernels.cl:243-362   PrefixScan{16,32}PerWi -> radix_scan_table_kernel
//   Tahoe/ClKernels/RadixSort32Kernels.cl:493-631   SortAndScatterKernel   -> radix_scatter_kernel
//   Tahoe/ClKernels/RadixSortKeyValueKernels.cl:182-248, :511-663 (key+value variants): same kernels,
//   instantiated for 8-byte elements whose key is the low dword.
// Only the behaviour is shared (per-workgroup contiguous runs, bucket-major table, stable local
// sort, carry per digit -- SURVEY.md Appendix A.1/A.2); the mechanism is different throughout:
// the in-tile ranking is a 64-lane ballot match (no packed counters, no LDS scans), the digit
// width is a template parameter (8 bits by default, 4 = the reference's), tiles are 4-16 K keys.
#pragma once
// non-template kernels: external linkage in the translation unit that launches them (adlhip.hip), internal (and so dropped, unused)
// in the one that only instantiates kernel templates (kernels_large.hip)
#ifndef ADLHIP_KERNEL
#define ADLHIP_KERNEL
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace adlhip {

// ------------------------------------------------------------------------------------------
// small wave / block helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// DPP move with zero fill: lanes whose source is outside the row / masked off receive 0.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_mov0(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

// 64-lane inclusive scan in six v_add_u32_dpp (no LDS traffic): row_shr 1/2/4/8 scans each row of 16,
// row_bcast:15 carries row 0 -> 1 and 2 -> 3, row_bcast:31 carries the first half into rows 2 and 3.
// All 64 lanes must be active.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += dpp_mov0<0x111, 0xf, 0xf>(v);   // row_shr:1
    v += dpp_mov0<0x112, 0xf, 0xf>(v);   // row_shr:2
    v += dpp_mov0<0x114, 0xf, 0xf>(v);   // row_shr:4
    v += dpp_mov0<0x118, 0xf, 0xf>(v);   // row_shr:8
    v += dpp_mov0<0x142, 0xa, 0xf>(v);   // row_bcast:15 -> rows 1, 3
    v += dpp_mov0<0x143, 0xc, 0xf>(v);   // row_bcast:31 -> rows 2, 3
    return v;
}

// Exclusive scan across the block of one value per thread.  `wsum` is LDS scratch of NT/64 + 1
// words.  Contains two barriers; every thread of the block must call it.  Returns the exclusive
// prefix; *total (if non-null) receives the block total.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* wsum, uint32_t* total)
{
    constexpr int NW = NT / 64;
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    uint32_t inc = wave_incl_scan_u32(v);
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t t = wsum[i];
        if (i < w) woff += t;
        tot += t;
    }
    __syncthreads();   // wsum may be reused by the caller right away
    if (total) *total = tot;
    return woff + inc - v;
}

// Digit extraction.  Digits never straddle a 32-bit word: passes start at multiples of 4 and an
// 8-bit pass starts at a multiple of 8, so for 64-bit elements the word select is wave-uniform.
template <int NBITS>
__device__ __forceinline__ uint32_t digit_of(uint32_t e, int start_bit)
{
    return (e >> start_bit) & ((1u << NBITS) - 1u);
}
template <int NBITS>
__device__ __forceinline__ uint32_t digit_of(uint64_t e, int start_bit)
{
    return (uint32_t)(e >> start_bit) & ((1u << NBITS) - 1u);   // any position: a digit may straddle bit 32 (placed digits)
}

// 64-lane "match any": the mask of lanes whose digit equals this lane's.  One ballot per digit bit;
// every lane of the wave must be active.
template <int NBITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t d)
{
    uint64_t m = ~0ull;
#pragma unroll
    for (int b = 0; b < NBITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// popcount(mask & lanes below me)
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ------------------------------------------------------------------------------------------
// Diagnostic build only (-DADLHIP_STAMPS, never in the shipped library): wave 0 / lane 0 of every
// tile records s_memtime at phase boundaries into a side buffer that no kernel reads.
// ------------------------------------------------------------------------------------------
#ifdef ADLHIP_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;   // [tiles][16]
__device__ __forceinline__ void stamp_at(uint32_t tile, int slot)
{
    if (threadIdx.x == 0 && g_stamp_buf) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        g_stamp_buf[(size_t)tile * 16 + slot] = t;
    }
}
#define ADLHIP_STAMP(tile, slot) stamp_at(tile, slot)
#else
#define ADLHIP_STAMP(tile, slot) ((void)0)
#endif

// ------------------------------------------------------------------------------------------
// Tile geometry shared by the scatter kernels
// ------------------------------------------------------------------------------------------
template <typename E, int NBITS, int NT, int K>
struct TileCfg {
    static constexpr int BINS = 1 << NBITS;
    static constexpr int NW = NT / 64;
    static constexpr int TILE = NT * K;
    static_assert(NT % 64 == 0, "block must be whole waves");
    static_assert(BINS <= NT, "one thread per bin");
    // LDS carve (dynamic shared memory, 16-byte aligned base; every offset a multiple of 16)
    static constexpr size_t OFF_ELEMS = 0;
    static constexpr size_t OFF_WCNT = OFF_ELEMS + sizeof(E) * TILE;
    static constexpr size_t OFF_GOFF = OFF_WCNT + sizeof(uint32_t) * NW * BINS;
    static constexpr size_t OFF_WSUM = OFF_GOFF + sizeof(uint32_t) * BINS;
    static constexpr size_t OFF_MISC = OFF_WSUM + 16 * ((NW + 1 + 3) / 4) * 4;
    // one-sweep pass: tile positions of the (wave, digit) runs as 16-bit values (a tile has fewer than 65536 elements)
    static_assert(TILE <= 65536, "16-bit tile positions");
    static constexpr size_t OFF_WPOS = OFF_MISC + 64;
    static constexpr size_t LDS_BYTES = OFF_WPOS + sizeof(uint16_t) * NW * BINS;
};

// Where a tile's elements come from and go to.
//   AosIO<E>: one array of E (u32 keys, u64 keys, or 8-byte {key, value} pairs).
//   SoaIO   : separate u32 key and u32 value arrays -- the layout of the reference's never-launched
//             SortAndScatterKernel(gSrc, gSrcVal, ...) (RadixSortKeyValueKernels.cl:354-509).
//             A pair travels through registers and LDS as one u64 {key low, value high}, so the tile body
//             is the one the AoS pairs use; only the global loads/stores differ.
// Key loads of the one-tile-per-workgroup sweeps.  Round 4 A/B (profiles/r4_nt_loads_ab.txt): non-temporal loads help a sweep that
// reads COLD data (first look-back pass of 64 Mi pairs 0.255 -> 0.245 ms) and hurt one that reads what the sweep before it has just
// written (wave-per-segment finish of pairs 0.198 -> 0.226 ms, hybrid second pass 0.227 -> 0.248): plain loads stay the default
// here (-DADLHIP_NT_LOADS=1 builds the partner); the persistent passes (persist_kernels.hpp) choose per pass.
#ifndef ADLHIP_NT_LOADS
#define ADLHIP_NT_LOADS 0
#endif
template <typename T>
__device__ __forceinline__ T load_once(const T* p)
{
#if ADLHIP_NT_LOADS
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}

template <typename E>
struct AosIO {
    typedef E elem_t;
    const E* src;
    E* dst;
    // cursor = one 64-bit pointer per tile; elements are then read at constant offsets from it
    struct Cursor {
        const E* p;
        __device__ __forceinline__ E at(int off) const { return load_once(p + off); }
    };
    __device__ __forceinline__ Cursor cursor(size_t base) const { return Cursor{src + base}; }
    __device__ __forceinline__ void store(size_t i, E v) const { dst[i] = v; }
    // Destination as a raw buffer: stores take a 32-bit BYTE offset and the hardware drops any store beyond
    // n elements (no 64-bit address arithmetic, no compare + exec masking per element).
    static constexpr uint32_t kStoreScale = (uint32_t)sizeof(E);   // bytes per element in a destination array
    struct Dst {
        __amdgpu_buffer_rsrc_t r;
    };
    __device__ __forceinline__ Dst make_dst(uint32_t n) const
    {
        return Dst{__builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(n * (uint32_t)sizeof(E)), 0x00020000)};
    }
    static __device__ __forceinline__ void store_at(const Dst& d, uint32_t byte_off, E v)
    {
        if constexpr (sizeof(E) == 4) {
            __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v, d.r, (int)byte_off, 0, 0);
        } else {
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t t = {(uint32_t)v, (uint32_t)((uint64_t)v >> 32)};
            __builtin_amdgcn_raw_buffer_store_b64(t, d.r, (int)byte_off, 0, 0);
        }
    }
};
struct SoaIO {
    typedef uint64_t elem_t;
    const uint32_t* ksrc;
    const uint32_t* vsrc;
    uint32_t* kdst;
    uint32_t* vdst;
    struct Cursor {
        const uint32_t* k;
        const uint32_t* v;
        __device__ __forceinline__ uint64_t at(int off) const { return (uint64_t)load_once(k + off) | ((uint64_t)load_once(v + off) << 32); }
    };
    __device__ __forceinline__ Cursor cursor(size_t base) const { return Cursor{ksrc + base, vsrc + base}; }
    __device__ __forceinline__ void store(size_t i, uint64_t v) const
    {
        kdst[i] = (uint32_t)v;
        vdst[i] = (uint32_t)(v >> 32);
    }
    static constexpr uint32_t kStoreScale = 4u;   // two arrays of 4-byte elements
    struct Dst {
        __amdgpu_buffer_rsrc_t k, v;
    };
    __device__ __forceinline__ Dst make_dst(uint32_t n) const
    {
        return Dst{__builtin_amdgcn_make_buffer_rsrc(kdst, 0, (int)(n * 4u), 0x00020000),
                   __builtin_amdgcn_make_buffer_rsrc(vdst, 0, (int)(n * 4u), 0x00020000)};
    }
    static __device__ __forceinline__ void store_at(const Dst& d, uint32_t byte_off, uint64_t v)
    {
        __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v, d.k, (int)byte_off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32((uint32_t)(v >> 32), d.v, (int)byte_off, 0, 0);
    }
};

// Buffer stores need every element of a destination array within reach of a 32-bit byte offset; they are USED
// for arrays up to 256 MiB, where they measured faster than pointer stores (64Mi u32 keys: +2.8 %, 32Mi: +3.5 %);
// at 512 MiB the two are level and at 1 GiB pointer stores win by 1.3 % (profiles/r1_pass_kernel_experiments.txt).
template <typename IO>
__device__ __forceinline__ bool dst_fits32(uint32_t n)
{
#ifdef ADLHIP_NO_BUFSTORE   // diagnostic builds: always take the pointer path
    return false;
#else
    return (uint64_t)n * IO::kStoreScale <= (256ull << 20);
#endif
}

// Write-out of a tile-sorted tile: consecutive threads -> consecutive tile positions -> contiguous runs per
// digit (reference paper eq. 2: dst = run start + position - tile offset of the digit).  s_goff[d] holds
// (run start - tile offset) of digit d, in BYTES of the destination array when `scaled` (the buffer-store
// path: one v_add3 per element, bounds checked by the hardware), in elements otherwise (arrays of 4 GiB and more).
template <typename IO, int NBITS, int NT, int K, int UNROLL>
__device__ __forceinline__ void write_out_tile(const IO& io, const typename IO::elem_t* s_elems, const uint32_t* s_goff,
                                               uint32_t valid, uint32_t n_total, int start_bit, bool scaled)
{
    typedef typename IO::elem_t E;
    constexpr uint32_t SC = IO::kStoreScale;
    const int tid = (int)threadIdx.x;
    if (scaled) {
        const typename IO::Dst dst = io.make_dst(n_total);
        const uint32_t tsc = (uint32_t)tid * SC;
        if (valid == (uint32_t)(NT * K)) {
#pragma unroll UNROLL
            for (int i = 0; i < K; ++i) {
                const E v = s_elems[tid + i * NT];
                const uint32_t d = digit_of<NBITS>(v, start_bit);
                IO::store_at(dst, s_goff[d] + tsc + (uint32_t)(i * NT) * SC, v);
            }
        } else {
#pragma unroll UNROLL
            for (int i = 0; i < K; ++i) {
                if (i * NT < (int)valid - tid) {   // tile position tid + i*NT exists
                    const E v = s_elems[tid + i * NT];
                    const uint32_t d = digit_of<NBITS>(v, start_bit);
                    IO::store_at(dst, s_goff[d] + tsc + (uint32_t)(i * NT) * SC, v);
                }
            }
        }
    } else {
#pragma unroll UNROLL
        for (int i = 0; i < K; ++i) {
            if (i * NT < (int)valid - tid) {
                const E v = s_elems[tid + i * NT];
                const uint32_t d = digit_of<NBITS>(v, start_bit);
                const uint32_t g = s_goff[d] + (uint32_t)(tid + i * NT);
                if (g < n_total) io.store((size_t)g, v);   // always true for a sound offset
            }
        }
    }
}

// Stable rank of each of a lane's K elements among the wave's elements with the same digit, in
// (item, lane) order; my_wcnt[digit] ends up holding the wave's count per digit (it must be zero on
// entry).  RANK == 1: one returning DS atomic per element -- on gfx950 a returning DS atomic issued
// by one wave-instruction resolves colliding lanes in ascending lane order and the DS ops of a wave
// execute in issue order, so the returned value IS the stable rank (checked at device creation by
// lds_order_selftest_kernel; if that ever fails the host selects RANK == 0).  RANK == 0: 64-lane
// ballot match: peers = lanes with my digit, rank = wave's running count (read by all peers, bumped by
// the lowest peer) + number of lower peers.
template <typename E, int NBITS, int K, int RANK>
__device__ __forceinline__ void rank_in_wave(const E (&e)[K], uint32_t (&rnk)[K], uint32_t* my_wcnt, int start_bit)
{
    if constexpr (RANK == 1) {
        // 64 lanes on ONE counter serialise in the DS atomic unit (constant / sorted / low-entropy input made
        // sorts up to 3.5x slower).  One check per wave and tile: if all 64*K elements of this wave share a
        // digit, their ranks are simply item*64 + lane.  Costs ~1 VALU per element on random data and keeps the
        // K atomics below back to back.
        const uint32_t dg0 = digit_of<NBITS>(e[0], start_bit);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dg0);
        // screen on ONE element per lane (one compare on random data, and the ranking below can start as soon as
        // e[0] has landed instead of waiting for all K loads); the exact test runs only when the screen passes
        bool same = false;
        if (__all(dg0 == d0)) {
            same = true;
#pragma unroll
            for (int j = 1; j < K; ++j) same &= digit_of<NBITS>(e[j], start_bit) == d0;
        }
        // second screen, again on ONE element per lane: how many lanes start a run (digit differs from the left neighbour's;
        // row_shr:1, the first lane of each row of 16 always counts).  Few runs = sorted / clustered input: a wave then holds
        // two or three digits and every returning atomic would still queue ~30 lanes on one counter (~25 cycles each: the
        // second MSD pass of the large sort took 0.30 ms instead of 0.12 on sorted keys).
        const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)~dg0, (int)dg0, 0x111, 0xf, 0xf, false);
        const bool few_runs = __popcll(__ballot(left != dg0)) <= 8;
        // third screen: ONE digit that most lanes share among scattered others (keys dominated by one value of this digit: 90 % of
        // the keys under one top byte made the one-sweep sort 0.80 ms instead of 0.60) -- that digit is served as in the few-runs
        // case, one returning add per instruction, the other lanes take the plain atomic
        const bool dominant = __popcll(__ballot(dg0 == d0)) >= 40;
        const int peel = few_runs ? 4 : (dominant ? 1 : 0);
        if (__all(same)) {
#pragma unroll
            for (int j = 0; j < K; ++j) rnk[j] = (uint32_t)(j * 64 + lane_id());
            if (lane_id() == 0) __hip_atomic_store(&my_wcnt[d0], (uint32_t)(64 * K), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if (peel) {
            // per instruction: the lanes of one digit are served together -- ONE returning add of their number by their first
            // lane, every lane's rank = the value it returned + the lanes of the group below it (the order the lane-ordered
            // atomics would have produced).  Up to four digits that way; lanes left over take the plain atomic.
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t d = digit_of<NBITS>(e[j], start_bit);
                uint64_t todo = __ballot(true);
                uint32_t r = 0u;
#pragma unroll 1
                for (int it = 0; it < peel && todo; ++it) {
                    // few runs: the digit of the first lane still to do; one dominant digit: that digit (d0), wherever it sits
                    const uint64_t m0 = few_runs ? todo : __ballot(d == d0);
                    if (!m0) break;
                    const int lead = __builtin_ctzll(m0);
                    const uint32_t dl = (uint32_t)__builtin_amdgcn_readlane((int)d, lead);
                    const uint64_t m = __ballot(d == dl);
                    uint32_t old = 0u;
                    if (lane_id() == lead)
                        old = __hip_atomic_fetch_add(&my_wcnt[dl], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    old = (uint32_t)__builtin_amdgcn_readlane((int)old, lead);
                    if (d == dl) r = old + mbcnt64(m);
                    todo &= ~m;
                }
                if ((todo >> lane_id()) & 1ull) r = __hip_atomic_fetch_add(&my_wcnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rnk[j] = r;
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const uint32_t d = digit_of<NBITS>(e[j], start_bit);
                rnk[j] = __hip_atomic_fetch_add(&my_wcnt[d], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    } else {
        // (round 4) the lowest peer alone adds -- one lane per digit and instruction, so nothing collides and no order is assumed --
        // with a RETURNING add; the K adds go out back to back (items in issue order) and the peers fetch their group's value from
        // that lane afterwards (ds_bpermute).  Round 3 read the counter in every peer and waited for it item by item.
        // (four items at a time: with all K in flight the kernel spilled and ran twice as long)
        constexpr int B = K % 4 == 0 ? 4 : (K % 2 == 0 ? 2 : 1);
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += B) {
            uint32_t old[B], info[B];   // info = leader lane << 8 | peers below
#pragma unroll
            for (int jj = 0; jj < B; ++jj) {
                const uint32_t d = digit_of<NBITS>(e[j0 + jj], start_bit);
                const uint64_t m = match_digit<NBITS>(d);
                const uint32_t below = mbcnt64(m);
                info[jj] = ((uint32_t)__builtin_ctzll(m) << 8) | below;
                old[jj] = 0u;
                if (below == 0u)
                    old[jj] = __hip_atomic_fetch_add(&my_wcnt[d], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
#pragma unroll
            for (int jj = 0; jj < B; ++jj) rnk[j0 + jj] = (uint32_t)__shfl((int)old[jj], (int)(info[jj] >> 8)) + (info[jj] & 0xffu);
        }
    }
}

// Stable in-tile ranking + local scatter + coalesced write-out of ONE tile.
//
//  * keys are loaded wave-striped: wave w owns tile elements [w*64*K, (w+1)*64*K); its item j, lane l
//    is element w*64*K + j*64 + l, so one load instruction reads 64 consecutive elements and tile
//    order == (wave, item, lane) order -- the order the ranking below preserves (stability).
//  * ranking: per item, match_digit() gives the peers; rank-in-wave = wave's running count for the
//    digit (one LDS word per (wave, digit), read by all peers, bumped by the lowest peer; DS ops of a
//    wave execute in order, so item j+1 sees item j's bump) + number of lower peers (mbcnt).
//  * thread b then folds the NW per-wave counts of digit b into wave offsets, the block scans the
//    digit totals into tile offsets, and `bin_offset(b, count)` -- supplied by the caller, run by
//    thread b only -- returns the global index at which this tile's digit-b run starts.
//  * elements go to LDS at their tile-sorted position and are read back in order, so that
//    consecutive lanes store consecutive elements of a digit's run (PDF eq. 2 of the reference paper:
//    dst = run start + position - tile offset of the digit).
//
// `valid` < TILE only for the globally last tile; the missing slots are padded with all-ones keys,
// which rank after every real element (max digit, highest indices, stable) and are never stored
// -- the reference's key-value kernel pads the same way (RadixSortKeyValueKernels.cl:554-563).
template <typename IO, int NBITS, int NT, int K, int RANK, typename BinOffsetFn>
__device__ __forceinline__ void sort_scatter_tile(const IO& io, uint32_t tile_base, uint32_t valid, uint32_t n_total,
                                                  int start_bit, unsigned char* smem, BinOffsetFn&& bin_offset)
{
    typedef typename IO::elem_t E;
    using C = TileCfg<E, NBITS, NT, K>;
    constexpr int BINS = C::BINS;
    constexpr int NW = C::NW;
    E* s_elems = reinterpret_cast<E*>(smem + C::OFF_ELEMS);
    uint32_t* s_wcnt = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);   // [NW][BINS]
    uint32_t* s_goff = reinterpret_cast<uint32_t*>(smem + C::OFF_GOFF);   // [BINS]
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + C::OFF_WSUM);

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint32_t* my_wcnt = s_wcnt + w * BINS;

    const bool scaled = dst_fits32<IO>(n_total);
    const uint32_t stamp_tile = tile_base / (uint32_t)C::TILE;
    (void)stamp_tile;
    ADLHIP_STAMP(stamp_tile, 0);
    // zero this wave's digit counters (wave-private row: no barrier needed before its own DS ops)
#pragma unroll
    for (int b = lane; b < BINS; b += 64) my_wcnt[b] = 0u;

    // load, wave-striped: one 64-bit pointer per tile + constant offsets; the tail predicate compares a
    // per-lane remainder with constants, so nothing per-element is loop-invariant (the compiler had
    // hoisted 2K address registers out of the tile loop otherwise)
    E e[K];
    {
        const uint32_t wbase = (uint32_t)(w * 64 * K + lane);
        const typename IO::Cursor p = io.cursor((size_t)tile_base + wbase);
        if (valid == (uint32_t)C::TILE) {
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = p.at(j * 64);
        } else {
            const int rem = (int)valid - (int)wbase;   // element j of this lane's column exists iff j*64 < rem
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = (j * 64 < rem) ? p.at(j * 64) : ~E(0);
        }
    }

#ifdef ADLHIP_STAMPS
    {   // force the loads to land so the stamp separates load latency from ranking
        E acc = 0;
#pragma unroll
        for (int j = 0; j < K; ++j) acc ^= e[j];
        if (acc == (E)0x1234567) my_wcnt[0] = 1u;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#endif
    ADLHIP_STAMP(stamp_tile, 1);
    // rank within the wave; two 16-bit ranks (< 64*K) per register
    uint32_t rnk2[(K + 1) / 2];
    {
        uint32_t rnk[K];
        rank_in_wave<E, NBITS, K, RANK>(e, rnk, my_wcnt, start_bit);
#pragma unroll
        for (int j = 0; j < K; j += 2) rnk2[j >> 1] = rnk[j] | ((j + 1 < K ? rnk[j + 1] : 0u) << 16);
    }
    // opaque from here on: otherwise the 32-bit ranks and the per-element LDS addresses of the ranking
    // phase stay live across the barriers (+50 VGPRs)
#pragma unroll
    for (int j = 0; j < (K + 1) / 2; ++j) asm volatile("" : "+v"(rnk2[j]));
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(e[j]));
    ADLHIP_STAMP(stamp_tile, 2);
    __syncthreads();
    ADLHIP_STAMP(stamp_tile, 3);

    // thread b: wave offsets for digit b, tile total, tile offset, global run start
    uint32_t cnt_b = 0u;
    uint32_t wc[NW];
    if (tid < BINS) {
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            wc[i] = s_wcnt[i * BINS + tid];
            cnt_b += wc[i];
        }
    }
    const uint32_t toff = block_excl_scan_u32<NT>(cnt_b, s_wsum, nullptr);
    if (tid < BINS) {
        uint32_t run = toff;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            s_wcnt[i * BINS + tid] = run;   // tile position of (wave i, digit b)'s first element
            run += wc[i];
        }
        ADLHIP_STAMP(stamp_tile, 4);
        const uint32_t gstart = bin_offset(tid, cnt_b);
        ADLHIP_STAMP(stamp_tile, 5);
        // dst index = goff[d] + tile position (mod 2^32); in bytes for the buffer-store path
        s_goff[tid] = scaled ? (gstart - toff) * IO::kStoreScale : gstart - toff;
    }
    __syncthreads();
    ADLHIP_STAMP(stamp_tile, 6);

    // local scatter into tile-sorted order: the LDS reads of the (wave, digit) positions go out CH at a
    // time ahead of the CH writes that use them (they may alias as far as the compiler knows)
    {
        constexpr int CH = K < 8 ? K : 8;
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += CH) {
            uint32_t pos[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) pos[j] = my_wcnt[digit_of<NBITS>(e[(j0 + j < K) ? j0 + j : K - 1], start_bit)];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j0 + j < K) {   // K need not be a multiple of CH
                    const uint32_t r = (rnk2[(j0 + j) >> 1] >> (16 * ((j0 + j) & 1))) & 0xffffu;
                    s_elems[pos[j] + r] = e[j0 + j];
                }
            }
        }
    }
    ADLHIP_STAMP(stamp_tile, 7);
    __syncthreads();
    ADLHIP_STAMP(stamp_tile, 8);

    write_out_tile<IO, NBITS, NT, K, 4>(io, s_elems, s_goff, valid, n_total, start_bit, scaled);
    ADLHIP_STAMP(stamp_tile, 9);
    __syncthreads();   // LDS is reused by the next tile
    ADLHIP_STAMP(stamp_tile, 10);
}

// ------------------------------------------------------------------------------------------
// Three-kernel pass ("sort.algo" = 1): count -> table scan -> sort+scatter
// ------------------------------------------------------------------------------------------

// Per-workgroup digit histogram over the workgroup's contiguous run of elements.
// table is bucket-major like the reference's (histogramOut[bucket*nWGs + wg],
// RadixSort32Kernels.cl:233): row b holds, for every workgroup, its count of digit b.
template <typename E, int NBITS, int NT>
__global__ __launch_bounds__(NT) void radix_count_kernel(const E* __restrict__ src,
                                                         uint32_t* __restrict__ table, uint32_t n,
                                                         int n_wgs, int start_bit, uint32_t elems_per_wg, int wg_major)
{
    constexpr int BINS = 1 << NBITS;
    constexpr int NW = NT / 64;
    // 8-bit digits: one private histogram per wave.  4-bit digits: one private column per lane
    // (16 x 64, bank = lane) so that 64 lanes never collide on an address.
    constexpr int COPIES = (NBITS <= 4) ? 64 : NW;
    __shared__ uint32_t hist[BINS * COPIES];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    for (int i = tid; i < BINS * COPIES; i += NT) hist[i] = 0u;
    __syncthreads();

    // Which range a workgroup counts: workgroups b and b + 8 run on one XCD (speed only), and the table is bucket-major -- thread d
    // writes ONE word into row d at the column of its range.  With range = workgroup index the columns an XCD writes are 8 apart
    // (32 bytes): every 64-byte piece of a row is written by two XCDs and leaves each L2 half-filled.  With the ranges dealt so that
    // an XCD's workgroups own CONSECUTIVE columns, whole lines are completed in one L2 (64 Mi keys, cold: 55 -> 4x us).
    uint32_t wg = blockIdx.x;
    if ((n_wgs & 7) == 0) wg = (blockIdx.x & 7u) * ((uint32_t)n_wgs >> 3) + (blockIdx.x >> 3);
    const uint64_t begin64 = (uint64_t)wg * elems_per_wg;
    if (begin64 < n) {
        const uint32_t begin = (uint32_t)begin64;
        const uint32_t end = (uint32_t)((begin64 + elems_per_wg < n) ? begin64 + elems_per_wg : n);
        constexpr int VEC = 16 / (int)sizeof(E);
        struct alignas(16) Vec { E v[VEC]; };
        auto bump = [&](E x) {
            const uint32_t d = digit_of<NBITS>(x, start_bit);
            if (NBITS <= 4) atomicAdd(&hist[d * 64 + lane], 1u);   // one private column per lane: never collides
            else atomicAdd(&hist[w * BINS + d], 1u);
        };
        // 4 vectors at once; if every element of every active lane has the same digit (constant / sorted
        // input) one lane adds the total instead of 64 lanes serialising on one LDS word
        auto bump4 = [&](const auto& a, const auto& b, const auto& c, const auto& d4) {
            constexpr int V = 16 / (int)sizeof(E);
            if (NBITS > 4) {
                const uint32_t da = digit_of<NBITS>(a.v[0], start_bit);
                const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)da);
                if (__all(da == d0)) {   // one-key-per-lane screen: the full test below runs only when it might succeed
                    bool same = true;
#pragma unroll
                    for (int k = 0; k < V; ++k)
                        same &= (digit_of<NBITS>(a.v[k], start_bit) == d0) & (digit_of<NBITS>(b.v[k], start_bit) == d0) &
                                (digit_of<NBITS>(c.v[k], start_bit) == d0) & (digit_of<NBITS>(d4.v[k], start_bit) == d0);
                    if (__all(same)) {
                        const uint64_t act = __ballot(true);
                        if (mbcnt64(act) == 0u) atomicAdd(&hist[w * BINS + d0], (uint32_t)(4 * V * __popcll(act)));
                        return;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < V; ++k) { bump(a.v[k]); bump(b.v[k]); bump(c.v[k]); bump(d4.v[k]); }
        };
        const uint32_t nvec = (end - begin) / VEC;
        // every key is read once: NON-TEMPORAL 16-byte loads.  Round 4, 64 Mi u32 keys cold in every cache (tools/r4_probe hist,
        // profiles/r4_hist_read_variants.txt): this kernel 55.2 us with plain loads; a bare read of the same bytes 48.9 us plain and
        // 42.0-43.5 us non-temporal (6.2-6.4 TB/s); the histogram on non-temporal loads 42.8-43.9 us = 6.1-6.3 TB/s = 0.77 of 8 TB/s --
        // the LDS atomics (8.8 cycles per 64 keys, tools/r4_lds_bench: 17 us of LDS time) hide behind the stream entirely.
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u* vraw = reinterpret_cast<const v4u*>(src + begin);
        auto vld = [&](uint32_t at) -> Vec {
            const v4u r = __builtin_nontemporal_load(vraw + at);
            Vec o;
            __builtin_memcpy(&o, &r, 16);
            return o;
        };
        uint32_t i = (uint32_t)tid;
        // 4 independent 16-byte loads in flight per lane
        if (i + 3u * NT < nvec) {   // software-pipelined: next loads in flight while the current vectors are counted
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
            Vec a = vld(i);
#pragma unroll
            for (int k = 0; k < VEC; ++k) bump(a.v[k]);
        }
        for (uint32_t s = begin + nvec * VEC + (uint32_t)tid; s < end; s += NT) bump(src[s]);
    }
    __syncthreads();
    if (tid < BINS) {
        uint32_t sum = 0u;
        if (NBITS <= 4) {
            for (int c = 0; c < 64; ++c) sum += hist[tid * 64 + ((c + tid) & 63)];
        } else {
#pragma unroll
            for (int c = 0; c < COPIES; ++c) sum += hist[c * BINS + tid];
        }
        // wg_major (few workgroups, the scatter kernel scans the raw counts itself): one coalesced row per workgroup
        table[wg_major ? (size_t)wg * BINS + tid : (size_t)tid * n_wgs + wg] = sum;
    }
}

// One workgroup per digit row: exclusive scan of the row in place + row total.  Together with the
// scan of the totals done in the scatter kernel's prologue this yields, for (digit b, workgroup m),
// sum_{i<b} sum_j c_j^i + sum_{j<m} c_j^b  -- the reference's flat bucket-major scan
// (RadixSort32Kernels.cl:243-362; paper eq. 1).
template <int NT>
__global__ __launch_bounds__(NT) void radix_scan_table_kernel(uint32_t* __restrict__ table,
                                                              uint32_t* __restrict__ totals, int n_wgs)
{
    __shared__ uint32_t wsum[NT / 64 + 1];
    uint32_t* row = table + (size_t)blockIdx.x * n_wgs;
    uint32_t carry = 0u;
    for (int base = 0; base < n_wgs; base += NT) {
        const int i = base + (int)threadIdx.x;
        const uint32_t v = i < n_wgs ? row[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_u32<NT>(v, wsum, &tot);
        if (i < n_wgs) row[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Stable local sort + scatter of the workgroup's run of tiles, carrying per-digit offsets from tile
// to tile (RadixSort32Kernels.cl:493-631 behaviour).
template <typename IO, int NBITS, int NT, int K, int RANK>
__global__ __launch_bounds__(NT) void radix_scatter_kernel(IO io, const uint32_t* __restrict__ table,
                                                           const uint32_t* __restrict__ totals, uint32_t n,
                                                           int n_wgs, int start_bit, uint32_t tiles_per_wg,
                                                           uint32_t num_tiles)
{
    typedef typename IO::elem_t E;
    using C = TileCfg<E, NBITS, NT, K>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + C::OFF_WSUM);
    const int tid = (int)threadIdx.x;
    const uint32_t wg = blockIdx.x;

    // digit bases = exclusive scan of the digit totals; carry = base + scanned table entry.
    // totals == nullptr (few workgroups, launch-bound sizes): `table` holds the RAW counts, workgroup-major
    // (table[wg][digit]), and thread b sums its column here -- n_wgs coalesced, independent loads -- which saves the
    // table-scan launch and its kernel boundary.
    uint32_t tot_b = 0u, pre_b = 0u;
    if (tid < C::BINS) {
        if (totals) {
            tot_b = totals[tid];
            pre_b = table[(size_t)tid * n_wgs + wg];
        } else {
            // fixed trip counts (16, or 64: n_wgs <= 64 here), every load independent of the others: one or two batches
            // of loads in flight instead of a remainder loop that pays one memory latency per workgroup
            auto walk = [&](auto trips) {
#pragma unroll
                for (int j = 0; j < decltype(trips)::value; ++j) {
                    const int jj = j < n_wgs ? j : n_wgs - 1;
                    const uint32_t v = table[(size_t)jj * C::BINS + tid];
                    tot_b += (j < n_wgs) ? v : 0u;
                    pre_b += (j < (int)wg) ? v : 0u;   // wg < n_wgs
                }
            };
            if (n_wgs <= 16) walk(std::integral_constant<int, 16>());
            else walk(std::integral_constant<int, 64>());
        }
    }
    const uint32_t base_b = block_excl_scan_u32<NT>(tot_b, s_wsum, nullptr);
    uint32_t carry = base_b + pre_b;

    const uint32_t t0 = wg * tiles_per_wg;
    const uint32_t t1 = (t0 + tiles_per_wg < num_tiles) ? t0 + tiles_per_wg : num_tiles;
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t tile_base = t * (uint32_t)C::TILE;
        const uint32_t left = n - tile_base;
        const uint32_t valid = left < (uint32_t)C::TILE ? left : (uint32_t)C::TILE;
        sort_scatter_tile<IO, NBITS, NT, K, RANK>(io, tile_base, valid, n, start_bit, smem,
                                           [&](int /*b*/, uint32_t cnt) {
                                               const uint32_t g = carry;
                                               carry += cnt;
                                               return g;
                                           });
    }
}

// ------------------------------------------------------------------------------------------
// Small inputs (n <= NT*K): the whole LSD sort in ONE workgroup and ONE launch -- all passes run on
// registers + LDS, the data touches global memory once in and once out.  Replaces the 9-12 dependent
// launches (~55-90 us of launch latency) the general paths need regardless of n; the reference's Demo
// sizes 1K..16K land here.
// ------------------------------------------------------------------------------------------
struct SmallPlan {
    int num_passes;
    uint8_t start_bit[16];
    uint8_t nbits[16];
};

template <typename E, int NT, int K, int RANK, bool FULL>
__device__ __forceinline__ void small_sort_body(E* __restrict__ data, uint32_t n, const SmallPlan& plan)
{
    constexpr int BINS = 256;
    constexpr int NW = NT / 64;
    constexpr int TILE = NT * K;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + sizeof(E) * TILE);   // [NW][BINS]
    uint32_t* __restrict__ s_wsum = s_wcnt + NW * BINS;

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    // Only ceil(n / NT) of the K items per thread are used, so that n < NT*K elements are spread over ALL waves
    // (wave w owns elements [w*64*keff, (w+1)*64*keff)) and nothing is spent on empty slots: with the full K items
    // 1 Ki keys sat in one wave, and thousands of all-ones pads queued up on one LDS counter in every pass
    // (1 Ki keys took 18 us, 4 Ki + 1 keys 54 us; a full 16 Ki tile 19 us).
    const int keff = FULL ? K : (int)((n + (uint32_t)NT - 1u) / (uint32_t)NT);   // 1..K (FULL: a compile-time K, no per-item branches)
    const uint32_t wbase = (uint32_t)(w * 64 * keff + lane);

    // wave-striped load; the < NT slots between n and keff*NT are padded with all-ones (they rank last, are never stored)
    E e[K];
    {
        const int rem = (int)n - (int)wbase;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (FULL || j < keff) e[j] = (j * 64 < rem) ? data[wbase + (uint32_t)(j * 64)] : ~E(0);
    }

    for (int p = 0; p < plan.num_passes; ++p) {
        const int sb = plan.start_bit[p];
        const uint32_t mask = (1u << plan.nbits[p]) - 1u;
        auto digit = [&](E x) -> uint32_t {
            if constexpr (sizeof(E) == 8) return (uint32_t)((uint64_t)x >> sb) & mask;
            else return ((uint32_t)x >> sb) & mask;
        };
        // pads (all-ones) have the top digit in every pass and start behind every real element, so they
        // stay at the end of the tile through all (stable) passes
#pragma unroll
        for (int b = lane; b < BINS; b += 64) my_wcnt[b] = 0u;
        uint32_t rnk[K];
        if constexpr (RANK == 1) {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (FULL || j < keff)
                    rnk[j] = __hip_atomic_fetch_add(&my_wcnt[digit(e[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (FULL || j < keff) {   // wave-uniform: all 64 lanes take part in the ballots
                    const uint32_t d = digit(e[j]);
                    const uint64_t m = match_digit<8>(d);
                    const uint32_t below = mbcnt64(m);
                    const uint32_t old = __hip_atomic_load(&my_wcnt[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (below == 0u)
                        __hip_atomic_fetch_add(&my_wcnt[d], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    rnk[j] = old + below;
                }
            }
        }
        __syncthreads();
        // thread b (< 256): fold the per-wave counts of digit b, scan over digits, write tile positions
        uint32_t cnt_b = 0u;
        uint32_t wc[NW];
        if (tid < BINS) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                wc[i] = s_wcnt[i * BINS + tid];
                cnt_b += wc[i];
            }
        }
        const uint32_t toff = block_excl_scan_u32<NT>(cnt_b, s_wsum, nullptr);
        if (tid < BINS) {
            uint32_t run = toff;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                s_wcnt[i * BINS + tid] = run;
                run += wc[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (FULL || j < keff) s_elems[my_wcnt[digit(e[j])] + rnk[j]] = e[j];
        __syncthreads();
        // back to registers in wave-striped order = the order the next pass must preserve
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (FULL || j < keff) e[j] = s_elems[wbase + (uint32_t)(j * 64)];
        __syncthreads();
    }
    {
        const int rem = (int)n - (int)wbase;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if ((FULL || j < keff) && j * 64 < rem) data[wbase + (uint32_t)(j * 64)] = e[j];
    }
}

template <typename E, int NT, int K, int RANK>
__global__ __launch_bounds__(NT) void small_sort_kernel(E* __restrict__ data, uint32_t n, SmallPlan plan)
{
    if (n + (uint32_t)NT > (uint32_t)(NT * K)) small_sort_body<E, NT, K, RANK, true>(data, n, plan);   // all K items in use
    else small_sort_body<E, NT, K, RANK, false>(data, n, plan);
}

// ------------------------------------------------------------------------------------------
// Generic exclusive scan (Pprims::scan): reduce -> scan partials -> apply
// Tahoe/ClKernels/PrefixScanKernels.cl:70-143 behaviour, without its 4096-block limit.
// ------------------------------------------------------------------------------------------
constexpr int kScanNT = 256;
constexpr int kScanRounds = 4;                                  // 16-byte loads per thread per tile
constexpr int kScanTile = kScanNT * 4 * kScanRounds;            // 4096 elements

// Scans one tile [base, base+tile) of src into dst (exclusive, + carry_in); elements >= n_total are
// treated as 0 and not stored.  Returns the tile total.  Two barriers.
__device__ __forceinline__ uint32_t scan_tile(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                              size_t base, size_t n_total, uint32_t carry_in,
                                              uint32_t* s_wtot /*[kScanRounds*NW + 1]*/)
{
    constexpr int NW = kScanNT / 64;
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint4 v[kScanRounds];
    uint32_t ex[kScanRounds];
    const bool full = base + (size_t)kScanTile <= n_total && ((reinterpret_cast<uintptr_t>(src + base) & 15u) == 0);
#pragma unroll
    for (int r = 0; r < kScanRounds; ++r) {
        const size_t i = base + ((size_t)(r * kScanNT + tid)) * 4;
        if (full) {
            v[r] = *reinterpret_cast<const uint4*>(src + i);
        } else {
            v[r].x = i + 0 < n_total ? src[i + 0] : 0u;
            v[r].y = i + 1 < n_total ? src[i + 1] : 0u;
            v[r].z = i + 2 < n_total ? src[i + 2] : 0u;
            v[r].w = i + 3 < n_total ? src[i + 3] : 0u;
        }
        const uint32_t s = v[r].x + v[r].y + v[r].z + v[r].w;
        const uint32_t inc = wave_incl_scan_u32(s);
        ex[r] = inc - s;
        if (lane == 63) s_wtot[r * NW + w] = inc;
    }
    __syncthreads();
    if (tid < 64) {   // one wave scans the kScanRounds*NW segment totals (<= 64 of them)
        const uint32_t t = tid < kScanRounds * NW ? s_wtot[tid] : 0u;
        const uint32_t inc = wave_incl_scan_u32(t);
        if (tid < kScanRounds * NW) s_wtot[tid] = inc - t;
        if (tid == kScanRounds * NW - 1) s_wtot[kScanRounds * NW] = inc;
    }
    __syncthreads();
    const uint32_t total = s_wtot[kScanRounds * NW];
    const bool dfull = full && ((reinterpret_cast<uintptr_t>(dst + base) & 15u) == 0);
#pragma unroll
    for (int r = 0; r < kScanRounds; ++r) {
        const size_t i = base + ((size_t)(r * kScanNT + tid)) * 4;
        uint32_t o = carry_in + s_wtot[r * NW + w] + ex[r];
        uint4 out;
        out.x = o; o += v[r].x;
        out.y = o; o += v[r].y;
        out.z = o; o += v[r].z;
        out.w = o;
        if (dfull) {
            *reinterpret_cast<uint4*>(dst + i) = out;
        } else {
            if (i + 0 < n_total) dst[i + 0] = out.x;
            if (i + 1 < n_total) dst[i + 1] = out.y;
            if (i + 2 < n_total) dst[i + 2] = out.z;
            if (i + 3 < n_total) dst[i + 3] = out.w;
        }
    }
    __syncthreads();   // s_wtot reused by the caller's next tile
    return total;
}

// Block sums: partial[block] = sum of the block's tile.
ADLHIP_KERNEL __global__ __launch_bounds__(kScanNT) void scan_reduce_kernel(const uint32_t* __restrict__ src,
                                                              uint32_t* __restrict__ partial, size_t n)
{
    __shared__ uint32_t wsum[kScanNT / 64];
    const size_t base = (size_t)blockIdx.x * kScanTile;
    const int tid = (int)threadIdx.x;
    uint32_t s = 0u;
    const bool full = base + (size_t)kScanTile <= n && ((reinterpret_cast<uintptr_t>(src + base) & 15u) == 0);
#pragma unroll
    for (int r = 0; r < kScanRounds; ++r) {
        const size_t i = base + ((size_t)(r * kScanNT + tid)) * 4;
        if (full) {
            const uint4 v = *reinterpret_cast<const uint4*>(src + i);
            s += v.x + v.y + v.z + v.w;
        } else {
            for (int k = 0; k < 4; ++k) if (i + k < n) s += src[i + k];
        }
    }
    s = wave_incl_scan_u32(s);
    if ((tid & 63) == 63) wsum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        uint32_t t = 0u;
        for (int i = 0; i < kScanNT / 64; ++i) t += wsum[i];
        partial[blockIdx.x] = t;
    }
}

// Single workgroup: exclusive scan of data[0..count) in place (tile loop with carry); the grand total
// goes to data[count] (the reference writes it to the same slot, PrefixScanKernels.cl:139-142).
// With src != data it is the whole scan for small n.
ADLHIP_KERNEL __global__ __launch_bounds__(kScanNT) void scan_single_kernel(const uint32_t* __restrict__ src,
                                                              uint32_t* __restrict__ dst, size_t count,
                                                              uint32_t* __restrict__ total_out)
{
    __shared__ uint32_t s_wtot[kScanRounds * (kScanNT / 64) + 1];
    uint32_t carry = 0u;
    for (size_t base = 0; base < count; base += kScanTile)
        carry += scan_tile(src, dst, base, count, carry, s_wtot);
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

// Block b rescans its tile with the scanned block sum as carry-in
// (LocalScan + AddOffset of the reference fused: one read, one write).
ADLHIP_KERNEL __global__ __launch_bounds__(kScanNT) void scan_apply_kernel(const uint32_t* __restrict__ src,
                                                             uint32_t* __restrict__ dst,
                                                             const uint32_t* __restrict__ partial_ex, size_t n)
{
    __shared__ uint32_t s_wtot[kScanRounds * (kScanNT / 64) + 1];
    const size_t base = (size_t)blockIdx.x * kScanTile;
    scan_tile(src, dst, base, n, partial_ex[blockIdx.x], s_wtot);
}

// ------------------------------------------------------------------------------------------
// bandwidth probes + fill
// ------------------------------------------------------------------------------------------
ADLHIP_KERNEL __global__ __launch_bounds__(256) void probe_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src,
                                                         size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < nvec; i += stride) dst[i] = src[i];
}

// The same two probes with cache-policy hints: NTL = non-temporal loads, NTS = non-temporal stores (bench.py reports the plain and
// the hinted rates side by side; which one is the honest ceiling for a sweep depends on whether the sweep's data will be read again)
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void probe_copy_hint_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t nvec)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u* s = reinterpret_cast<const v4u*>(src);
    v4u* d = reinterpret_cast<v4u*>(dst);
    auto ld = [&](size_t i) -> v4u { return NTL ? __builtin_nontemporal_load(s + i) : s[i]; };
    auto st = [&](size_t i, v4u v) {
        if (NTS) __builtin_nontemporal_store(v, d + i);
        else d[i] = v;
    };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const v4u a = ld(i), b = ld(i + stride), c = ld(i + 2 * stride), e = ld(i + 3 * stride);
        st(i, a); st(i + stride, b); st(i + 2 * stride, c); st(i + 3 * stride, e);
    }
    for (; i < nvec; i += stride) st(i, ld(i));
}
template <bool NTL>
__global__ __launch_bounds__(256) void probe_read_hint_kernel(const uint4* __restrict__ src, size_t nvec, unsigned long long* __restrict__ sink)
{
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u* s = reinterpret_cast<const v4u*>(src);
    auto ld = [&](size_t i) -> v4u { return NTL ? __builtin_nontemporal_load(s + i) : s[i]; };
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0u;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const v4u a = ld(i), b = ld(i + stride), c = ld(i + 2 * stride), e = ld(i + 3 * stride);
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ e.x ^ e.y ^ e.z ^ e.w;
    }
    for (; i < nvec; i += stride) { const v4u a = ld(i); acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);   // practically never; keeps the loads alive
}

ADLHIP_KERNEL __global__ __launch_bounds__(256) void probe_read_kernel(const uint4* __restrict__ src, size_t nvec,
                                                         unsigned long long* __restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0u;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < nvec; i += stride) { uint4 a = src[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x9e3779b9u) atomicAdd(sink, 1ull);   // practically never; keeps the loads alive
}

ADLHIP_KERNEL __global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ dst, uint32_t pattern, size_t count)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = pattern;
}

// count copies of a 16-byte pattern (8-byte patterns are doubled by the host: count is then in 16-byte units
// plus an optional 8-byte tail written by thread 0)
ADLHIP_KERNEL __global__ __launch_bounds__(256) void fill_pattern16_kernel(uint4* __restrict__ dst, uint4 pattern, size_t count,
                                                             uint2* __restrict__ tail8, uint2 tail_pattern)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dst[i] = pattern;
    if (tail8 && blockIdx.x == 0 && threadIdx.x == 0) *tail8 = tail_pattern;
}

// Device self-test for the RANK == 1 path: do returning DS atomics of one wave-instruction resolve colliding
// lanes in ascending lane order, and do successive instructions of a wave apply in order?  Compares ds_add_rtn
// ranks with ballot/mbcnt ranks over pseudo-random digits for 1, 2, 16 and 256 bins; *mismatches stays 0 iff they
// agree everywhere.  Shaped like the kernels that rely on it: up to 1024-thread workgroups (every wave hammering
// its own counters, so the LDS unit is contended the way it is in a pass), sixteen atomics back to back per lane,
// and rounds with a pseudo-random subset of the lanes switched off (partial waves: tail tiles, segment ends).
// It also runs on its own stream beside real sorts (adlhip_selftest_lds_order; tests/test_gpu_parity.py).
ADLHIP_KERNEL __global__ __launch_bounds__(1024) void lds_order_selftest_kernel(uint32_t* __restrict__ mismatches, uint32_t salt)
{
    __shared__ uint32_t c_atomic[16][256];
    __shared__ uint32_t c_ballot[16][256];
    const int lane = lane_id();
    const int w = (int)(threadIdx.x >> 6);
    uint32_t bad = 0u;
    for (int lg = 0; lg <= 8; lg += (lg < 1 ? 1 : (lg < 4 ? 3 : 4))) {   // bins = 1, 2, 16, 256
        const uint32_t bins = 1u << lg;
        for (int part = 0; part < 2; ++part) {   // all lanes on / a pseudo-random subset on
            for (int b = lane; b < 256; b += 64) { c_atomic[w][b] = 0u; c_ballot[w][b] = 0u; }
            uint32_t d[16], got[16];
            bool on[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t h = (blockIdx.x * blockDim.x + threadIdx.x) * 16u + (uint32_t)j + bins * 0x9E3779B9u + salt * 0x85EBCA6Bu;
                h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
                d[j] = h & (bins - 1u);
                on[j] = part == 0 || ((h >> 20) & 3u) != 0u;
            }
            // the atomics back to back, as the ranking loops issue them
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (on[j]) got[j] = __hip_atomic_fetch_add(&c_atomic[w][d[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (on[j]) {   // ballots inside the branch see the active lanes only (~ballot would count the others: mask them)
                    const uint64_t m = match_digit<8>(d[j]) & __ballot(true);
                    const uint32_t below = mbcnt64(m);
                    const uint32_t old = __hip_atomic_load(&c_ballot[w][d[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    if (below == 0u)
                        __hip_atomic_fetch_add(&c_ballot[w][d[j]], (uint32_t)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    bad += (got[j] != old + below) ? 1u : 0u;
                }
            }
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Synthetic inputs, reproducible by index (same function as the oracle's generator).
__device__ __forceinline__ uint64_t splitmix64_at(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// kind: 0 = u32 keys, 1 = {key32, index} pairs, 2 = u64 keys
ADLHIP_KERNEL __global__ __launch_bounds__(256) void generate_keys_kernel(void* __restrict__ dst, size_t n, uint64_t base,
                                                            uint64_t first_index, int kind)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t r = splitmix64_at(base + i);
        if (kind == 0) reinterpret_cast<uint32_t*>(dst)[i] = (uint32_t)(r >> 32);
        else if (kind == 1) reinterpret_cast<uint64_t*>(dst)[i] = (r >> 32) | ((uint64_t)(uint32_t)(first_index + i) << 32);
        else reinterpret_cast<uint64_t*>(dst)[i] = r;
    }
}

}  // namespace adlhip
