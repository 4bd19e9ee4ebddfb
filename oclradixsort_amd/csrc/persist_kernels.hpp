// persist_kernels.hpp -- gfx950 device code: the cursor-placed MSD pass of the large keys-only sort as a PERSISTENT,
// software-pipelined kernel (round 4).
//
// msd_bucket_scatter_kernel (hybrid_kernels.hpp) runs one tile per workgroup: load -> rank -> LDS scatter -> write-out,
// strictly one after the other, two workgroups per CU.  While a workgroup ranks, the memory system has nothing of its to
// do; while it waits for its keys, the LDS idles.  The counters said so (profiles/r3_pmc_msd2_u32_all_counters.txt: mean
// read latency 2812 cycles with 4 waves per SIMD) and so did the pass-shaped copy without ranking (107-117 us against the
// pass's 145-150 us at 64 Mi keys, profiles/r1_scatter_granularity_probe.txt, r2_writeout_shape_probe.txt).
//
// Here a workgroup stays and takes tiles g, g + G, g + 2G, ...: the keys of tile i + 1 are requested (16 bytes per lane)
// BEFORE tile i is ranked, into a second set of registers, and are in flight through tile i's whole body; the stores of
// tile i drain while tile i + 1 is ranked.  Barriers are LDS-only (s_waitcnt lgkmcnt(0); s_barrier) so that they do not
// wait for the prefetch.  Two barriers per tile: the third one of the one-shot kernel (LDS free for the next tile) is
// implied by the next tile's first barrier.
//
// Placement is the cursor form's (see msd_bucket_scatter_kernel): wave 0 reserves the tile's 256 runs with one returning
// atomic each on the destination cursors; keys only, so neither pass needs to be stable and lanes may take their keys in any
// order (16-byte loads: lane l of wave w holds elements w*64K + q*64V + l*V + c, V = 16 / sizeof(E)).
//
// Pass 2 (source = the 256 bucket slabs of pass 1): tickets are (bucket, tile) pairs, and workgroup g serves only buckets with
// bucket % 8 == g % 8 -- workgroups g and g + 8 share an XCD (speed only, never correctness), so all runs of one segment slab
// are written through ONE L2 and abutting partial lines are completed there before they are written back.
//
// Reference behaviour: Tahoe/ClKernels/RadixSort32Kernels.cl:493-631 (SortAndScatterKernel: local sort + scatter with a
// running per-workgroup carry); the mechanism is different throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hybrid_kernels.hpp"

namespace adlhip {

template <typename E, int NT, int K>
struct PersistCfg {
    static constexpr int NW = NT / 64;
    static constexpr int TILE = NT * K;
    static constexpr int VEC = 16 / (int)sizeof(E);
    static_assert(K % VEC == 0, "whole 16-byte loads");
    static_assert(TILE <= 65536, "16-bit tile positions");
    static constexpr size_t OFF_ELEMS = 0;
    static constexpr size_t OFF_WCNT = sizeof(E) * TILE;                       // u32 [NW][256]
    static constexpr size_t OFF_WPOS = OFF_WCNT + sizeof(uint32_t) * NW * 256;  // u16 [NW][256]
    static constexpr size_t OFF_GOFF = OFF_WPOS + sizeof(uint16_t) * NW * 256;  // u32 [256]
    static constexpr size_t OFF_PREF = OFF_GOFF + sizeof(uint32_t) * 256;       // u32 [260]: tiles before (permuted) bucket p
    static constexpr size_t OFF_BCNT = OFF_PREF + sizeof(uint32_t) * 260;       // u32 [256]: elements of (permuted) bucket p
    static constexpr size_t OFF_WSUM = OFF_BCNT + sizeof(uint32_t) * 256;       // block scan scratch
    static constexpr size_t LDS_BYTES = OFF_WSUM + 64;
};

struct PersistTile {
    uint32_t base;          // index of the tile's first element in the source array
    uint32_t valid;         // elements of the tile (0: no tile)
    uint32_t cursor_base;   // first of the tile's 256 destination cursors
};

// rank_in_wave (radix_kernels.hpp) for a BATCH of B of a lane's elements: the counters carry on from the batches before (the
// all-the-same shortcut adds to its counter instead of storing), so a tile can be ranked eight elements at a time and its ranks
// packed two to a register as they arrive -- with all K returning atomics of a tile in flight at once their K result registers
// are live next to the K keys and the K prefetched keys, and the kernel spills.  Same three screens, per batch: every lane's
// elements share one digit (one add per wave), few runs / one dominant digit (one add per digit and instruction), else one
// returning atomic per element.  Ranks are handed out in (item, lane) order.
template <typename E, int B>
__device__ __forceinline__ void rank_batch(const E* e, uint32_t* rnk, uint32_t* my_wcnt, int start_bit)
{
    const uint32_t dg0 = digit_of<8>(e[0], start_bit);
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dg0);
    bool same = false;
    if (__all(dg0 == d0)) {
        same = true;
#pragma unroll
        for (int j = 1; j < B; ++j) same &= digit_of<8>(e[j], start_bit) == d0;
    }
    const uint32_t left = (uint32_t)__builtin_amdgcn_update_dpp((int)~dg0, (int)dg0, 0x111, 0xf, 0xf, false);
    const bool few_runs = __popcll(__ballot(left != dg0)) <= 8;
    const bool dominant = __popcll(__ballot(dg0 == d0)) >= 40;
    const int peel = few_runs ? 4 : (dominant ? 1 : 0);
    if (__all(same)) {
        uint32_t old = 0u;
        if (lane_id() == 0) old = __hip_atomic_fetch_add(&my_wcnt[d0], (uint32_t)(64 * B), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
#pragma unroll
        for (int j = 0; j < B; ++j) rnk[j] = old + (uint32_t)(j * 64 + lane_id());
    } else if (peel) {
#pragma unroll
        for (int j = 0; j < B; ++j) {
            const uint32_t d = digit_of<8>(e[j], start_bit);
            uint64_t todo = __ballot(true);
            uint32_t r = 0u;
#pragma unroll 1
            for (int it = 0; it < peel && todo; ++it) {
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
        for (int j = 0; j < B; ++j)
            rnk[j] = __hip_atomic_fetch_add(&my_wcnt[digit_of<8>(e[j], start_bit)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// PASS: 1 = the input array, first digit; 2 = the bucket slabs of pass 1, second digit (a.src_counts != nullptr).
// WPE = waves per SIMD the launch wants resident (workgroups per CU x NT / 256): bounds the registers.
// D16: the destination slabs hold uint16_t (a.dst16; compile-time here: a run-time test would put a branch around every store)
// LAUX / SAUX: cache policy of the key loads / the run stores (buffer aux bits: 0 = default, 2 = nt)
template <typename E, int NT, int K, int PASS, int WPE, bool D16, int LAUX = 0, int SAUX = 0>
__global__ __launch_bounds__(NT, WPE) void msd_scatter_persist_kernel(BucketPass<E> a)
{
    using C = PersistCfg<E, NT, K>;
    constexpr int NW = C::NW;
    constexpr int TILE = C::TILE;
    constexpr int VEC = C::VEC;
    constexpr int Q = K / VEC;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem + C::OFF_ELEMS);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);
    uint32_t* __restrict__ s_goff = reinterpret_cast<uint32_t*>(smem + C::OFF_GOFF);
    uint32_t* __restrict__ s_pref = reinterpret_cast<uint32_t*>(smem + C::OFF_PREF);
    uint32_t* __restrict__ s_bcnt = reinterpret_cast<uint32_t*>(smem + C::OFF_BCNT);
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    if (a.zero_me && blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.zero_me, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // A run that did not fit its slab has raised the flag: the sort's result will come from the safety net (hybrid_kernels.hpp
    // net_sort), whatever is written from here on is never read.  The second pass leaves at once when the first has raised it; both
    // look at the flag once per tile (below) -- keys that are all equal overflow a slab within the first per cent of the input.
    // Either way the decision is the same for every wave of the workgroup -- the flag can rise between two waves' loads, and a
    // workgroup of which some waves have left would go on with stale counters and tickets: one thread's (one tile ago: every wave's)
    // view goes through LDS and a barrier.
    uint32_t* __restrict__ s_give = reinterpret_cast<uint32_t*>(smem + C::OFF_WSUM);   // [8 + 1] (the block scan's scratch: prologue only)
    {   // (pass 1 too: the sample kernel raises the flag for keys that repeat a few thousand values)
        if (tid == 0) s_give[8] = large_sort_gave_up(a.flag);
        __syncthreads();
        if (s_give[8]) return;
        __syncthreads();   // (the scan below reuses the scratch)
    }
    int start_bit = a.start_bit;
    Msd2Placement place{0, 0ull};
    if (a.sample) {
        place = msd2_placement(a.sample);
        start_bit = place.top - 8 * a.which_digit;
    }
    if constexpr (PASS >= 2) start_bit += 8 - (int)a.seg_shift;   // narrow second digit: the field sits 8 - w bits higher

    // ---- tickets ---------------------------------------------------------------------------------------------------------------
    const uint32_t G = gridDim.x, g = blockIdx.x;
    uint32_t ticket, t_end, t_step, t_origin = 0u;
    if constexpr (PASS == 1) {
        ticket = g;
        t_end = (a.n + (uint32_t)TILE - 1u) / (uint32_t)TILE;
        t_step = G;
    } else {
        // permuted bucket p = (b % 8) * 32 + b / 8: the buckets of one XCD are consecutive
        uint32_t* s_wsum = reinterpret_cast<uint32_t*>(smem + C::OFF_WSUM);
        uint32_t tiles = 0u;
        if (tid < 256) {
            const uint32_t b = (((uint32_t)tid & 31u) << 3) | ((uint32_t)tid >> 5);
            uint32_t cnt = a.src_counts[b << a.src_count_shift];
            if (cnt > a.src_stride) cnt = a.src_stride;   // the bucket overflowed: the flag is set, only stay in bounds
            s_bcnt[tid] = cnt;
            tiles = (cnt + (uint32_t)TILE - 1u) / (uint32_t)TILE;
        }
        uint32_t total;
        const uint32_t ex = block_excl_scan_u32<NT>(tiles, s_wsum, &total);
        if (tid < 256) s_pref[tid] = ex;
        if (tid == 0) s_pref[256] = total;
        __syncthreads();
        const uint32_t x = g & 7u;
        t_origin = 32u * x;
        ticket = s_pref[t_origin] + (g >> 3);
        t_end = s_pref[t_origin + 32u];
        t_step = G >> 3;
    }
    if (ticket >= t_end) return;

    auto tile_of = [&](uint32_t t) -> PersistTile {
        PersistTile r;
        if constexpr (PASS == 1) {
            r.base = t * (uint32_t)TILE;
            const uint32_t left = a.n - r.base;
            r.valid = left < (uint32_t)TILE ? left : (uint32_t)TILE;
            r.cursor_base = 0u;
        } else {
            // the bucket (of this XCD's 32) whose tiles include ticket t: s_pref[p] <= t < s_pref[p + 1]
            uint32_t p = t_origin;
#pragma unroll
            for (uint32_t s = 16u; s >= 1u; s >>= 1)
                if (s_pref[p + s] <= t) p += s;
            const uint32_t b = ((p & 31u) << 3) | (p >> 5);
            const uint32_t off = (t - s_pref[p]) * (uint32_t)TILE;
            const uint32_t room = s_bcnt[p] - off;
            r.base = b * a.src_stride + off;
            r.valid = room < (uint32_t)TILE ? room : (uint32_t)TILE;
            r.cursor_base = b * 256u;
        }
        return r;
    };
    const uint32_t wbase0 = (uint32_t)(w * 64 * K + lane * VEC);   // tile position of this lane's first element
    // Every vector-memory operation between a tile's prefetch and the wait for it must be UNCONDITIONAL and countable: gfx9 has
    // one counter (vmcnt) for loads, stores and atomics, in issue order, and the compiler turns "wait for these loads" into
    // s_waitcnt vmcnt(N) with N = the operations it can PROVE were issued after them.  A store behind a branch, or inside a loop,
    // counts as "maybe not issued": N = 0, and the wait drains the tile's stores as well -- which is all the persistent form is
    // meant to avoid.  So: tiles load with 16-byte BUFFER loads whose resource ends at the tile's last element (what lies beyond
    // reads as zero, no branch); the write-out is K unrolled buffer stores, and a position beyond the tile's elements gets an
    // offset outside the destination resource (the hardware drops the store).
    auto load_tile = [&](const PersistTile& t, E(&e)[K]) {
        // (the thread's tile position is hidden from loop-invariant code motion: the compiler otherwise keeps K addresses per phase
        // in registers across the whole tile loop, and the kernel spills)
        uint32_t wbase = wbase0;
        asm volatile("" : "+v"(wbase));
        // ONE load path for full and partial tiles: 16-byte buffer loads through a resource that ends at the tile's last element --
        // multi-dword buffer loads are range-checked per dword, what lies beyond reads as zero and is never looked at (pads are
        // told by their index).  Two paths into the same registers cost more than a branch: where they join, the compiler must
        // assume the other path's loads in flight and waits for them -- and with them for the previous tile's stores.
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<E*>(a.src + (size_t)t.base), 0,
                                                                           (int)(t.valid * (uint32_t)sizeof(E)), 0x00020000);
        const int off = (int)(wbase * (uint32_t)sizeof(E));
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, q * 64 * 16, LAUX);
            if constexpr (VEC == 4) {
                e[q * 4 + 0] = v.x; e[q * 4 + 1] = v.y; e[q * 4 + 2] = v.z; e[q * 4 + 3] = v.w;
            } else {
                e[q * 2 + 0] = (E)v.x | ((E)v.y << 32);
                e[q * 2 + 1] = (E)v.z | ((E)v.w << 32);
            }
        }
    };

    uint32_t* my_wcnt = s_wcnt + w * 256;
    uint16_t* __restrict__ my_wpos = reinterpret_cast<uint16_t*>(smem + C::OFF_WPOS) + w * 256;
    constexpr uint32_t SC = D16 ? 2u : (uint32_t)sizeof(E);   // bytes per element of the destination
    const __amdgpu_buffer_rsrc_t dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, (int)(a.dst_total * SC), 0x00020000);

    // The first tile's keys are waited for HERE, before the loop (the empty asm "uses" them): a wait for them inside the loop would
    // sit behind the next tile's prefetch on every iteration and so drain everything older -- the previous tile's stores.
    E cur[K], nxt[K];
    PersistTile tc = tile_of(ticket), tn{0u, 0u, 0u};
    load_tile(tc, cur);
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(cur[j]));
    uint32_t flag_prev = 0u;   // the overflow flag as this wave loaded it during the previous tile
    for (;;) {
        const uint32_t ticket_n = ticket + t_step;
        const bool has_next = ticket_n < t_end;
        // the overflow flag, requested with the next tile's keys (one more countable vector-memory operation in front of this
        // tile's stores: the wait at the bottom stays s_waitcnt vmcnt(K)) and looked at ONE TILE LATER, behind that tile's first
        // barrier: by then it has long arrived, and every wave sees every wave's view
        const uint32_t flag_now = __hip_atomic_load(a.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (has_next) {
            tn = tile_of(ticket_n);
            load_tile(tn, nxt);   // in flight through the whole body below
        }
        const uint32_t valid = tc.valid;
        const bool full = valid == (uint32_t)TILE;
        uint32_t tid_v = (uint32_t)tid;   // hidden from loop-invariant code motion, see load_tile
        asm volatile("" : "+v"(tid_v));
        const uint32_t wbase = (tid_v >> 6) * (uint32_t)(64 * K) + (tid_v & 63u) * (uint32_t)VEC;
        const int rem = (int)valid - (int)wbase;   // element (q, c) of this lane exists iff q*64*VEC + c < rem
        if (PASS == 1 && a.sample && place.top < (int)(8 * sizeof(E))) {
            // a key outside the sampled range would land in a wrong bucket: let the safety net sort instead
            const E pre = (E)place.prefix;
            E bad = E(0);
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int c = 0; c < VEC; ++c)
                    if (full || q * 64 * VEC + c < rem) bad |= (cur[q * VEC + c] >> place.top) ^ pre;
            if (bad != E(0)) __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- rank (returning DS atomics on the wave's own counters; pads of a partial tile take no part) -----------------------
#pragma unroll
        for (int b = lane; b < 256; b += 64) my_wcnt[b] = 0u;
        uint32_t rnk2[(K + 1) / 2];
        if (full) {
            constexpr int B = 8;
            static_assert(K % B == 0, "whole batches");
#pragma unroll
            for (int j0 = 0; j0 < K; j0 += B) {
                uint32_t rnk[B];
                rank_batch<E, B>(cur + j0, rnk, my_wcnt, start_bit);
#pragma unroll
                for (int j = 0; j < B; j += 2) rnk2[(j0 + j) >> 1] = rnk[j] | (rnk[j + 1] << 16);
            }
        } else {
#pragma unroll
            for (int j = 0; j < K; j += 2) {
                uint32_t r0 = 0u, r1 = 0u;
                if ((j / VEC) * 64 * VEC + (j % VEC) < rem)
                    r0 = __hip_atomic_fetch_add(&my_wcnt[digit_of<8>(cur[j], start_bit)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (((j + 1) / VEC) * 64 * VEC + ((j + 1) % VEC) < rem)
                    r1 = __hip_atomic_fetch_add(&my_wcnt[digit_of<8>(cur[j + 1], start_bit)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                rnk2[j >> 1] = r0 | (r1 << 16);
            }
        }
#pragma unroll
        for (int j = 0; j < (K + 1) / 2; ++j) asm volatile("" : "+v"(rnk2[j]));
        if (lane == 0) s_give[w] = flag_prev;
        lds_barrier();
        {   // leave together: nothing this workgroup writes from here on would be read
            const u32x4 g0 = *reinterpret_cast<const u32x4*>(s_give), g1 = *reinterpret_cast<const u32x4*>(s_give + 4);
            static_assert(NW <= 8, "one word per wave");
            uint32_t any = 0u;
            if (NW > 0) any |= g0.x; if (NW > 1) any |= g0.y; if (NW > 2) any |= g0.z; if (NW > 3) any |= g0.w;
            if (NW > 4) any |= g1.x; if (NW > 5) any |= g1.y; if (NW > 6) any |= g1.z; if (NW > 7) any |= g1.w;
            if (any) break;
        }
        // ---- every wave: counts of all waves for its lanes' digits -> tile offsets -> its own (wave, digit) positions -------------
        u32x4 cnt4 = {0u, 0u, 0u, 0u};
        u32x4 toff4 = {0u, 0u, 0u, 0u};
        {
            u32x4 pre4 = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(s_wcnt + i * 256 + 4 * lane);
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
        if (w == 0) {
            uint32_t* cp = a.cursors + ((size_t)(tc.cursor_base + 4u * (uint32_t)lane) << a.cursor_shift);
            const size_t step = (size_t)1 << a.cursor_shift;
            if (cnt4.x) at4.x = __hip_atomic_fetch_add(cp + 0 * step, cnt4.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cnt4.y) at4.y = __hip_atomic_fetch_add(cp + 1 * step, cnt4.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cnt4.z) at4.z = __hip_atomic_fetch_add(cp + 2 * step, cnt4.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cnt4.w) at4.w = __hip_atomic_fetch_add(cp + 3 * step, cnt4.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- scatter into tile-sorted order ------------------------------------------------------------------------------------------
        {
            constexpr int CH = K < 8 ? K : 8;
#pragma unroll
            for (int j0 = 0; j0 < K; j0 += CH) {
                uint32_t pos[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) pos[j] = my_wpos[digit_of<8>(cur[(j0 + j < K) ? j0 + j : K - 1], start_bit)];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const int jj = j0 + j;
                    if (jj < K) {
                        const uint32_t r = (rnk2[jj >> 1] >> (16 * (jj & 1))) & 0xffffu;
                        if (full || (jj / VEC) * 64 * VEC + (jj % VEC) < rem) s_elems[pos[j] + r] = cur[jj];
                    }
                }
            }
        }
        // every wave "uses" the cursor values here, not only wave 0: a register that MAY hold a pending atomic's result where the
        // paths join again would make the compiler wait for it at that register's next write -- half-way through the next tile's
        // ranking, behind that tile's prefetch, which drains this tile's oldest stores
        asm volatile("" : "+v"(at4.x), "+v"(at4.y), "+v"(at4.z), "+v"(at4.w));
        if (w == 0) {
            const bool over = (at4.x + cnt4.x > a.dst_stride) | (at4.y + cnt4.y > a.dst_stride) | (at4.z + cnt4.z > a.dst_stride) |
                              (at4.w + cnt4.w > a.dst_stride);
            if (over) __hip_atomic_fetch_or(a.flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t d0 = tc.cursor_base + 4u * (uint32_t)lane;
            if constexpr (PASS >= 2)   // slab of (bucket, digit): slot (bucket << w) | (digit & (2^w - 1)); w = 8: the same number
                d0 = ((tc.cursor_base >> 8) << a.seg_shift) + ((4u * (uint32_t)lane) & ((1u << a.seg_shift) - 1u));
            u32x4 go;   // destination BYTE offset = goff[digit] + tile position * SC
            go.x = (d0 + 0u) * a.dst_stride + at4.x - toff4.x;
            go.y = (d0 + 1u) * a.dst_stride + at4.y - toff4.y;
            go.z = (d0 + 2u) * a.dst_stride + at4.z - toff4.z;
            go.w = (d0 + 3u) * a.dst_stride + at4.w - toff4.w;
            *reinterpret_cast<u32x4*>(s_goff + 4 * lane) = go * SC;
        }
        lds_barrier();
        // ---- write-out: consecutive lanes -> consecutive tile positions -> contiguous runs per digit ----------------------------------
        {
            const uint32_t tsc = tid_v * SC;
            constexpr int WCH = K < 8 ? K : 8;   // LDS reads of a chunk go out together; chunks are kept apart (registers)
#pragma unroll
            for (int i0 = 0; i0 < K; i0 += WCH) {
                E v[WCH];
                uint32_t off[WCH];
#pragma unroll
                for (int i = 0; i < WCH; ++i) v[i] = s_elems[tid_v + (uint32_t)((i0 + i) * NT)];
#pragma unroll
                for (int i = 0; i < WCH; ++i) {
                    off[i] = s_goff[digit_of<8>(v[i], start_bit)] + tsc + (uint32_t)((i0 + i) * NT) * SC;
                    if (tid_v + (uint32_t)((i0 + i) * NT) >= valid) off[i] = 0xfffffff0u;   // beyond the resource: dropped
                }
#pragma unroll
                for (int i = 0; i < WCH; ++i) {
                    if constexpr (D16) {
                        __builtin_amdgcn_raw_buffer_store_b16((uint16_t)v[i], dst_rsrc, (int)off[i], 0, SAUX);
                    } else if constexpr (sizeof(E) == 4) {
                        __builtin_amdgcn_raw_buffer_store_b32((uint32_t)v[i], dst_rsrc, (int)off[i], 0, SAUX);
                    } else {
                        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
                        const u32x2_t t2 = {(uint32_t)v[i], (uint32_t)((uint64_t)v[i] >> 32)};
                        __builtin_amdgcn_raw_buffer_store_b64(t2, dst_rsrc, (int)off[i], 0, SAUX);
                    }
                }
                asm volatile("" ::: "memory");
            }
        }
        if (!has_next) break;
        flag_prev = flag_now;
        // (no barrier here: the next tile touches s_elems / s_goff / the positions only behind ITS first barrier, which every wave
        // reaches after its write-out; the wave's own counters are read by the other waves before this tile's second barrier)
        // the prefetched keys become the current ones: s_waitcnt vmcnt(K) -- the K stores above stay in flight
#pragma unroll
        for (int j = 0; j < K; ++j) {
            cur[j] = nxt[j];
            asm volatile("" : "+v"(cur[j]));
        }
        tc = tn;
        ticket = ticket_n;
    }
}

}   // namespace adlhip
