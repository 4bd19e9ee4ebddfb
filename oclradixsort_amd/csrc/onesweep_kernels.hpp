// onesweep_kernels.hpp -- single-sweep-per-digit LSD radix sort pass for gfx950 ("sort.algo" = 0).
//
// Same three logical steps as the reference's pass (histogram -> global prefix scan -> local sort +
// scatter; Tahoe/ParallelPrimitives/Pprims.cpp:357-398) but arranged so that every element is read
// once per pass instead of twice:
//   * ONE up-front kernel histograms every digit position of the sort (the digit histograms of an
//     LSD sort do not depend on element order), a tiny kernel reduces + scans them into per-pass
//     global digit bases;
//   * each pass is one kernel: a tile is ranked + locally sorted exactly as in radix_kernels.hpp and
//     obtains "how many elements with my digit precede my tile" by decoupled look-back over
//     per-tile status words instead of from a pre-scanned table.
//
// Inter-workgroup protocol (MI355X: 8 XCDs with private, mutually non-coherent L2s):
//   status[tile][digit] is ONE 32-bit word = {2-bit flag, 30-bit count}; it is written with a single
//   relaxed agent-scope atomic store and polled with relaxed agent-scope atomic loads (sc1 accesses:
//   served past the CU's L1).  The word itself is the only thing handed over -- no other memory is
//   published through it -- so no release/acquire fence is needed ("the data is the flag").
//   Tile ids come from an atomic ticket, so a tile only ever waits for tiles whose workgroups already
//   run: forward progress does not depend on dispatch order or residency.  Every spin is bounded; a
//   wait that exceeds its bound raises the device fault word, which adlhip_sync() reports.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_kernels.hpp"

namespace adlhip {

constexpr int kOsNT = 256;
constexpr int kOsK = 16;
constexpr uint32_t kOsTile = kOsNT * kOsK;

constexpr int kHistNT = 1024;
constexpr uint32_t kHistChunk = 64 * 1024;   // minimum elements per histogram workgroup

struct PassDesc {
    int num_passes;
    uint8_t start_bit[16];
    uint8_t nbits[16];
};

constexpr uint32_t kFlagAgg = 1u << 30;    // tile's own digit count is available
constexpr uint32_t kFlagPfx = 2u << 30;    // inclusive prefix over tiles 0..t is available
constexpr uint32_t kValMask = (1u << 30) - 1u;
constexpr uint32_t kSpinBound = 1u << 20;  // polls (each followed by s_sleep) before giving up

constexpr int kLookbackWindow = 4;   // predecessor status words fetched per round trip

// Poll one status word until its flag is set.  Bounded: after kSpinBound polls, or as soon as any
// other waiter has given up (so a broken hand-off drains in one bound, not one bound per tile), raise
// the fault word and return a terminating value; adlhip_sync() then reports the sort as invalid.
__device__ __forceinline__ uint32_t wait_status(const uint32_t* p, uint32_t* fault, int start_bit)
{
    uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t spins = 0u;
    while ((v >> 30) == 0u) {
        ++spins;
        if (spins > kSpinBound ||
            ((spins & 1023u) == 0u && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            atomicOr(fault, 0x10000u | (uint32_t)start_bit);
            return kFlagPfx;
        }
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return v;
}

// Decoupled look-back for one digit: sum the counts of the preceding tiles back to the nearest tile
// that already knows its inclusive prefix.  W predecessors are fetched per round trip (independent
// loads in flight together), because a walk costs (number of round trips) x (fabric latency) and
// with hundreds of tiles in flight the nearest finished prefix is typically several tiles back.
template <int BINS, int W>
__device__ __forceinline__ uint32_t lookback_exclusive(const uint32_t* status, uint32_t tile, int b,
                                                       uint32_t* fault, int start_bit)
{
    uint32_t excl = 0u;
    int t = (int)tile - 1;
    for (;;) {
        uint32_t v[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int ti = t - i;
            v[i] = ti >= 0 ? __hip_atomic_load(status + (size_t)ti * BINS + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : kFlagPfx;   // before tile 0: nothing precedes
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            uint32_t x = v[i];
            if ((x >> 30) == 0u) x = wait_status(status + (size_t)(t - i) * BINS + b, fault, start_bit);
            excl += x & kValMask;
            if (x & kFlagPfx) return excl;
        }
        t -= W;
    }
}

// All digit histograms of the sort in one read of the data: partial[(wg*P + p)*256 + d].
template <typename E>
__global__ __launch_bounds__(kHistNT) void onesweep_hist_kernel(const E* __restrict__ src,
                                                                uint32_t* __restrict__ partial, uint32_t n,
                                                                uint32_t chunk, PassDesc desc)
{
    __shared__ uint32_t hist[16 * 256];
    const int tid = (int)threadIdx.x;
    const int P = desc.num_passes;
    for (int i = tid; i < P * 256; i += kHistNT) hist[i] = 0u;
    __syncthreads();

    const uint64_t begin64 = (uint64_t)blockIdx.x * chunk;
    if (begin64 < n) {
        const uint32_t begin = (uint32_t)begin64;
        const uint32_t end = (uint32_t)((begin64 + chunk < n) ? begin64 + chunk : n);
        auto bump = [&](E x) {
#pragma unroll 4
            for (int p = 0; p < P; ++p) {
                const int sb = desc.start_bit[p];
                uint32_t wsel;
                if constexpr (sizeof(E) == 8) wsel = (sb & 32) ? (uint32_t)((uint64_t)x >> 32) : (uint32_t)x;
                else wsel = (uint32_t)x;
                const uint32_t d = (wsel >> (sb & 31)) & ((1u << desc.nbits[p]) - 1u);
                atomicAdd(&hist[p * 256 + d], 1u);
            }
        };
        constexpr int VEC = 16 / (int)sizeof(E);
        struct alignas(16) Vec { E v[VEC]; };
        const uint32_t nvec = (end - begin) / VEC;
        const Vec* vsrc = reinterpret_cast<const Vec*>(src + begin);
        uint32_t i = (uint32_t)tid;
        for (; i + 3u * kHistNT < nvec; i += 4u * kHistNT) {
            Vec a = vsrc[i], b = vsrc[i + kHistNT], c = vsrc[i + 2 * kHistNT], d4 = vsrc[i + 3 * kHistNT];
#pragma unroll
            for (int k = 0; k < VEC; ++k) { bump(a.v[k]); bump(b.v[k]); bump(c.v[k]); bump(d4.v[k]); }
        }
        for (; i < nvec; i += kHistNT) {
            Vec a = vsrc[i];
#pragma unroll
            for (int k = 0; k < VEC; ++k) bump(a.v[k]);
        }
        for (uint32_t s = begin + nvec * VEC + (uint32_t)tid; s < end; s += kHistNT) bump(src[s]);
    }
    __syncthreads();
    uint32_t* out = partial + (size_t)blockIdx.x * P * 256;
    for (int i = tid; i < P * 256; i += kHistNT) out[i] = hist[i];
}

// One workgroup per pass: sum the partial histograms over workgroups, exclusive-scan the 256 totals.
// gbase[p*256 + d] = number of elements whose pass-p digit is < d.
__global__ __launch_bounds__(1024) void onesweep_hist_reduce_kernel(const uint32_t* __restrict__ partial,
                                                                    uint32_t* __restrict__ gbase,
                                                                    uint32_t n_wgs, int P)
{
    __shared__ uint32_t red[4][256];
    __shared__ uint32_t wsum[1024 / 64 + 1];
    const int p = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    const int d = tid & 255;
    const int g = tid >> 8;
    uint32_t s = 0u;
#pragma unroll 8
    for (uint32_t wg = (uint32_t)g; wg < n_wgs; wg += 4u) s += partial[((size_t)wg * P + p) * 256 + d];
    red[g][d] = s;
    __syncthreads();
    uint32_t tot = 0u;
    if (tid < 256) tot = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    const uint32_t ex = block_excl_scan_u32<1024>(tot, wsum, nullptr);
    if (tid < 256) gbase[p * 256 + tid] = ex;
}

// One tile per workgroup; ticket-ordered tile ids; decoupled look-back per digit.
template <typename E, int NBITS, int NT, int K, int RANK>
__global__ __launch_bounds__(NT) void onesweep_pass_kernel(const E* __restrict__ src, E* __restrict__ dst,
                                                           const uint32_t* __restrict__ gbase,
                                                           uint32_t* status, uint32_t* ticket, uint32_t* fault,
                                                           uint32_t n, int start_bit, uint32_t num_tiles)
{
    using C = TileCfg<E, NBITS, NT, K>;
    constexpr int BINS = C::BINS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* s_misc = reinterpret_cast<uint32_t*>(smem + C::OFF_MISC);

#ifdef ADLHIP_STAMPS
    unsigned long long t_entry = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry)::"memory");
#endif
    if (threadIdx.x == 0) s_misc[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_misc[0];
#ifdef ADLHIP_STAMPS
    if (threadIdx.x == 0 && g_stamp_buf && tile < num_tiles) g_stamp_buf[(size_t)tile * 16 + 11] = t_entry;
#endif
    if (tile >= num_tiles) return;   // cannot happen (grid == num_tiles); keeps indices in range

    const uint32_t tile_base = tile * (uint32_t)C::TILE;
    const uint32_t left = n - tile_base;
    const uint32_t valid = left < (uint32_t)C::TILE ? left : (uint32_t)C::TILE;

    sort_scatter_tile<E, NBITS, NT, K, RANK>(
        src, dst, tile_base, valid, n, start_bit, smem, [&](int b, uint32_t cnt) -> uint32_t {
            uint32_t* mine = status + (size_t)tile * BINS + b;
            uint32_t excl = 0u;
            if (tile == 0u) {
                __hip_atomic_store(mine, kFlagPfx | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(mine, kFlagAgg | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                excl = lookback_exclusive<BINS, kLookbackWindow>(status, tile, b, fault, start_bit);
                __hip_atomic_store(mine, kFlagPfx | ((excl + cnt) & kValMask), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            return gbase[b] + excl;
        });
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Look-back for the 4 consecutive digits one bookkeeping lane owns.  A tile's 256 status words are one
// 1-KiB row written by ONE wave-wide 16-byte-per-lane sc1 store, so a row is read back with one
// wave-wide 16-byte sc1 buffer load; each 32-bit word is self-describing, so tearing between words
// of a row is harmless.  W rows are in flight per round trip.
template <int BINS, int W>
__device__ __forceinline__ u32x4 lookback_exclusive4(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile, int lane,
                                                     uint32_t* fault, int start_bit)
{
    u32x4 excl = {0u, 0u, 0u, 0u};
    uint32_t done = 0u;
    int t = (int)tile - 1;
    for (;;) {
        u32x4 v[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int ti = t - i;
            const uint32_t off = ((uint32_t)(ti > 0 ? ti : 0) * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u;
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16 /* sc1 */);
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int ti = t - i;
            u32x4 x = v[i];
            if (ti < 0) {   // before tile 0: nothing precedes
                x = (u32x4){kFlagPfx, kFlagPfx, kFlagPfx, kFlagPfx};
            } else {
                uint32_t spins = 0u;
                while (((x.x >> 30) == 0u) | ((x.y >> 30) == 0u) | ((x.z >> 30) == 0u) | ((x.w >> 30) == 0u)) {
                    ++spins;
                    if (spins > kSpinBound ||
                        ((spins & 1023u) == 0u &&
                         __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        atomicOr(fault, 0x20000u | (uint32_t)start_bit);
                        return excl;   // results are invalid; the host is told at adlhip_sync()
                    }
                    __builtin_amdgcn_s_sleep(2);
                    const uint32_t off = ((uint32_t)ti * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u;
                    x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
                }
            }
            if (!(done & 1u)) { excl.x += x.x & kValMask; done |= (x.x & kFlagPfx) ? 1u : 0u; }
            if (!(done & 2u)) { excl.y += x.y & kValMask; done |= (x.y & kFlagPfx) ? 2u : 0u; }
            if (!(done & 4u)) { excl.z += x.z & kValMask; done |= (x.z & kFlagPfx) ? 4u : 0u; }
            if (!(done & 8u)) { excl.w += x.w & kValMask; done |= (x.w & kFlagPfx) ? 8u : 0u; }
            if (done == 15u) return excl;
        }
        t -= W;
    }
}

constexpr int kLookbackBlock = 16;   // tiles per look-back block (second-level prefix every 16 tiles)
constexpr int kBlockPolls = 6;       // polls of a block prefix before walking past it

// One 16-byte status row for this lane's 4 digits, waited for until all 4 flags are set (bounded).
__device__ __forceinline__ bool row_ready(const u32x4& x)
{
    return ((x.x >> 30) != 0u) & ((x.y >> 30) != 0u) & ((x.z >> 30) != 0u) & ((x.w >> 30) != 0u);
}

__device__ __forceinline__ u32x4 wait_row(__amdgpu_buffer_rsrc_t rsrc, uint32_t off, u32x4 x, uint32_t* fault,
                                          int start_bit)
{
    uint32_t spins = 0u;
    while (!row_ready(x)) {
        ++spins;
        if (spins > kSpinBound ||
            ((spins & 1023u) == 0u && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            atomicOr(fault, 0x40000u | (uint32_t)start_bit);
            return (u32x4){kFlagAgg, kFlagAgg, kFlagAgg, kFlagAgg};   // results invalid; host is told at sync
        }
        __builtin_amdgcn_s_sleep(1);
        x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
    }
    return x;
}

// excl += counts of tile rows hi, hi-1, ..., lo (8 rows in flight per round trip).  Tile rows are
// published unconditionally at each tile's barrier A, so these waits never chain.
template <int BINS>
__device__ __forceinline__ void add_tile_rows(__amdgpu_buffer_rsrc_t rsrc, int hi, int lo, uint32_t lane_off,
                                              u32x4& excl, uint32_t* fault, int start_bit)
{
    for (int r = hi; r >= lo; r -= 8) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rr = (r - k) > lo ? (r - k) : lo;
            v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (uint32_t)rr * (uint32_t)(BINS * 4) + lane_off, 0, 16);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (r - k >= lo) {
                const u32x4 x = wait_row(rsrc, (uint32_t)(r - k) * (uint32_t)(BINS * 4) + lane_off, v[k], fault, start_bit);
                excl += x & kValMask;
            }
        }
    }
}

// Two-level decoupled look-back for the 4 digits of one bookkeeping lane.
//   level 1: every tile publishes its digit counts (one 16-byte-per-lane row) -- no dependencies;
//   level 2: the last tile of every block of kLookbackBlock tiles publishes the inclusive prefix
//            through the end of its block.
// Exclusive prefix of tile t in block b = counts of the earlier tiles of block b (read together, they
// never wait on a chain) + prefix of block b-1.  If that block prefix is not there after a few polls,
// add block b-1's tile rows instead and look one block further back.  The serial dependency runs over
// blocks, not tiles: 16x fewer hops per pass than a tile-by-tile look-back, which at ~40 tiles/us is
// what the fabric round trip can sustain.
template <int BINS, int B>
__device__ __forceinline__ u32x4 lookback_blocked(__amdgpu_buffer_rsrc_t rsrc, uint32_t bp_base, uint32_t tile, int lane,
                                                  uint32_t* fault, int start_bit)
{
    u32x4 excl = {0u, 0u, 0u, 0u};
    const uint32_t lane_off = 16u * (uint32_t)lane;
    const int blk = (int)(tile / (uint32_t)B);
    add_tile_rows<BINS>(rsrc, (int)tile - 1, blk * B, lane_off, excl, fault, start_bit);
    ADLHIP_STAMP(tile, 12);
#ifdef ADLHIP_STAMPS
    unsigned long long polls = 0, walked = 0;
#endif
    for (int c = blk - 1; c >= 0; --c) {
        const uint32_t off = bp_base + (uint32_t)c * (uint32_t)(BINS * 4) + lane_off;
        u32x4 bp = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
#ifdef ADLHIP_STAMPS
        if (c == blk - 1) { if (row_ready(bp)) polls += 1000; ADLHIP_STAMP(tile, 13); }
#endif
        for (int poll = 0; poll < kBlockPolls && !row_ready(bp); ++poll) {
            __builtin_amdgcn_s_sleep(4);
            bp = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 16);
#ifdef ADLHIP_STAMPS
            ++polls;
#endif
        }
        if (row_ready(bp)) {
            excl += bp & kValMask;
            break;
        }
        add_tile_rows<BINS>(rsrc, c * B + B - 1, c * B, lane_off, excl, fault, start_bit);
#ifdef ADLHIP_STAMPS
        ++walked;
#endif
    }
#ifdef ADLHIP_STAMPS
    if (threadIdx.x == 0 && g_stamp_buf) { g_stamp_buf[(size_t)tile * 16 + 14] = polls + 1; g_stamp_buf[(size_t)tile * 16 + 15] = walked + 1; }
#endif
    return excl;
}

// Persistent one-sweep pass.  Each workgroup pulls tiles from an atomic ticket until none are left.
// Per tile:
//   rank (one returning DS atomic per element)                                   | barrier A
//   wave 0 ("bookkeeping wave", 4 digits per lane, 16-byte LDS/status accesses): fold the per-wave
//     counts, PUBLISH the tile's digit counts at once, scan them into tile offsets,
//     write (wave, digit) tile positions back                                     | barrier B
//   all waves: scatter elements to their tile-sorted LDS slot, then immediately issue the loads of
//     the NEXT tile (ticket taken at the top of the iteration) so HBM latency hides behind the rest;
//   wave 0 meanwhile: decoupled look-back (as late as possible, so predecessors' counts published
//     at THEIR barrier A have had time to become visible), publish inclusive prefix, global offsets
//                                                                                 | barrier C
//   write-out: consecutive lanes store consecutive elements of a digit's run.
// Forward progress: a tile waits only for tiles with smaller tickets, whose workgroups are running.
template <typename E, int NBITS, int NT, int K, int RANK, bool PERSIST>
__global__ __launch_bounds__(NT) void onesweep_persistent_kernel(const E* __restrict__ src, E* __restrict__ dst,
                                                                 const uint32_t* __restrict__ gbase,
                                                                 uint32_t* status, uint32_t status_bytes,
                                                                 uint32_t* ticket, uint32_t* fault, uint32_t n,
                                                                 int start_bit, uint32_t num_tiles)
{
    using C = TileCfg<E, NBITS, NT, K>;
    constexpr int BINS = C::BINS;
    constexpr int NW = C::NW;
    constexpr int BK_LANES = BINS / 4;   // bookkeeping lanes: 4 digits each
    static_assert(BK_LANES <= 64, "one wave keeps the books");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem + C::OFF_ELEMS);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);   // [NW][BINS]
    uint32_t* __restrict__ s_goff = reinterpret_cast<uint32_t*>(smem + C::OFF_GOFF);   // [BINS]
    uint32_t* __restrict__ s_misc = reinterpret_cast<uint32_t*>(smem + C::OFF_MISC);

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    const bool bk = (w == 0) && (lane < BK_LANES);

    if (tid == 0) s_misc[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    uint32_t tile = s_misc[0];
    if (tile >= num_tiles) return;
    __syncthreads();   // everyone has read s_misc[0] before it is rewritten

    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(status, 0, (int)status_bytes, 0x00020000);
    u32x4 gb = {0u, 0u, 0u, 0u};
    if (bk) gb = *reinterpret_cast<const u32x4*>(gbase + 4 * lane);

    const uint32_t wbase = (uint32_t)(w * 64 * K + lane);
    E e[K];
    auto load_tile = [&](uint32_t t) {
        // one 64-bit pointer per tile + constant offsets; the tail predicate compares a per-lane
        // remainder with constants, so nothing per-element is loop-invariant (and hoisted into VGPRs)
        const uint32_t tile_base = t * (uint32_t)C::TILE;
        const E* p = src + (size_t)tile_base + wbase;
        const uint32_t left = n - tile_base;
        if (left >= (uint32_t)C::TILE) {
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = p[j * 64];
        } else {
            const int rem = (int)left - (int)wbase;   // elements of this lane's column that exist: j*64 < rem
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = (j * 64 < rem) ? p[j * 64] : ~E(0);
        }
    };
    load_tile(tile);

    for (;;) {
        uint32_t next_ticket = 0xffffffffu;
        ADLHIP_STAMP(tile, 0);
        if (PERSIST && tid == 0) next_ticket = atomicAdd(ticket, 1u);   // its latency hides behind the ranking

#pragma unroll
        for (int b = lane; b < BINS; b += 64) my_wcnt[b] = 0u;
        uint32_t rnk2[(K + 1) / 2];   // two 16-bit in-wave ranks (< 64*K) per register
        {
            uint32_t rnk[K];
            rank_in_wave<E, NBITS, K, RANK>(e, rnk, my_wcnt, start_bit);
#pragma unroll
            for (int j = 0; j < K; j += 2) rnk2[j >> 1] = rnk[j] | ((j + 1 < K ? rnk[j + 1] : 0u) << 16);
        }
        // make the packed ranks and the elements opaque here: otherwise the compiler keeps the 32-bit
        // ranks and the per-element LDS addresses of the ranking phase alive across the barriers
#pragma unroll
        for (int j = 0; j < (K + 1) / 2; ++j) asm volatile("" : "+v"(rnk2[j]));
#pragma unroll
        for (int j = 0; j < K; ++j) asm volatile("" : "+v"(e[j]));
        ADLHIP_STAMP(tile, 2);
        __syncthreads();   // A
        ADLHIP_STAMP(tile, 3);

        u32x4 cnt4 = {0u, 0u, 0u, 0u};
        u32x4 toff4 = {0u, 0u, 0u, 0u};
        if (w == 0) {   // whole wave: the DPP scan needs all 64 lanes active
            if (lane < BK_LANES) {
#pragma unroll
                for (int i = 0; i < NW; ++i) cnt4 += *reinterpret_cast<const u32x4*>(s_wcnt + i * BINS + 4 * lane);
                // publish this tile's digit counts right away
                __builtin_amdgcn_raw_buffer_store_b128(cnt4 | kFlagAgg, rsrc,
                                                       (tile * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u, 0, 16);
            }
            const uint32_t s4 = cnt4.x + cnt4.y + cnt4.z + cnt4.w;
            const uint32_t ex = wave_incl_scan_u32(s4) - s4;
            toff4.x = ex;
            toff4.y = ex + cnt4.x;
            toff4.z = toff4.y + cnt4.y;
            toff4.w = toff4.z + cnt4.z;
            if (lane < BK_LANES) {
                u32x4 run = toff4;
#pragma unroll
                for (int i = 0; i < NW; ++i) {   // second sweep over the rows keeps one row live, not NW
                    u32x4* row = reinterpret_cast<u32x4*>(s_wcnt + i * BINS + 4 * lane);
                    const u32x4 ci = *row;
                    *row = run;   // tile position of (wave i, digit)'s first element
                    run += ci;
                }
            }
            if (lane == 0) s_misc[0] = next_ticket;
        }
        ADLHIP_STAMP(tile, 4);
        __syncthreads();   // B
        ADLHIP_STAMP(tile, 5);
        const uint32_t next_tile = PERSIST ? s_misc[0] : 0xffffffffu;
        const bool have_next = PERSIST && next_tile < num_tiles;

        {   // LDS reads of the (wave, digit) positions go out CH at a time ahead of the CH writes that use
            // them (reads and writes may alias as far as the compiler knows, so it will not batch them itself)
            constexpr int CH = K < 8 ? K : 8;
#pragma unroll
            for (int j0 = 0; j0 < K; j0 += CH) {
                uint32_t pos[CH];
#pragma unroll
                for (int j = 0; j < CH; ++j) pos[j] = my_wcnt[digit_of<NBITS>(e[j0 + j], start_bit)];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    const uint32_t r = (rnk2[(j0 + j) >> 1] >> (16 * ((j0 + j) & 1))) & 0xffffu;
                    s_elems[pos[j] + r] = e[j0 + j];
                }
            }
        }
        const uint32_t tile_base = tile * (uint32_t)C::TILE;
        const uint32_t left = n - tile_base;
        const uint32_t valid = left < (uint32_t)C::TILE ? left : (uint32_t)C::TILE;

        ADLHIP_STAMP(tile, 6);
        if (bk) {
            u32x4 excl = {0u, 0u, 0u, 0u};
            const uint32_t bp_base = num_tiles * (uint32_t)(BINS * 4);   // block rows follow the tile rows
            if (tile != 0u) excl = lookback_blocked<BINS, kLookbackBlock>(rsrc, bp_base, tile, lane, fault, start_bit);
            if ((tile % (uint32_t)kLookbackBlock) == (uint32_t)(kLookbackBlock - 1))   // last tile of a block
                __builtin_amdgcn_raw_buffer_store_b128(((excl + cnt4) & kValMask) | kFlagPfx, rsrc,
                                                       bp_base + (tile / (uint32_t)kLookbackBlock) * (uint32_t)(BINS * 4) +
                                                           16u * (uint32_t)lane, 0, 16);
            *reinterpret_cast<u32x4*>(s_goff + 4 * lane) = gb + excl - toff4;
        }
        ADLHIP_STAMP(tile, 7);
        // prefetch the next tile: in flight during (the other waves') look-back wait and the write-out.
        // Wave 0 issues it after its look-back: vmcnt retires in order, so older key loads would
        // otherwise sit in front of every status-row wait.
        if (have_next) load_tile(next_tile);
        ADLHIP_STAMP(tile, 8);
        __syncthreads();   // C
        ADLHIP_STAMP(tile, 9);

#pragma unroll 4
        for (int i = 0; i < K; ++i) {
            if (i * NT < (int)valid - tid) {   // tile position tid + i*NT exists
                const E v = s_elems[tid + i * NT];
                const uint32_t d = digit_of<NBITS>(v, start_bit);
                const uint32_t g = s_goff[d] + (uint32_t)(tid + i * NT);
                if (g < n) dst[(size_t)g] = v;   // always true for a sound offset (guards a faulted look-back)
            }
        }
        ADLHIP_STAMP(tile, 10);
        if (!have_next) break;
        tile = next_tile;
        // no barrier here: the next iteration only touches wave-private LDS before its barrier A
    }
}

// counts[k] = number of keys whose top log2(num_buckets) bits equal k, from the 256 top-byte totals.
__global__ __launch_bounds__(256) void fold_buckets_kernel(const uint32_t* __restrict__ totals,
                                                           uint32_t* __restrict__ counts, int num_buckets)
{
    __shared__ uint32_t t[256];
    t[threadIdx.x] = totals[threadIdx.x];
    __syncthreads();
    const int per = 256 / num_buckets;
    if ((int)threadIdx.x < num_buckets) {
        uint32_t s = 0u;
        for (int i = 0; i < per; ++i) s += t[threadIdx.x * per + i];
        counts[threadIdx.x] = s;
    }
}

}  // namespace adlhip
