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

    if (threadIdx.x == 0) s_misc[0] = atomicAdd(ticket, 1u);
    __syncthreads();
    const uint32_t tile = s_misc[0];
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
                uint32_t t = tile - 1u;
                for (;;) {
                    const uint32_t* p = status + (size_t)t * BINS + b;
                    uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    uint32_t spins = 0u;
                    while ((v >> 30) == 0u) {
                        ++spins;
                        // give up after the bound, or as soon as any other waiter has given up (so a
                        // broken hand-off drains in one bound, not one bound per tile)
                        if (spins > kSpinBound ||
                            ((spins & 1023u) == 0u &&
                             __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                            atomicOr(fault, 0x10000u | (uint32_t)start_bit);
                            v = kFlagPfx;   // results are invalid; the host is told at adlhip_sync()
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    excl += v & kValMask;
                    if (v & kFlagPfx) break;
                    --t;   // tile 0 always publishes a prefix, so t never underflows
                }
                __hip_atomic_store(mine, kFlagPfx | ((excl + cnt) & kValMask), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            return gbase[b] + excl;
        });
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
