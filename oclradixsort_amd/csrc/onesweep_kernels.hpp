// onesweep_kernels.hpp -- single-sweep-per-digit LSD radix sort pass for gfx950 ("sort.algo" = 0).
//
// Same three logical steps as the reference's pass (histogram -> global prefix scan -> local sort +
// scatter; Tahoe/ParallelPrimitives/Pprims.cpp:357-398) but arranged so that every element is read
// once per pass instead of twice:
//   * ONE up-front kernel histograms every digit position of the sort (digit histograms of an LSD
//     sort do not depend on element order); tiny kernels turn them into per-pass offset tables;
//   * each pass is one kernel: a tile is ranked and locally sorted in LDS and learns "how many
//     elements with my digit precede my tile" by decoupled look-back over per-tile status rows
//     instead of from a pre-scanned table.
//
// MI355X specifics that shape it (measured with in-kernel stamps, profiles/):
//   * A status hand-off between workgroups costs ~1K cycles round trip under load (8 XCDs, private
//     L2s: the words travel through the fabric), while at full rate a new tile reaches its look-back
//     every ~40-80 cycles.  A single serial prefix chain over all tiles cannot keep up (measured: the
//     predecessor's prefix was ready on first poll 0.5 % of the time and tiles spent 40-50 % of their
//     life in look-back).  So every pass runs kChains = 16 INDEPENDENT look-back chains: chain c of
//     pass p is the set of elements whose PREVIOUS digit has top nibble c -- contiguous in the array
//     that pass p-1 produced (pass 0: sixteen equal slices of the input).  The per-chain digit bases
//     come from joint histograms (previous top nibble x digit) that are order-independent and hence
//     part of the one up-front histogram kernel.  Tile arrivals per chain are 16x sparser, and a
//     look-back typically ends after one or two status rows.
//   * status[tile][digit] is ONE 32-bit word = {2-bit flag, 30-bit count}; a tile's 256 words are one
//     1-KiB row written by ONE wave-wide 16-byte-per-lane sc1 (write-through) buffer store and read
//     with wave-wide 16-byte sc1 buffer loads (served past the CU's L1).  Each word is
//     self-describing, so nothing else is handed over and no release/acquire fence is needed.
//   * one "bookkeeping wave" (4 digits per lane, 16-byte LDS accesses, DPP scan) folds the per-wave
//     counts, publishes them immediately, and does the look-back as late as possible (after the LDS
//     scatter) so that predecessors' rows have had time to become visible.
//   * Tile ids come from atomic tickets (one counter per chain), so a tile only ever waits for tiles
//     whose workgroups already run: forward progress does not depend on dispatch order or residency.
//     Every spin is bounded; a wait that exceeds its bound raises the device fault word, which
//     adlhip_sync() reports.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_kernels.hpp"

#ifndef ADLHIP_WRITE_UNROLL
#define ADLHIP_WRITE_UNROLL 8
#endif
// Diagnostic builds only (never the shipped library; results are WRONG, only the timing is of interest):
// -DADLHIP_NO_LOOKBACK replaces the look-back by the analytic prefix of uniform keys (nothing polled or published)


namespace adlhip {

constexpr int kMaxPassesFwd = 16;
#ifndef ADLHIP_CHAIN_BITS
#define ADLHIP_CHAIN_BITS 4
#endif
constexpr int kChainBits = ADLHIP_CHAIN_BITS;
constexpr int kChains = 1 << kChainBits;   // independent look-back chains per pass (top bits of the previous digit)
constexpr int kTicketStride = 32;    // u32 words between two chains' ticket counters: one 128-byte line each (sixteen counters
                                     // in ONE line queue every ticket of a pass on the same atomic unit)
constexpr int kTicketVecs = kMaxPassesFwd * kChains * kTicketStride * 4 / 16;   // 16-byte vectors of the ticket area
constexpr int kHistNT = 1024;
constexpr uint32_t kHistChunk = 64 * 1024;   // minimum elements per histogram workgroup
constexpr int kMaxPasses = 16;
static_assert(kMaxPasses == kMaxPassesFwd, "ticket area");

struct PassDesc {
    int num_passes;
    uint8_t start_bit[kMaxPasses];
    uint8_t nbits[kMaxPasses];
};

// Offset tables of one pass, built on the device by onesweep_tables_kernel.
struct PassTable {
    uint32_t cbase[kChains][256];        // global index of the first element of (chain c, digit d)
    uint32_t chunk_start[kChains + 1];   // element index (in the pass's input order) where chain c starts
    uint32_t tile_start[kChains + 1];    // first tile id of chain c; [kChains] = number of tiles of the pass
    uint32_t pad[30];
};
static_assert(sizeof(PassTable) % 16 == 0, "tables are read with 16-byte loads");

constexpr uint32_t kFlagAgg = 1u << 30;    // tile's own digit count is available
constexpr uint32_t kFlagPfx = 2u << 30;    // inclusive prefix over the chain's tiles 0..t is available
constexpr uint32_t kValMask = (1u << 30) - 1u;
constexpr uint32_t kSpinBound = 1u << 20;  // polls (each followed by s_sleep) before giving up
constexpr int kLookbackWindow = 4;         // predecessor status rows fetched per round trip

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Device fault words: fault[0] is "live" -- waiters of the sort that is running poll it so that a broken hand-off
// drains in one bound; the first kernel of every sort clears it, so one failure cannot poison later sorts.
// fault[1] is sticky: it keeps every code raised since the host last looked (adlhip_sync / adlhip_fault_check).
__device__ __forceinline__ void raise_fault(uint32_t* fault, uint32_t code)
{
    atomicOr(fault, code);
    atomicOr(fault + 1, code);
}

// joint-histogram bins of a pass: (chain, digit) -> kChains << nbits
__host__ __device__ inline uint32_t joint_bins(int nbits) { return (uint32_t)kChains << nbits; }

// ------------------------------------------------------------------------------------------
// Up-front histograms: for every pass p the joint histogram of (chain, digit):
//   p >= 1: chain = top nibble of pass p-1's digit (bits [start_p - 4, start_p)), digit = pass p's;
//   p == 0: chain = which sixteenth ("slice") of the input the element lies in (workgroup-uniform).
// partial[wg * total_bins + bin_off[p] + (p == 0 ? chain * (1 << nbits[0]) + digit : digit * 16 + chain)]
// ------------------------------------------------------------------------------------------
// P (number of passes) is a template parameter so that the loop over passes unrolls with constant
// indices: the pass descriptors then sit in SGPRs (indexed dynamically they were re-loaded from the
// kernarg segment for every key).
// The body of one histogram workgroup, `vblock` of `vgrid`, run by NT threads: the kernel below (one workgroup each), or the large
// sort's safety net (hybrid_kernels.hpp coop_onesweep_sort), whose resident workgroups take the histogram workgroups in turns.
template <typename E, int P, int NT>
__device__ __forceinline__ void onesweep_hist_body(const E* __restrict__ src, uint32_t* __restrict__ partial, uint32_t n, uint32_t chunk,
                                                   uint32_t slice0, const PassDesc& desc, uint32_t total_bins,
                                                   u32x4* __restrict__ tickets, u32x4* __restrict__ status, size_t status_vecs,
                                                   uint32_t* __restrict__ fault, uint32_t vblock, uint32_t vgrid, unsigned char* smem)
{
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    const int tid = (int)threadIdx.x;
    // Zero the chain tickets (1 KiB) and the status rows of all passes on the way: fire-and-forget stores that
    // drain under the key stream, instead of two memset launches in front of every sort.
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        if (vblock == 0) {
            for (int i = tid; i < kTicketVecs; i += NT) tickets[i] = z;
            if (tid == 0) fault[0] = 0u;   // the live fault word belongs to the sort that starts here
        }
        const size_t stride = (size_t)vgrid * NT;
        for (size_t i = (size_t)vblock * NT + (size_t)tid; i < status_vecs; i += stride) status[i] = z;
    }
    for (uint32_t i = (uint32_t)tid; i < total_bins; i += NT) hist[i] = 0u;
    __syncthreads();

    const uint64_t begin64 = (uint64_t)vblock * chunk;
    if (begin64 < n) {
        const uint32_t begin = (uint32_t)begin64;
        const uint32_t end = (uint32_t)((begin64 + chunk < n) ? begin64 + chunk : n);
        // pass 0: the workgroup's range lies inside ONE input slice (chunk divides slice0)
        const uint32_t chain0 = begin / slice0;
        auto bin_of = [&](E x, int p) -> uint32_t {
            const int sb = desc.start_bit[p];
            const int nb = desc.nbits[p];
            // pass 0 (the chain is workgroup-uniform; it starts at bit 0 in a sort, at bit 24 as the one-pass MSB partition):
            // bin = chain * 2^nb + digit.  With the uniform chain in the LOW bits every lane would hit one of 4 LDS banks.
            if (p == 0) return (chain0 << nb) | ((uint32_t)(x >> sb) & ((1u << nb) - 1u));
            // p >= 1: bin = digit * kChains + chain = the nb + kChainBits contiguous key bits from sb - kChainBits (one v_bfe_u32)
            uint32_t v;
            if constexpr (sizeof(E) == 8) v = (uint32_t)((uint64_t)x >> (sb - kChainBits));
            else v = (uint32_t)x >> (sb - kChainBits);
            return v & ((1u << (nb + kChainBits)) - 1u);
        };
        auto bump = [&](E x) {
            uint32_t off = 0u;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                atomicAdd(&hist[off + bin_of(x, p)], 1u);
                off += joint_bins(desc.nbits[p]);
            }
        };
        // 4 vectors at once.  Guards against same-address LDS atomics (64 lanes on one word serialise; sorted input
        // made this kernel 4.7x slower), decided per pass from a screen on ONE key per lane -- the number of lanes
        // whose bin differs from their left neighbour's ("run heads"; the first lane of each row of 16 always counts):
        //   * 4 heads and the exact test agrees -> the whole wave is in one bin: one lane adds the total.  The exact
        //     test is shared by all passes: `diff` ORs key ^ (first lane's first key) over the lane's 16 keys, and
        //     pass p is uniform iff no lane has a diff bit inside p's bin bits;
        //   * <= 16 heads (sorted / clustered input) -> run-aggregated adds: per instruction every run of equal
        //     neighbouring lanes is added once, by its head, with the run length as the addend;
        //   * otherwise (always, on random keys) plain atomics.  The screen costs ~5 instructions per pass and 16 keys.
        typedef typename std::conditional<sizeof(E) == 8, uint64_t, uint32_t>::type K;
        const int lane = tid & 63;
        auto bump4 = [&](const auto& a, const auto& b, const auto& c, const auto& d4) {
            constexpr int V = 16 / (int)sizeof(E);
            auto mask_of = [&](int p) -> K {
                const int sb = desc.start_bit[p];
                const int nb = desc.nbits[p];
                return (p == 0) ? ((K)((1u << nb) - 1u) << sb) : ((K)((1u << (nb + kChainBits)) - 1u) << (sb - kChainBits));
            };
            auto plain = [&](int p, uint32_t off) {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    atomicAdd(&hist[off + bin_of(a.v[k], p)], 1u);
                    atomicAdd(&hist[off + bin_of(b.v[k], p)], 1u);
                    atomicAdd(&hist[off + bin_of(c.v[k], p)], 1u);
                    atomicAdd(&hist[off + bin_of(d4.v[k], p)], 1u);
                }
            };
            // left neighbour's bin (row_shr:1); the first lane of a row keeps ~bin, i.e. always starts a run
            auto left_of = [](uint32_t bin) -> uint32_t {
                return (uint32_t)__builtin_amdgcn_update_dpp((int)~bin, (int)bin, 0x111, 0xf, 0xf, false);
            };
            auto add_runs = [&](uint32_t* h, uint32_t bin) {   // all 64 lanes active
                const bool head = left_of(bin) != bin;
                const uint64_t heads = __ballot(head);
                // distance to the next head above this lane (bit 63 - lane = a virtual head at lane 64)
                const uint64_t above = ((heads >> 1) >> lane) | (1ull << (63 - lane));
                const uint32_t len = (uint32_t)__builtin_ctzll(above) + 1u;
                if (head) atomicAdd(&h[bin], len);
            };
            auto runs = [&](int p, uint32_t off) {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    add_runs(hist + off, bin_of(a.v[k], p));
                    add_runs(hist + off, bin_of(b.v[k], p));
                    add_runs(hist + off, bin_of(c.v[k], p));
                    add_runs(hist + off, bin_of(d4.v[k], p));
                }
            };
            const bool full = __ballot(true) == ~0ull;   // the last iteration may run with some lanes off: plain only
            uint32_t nheads[P];
            bool any_uniform = false;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const uint32_t b0 = bin_of(a.v[0], p);
                nheads[p] = full ? (uint32_t)__popcll(__ballot(left_of(b0) != b0)) : 64u;
                any_uniform |= nheads[p] == 4u;
            }
            K f0 = 0, diff = 0;
            if (any_uniform) {
                if constexpr (sizeof(E) == 8) {
                    const uint64_t k0 = (uint64_t)a.v[0];
                    f0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(k0 >> 32)) << 32) |
                         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)k0);
                } else {
                    f0 = (K)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a.v[0]);
                }
#pragma unroll
                for (int k = 0; k < V; ++k)
                    diff |= ((K)a.v[k] ^ f0) | ((K)b.v[k] ^ f0) | ((K)c.v[k] ^ f0) | ((K)d4.v[k] ^ f0);
            }
            uint32_t off = 0u;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (nheads[p] == 4u && __all((diff & mask_of(p)) == 0)) {
                    if (lane == 0) atomicAdd(&hist[off + bin_of((E)f0, p)], (uint32_t)(4 * V * 64));
                } else if (nheads[p] <= 16u) {
                    runs(p, off);
                } else {
                    plain(p, off);
                }
                off += joint_bins(desc.nbits[p]);
            }
        };
        constexpr int VEC = 16 / (int)sizeof(E);
        struct alignas(16) Vec { E v[VEC]; };
        const uint32_t nvec = (end - begin) / VEC;
        const Vec* vsrc = reinterpret_cast<const Vec*>(src + begin);
        uint32_t i = (uint32_t)tid;
        // software-pipelined: the next four vectors are in flight while the current four are histogrammed
        if (i + 3u * NT < nvec) {
            Vec a = vsrc[i], b = vsrc[i + NT], c = vsrc[i + 2 * NT], d4 = vsrc[i + 3 * NT];
            i += 4u * NT;
            for (; i + 3u * NT < nvec; i += 4u * NT) {
                const Vec na = vsrc[i], nb = vsrc[i + NT], nc = vsrc[i + 2 * NT], nd = vsrc[i + 3 * NT];
                bump4(a, b, c, d4);
                a = na; b = nb; c = nc; d4 = nd;
            }
            bump4(a, b, c, d4);
        }
        for (; i < nvec; i += NT) {
            Vec a = vsrc[i];
#pragma unroll
            for (int k = 0; k < VEC; ++k) bump(a.v[k]);
        }
        for (uint32_t s = begin + nvec * VEC + (uint32_t)tid; s < end; s += NT) bump(src[s]);
    }
    __syncthreads();
    uint32_t* out = partial + (size_t)vblock * total_bins;
    for (uint32_t i = (uint32_t)tid; i < total_bins; i += NT) out[i] = hist[i];
}

template <typename E, int P>
__global__ __launch_bounds__(kHistNT) void onesweep_hist_kernel(const E* __restrict__ src,
                                                                uint32_t* __restrict__ partial, uint32_t n,
                                                                uint32_t chunk, uint32_t slice0, PassDesc desc,
                                                                uint32_t total_bins, u32x4* __restrict__ tickets,
                                                                u32x4* __restrict__ status, size_t status_vecs,
                                                                uint32_t* __restrict__ fault)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    onesweep_hist_body<E, P, kHistNT>(src, partial, n, chunk, slice0, desc, total_bins, tickets, status, status_vecs, fault, blockIdx.x,
                                      gridDim.x, smem);
}

// Sum the partial joint histograms over workgroups: joint[bin], all passes (grid = ceil(total_bins/256)).
template <int NT>
__device__ __forceinline__ void onesweep_hist_reduce_body(const uint32_t* __restrict__ partial, uint32_t* __restrict__ joint, uint32_t n_wgs,
                                                          uint32_t total_bins, uint32_t vblock)
{
    constexpr int G = NT / 256;   // groups of 256 threads, each sums every G-th workgroup's partials
    __shared__ uint32_t red[G][256];
    const int tid = (int)threadIdx.x;
    const uint32_t bin = vblock * 256u + (uint32_t)(tid & 255);
    const int g = tid >> 8;
    uint32_t s = 0u;
    if (bin < total_bins) {
#pragma unroll 8
        for (uint32_t wg = (uint32_t)g; wg < n_wgs; wg += (uint32_t)G) s += partial[(size_t)wg * total_bins + bin];
    }
    red[g][tid & 255] = s;
    __syncthreads();
    if (tid < 256 && bin < total_bins) {
        uint32_t t = 0u;
#pragma unroll
        for (int i = 0; i < G; ++i) t += red[i][tid];
        joint[bin] = t;
    }
    __syncthreads();   // (red is written again by the caller's next block)
}
ADLHIP_KERNEL __global__ __launch_bounds__(1024) void onesweep_hist_reduce_kernel(const uint32_t* __restrict__ partial,
                                                                    uint32_t* __restrict__ joint, uint32_t n_wgs,
                                                                    uint32_t total_bins)
{
    onesweep_hist_reduce_body<1024>(partial, joint, n_wgs, total_bins, blockIdx.x);
}

// One workgroup (256 threads) per pass: joint histogram -> PassTable.
//   gbase[d]       = #elements with digit < d                         (exclusive scan of the digit totals)
//   cbase[c][d]    = gbase[d] + sum_{c' < c} joint[c'][d]
//   chunk_start[c] = sum of the joint rows < c  (pass 0: the input slices; pass p: the number of
//                    elements whose pass p-1 digit has a top nibble < c, i.e. where they start in the
//                    array pass p-1 produced)
//   tile_start[c]  = sum_{c' < c} ceil(len_c' / tile)
template <int NT>
__device__ __forceinline__ void onesweep_tables_body(const uint32_t* __restrict__ joint, PassTable* __restrict__ tables, const PassDesc& desc,
                                                     uint32_t tile, int p)
{
    __shared__ uint32_t wsum[NT / 64 + 1];
    __shared__ uint32_t rowpart[NT / 64][kChains];   // per wave: partial chain lengths (threads beyond the bins contribute zeros)
    const int tid = (int)threadIdx.x;
    const int nb = desc.nbits[p];
    const uint32_t bins = 1u << nb;
    uint32_t off = 0u;
    for (int q = 0; q < p; ++q) off += joint_bins(desc.nbits[q]);
    const uint32_t* J = joint + off;   // pass 0: [kChains][bins]; later passes: [bins][kChains]
    PassTable* T = tables + p;

    uint32_t col[kChains];
    uint32_t tot = 0u;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
        col[c] = ((uint32_t)tid < bins) ? J[p == 0 ? (uint32_t)c * bins + (uint32_t)tid : (uint32_t)tid * (uint32_t)kChains + (uint32_t)c] : 0u;
        tot += col[c];
    }
    // chain lengths = row sums: one DPP wave scan per chain, the four wave totals are added by the thread that needs them
    // (sixteen block-wide scans = 32 barriers before)
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
        const uint32_t r = wave_incl_scan_u32(col[c]);
        if ((tid & 63) == 63) rowpart[tid >> 6][c] = r;
    }
    uint32_t total;
    const uint32_t gb = block_excl_scan_u32<NT>(tot, wsum, &total);   // two barriers: rowpart is visible after them
    {   // every element has the same digit here: the pass would move nothing (read by the safety net's passes, which then skip it)
        const int constant = __syncthreads_or((uint32_t)tid < bins && tot == total && total != 0u);
        if (tid == 0) T->pad[0] = constant ? 1u : 0u;
    }
    uint32_t run = gb;
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
        if ((uint32_t)tid < bins) T->cbase[c][tid] = run;
        run += col[c];
    }
    if (tid <= kChains) {   // thread c: where chain c starts (elements and tiles); thread kChains: the totals
        uint32_t es = 0u, ts = 0u;
        for (int c = 0; c < tid; ++c) {
            uint32_t len = 0u;
#pragma unroll
            for (int wv = 0; wv < NT / 64; ++wv) len += rowpart[wv][c];
            es += len;
            ts += (len + tile - 1u) / tile;
        }
        T->chunk_start[tid] = es;   // [kChains] == n
        T->tile_start[tid] = ts;
    }
    __syncthreads();   // (the scratch is written again by the caller's next pass)
}
ADLHIP_KERNEL __global__ __launch_bounds__(256) void onesweep_tables_kernel(const uint32_t* __restrict__ joint,
                                                              PassTable* __restrict__ tables, PassDesc desc,
                                                              uint32_t tile)
{
    onesweep_tables_body<256>(joint, tables, desc, tile, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// look-back
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool row_ready(const u32x4& x)
{
    return ((x.x >> 30) != 0u) & ((x.y >> 30) != 0u) & ((x.z >> 30) != 0u) & ((x.w >> 30) != 0u);
}

// Decoupled look-back for the 4 consecutive digits one bookkeeping lane owns, over the rows
// tile-1, tile-2, ... down to `first` (the chain's first tile): sum counts until a row that carries
// an inclusive prefix.  W rows are in flight per round trip.
template <int BINS, int W>
__device__ __forceinline__ u32x4 lookback_exclusive4(__amdgpu_buffer_rsrc_t rsrc, uint32_t tile, uint32_t first,
                                                     int lane, uint32_t* fault, int start_bit)
{
    u32x4 excl = {0u, 0u, 0u, 0u};
    uint32_t done = 0u;
    int t = (int)tile - 1;
    const int lo = (int)first;
    for (;;) {
        u32x4 v[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int ti = (t - i) > lo ? (t - i) : lo;
            v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((uint32_t)ti * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u, 0,
                                                         16 /* sc1 */);
        }
#pragma unroll
        for (int i = 0; i < W; ++i) {
            const int ti = t - i;
            if (ti < lo) return excl;   // walked past the chain's first tile (cannot happen: it carries a prefix)
            u32x4 x = v[i];
            uint32_t spins = 0u;
            while (!row_ready(x)) {
                ++spins;
                // give up after the bound, or as soon as any other waiter has given up (so a broken
                // hand-off drains in one bound, not one bound per tile)
                if (spins > kSpinBound ||
                    ((spins & 1023u) == 0u && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                    raise_fault(fault, 0x20000u | (uint32_t)start_bit);
                    return excl;   // results are invalid; the host is told at adlhip_sync() / adlhip_fault_check()
                }
                __builtin_amdgcn_s_sleep(1);
                x = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ((uint32_t)ti * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u,
                                                          0, 16);
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

// ------------------------------------------------------------------------------------------
// The pass kernel: one tile per workgroup.
//   ticket (chain chosen by blockIdx % 16, next chains probed when one is exhausted)
//   load (wave-striped) -> rank (one returning DS atomic per element)                 | barrier A
//   every wave: fold the per-wave counts of all waves (4 digits per lane), DPP-scan the digit totals into tile offsets and
//     write the 16-bit tile positions of its own (wave, digit) runs; wave 0 also PUBLISHES the tile's digit counts at once
//   all waves: scatter elements to their tile-sorted LDS slot (no barrier in between: a wave reads only its own positions);
//   wave 0 then: look-back over its chain, publish inclusive prefix, global offsets    | barrier C
//   write-out: consecutive lanes store consecutive elements of a digit's run.
// ------------------------------------------------------------------------------------------
// One tile of a pass: takes a ticket (home chain `home`, then the others in turn), ranks, scatters, looks back, writes out.
// Returns false -- for every thread of the workgroup alike -- when every chain is fully ticketed.  Called once per workgroup by
// the kernel below, and in a loop by the resident workgroups of the large sort's safety net (hybrid_kernels.hpp
// coop_onesweep_sort): a tile only ever waits for tiles whose tickets were taken before its own, i.e. for workgroups that are
// already at work, so the loop cannot deadlock; LDS is reused safely because a wave reaches the next tile's second barrier only
// after every wave has left this tile's write-out.
template <typename IO, int NBITS, int NT, int K, int RANK>
__device__ __forceinline__ bool onesweep_tile(const IO& io, const PassTable* __restrict__ table, uint32_t* status, uint32_t status_bytes,
                                              uint32_t* tickets, uint32_t* fault, uint32_t n, int start_bit, uint32_t home,
                                              unsigned char* smem)
{
    typedef typename IO::elem_t E;
    using C = TileCfg<E, NBITS, NT, K>;
    constexpr int BINS = C::BINS;
    constexpr int NW = C::NW;
    constexpr int BK_LANES = BINS / 4;   // bookkeeping lanes: 4 digits each
    static_assert(BK_LANES <= 64, "one wave keeps the books");
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem + C::OFF_ELEMS);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + C::OFF_WCNT);   // [NW][BINS]
    uint32_t* __restrict__ s_goff = reinterpret_cast<uint32_t*>(smem + C::OFF_GOFF);   // [BINS]
    uint32_t* __restrict__ s_misc = reinterpret_cast<uint32_t*>(smem + C::OFF_MISC);

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    const bool bk = (w == 0) && (lane < BK_LANES);

#ifdef ADLHIP_STAMPS
    // diagnostic: time and place of the workgroup's first instruction (stored once the tile id is known)
    unsigned long long st_entry = 0ull, st_rt = 0ull;
    uint32_t st_hw = 0u, st_xcc = 0u;
    if (tid == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_entry)::"memory");
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt)::"memory");
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(st_hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(st_xcc));
    }
#endif
    // ---- ticket: (chain, index in chain) -------------------------------------------------------
    // ONE global round trip before the tile's keys can be requested: wave 0 fetches the pass's 34 table words
    // (lanes 0..16 tile_start, 17..33 chunk_start) and takes a ticket of its home chain (lane 34) in the same
    // breath -- speculatively: a ticket beyond the chain's tile count is simply void.  (Done one after the other
    // -- tile count, ticket, table words -- this preamble cost three dependent round trips per tile.)
    if (w == 0) {
        const uint32_t c0 = home % (uint32_t)kChains;
        uint32_t v = 0u;
        if (lane <= kChains) v = table->tile_start[lane];
        else if (lane <= 2 * kChains + 1) v = table->chunk_start[lane - (kChains + 1)];
        else if (lane == 2 * kChains + 2) v = atomicAdd(&tickets[c0 * (uint32_t)kTicketStride], 1u);
        auto at = [&](uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
        uint32_t chain = c0;
        uint32_t index = at(2 * kChains + 2);
        if (index >= at(c0 + 1u) - at(c0)) {   // home chain exhausted (or empty): probe the others in turn
            chain = 0xffffffffu;
            for (int k = 1; k < kChains; ++k) {
                const uint32_t c = (c0 + (uint32_t)k) % (uint32_t)kChains;
                const uint32_t tiles_c = at(c + 1u) - at(c);
                if (tiles_c == 0u) continue;
                uint32_t i = 0u;
                if (lane == 0) i = atomicAdd(&tickets[c * (uint32_t)kTicketStride], 1u);
                i = (uint32_t)__builtin_amdgcn_readfirstlane((int)i);
                if (i < tiles_c) { chain = c; index = i; break; }
            }
        }
        if (lane == 0) {
            s_misc[0] = chain;
            s_misc[1] = index;
            if (chain != 0xffffffffu) {
                const uint32_t elem0 = at(kChains + 1u + chain) + index * (uint32_t)C::TILE;
                s_misc[2] = at(chain);                              // first status row of the chain
                s_misc[3] = elem0;
                s_misc[4] = at(kChains + 2u + chain) - elem0;       // elements from the tile's start to the chain's end
            }
        }
    }
    __syncthreads();
    const uint32_t chain = s_misc[0];
    if (chain == 0xffffffffu) return false;   // every chain is fully ticketed (the grid is an upper bound on the tile count)
    const uint32_t index = s_misc[1];
    const uint32_t first_row = s_misc[2];
    const uint32_t tile = first_row + index;                  // status row of this tile
    const uint32_t elem0 = s_misc[3];
    const uint32_t left = s_misc[4];
    const uint32_t valid = left < (uint32_t)C::TILE ? left : (uint32_t)C::TILE;
    ADLHIP_STAMP(tile, 0);
#ifdef ADLHIP_STAMPS
    if (tid == 0 && g_stamp_buf) {
        g_stamp_buf[(size_t)tile * 16 + 11] = st_entry;
        g_stamp_buf[(size_t)tile * 16 + 12] = (unsigned long long)st_hw | ((unsigned long long)st_xcc << 32);
        g_stamp_buf[(size_t)tile * 16 + 13] = st_rt;
    }
#endif

    const bool scaled = dst_fits32<IO>(n);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(status, 0, (int)status_bytes, 0x00020000);
    u32x4 gb = {0u, 0u, 0u, 0u};
    if (bk) gb = *reinterpret_cast<const u32x4*>(&table->cbase[chain][4 * lane]);

    // ---- load, wave-striped: one 64-bit pointer per tile + constant offsets; the tail predicate
    // compares a per-lane remainder with constants, so nothing per-element is loop-invariant ----------
    const uint32_t wbase = (uint32_t)(w * 64 * K + lane);
    E e[K];
    {
        const typename IO::Cursor p = io.cursor((size_t)elem0 + wbase);
        if (valid == (uint32_t)C::TILE) {
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = p.at(j * 64);
        } else {
            const int rem = (int)valid - (int)wbase;   // element j of this lane's column exists iff j*64 < rem
#pragma unroll
            for (int j = 0; j < K; ++j) e[j] = (j * 64 < rem) ? p.at(j * 64) : ~E(0);
        }
    }

    // ---- rank -------------------------------------------------------------------------------------
#pragma unroll
    for (int b = lane; b < BINS; b += 64) my_wcnt[b] = 0u;
    uint32_t rnk2[(K + 1) / 2];   // two 16-bit in-wave ranks (< 64*K) per register
    {
        uint32_t rnk[K];
        rank_in_wave<E, NBITS, K, RANK>(e, rnk, my_wcnt, start_bit);
#pragma unroll
        for (int j = 0; j < K; j += 2) rnk2[j >> 1] = rnk[j] | ((j + 1 < K ? rnk[j + 1] : 0u) << 16);
    }
    // make the packed ranks and the elements opaque here: otherwise the compiler keeps the 32-bit ranks
    // and the per-element LDS addresses of the ranking phase alive across the barriers (+50 VGPRs)
#pragma unroll
    for (int j = 0; j < (K + 1) / 2; ++j) asm volatile("" : "+v"(rnk2[j]));
#pragma unroll
    for (int j = 0; j < K; ++j) asm volatile("" : "+v"(e[j]));
    ADLHIP_STAMP(tile, 2);
    __syncthreads();   // A
    ADLHIP_STAMP(tile, 3);

    // ---- every wave: fold the per-wave counts, scan the digit totals, write the tile positions of ITS OWN (wave, digit)
    // runs.  (One bookkeeping wave used to do this for all eight while seven waited: 2.2K cycles + a barrier of a 28K-cycle
    // tile.  Each wave now reads the NW count rows itself -- 8 x 16 bytes per lane -- and writes one row of 16-bit
    // positions that only it will read, so no barrier separates this from the LDS scatter.)
    u32x4 cnt4 = {0u, 0u, 0u, 0u};
    u32x4 toff4 = {0u, 0u, 0u, 0u};
    uint16_t* __restrict__ my_wpos = reinterpret_cast<uint16_t*>(smem + C::OFF_WPOS) + w * BINS;
    {
        u32x4 pre4 = {0u, 0u, 0u, 0u};   // elements of my digits in the waves before mine
        if (lane < BK_LANES) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(s_wcnt + i * BINS + 4 * lane);
                cnt4 += r;
                if (i < w) pre4 += r;
            }
            if (w == 0) {
                // publish this tile's digit counts right away (the chain's first tile: they ARE its prefix)
                const uint32_t flag = index == 0u ? kFlagPfx : kFlagAgg;
                __builtin_amdgcn_raw_buffer_store_b128(cnt4 | flag, rsrc, (tile * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u, 0,
                                                       16 /* sc1 */);
            }
        }
        const uint32_t s4 = cnt4.x + cnt4.y + cnt4.z + cnt4.w;
        const uint32_t ex = wave_incl_scan_u32(s4) - s4;   // whole wave: the DPP scan needs all 64 lanes active
        toff4.x = ex;
        toff4.y = ex + cnt4.x;
        toff4.z = toff4.y + cnt4.y;
        toff4.w = toff4.z + cnt4.z;
        if (lane < BK_LANES) {
            const u32x4 p4 = toff4 + pre4;   // < TILE <= 65536
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 packed = {p4.x | (p4.y << 16), p4.z | (p4.w << 16)};
            *reinterpret_cast<u32x2*>(my_wpos + 4 * lane) = packed;
        }
    }
    ADLHIP_STAMP(tile, 4);
    ADLHIP_STAMP(tile, 5);

    // ---- scatter into tile-sorted order -----------------------------------------------------------
    {   // the LDS reads of the (wave, digit) positions go out CH at a time ahead of the CH writes that use
        // them (as far as the compiler knows they may alias, so it will not batch them itself)
        constexpr int CH = K < 8 ? K : 8;
#pragma unroll
        for (int j0 = 0; j0 < K; j0 += CH) {
            uint32_t pos[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) pos[j] = my_wpos[digit_of<NBITS>(e[(j0 + j < K) ? j0 + j : K - 1], start_bit)];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (j0 + j < K) {   // K need not be a multiple of CH
                    const uint32_t r = (rnk2[(j0 + j) >> 1] >> (16 * ((j0 + j) & 1))) & 0xffffu;
                    s_elems[pos[j] + r] = e[j0 + j];
                }
            }
        }
    }
    ADLHIP_STAMP(tile, 6);

    // ---- look-back (as late as possible) -------------------------------------------------------------
    if (bk) {
        u32x4 excl = {0u, 0u, 0u, 0u};
#ifdef ADLHIP_NO_LOOKBACK
        excl.x = excl.y = excl.z = excl.w = index * (uint32_t)(C::TILE / BINS);
#else
        if (index != 0u) {
            excl = lookback_exclusive4<BINS, kLookbackWindow>(rsrc, tile, first_row, lane, fault, start_bit);
            __builtin_amdgcn_raw_buffer_store_b128(((excl + cnt4) & kValMask) | kFlagPfx, rsrc,
                                                   (tile * (uint32_t)BINS + 4u * (uint32_t)lane) * 4u, 0, 16);
        }
#endif
        const u32x4 go = gb + excl - toff4;   // dst index = goff[d] + tile position (mod 2^32); bytes on the buffer-store path
        *reinterpret_cast<u32x4*>(s_goff + 4 * lane) = scaled ? go * (uint32_t)IO::kStoreScale : go;
    }
    ADLHIP_STAMP(tile, 7);
    __syncthreads();   // C
    ADLHIP_STAMP(tile, 9);

    // ---- write-out ---------------------------------------------------------------------------------
    write_out_tile<IO, NBITS, NT, K, ADLHIP_WRITE_UNROLL>(io, s_elems, s_goff, valid, n, start_bit, scaled);
    ADLHIP_STAMP(tile, 10);
#ifdef ADLHIP_STAMPS
    if (tid == 0 && g_stamp_buf) {
        unsigned long long rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
        g_stamp_buf[(size_t)tile * 16 + 14] = rt;
    }
#endif
    return true;
}

template <typename IO, int NBITS, int NT, int K, int RANK>
__global__ __launch_bounds__(NT) void onesweep_chain_kernel(IO io, const PassTable* __restrict__ table, uint32_t* status,
                                                            uint32_t status_bytes, uint32_t* tickets, uint32_t* fault,
                                                            uint32_t n, int start_bit, const uint32_t* __restrict__ dyn_start_bit)
{
    if (dyn_start_bit) {   // the mid-size sort picks its digit position on the device (MidDyn: start_bit, low_bits, mode)
        if (dyn_start_bit[2] != 0u) return;   // ... and may hand the input to its cooperative LSD kernel instead
        start_bit = (int)dyn_start_bit[0];
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    onesweep_tile<IO, NBITS, NT, K, RANK>(io, table, status, status_bytes, tickets, fault, n, start_bit, blockIdx.x, smem);
}

// totals[d] = sum over the 16 chains of pass 0's joint histogram (chain-major: joint[chain * 256 + digit]; 8-bit digits)
ADLHIP_KERNEL __global__ __launch_bounds__(256) void fold_joint_kernel(const uint32_t* __restrict__ joint, uint32_t* __restrict__ totals)
{
    uint32_t s = 0u;
#pragma unroll
    for (int c = 0; c < kChains; ++c) s += joint[c * 256 + (int)threadIdx.x];
    totals[threadIdx.x] = s;
}

// counts[k] = number of keys whose top log2(num_buckets) bits equal k, from the 256 top-byte totals.
ADLHIP_KERNEL __global__ __launch_bounds__(256) void fold_buckets_kernel(const uint32_t* __restrict__ totals,
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
