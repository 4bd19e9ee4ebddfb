// hybrid_kernels.hpp -- "sort.algo" = 2: two MSD one-sweep passes + one LDS-resident finish, for 32-bit keys.
//
// The LSD sort of Pprims::radixSort (Tahoe/ParallelPrimitives/Pprims.cpp:304-406) moves every element through
// global memory once per digit.  A pass of this library costs ~120 us at 64 Mi keys whatever its digit width
// (profiles/r2_pass_time_vs_digit_width.txt), so the lever left is the NUMBER of global passes:
//   A  one-sweep pass on the TOP 7 bits of the sorted range        (128 buckets)
//   B  one-sweep pass on the next 7 bits, bucket by bucket          (16384 segments of n / 16384 elements)
//   C  every segment is loaded into LDS once, finished there on the remaining <= 18 bits (two local passes
//      of <= 9 bits) and written back in place, fully coalesced.
// 3 sweeps + one histogram read instead of 4 + 1.  The result is the same array any stable sort produces
// (total order + stability => unique output), so parity with the reference's CPU sort
// (Tahoe/Algorithm/Sort/RadixSort.cpp:10-104) is unaffected.
//
// What makes it safe for ANY input: the one up-front histogram (14 bits, ONE LDS atomic per key) yields every
// segment's size before anything is moved; if a segment would not fit the finishing kernel's LDS tile the
// tables kernel raises a device-side mode word and the classic LSD passes run instead (their launches are
// always enqueued; each kernel of the path not taken returns at its first instruction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "radix_kernels.hpp"
#include "onesweep_kernels.hpp"

namespace adlhip {

constexpr int kMsdBits = 7;                     // digit width of passes A and B
constexpr int kMsdBuckets = 1 << kMsdBits;      // 128
constexpr int kSegments = 1 << (2 * kMsdBits);  // 16384
constexpr int kLocalPassesMax = 3;               // local passes of the finishing kernel (LBITS bits each at most)

constexpr uint32_t kModeHybrid = 0u;            // mode word values (work buffer)
constexpr uint32_t kModeClassic = 1u;

// ------------------------------------------------------------------------------------------
// C: finish segments in LDS.  A workgroup takes segments blockIdx.x, blockIdx.x + gridDim.x, ... (the grid is
// sized to the resident slots: a workgroup per segment spent a fifth of its life being launched, and nothing
// was loading while a workgroup sorted).  The segment [seg_start[s], seg_start[s+1]) holds at most NT*K elements
// (guaranteed by the mode word; checked again here).  `low_bits` (1..3*LBITS) bits starting at bit 0 are sorted
// with up to three stable local passes of at most LBITS bits: per-wave returning DS atomics give the in-wave
// rank (radix_kernels.hpp rank_in_wave has the argument for why that is the stable rank), the per-wave counts
// are folded and scanned by the threads that own the bins, elements go to LDS at their sorted position.  The
// last pass leaves the tile in LDS and the workgroup writes it back over the segment, consecutive threads to
// consecutive addresses.  In place: everything of a segment was read before its first store.
// Software pipeline: the keys of the NEXT segment are requested before the local passes of the current one
// (they land in registers while the workgroup sorts; vmcnt retires in order and these loads are older than the
// current segment's stores, so consuming them never waits for a store), its bounds one segment earlier still.
// Slots beyond the segment's size take no part (no pad keys: hundreds of pads on one LDS counter serialise).
// ------------------------------------------------------------------------------------------
template <typename E, int NT, int K, int LBITS>
__global__ __launch_bounds__(NT) void segment_sort_kernel(E* data, const uint32_t* __restrict__ seg_start,
                                                          uint32_t num_segments, uint32_t low_bits,
                                                          const uint32_t* __restrict__ gate, uint32_t gate_value,
                                                          uint32_t* fault)
{
    if (gate && *gate != gate_value) return;
    constexpr int NW = NT / 64;
    constexpr int CAP = NT * K;
    constexpr int BINS = 1 << LBITS;
    constexpr int BPT = (BINS + NT - 1) / NT;          // bins per bookkeeping thread (2 at 256 threads, else 1)
    constexpr int BK_THREADS = BINS / BPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    E* __restrict__ s_elems = reinterpret_cast<E*>(smem);
    uint32_t* __restrict__ s_wcnt = reinterpret_cast<uint32_t*>(smem + sizeof(E) * CAP);   // [NW][BINS]
    uint32_t* __restrict__ s_wsum = s_wcnt + NW * BINS;

    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    uint32_t* my_wcnt = s_wcnt + w * BINS;
    const uint32_t stride = gridDim.x;

    // bounds of segment s (a segment beyond the tile is skipped and reported: never sorted wrongly in silence)
    auto bounds = [&](uint32_t s, uint32_t& begin, uint32_t& m) {
        begin = 0u;
        m = 0u;
        if (s < num_segments) {
            begin = seg_start[s];
            m = seg_start[s + 1] - begin;
            if (m > (uint32_t)CAP) {
                if (tid == 0) atomicOr(fault + 1, 0x40000u);   // sticky word
                m = 0u;
            }
        }
    };
    // wave-striped: with keff = ceil(m / NT) items per thread in use, wave w owns elements [w*64*keff, (w+1)*64*keff)
    auto fetch = [&](E (&x)[K], uint32_t begin, uint32_t m) {
        const int keff = (int)((m + (uint32_t)NT - 1u) / (uint32_t)NT);
        const uint32_t wbase = (uint32_t)(w * 64 * keff + lane);
        const E* p = data + begin + wbase;
        const int rem = (int)m - (int)wbase;
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (j < keff) x[j] = (j * 64 < rem) ? p[j * 64] : E(0);
    };

    uint32_t seg = blockIdx.x;
    uint32_t begin, m, nbegin, nm;
    bounds(seg, begin, m);
    bounds(seg + stride, nbegin, nm);
    E e[K];
    fetch(e, begin, m);

    // digits as even as possible: low_bits = 18, LBITS = 9 -> 9 + 9; 18 with LBITS = 8 -> 6 + 6 + 6
    const int npass = ((int)low_bits + LBITS - 1) / LBITS;
    for (; seg < num_segments; seg += stride) {
        uint32_t nnbegin, nnm;
        bounds(seg + 2u * stride, nnbegin, nnm);
        E en[K];
        fetch(en, nbegin, nm);   // in flight during the local passes below
        if (m != 0u) {
            const int keff = (int)((m + (uint32_t)NT - 1u) / (uint32_t)NT);
            const uint32_t wbase = (uint32_t)(w * 64 * keff + lane);
            const int rem = (int)m - (int)wbase;          // item j of this lane exists iff j*64 < rem
            int wave_valid = (int)m - w * 64 * keff;      // elements of this wave: a prefix in (item, lane) order
            wave_valid = wave_valid < 0 ? 0 : (wave_valid > 64 * keff ? 64 * keff : wave_valid);
            int sb = 0;
            for (int p = 0; p < npass; ++p) {
                const int nb = ((int)low_bits - sb + (npass - p - 1)) / (npass - p);
                const uint32_t mask = (1u << nb) - 1u;
                const int bins = 1 << nb;
                auto digit = [&](E x) -> uint32_t { return ((uint32_t)x >> sb) & mask; };
                for (int b = lane; b < bins; b += 64) my_wcnt[b] = 0u;
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
                __syncthreads();
                // bookkeeping threads: fold the per-wave counts of their bins, scan over bins, write tile positions back
                uint32_t wc[NW][BPT];
                uint32_t mine = 0u;
                const bool bk = tid < BK_THREADS && tid * BPT < bins;
                if (bk) {
#pragma unroll
                    for (int i = 0; i < NW; ++i)
#pragma unroll
                        for (int q = 0; q < BPT; ++q) {
                            wc[i][q] = s_wcnt[i * BINS + tid * BPT + q];
                            mine += wc[i][q];
                        }
                }
                const uint32_t toff = block_excl_scan_u32<NT>(mine, s_wsum, nullptr);
                if (bk) {
                    uint32_t run = toff;
#pragma unroll
                    for (int q = 0; q < BPT; ++q)
#pragma unroll
                        for (int i = 0; i < NW; ++i) {
                            s_wcnt[i * BINS + tid * BPT + q] = run;
                            run += wc[i][q];
                        }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (j < keff && j * 64 < rem) s_elems[my_wcnt[digit(e[j])] + rnk[j]] = e[j];
                __syncthreads();
                if (p + 1 < npass) {
#pragma unroll
                    for (int j = 0; j < K; ++j)
                        if (j < keff && j * 64 < rem) e[j] = s_elems[wbase + (uint32_t)(j * 64)];
                    __syncthreads();
                }
                sb += nb;
            }
            E* out = data + begin;
            for (uint32_t i = (uint32_t)tid; i < m; i += (uint32_t)NT) out[i] = s_elems[i];
        }
#pragma unroll
        for (int j = 0; j < K; ++j) e[j] = en[j];
        begin = nbegin; m = nm;
        nbegin = nnbegin; nm = nnm;
    }
}

// ------------------------------------------------------------------------------------------
// C, wave-sized segments: ONE WAVE per segment, no workgroup barrier anywhere.  A segment of at most 64*K
// elements lives in the wave's registers (lane l, item j <-> element j*64 + l: every load and store instruction
// moves 64 consecutive elements) and in the wave's private slice of LDS.  Per local pass: returning DS atomics on
// the wave's own bins give the stable rank (issue order = item order, colliding lanes in lane order), the lanes
// scan the bins (BINS / 64 per lane + one DPP scan), the elements go to LDS at bin start + rank and come back in
// order.  DS operations of one wave execute in issue order, so the phases need no barrier -- which is what made
// the workgroup-per-segment form slow (14 barriers per 4 Ki keys, ~20K cycles per segment).
// ------------------------------------------------------------------------------------------
template <typename E, int K, int LBITS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void segment_sort_wave_kernel(E* data, const uint32_t* __restrict__ seg_start,
                                                                        uint32_t num_segments, uint32_t low_bits,
                                                                        const uint32_t* __restrict__ gate, uint32_t gate_value,
                                                                        uint32_t* fault)
{
    if (gate && *gate != gate_value) return;
    constexpr int CAP = 64 * K;
    constexpr int BINS = 1 << LBITS;
    constexpr int BPL = BINS / 64;                       // bins per lane in the scan
    static_assert(BPL >= 1 && BPL <= 8, "bins per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)threadIdx.x & 63;
    const int w = (int)threadIdx.x >> 6;
    unsigned char* mine = smem + (size_t)w * (sizeof(E) * CAP + sizeof(uint32_t) * BINS);
    E* __restrict__ s_elems = reinterpret_cast<E*>(mine);
    uint32_t* __restrict__ s_cnt = reinterpret_cast<uint32_t*>(mine + sizeof(E) * CAP);

    const uint32_t seg = blockIdx.x * (uint32_t)WAVES + (uint32_t)w;
    if (seg >= num_segments) return;
    const uint32_t begin = seg_start[seg];
    const uint32_t m = seg_start[seg + 1] - begin;
    if (m == 0u) return;
    if (m > (uint32_t)CAP) {   // cannot happen when the mode word says "hybrid"; never sort wrongly in silence
        if (lane == 0) atomicOr(fault + 1, 0x40000u);   // sticky word
        return;
    }
    E* seg_ptr = data + begin;
    const int keff = (int)((m + 63u) >> 6);
    const int rem = (int)m - lane;                       // item j of this lane exists iff j*64 < rem
    E e[K];
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < keff) e[j] = (j * 64 < rem) ? seg_ptr[j * 64 + lane] : E(0);

    const int npass = ((int)low_bits + LBITS - 1) / LBITS;
    int sb = 0;
    for (int p = 0; p < npass; ++p) {
        const int nb = ((int)low_bits - sb + (npass - p - 1)) / (npass - p);
        const uint32_t mask = (1u << nb) - 1u;
        auto digit = [&](E x) -> uint32_t { return ((uint32_t)x >> sb) & mask; };
#pragma unroll
        for (int q = 0; q < BPL; ++q) s_cnt[q * 64 + lane] = 0u;
        uint32_t rnk[K];
        bool uniform;
        {
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)digit(e[0]));
            bool same = true;
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < keff) same &= (j * 64 >= rem) | (digit(e[j]) == d0);
            uniform = __all(same);
            if (!uniform) {
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (j < keff && j * 64 < rem)
                        rnk[j] = __hip_atomic_fetch_add(&s_cnt[digit(e[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        if (!uniform) {   // (a segment whose digits all agree keeps its order: nothing to move in this pass)
            // exclusive scan of the bins: lane l owns bins [l*BPL, (l+1)*BPL)
            uint32_t c[BPL];
            uint32_t tot = 0u;
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
                c[q] = s_cnt[lane * BPL + q];
                tot += c[q];
            }
            uint32_t run = wave_incl_scan_u32(tot) - tot;
#pragma unroll
            for (int q = 0; q < BPL; ++q) {
                s_cnt[lane * BPL + q] = run;
                run += c[q];
            }
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < keff && j * 64 < rem) s_elems[s_cnt[digit(e[j])] + rnk[j]] = e[j];
            if (p + 1 < npass) {
#pragma unroll
                for (int j = 0; j < K; ++j)
                    if (j < keff && j * 64 < rem) e[j] = s_elems[j * 64 + lane];
            }
        } else if (p + 1 == npass) {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (j < keff && j * 64 < rem) s_elems[j * 64 + lane] = e[j];
        }
        sb += nb;
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (j < keff && j * 64 < rem) seg_ptr[j * 64 + lane] = s_elems[j * 64 + lane];
}

}  // namespace adlhip
