// finish16_kernels.hpp -- gfx950 device code: the LDS finish of the large keys-only sort for whole u32 keys (round 4).
//
// After the two MSD passes a segment slab holds the low 16 bits of ~n / 65536 keys (uint16_t; the bits above are the segment's
// number).  ONE WAVE sorts one segment, as wave_segment_sort_kernel (hybrid_kernels.hpp) does -- stable 8-bit LSD passes in the
// wave's own slice of LDS, no workgroup barrier -- but the keys stay 16 bits wide all the way:
//   * they are loaded two to a dword (half the load instructions) and live two to a register;
//   * the LDS tile holds uint16_t (3 KiB instead of 6 for 1536 keys: the wave slots, not the LDS, bound the occupancy) in an
//     interleaved order -- position p sits at index (p / 128) * 128 + (p % 64) * 2 + (p / 64) % 2 -- so that ONE ds_read_b32 per lane
//     brings back rows 2j and 2j + 1 (positions 128 j + lane and 128 j + 64 + lane): half the read-backs, and the final stores are
//     still one dword per lane to consecutive addresses.
// The counters of round 3 (profiles/r3_pmc_msd2_u32_all_counters.txt) put the old finish at 72 % LDS-busy with 61 % of those cycles in
// bank conflicts; conflicts of 64 random bins over 32 banks cannot be laid out away, so what this kernel buys is fewer LDS
// instructions per key and more waves in flight to keep the LDS pipe full while others wait for memory.
//
// Reference behaviour: Tahoe/ClKernels/RadixSort32Kernels.cl:401-489 (sort4Bits1: the stable local sort of a block).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hybrid_kernels.hpp"

namespace adlhip {

template <int R2>
struct Finish16Cfg {
    static constexpr int CAP = 128 * R2;                     // keys a wave's tile holds
    static constexpr size_t PER_WAVE = 2 * CAP + 256 * 4;    // tile + 256 counters
};

__device__ __forceinline__ uint32_t finish16_index(uint32_t p) { return (p & ~127u) | ((p & 63u) << 1) | ((p >> 6) & 1u); }

// RR row pairs; the first RR - 1 are full (m > 128 (RR - 1)), only the last is tested per lane
// NTL / NTS: non-temporal loads of the slab / stores of the result.  ALG: 0 = count, scan, then a returning atomic per key hands out
// its slot; 1 = ONE returning atomic per key counts and ranks (its arrival number inside its bin, in position order), then scan and
// slot = bin start (a gather: cheaper than an atomic, tools/r4_lds_bench) + rank.
template <int RR, bool NTL, bool NTS, int ALG>
__device__ __forceinline__ void finish16_rows(const uint32_t* __restrict__ src32, uint32_t* __restrict__ out, uint32_t m, int lane,
                                              uint16_t* __restrict__ buf16, uint32_t* __restrict__ cnt, uint32_t low_bits, uint32_t hi)
{
    constexpr int L = RR - 1;
    // as loaded: dword i = 64 j + lane holds keys 2 i and 2 i + 1; after a pass: positions 128 j + lane and 128 j + 64 + lane
    const bool a_lo = 2u * (uint32_t)(64 * L + lane) < m, a_hi = 2u * (uint32_t)(64 * L + lane) + 1u < m;
    const bool b_lo = (uint32_t)(128 * L + lane) < m, b_hi = (uint32_t)(128 * L + 64 + lane) < m;
    uint32_t kp[RR];
#pragma unroll
    for (int j = 0; j < RR; ++j)
        if (j < L || a_lo) kp[j] = NTL ? __builtin_nontemporal_load(src32 + j * 64 + lane) : src32[j * 64 + lane];
    bool placed = false;   // wave-uniform: the keys sit in position order (a pass has run)
    const int npass = ((int)low_bits + 7) / 8;
    int sb = 0;
    const uint32_t* buf32 = reinterpret_cast<const uint32_t*>(buf16);
    for (int p = 0; p < npass; ++p) {
        const int nb = ((int)low_bits - sb + (npass - p - 1)) / (npass - p);
        const uint32_t mask = (1u << nb) - 1u;
        const bool v_lo = placed ? b_lo : a_lo, v_hi = placed ? b_hi : a_hi;
        auto dlo = [&](uint32_t x) -> uint32_t { return (x >> sb) & mask; };
        auto dhi = [&](uint32_t x) -> uint32_t { return (x >> (16 + sb)) & mask; };
        {   // a pass in which every key has the same digit changes nothing -- and would queue all 64 lanes of every atomic on ONE counter
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)dlo(kp[0]));
            bool same = true;
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                if (j < L || v_lo) same &= dlo(kp[j]) == d0;
                if (j < L || v_hi) same &= dhi(kp[j]) == d0;
            }
            if (__all(same)) {
                sb += nb;
                continue;
            }
        }
        const u32x4 z = {0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(cnt + 4 * lane) = z;
        uint32_t rk[RR];   // ALG 1: the two ranks of a register's keys
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            if constexpr (ALG == 0) {
                if (j < L || v_lo) __hip_atomic_fetch_add(&cnt[dlo(kp[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (j < L || v_hi) __hip_atomic_fetch_add(&cnt[dhi(kp[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            } else {   // position order: row 2j before row 2j + 1, colliding lanes in lane order (stable)
                uint32_t r0 = 0u, r1 = 0u;
                if (j < L || v_lo) r0 = __hip_atomic_fetch_add(&cnt[dlo(kp[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (j < L || v_hi) r1 = __hip_atomic_fetch_add(&cnt[dhi(kp[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                rk[j] = r0 | (r1 << 16);
            }
        }
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
        if constexpr (ALG == 0) {
            // slots: one returning atomic per key in position order (row 2j before row 2j + 1; colliding lanes are served in lane
            // order: stable, radix_kernels.hpp rank_in_wave), then its 16-bit store
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                uint32_t p0 = 0u, p1 = 0u;
                if (j < L || v_lo) p0 = __hip_atomic_fetch_add(&cnt[dlo(kp[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (j < L || v_hi) p1 = __hip_atomic_fetch_add(&cnt[dhi(kp[j])], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                if (j < L || v_lo) buf16[finish16_index(p0)] = (uint16_t)kp[j];
                if (j < L || v_hi) buf16[finish16_index(p1)] = (uint16_t)(kp[j] >> 16);
            }
        } else {
            const volatile uint32_t* start = cnt;
#pragma unroll
            for (int j = 0; j < RR; ++j) {
                uint32_t p0 = 0u, p1 = 0u;
                if (j < L || v_lo) p0 = start[dlo(kp[j])] + (rk[j] & 0xffffu);
                if (j < L || v_hi) p1 = start[dhi(kp[j])] + (rk[j] >> 16);
                if (j < L || v_lo) buf16[finish16_index(p0)] = (uint16_t)kp[j];
                if (j < L || v_hi) buf16[finish16_index(p1)] = (uint16_t)(kp[j] >> 16);
            }
        }
#pragma unroll
        for (int j = 0; j < RR; ++j)
            if (j < L || b_lo) kp[j] = buf32[j * 64 + lane];
        placed = true;
        sb += nb;
    }
    if (placed) {
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            if (j < L || b_lo) {
                if constexpr (NTS) __builtin_nontemporal_store(hi | (kp[j] & 0xffffu), out + 128 * j + lane);
                else out[128 * j + lane] = hi | (kp[j] & 0xffffu);
            }
            if (j < L || b_hi) {
                if constexpr (NTS) __builtin_nontemporal_store(hi | (kp[j] >> 16), out + 128 * j + 64 + lane);
                else out[128 * j + 64 + lane] = hi | (kp[j] >> 16);
            }
        }
    } else {   // no pass had anything to do: every key of the segment is the same, any order will do
#pragma unroll
        for (int j = 0; j < RR; ++j) {
            if (j < L || a_lo) out[2 * (64 * j + lane)] = hi | (kp[j] & 0xffffu);
            if (j < L || a_hi) out[2 * (64 * j + lane) + 1] = hi | (kp[j] >> 16);
        }
    }
}

template <int RR, int R2, bool NTL, bool NTS, int ALG>
__device__ __forceinline__ void finish16_dispatch(int rows2, const uint32_t* __restrict__ src32, uint32_t* __restrict__ out, uint32_t m,
                                                  int lane, uint16_t* __restrict__ buf16, uint32_t* __restrict__ cnt, uint32_t low_bits,
                                                  uint32_t hi)
{
    if constexpr (RR >= R2) {
        finish16_rows<R2, NTL, NTS, ALG>(src32, out, m, lane, buf16, cnt, low_bits, hi);
    } else {
        if (rows2 <= RR) finish16_rows<RR, NTL, NTS, ALG>(src32, out, m, lane, buf16, cnt, low_bits, hi);
        else finish16_dispatch<RR + 1, R2, NTL, NTS, ALG>(rows2, src32, out, m, lane, buf16, cnt, low_bits, hi);
    }
}

// slab: 65536 segment slabs of `stride` uint16_t (stride even); segment s holds seg_cnt[s] keys and goes to out[seg_off[s] ...).
// dyn[0] = bits the finish sorts (<= 16), dyn[1] = the keys' common prefix above the two digits (msd2_offsets_kernel).
template <int R2, int WAVES, bool NTL = false, bool NTS = false, int ALG = 0>
__global__ __launch_bounds__(64 * WAVES) void wave_finish16_kernel(const uint16_t* __restrict__ slab, uint32_t* __restrict__ out,
                                                                    const uint32_t* __restrict__ seg_off,
                                                                    const uint32_t* __restrict__ seg_cnt, uint32_t stride,
                                                                    const uint32_t* __restrict__ gate, const uint32_t* __restrict__ dyn,
                                                                    uint32_t num_segments, uint32_t* fault)
{
    if (gate && *gate != 0u) return;
    using C = Finish16Cfg<R2>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)threadIdx.x & 63;
    const int w = (int)threadIdx.x >> 6;
    unsigned char* mine = smem + (size_t)w * C::PER_WAVE;
    uint16_t* __restrict__ buf16 = reinterpret_cast<uint16_t*>(mine);
    uint32_t* __restrict__ cnt = reinterpret_cast<uint32_t*>(mine + 2 * C::CAP);
    const uint32_t seg = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (uint32_t)WAVES + (uint32_t)w));
    if (seg >= num_segments) return;
    const uint32_t low_bits = (uint32_t)__builtin_amdgcn_readfirstlane((int)dyn[0]);
    const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_cnt[seg]);
    const uint32_t begin = (uint32_t)__builtin_amdgcn_readfirstlane((int)seg_off[seg]);
    if (m == 0u) return;
    if (m > (uint32_t)C::CAP || low_bits > 16u) {   // never sort wrongly in silence
        if (lane == 0) atomicOr(fault + 1, 0x40000u);
        return;
    }
    const uint32_t hi = ((dyn[1] << 16) | seg) << low_bits;
    const uint32_t* src32 = reinterpret_cast<const uint32_t*>(slab + (size_t)seg * stride);
    finish16_dispatch<1, R2, NTL, NTS, ALG>((int)((m + 127u) >> 7), src32, out + begin, m, lane, buf16, cnt, low_bits, hi);
}

}   // namespace adlhip
