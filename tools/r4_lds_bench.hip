// tools/r4_lds_bench.hip -- round 4 diagnostic: LDS-array cycles per wave-instruction for the operations of the LDS finish, on
// random addresses (the finish's pattern: 64 random bins of 256 / random slots of 1536), all CUs busy, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; }

// MODE: 0 ds_add (no return) on 256 random counters; 1 ds_add_rtn; 2 ds_read_b32 gather of 256 random words; 3 ds_read_u16 gather of 256
// random halfwords; 4 ds_write_b32 to random 1536 slots; 5 ds_write_b16 to random 1536 slots; 6 ds_read_b32 linear; 7 ds_read_b64 linear;
// 8 ds_add_rtn on linear (conflict-free) counters; 9 ds_add on linear; 10 ds_write_b32 linear
template <int MODE>
__global__ __launch_bounds__(256) void lds_bench(uint32_t* out, int iters)
{
    __shared__ uint32_t lds[4][2048];
    uint32_t* mine = lds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    for (int i = lane; i < 2048; i += 64) mine[i] = 0;
    uint32_t s = 0x9E3779B9u * (blockIdx.x * 256 + threadIdx.x + 1);
    uint32_t idx[16];
    for (int j = 0; j < 16; ++j) {
        const uint32_t r = rnd(s);
        idx[j] = (MODE == 4 || MODE == 5) ? r % 1536u : (MODE >= 6 ? (uint32_t)(lane + 64 * j) : (r & 255u));
    }
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0 || MODE == 9) __hip_atomic_fetch_add(&mine[idx[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (MODE == 1 || MODE == 8) acc += __hip_atomic_fetch_add(&mine[idx[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (MODE == 2 || MODE == 6) acc += ((volatile uint32_t*)mine)[idx[j]];
            else if (MODE == 3) acc += ((volatile uint16_t*)mine)[idx[j]];
            else if (MODE == 4 || MODE == 10) ((volatile uint32_t*)mine)[idx[j]] = acc + j;
            else if (MODE == 5) ((volatile uint16_t*)mine)[idx[j]] = (uint16_t)(acc + j);
            else if (MODE == 7) { const unsigned long long v = ((volatile unsigned long long*)mine)[idx[j]]; acc += (uint32_t)v + (uint32_t)(v >> 32); }
        }
        // rotate the addresses a little so that nothing is loop-invariant
#pragma unroll
        for (int j = 0; j < 16; ++j) if (MODE < 6) idx[j] = (MODE == 4 || MODE == 5) ? (idx[j] * 5u + 1u) % 1536u : ((idx[j] * 5u + 1u) & 255u);
    }
    if (acc == 0x12345678u) out[0] = acc + mine[lane];
}

template <int MODE>
static int run(const char* name, uint32_t* d_out)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200, wgs = 256 * 8;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(lds_bench<MODE>, dim3(wgs), dim3(256), 0, 0, d_out, 10);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(lds_bench<MODE>, dim3(wgs), dim3(256), 0, 0, d_out, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per CU = 8 wgs * 4 waves * iters * 16; at ~2.1 GHz
    const double instr_per_cu = 8.0 * 4 * iters * 16;
    printf("%-44s %8.3f ms  %6.2f ns per wave-instruction per CU (= %5.1f cycles at 2.1 GHz)\n", name, ms, ms * 1e6 / instr_per_cu, ms * 1e6 / instr_per_cu * 2.1);
    return 0;
}
int main()
{
    uint32_t* d; CK(hipMalloc(&d, 64));
    run<0>("ds_add        256 random counters", d);
    run<1>("ds_add_rtn    256 random counters", d);
    run<2>("ds_read_b32   256 random words", d);
    run<3>("ds_read_u16   256 random halfwords", d);
    run<4>("ds_write_b32  1536 random slots", d);
    run<5>("ds_write_b16  1536 random slots", d);
    run<6>("ds_read_b32   linear", d);
    run<7>("ds_read_b64   linear", d);
    run<8>("ds_add_rtn    linear (no conflicts)", d);
    run<9>("ds_add        linear (no conflicts)", d);
    run<10>("ds_write_b32  linear", d);
    return 0;
}
