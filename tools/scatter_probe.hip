// Diagnostic (not product): what does the WRITE GRANULARITY of a radix pass cost on this MI355X?
// A pass with D-bit digits and T-key tiles writes T / 2^D keys per (tile, digit) run: 16 Ki-key tiles give
// 256-byte runs at 8 bits, 128 B at 9, 64 B at 10, 32 B at 11 bits.  This kernel moves the same bytes as a pass
// (reads n keys in tile order, writes n keys) with the look-back pass's address pattern -- 16 chains, tile t of
// chain c appends a run to every digit's (chain, digit) region -- but no ranking, so only the run length varies.
//   hipcc --offload-arch=gfx950 -O3 -o tools/scatter_probe tools/scatter_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NT = 512, K = 32, TILE = NT * K, CHAINS = 16;

// run = keys per (tile, digit) run; digits = TILE / run.  Tile `t` (chain c = t % 16, index i = t / 16, I tiles per
// chain): position p of the tile (digit d = p / run, r = p % run) goes to ((d * CHAINS + c) * I + i) * run + r.
__global__ __launch_bounds__(NT) void scatter_probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t run_log2,
                                                    uint32_t tiles_per_chain, uint32_t misalign)
{
    extern __shared__ uint32_t lds[];   // only to pin the occupancy of the real pass kernel (2 workgroups per CU)
    const uint32_t t = blockIdx.x, c = t % CHAINS, i = t / CHAINS;
    const uint32_t tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const uint32_t* p = src + (size_t)t * TILE + w * 64 * K + lane;
    uint32_t e[K];
#pragma unroll
    for (int j = 0; j < K; ++j) e[j] = p[j * 64];
    if (e[0] == 0x12345678u && e[K - 1] == 0x9abcdef0u) lds[tid] = e[1];   // keep the allocation alive, never true in practice
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < K; ++j) {
        const uint32_t pos = (uint32_t)j * NT + tid;
        const uint32_t d = pos >> run_log2, r = pos & ((1u << run_log2) - 1u);
        // misalign: every (digit, chain) region starts at its own odd 4-byte offset (0..31 keys), as real digit
        // boundaries do, so runs straddle 128-byte lines and complete each other's lines tile after tile
        const uint32_t skew = misalign ? ((d * 2654435761u + c * 40503u) >> 7) & 31u : 0u;
        const size_t g = ((((size_t)d * CHAINS + c) * tiles_per_chain + i) << run_log2) + r + skew;
        dst[g] = e[j];
    }
}
int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 0) : (size_t)1 << 26;
    const uint32_t tiles = (uint32_t)(n / TILE), tpc = tiles / CHAINS;
    uint32_t *a, *b;
    HK(hipMalloc(&a, n * 4 + 256)); HK(hipMalloc(&b, n * 4 + 256));
    HK(hipMemset(a, 1, n * 4)); HK(hipMemset(b, 2, n * 4));
    const size_t lds = (argc > 2 ? strtoull(argv[2], 0, 0) : 73) * 1024;   // KiB of LDS per workgroup = workgroups per CU
    HK(hipFuncSetAttribute((const void*)scatter_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    printf("LDS per workgroup %zu KiB -> %d workgroup(s) of %d threads per CU\n", lds >> 10, (int)(160 * 1024 / lds), NT);
    printf("n = %zu keys, %u tiles of %d, ping-pong between two %zu-MiB buffers (like consecutive passes)\n", n, tiles, TILE, n * 4 >> 20);
    for (uint32_t mis : {0u, 1u})
    for (uint32_t rl : {14u, 8u, 6u, 5u, 4u}) {
        for (int k = 0; k < 3; ++k) { scatter_probe<<<tiles, NT, lds>>>(a, b, rl, tpc, mis); scatter_probe<<<tiles, NT, lds>>>(b, a, rl, tpc, mis); }
        HK(hipEventRecord(e0));
        const int reps = 10;
        for (int k = 0; k < reps; ++k) { scatter_probe<<<tiles, NT, lds>>>(a, b, rl, tpc, mis); scatter_probe<<<tiles, NT, lds>>>(b, a, rl, tpc, mis); }
        HK(hipEventRecord(e1)); HK(hipEventSynchronize(e1));
        float ms; HK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (2 * reps);
        printf("%s run %5u keys = %6u B  (%5u digits/tile: %2d-bit digit at this tile): %7.1f us  %7.1f GB/s read+write\n", mis ? "misaligned" : "aligned   ", 1u << rl, 4u << rl,
               TILE >> rl, 14 - (int)rl, us, 2.0 * n * 4 / us / 1e3);
    }
    return 0;
}
