// Diagnostic (not product): does the 256 MiB Infinity Cache absorb an intermediate buffer that is written by one kernel and read
// by the next?  Moves 256 MiB from src to dst through an intermediate `mid`, chunk by chunk (kernel W: src chunk -> mid region,
// kernel R: mid region -> dst chunk), with mid either ONE region reused by every chunk or a different region per chunk.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mall_probe tools/mall_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ a, u32x4* __restrict__ b, size_t n16)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
int main()
{
    const size_t total = (size_t)256 << 20;
    uint32_t *src, *dst, *mid;
    HK(hipMalloc(&src, total)); HK(hipMalloc(&dst, total)); HK(hipMalloc(&mid, total + (64 << 20)));
    HK(hipMemset(src, 1, total));
    hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
    auto copy = [&](const void* s, void* d, size_t bytes) {
        hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (const u32x4*)s, (u32x4*)d, bytes / 16);
    };
    for (size_t chunk_mb : {4, 8, 16, 32, 64, 128, 256}) {
        const size_t chunk = chunk_mb << 20, steps = total / chunk;
        for (int rotate = 0; rotate < 2; ++rotate) {
            float best = 1e9;
            for (int r = 0; r < 6; ++r) {
                HK(hipEventRecord(a));
                for (size_t i = 0; i < steps; ++i) {
                    char* m = (char*)mid + (rotate ? i * chunk : 0);
                    copy((char*)src + i * chunk, m, chunk);
                    copy(m, (char*)dst + i * chunk, chunk);
                }
                HK(hipEventRecord(b)); HK(hipEventSynchronize(b));
                float ms; HK(hipEventElapsedTime(&ms, a, b));
                best = ms < best ? ms : best;
            }
            printf("chunk %4zu MiB x %3zu steps, intermediate %-22s : %8.1f us for 256 MiB src -> mid -> dst  (%zu launches)\n", chunk_mb, steps,
                   rotate ? "a new region per chunk" : "ONE region, reused", best * 1e3, 2 * steps);
        }
    }
    return 0;
}
