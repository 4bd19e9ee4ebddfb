// Diagnostic (not product): what does a plain device-to-device copy reach on this MI355X, over grid size,
// unroll, block size and store flavour?  Sets the ceiling a read-n/write-n pass can be compared with.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ void copyk(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < nvec; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) { v4u t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(&dst[i + u * stride])); }
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < nvec; i += stride) dst[i] = src[i];
}
// contiguous chunk per block (each block streams its own range), like the sort's tiles
template <int UNROLL>
__global__ void copy_chunked(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t nvec, size_t per_block)
{
    size_t b = (size_t)blockIdx.x * per_block;
    size_t e = b + per_block < nvec ? b + per_block : nvec;
    for (size_t i = b + threadIdx.x; i < e; i += (size_t)blockDim.x * UNROLL) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u * blockDim.x < e) v[u] = src[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) if (i + u * blockDim.x < e) dst[i + u * blockDim.x] = v[u];
    }
}
template <typename F> float timeit(F f, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}
int main()
{
    for (size_t mib : {256, 1024}) {
        size_t bytes = mib << 20, nvec = bytes / 16;
        uint4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes);
        hipMemset(s, 1, bytes); hipMemset(d, 2, bytes);
        printf("== %zu MiB copy (read+write bytes / time)\n", mib);
        float ms = timeit([&] { hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); }, 10);
        printf("hipMemcpyAsync d2d                        %7.1f GB/s\n", 2.0 * bytes / ms / 1e6);
        for (int bs : {256, 512, 1024})
            for (int mult : {2, 4, 8, 16, 32, 64}) {
                int grid = 256 * mult * 256 / bs;
                float m4 = timeit([&] { copyk<4, false><<<grid, bs>>>(d, s, nvec); }, 10);
                float m8 = timeit([&] { copyk<8, false><<<grid, bs>>>(d, s, nvec); }, 10);
                float n4 = timeit([&] { copyk<4, true><<<grid, bs>>>(d, s, nvec); }, 10);
                float m1 = timeit([&] { copyk<1, false><<<grid, bs>>>(d, s, nvec); }, 10);
                printf("block %4d grid %6d: u1 %7.1f  u4 %7.1f  u8 %7.1f  u4-nt %7.1f GB/s\n", bs, grid, 2.0 * bytes / m1 / 1e6,
                       2.0 * bytes / m4 / 1e6, 2.0 * bytes / m8 / 1e6, 2.0 * bytes / n4 / 1e6);
            }
        for (size_t chunk_kib : {64, 128, 256}) {
            size_t per_block = chunk_kib * 1024 / 16;
            int grid = (int)((nvec + per_block - 1) / per_block);
            float c4 = timeit([&] { copy_chunked<4><<<grid, 512>>>(d, s, nvec, per_block); }, 10);
            printf("chunked %3zu KiB per block (512 thr, grid %d): %7.1f GB/s\n", chunk_kib, grid, 2.0 * bytes / c4 / 1e6);
        }
        hipFree(s); hipFree(d);
    }
    return 0;
}
