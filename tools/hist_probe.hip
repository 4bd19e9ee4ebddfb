// Diagnostic (not part of the product): where does the time of the up-front histogram kernel go?
// Streams a 256-MiB u32 buffer through variants of "read + P LDS atomics per key" that differ in
// launch geometry, addressing and the number of histogram passes, and prints GB/s for each.
//   hipcc --offload-arch=gfx950 -O3 -o tools/hist_probe tools/hist_probe.hip && tools/hist_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static int g_bufs = 8;

template <int P>
__device__ inline void bump(uint32_t* hist, uint32_t x, uint32_t chain0)
{
    if constexpr (P >= 1) atomicAdd(&hist[(chain0 << 8) | (x & 255u)], 1u);
    if constexpr (P >= 2) atomicAdd(&hist[4096 + ((x >> 4) & 4095u)], 1u);
    if constexpr (P >= 3) atomicAdd(&hist[8192 + ((x >> 12) & 4095u)], 1u);
    if constexpr (P >= 4) atomicAdd(&hist[12288 + ((x >> 20) & 4095u)], 1u);
}

// MODE 0: every workgroup streams its own contiguous range; MODE 1: whole-grid stride
template <int NT, int P, int MODE>
__global__ __launch_bounds__(NT) void hist_variant(const uint4* __restrict__ src, size_t nvec, size_t per_wg,
                                                   uint32_t* __restrict__ out)
{
    extern __shared__ uint32_t hist[];
    const int tid = (int)threadIdx.x;
    constexpr int BINS = P * 4096;
    for (int i = tid; i < BINS; i += NT) hist[i] = 0u;
    __syncthreads();
    uint32_t acc = 0u;
    const uint32_t chain0 = blockIdx.x & 15u;
    size_t i, end, step;
    if (MODE == 0) { i = (size_t)blockIdx.x * per_wg + tid; end = i - tid + per_wg; if (end > nvec) end = nvec; step = NT; }
    else { step = (size_t)gridDim.x * NT; i = (size_t)blockIdx.x * NT + tid; end = nvec; }
    auto use = [&](const uint4& v) {
        if constexpr (P == 0) acc ^= v.x ^ v.y ^ v.z ^ v.w;
        else { bump<P>(hist, v.x, chain0); bump<P>(hist, v.y, chain0); bump<P>(hist, v.z, chain0); bump<P>(hist, v.w, chain0); }
    };
    if (i + 3 * step < end) {
        uint4 a = src[i], b = src[i + step], c = src[i + 2 * step], d = src[i + 3 * step];
        i += 4 * step;
        for (; i + 3 * step < end; i += 4 * step) {
            const uint4 na = src[i], nb = src[i + step], nc = src[i + 2 * step], nd = src[i + 3 * step];
            use(a); use(b); use(c); use(d);
            a = na; b = nb; c = nc; d = nd;
        }
        use(a); use(b); use(c); use(d);
    }
    for (; i < end; i += step) { const uint4 a = src[i]; use(a); }
    __syncthreads();
    if constexpr (P == 0) { if (acc == 0x9e3779b9u) atomicAdd(out, 1u); }
    else { uint32_t* o = out + (size_t)blockIdx.x * BINS; for (int k = tid; k < BINS; k += NT) o[k] = hist[k]; }
}

__global__ void fill(uint32_t* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = (i + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull; z ^= z >> 31; z *= 0x94d049bb133111ebull; z ^= z >> 29;
        p[i] = (uint32_t)z;
    }
}

__global__ __launch_bounds__(256) void copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t nvec)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < nvec; i += stride) dst[i] = src[i];
}

// ping-pong copy A -> B -> A ... inside a working set of 2 x bytes: how fast is a pass whose data stays on chip?
void run_copy(uint4* a, uint4* b, size_t nvec, int wgs)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(copy_kernel, dim3(wgs), dim3(256), 0, 0, (w & 1) ? a : b, (w & 1) ? b : a, nvec);
    CK(hipEventRecord(e0));
    const int K = 40;
    for (int w = 0; w < K; ++w) hipLaunchKernelGGL(copy_kernel, dim3(wgs), dim3(256), 0, 0, (w & 1) ? a : b, (w & 1) ? b : a, nvec);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= K;
    printf("ping-pong copy %6zu MiB -> %6zu MiB   %7.1f us  %7.1f GB/s (read + write)\n", nvec * 16 >> 20, nvec * 16 >> 20, ms * 1e3,
           2.0 * nvec * 16.0 / ms * 1e-6);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

// The histogram as the sort sees it: every launch follows a kernel that has just written 256 MiB (the previous
// sort's last pass), so dirty lines are still being written back while the histogram streams its input in.
template <int NT, int P>
void run_after_writer(const uint4* src0, size_t nvec, int wgs, uint32_t* out, uint4* wr_dst, const uint4* wr_src)
{
    const size_t lds = (size_t)P * 4096 * 4;
    auto kern = hist_variant<NT, P, 0>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    size_t per_wg = ((nvec + wgs - 1) / wgs + NT - 1) / NT * NT;
    const int K = 20;
    std::vector<hipEvent_t> ev(2 * K);
    for (auto& e : ev) CK(hipEventCreate(&e));
    for (int w = 0; w < K; ++w) {
        hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, wr_dst, wr_src, nvec);
        CK(hipEventRecord(ev[2 * w]));
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), lds, 0, src0 + (size_t)(w % (g_bufs - 2)) * nvec, nvec, per_wg, out);
        CK(hipEventRecord(ev[2 * w + 1]));
    }
    CK(hipDeviceSynchronize());
    double tot = 0;
    for (int w = 2; w < K; ++w) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * w], ev[2 * w + 1])); tot += ms; }
    const double ms = tot / (K - 2);
    printf("%d pass(es) right after a 256-MiB copy kernel, event-bracketed          %7.1f us  %7.1f GB/s\n", P, ms * 1e3, nvec * 16.0 / ms * 1e-6);
    for (auto& e : ev) CK(hipEventDestroy(e));
}

template <int NT, int P, int MODE>
void run(const char* label, const uint4* src0, size_t nvec, int wgs, uint32_t* out)
{
    // rotate over kBufs input buffers so that no launch finds its input in the 256-MB MALL
    auto srcof = [&](int w) { return src0 + (size_t)(w % g_bufs) * nvec; };
    const size_t lds = (size_t)P * 4096 * 4;
    auto kern = hist_variant<NT, P, MODE>;
    if (lds > 48 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    size_t per_wg = ((nvec + wgs - 1) / wgs + NT - 1) / NT * NT;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), lds, 0, srcof(w), nvec, per_wg, out);
    CK(hipEventRecord(e0));
    const int K = 20;
    for (int w = 0; w < K; ++w) hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), lds, 0, srcof(w + 3), nvec, per_wg, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= K;
    printf("%-44s NT=%4d wgs=%5d P=%d mode=%d  %7.1f us  %7.1f GB/s\n", label, NT, wgs, P, MODE, ms * 1e3, nvec * 16.0 / ms * 1e-6);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 0) : (size_t)1 << 26;
    uint32_t* src; uint32_t* out;
    if (argc > 2) g_bufs = atoi(argv[2]);
    CK(hipMalloc(&src, n * 4 * g_bufs));
    CK(hipMalloc(&out, (size_t)4096 * 16384 * 4));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, src, n * g_bufs);
    CK(hipDeviceSynchronize());
    const uint4* s = (const uint4*)src; const size_t nv = n / 4;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cu = prop.multiProcessorCount;
    printf("n = %zu u32 keys (%zu MiB) x %d rotating buffers, %d CUs\n", n, n * 4 >> 20, g_bufs, cu);
    if (argc > 3) {   // residency curve: ping-pong copies and repeated reads of ONE buffer of growing size
        for (size_t mib = 4; mib * 2 <= (n * 4 * (size_t)g_bufs) >> 20 && mib <= 1024; mib *= 2) {
            const size_t v = (mib << 20) / 16;
            run_copy((uint4*)src, (uint4*)src + v, v, cu * 8);
        }
        const int keep = g_bufs; g_bufs = 1;
        for (size_t mib = 4; mib <= (n * 4 * (size_t)keep) >> 20 && mib <= 1024; mib *= 2) {
            char lab[64]; snprintf(lab, sizeof lab, "repeated read of %zu MiB", mib);
            run<1024, 0, 1>(lab, s, (mib << 20) / 16, cu, out);
        }
        CK(hipFree(src)); CK(hipFree(out));
        return 0;
    }
    // read only
    run<256, 0, 1>("read, grid-stride, 8 WG/CU x 256", s, nv, cu * 8, out);
    run<1024, 0, 1>("read, grid-stride, 1 WG/CU x 1024", s, nv, cu, out);
    run<1024, 0, 0>("read, contiguous per WG, 1 WG/CU x 1024", s, nv, cu, out);
    run<1024, 0, 0>("read, contiguous per WG, 2 WG/CU x 1024", s, nv, cu * 2, out);
    run<256, 0, 0>("read, contiguous per WG, 8 WG/CU x 256", s, nv, cu * 8, out);
    // one histogram pass
    run<1024, 1, 0>("1 pass, contiguous, 1 WG/CU x 1024", s, nv, cu, out);
    run<1024, 1, 0>("1 pass, contiguous, 2 WG/CU x 1024", s, nv, cu * 2, out);
    run<1024, 1, 1>("1 pass, grid-stride, 1 WG/CU x 1024", s, nv, cu, out);
    run<256, 1, 0>("1 pass, contiguous, 8 WG/CU x 256", s, nv, cu * 8, out);
    run<256, 1, 1>("1 pass, grid-stride, 8 WG/CU x 256", s, nv, cu * 8, out);
    run<512, 1, 0>("1 pass, contiguous, 4 WG/CU x 512", s, nv, cu * 4, out);
    // two and four passes
    run<1024, 2, 0>("2 passes, contiguous, 1 WG/CU x 1024", s, nv, cu, out);
    run<1024, 4, 0>("4 passes, contiguous, 1 WG/CU x 1024", s, nv, cu, out);
    run<1024, 4, 0>("4 passes, contiguous, 2 WG/CU x 1024", s, nv, cu * 2, out);
    run<1024, 4, 1>("4 passes, grid-stride, 1 WG/CU x 1024", s, nv, cu, out);
    run<1024, 4, 1>("4 passes, grid-stride, 2 WG/CU x 1024", s, nv, cu * 2, out);
    run<512, 4, 0>("4 passes, contiguous, 2 WG/CU x 512", s, nv, cu * 2, out);
    run<512, 4, 0>("4 passes, contiguous, 4 WG/CU x 512 (2 fit)", s, nv, cu * 4, out);
    run<256, 4, 1>("4 passes, grid-stride, 8 WG/CU x 256 (2 fit)", s, nv, cu * 8, out);
    if (g_bufs >= 4) {   // last two buffers serve as the writer's source and destination
        uint4* wd = (uint4*)src + (size_t)(g_bufs - 1) * nv;
        const uint4* ws = s + (size_t)(g_bufs - 2) * nv;
        run_after_writer<1024, 0>(s, nv, cu, out, wd, ws);
        run_after_writer<1024, 1>(s, nv, cu, out, wd, ws);
        run_after_writer<1024, 4>(s, nv, cu, out, wd, ws);
    }
    CK(hipFree(src)); CK(hipFree(out));
    return 0;
}
