// Diagnostic (not product): what bounds the LDS finish of the large keys-only sort?  It reads 65536 segments of about 1024 keys
// out of 1280-key slabs and writes them back to back; with the sort taken out it still ran at 3.8 TB/s while a plain copy of the
// same bytes runs at 5.5 TB/s.  This probe moves the same bytes with different shapes:
//   mode 0  a wave per segment, dword loads and stores (lane l, item j <-> element j*64 + l): the kernel's shape
//   mode 1  dwordx4 loads (aligned: slabs start on 5120-byte boundaries), dword stores
//   mode 2  dwordx4 loads and dwordx4 stores over the 16-byte aligned body of the destination, dword head/tail
//   mode 3  mode 0, persistent: 256 x 4 workgroups, every wave loops over segments
//   mode 4  mode 0 with all 16 loads of the NEXT segment issued before the stores of the current one (persistent)
// each with the slabs clean (read twice in a row) and "dirty" (a kernel rewrites the slabs right before, as pass 2 does).
//   hipcc --offload-arch=gfx950 -O3 -o tools/finish_probe tools/finish_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int K = 20, STRIDE = 1280, SEGS = 65536;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int MODE>
__global__ __launch_bounds__(64 * WAVES) void finish_probe(const uint32_t* __restrict__ slab, uint32_t* __restrict__ out,
                                                           const uint32_t* __restrict__ off, const uint32_t* __restrict__ cnt)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t nw = gridDim.x * WAVES;
    uint32_t seg = blockIdx.x * WAVES + w;
    if (MODE == 0 || MODE == 3) {
        for (; seg < SEGS; seg += nw) {
            const uint32_t m = cnt[seg], begin = off[seg];
            const uint32_t* src = slab + (size_t)seg * STRIDE;
            const int keff = (m + 63) >> 6, rem = (int)m - lane;
            uint32_t e[K];
#pragma unroll
            for (int j = 0; j < K; ++j) if (j < keff) e[j] = (j * 64 < rem) ? src[j * 64 + lane] : 0u;
#pragma unroll
            for (int j = 0; j < K; ++j) if (j < keff && j * 64 < rem) out[begin + j * 64 + lane] = e[j];
            if (MODE == 0) break;
        }
    } else if (MODE == 1 || MODE == 2) {
        const uint32_t m = cnt[seg], begin = off[seg];
        const u32x4* src = reinterpret_cast<const u32x4*>(slab + (size_t)seg * STRIDE);
        constexpr int K4 = K / 4;   // 5 x (64 lanes x 16 bytes)
        u32x4 e[K4];
        const int groups = (m + 3) >> 2;
#pragma unroll
        for (int j = 0; j < K4; ++j) if (j * 64 + lane < groups) e[j] = src[j * 64 + lane];
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < K4; ++j) {
                const uint32_t i = (uint32_t)(j * 64 + lane) * 4u;
                if (i + 0 < m) out[begin + i + 0] = e[j].x;
                if (i + 1 < m) out[begin + i + 1] = e[j].y;
                if (i + 2 < m) out[begin + i + 2] = e[j].z;
                if (i + 3 < m) out[begin + i + 3] = e[j].w;
            }
        } else {   // through LDS, shifted so that LDS word (i + shift) <-> out[begin + i] and 16-byte groups line up
            __shared__ __attribute__((aligned(16))) uint32_t lds[WAVES][STRIDE + 8];
            uint32_t* buf = lds[w];
            const uint32_t shift = begin & 3u;
#pragma unroll
            for (int j = 0; j < K4; ++j) {
                const uint32_t i = (uint32_t)(j * 64 + lane) * 4u;
                if (i + 0 < m) buf[shift + i + 0] = e[j].x;
                if (i + 1 < m) buf[shift + i + 1] = e[j].y;
                if (i + 2 < m) buf[shift + i + 2] = e[j].z;
                if (i + 3 < m) buf[shift + i + 3] = e[j].w;
            }
            uint32_t* base = out + (begin - shift);   // 16-byte aligned
            const uint32_t end = shift + m;
#pragma unroll
            for (int j = 0; j < K4 + 1; ++j) {
                const uint32_t g = (uint32_t)(j * 64 + lane) * 4u;
                if (g >= end) continue;
                if (g >= shift && g + 4 <= end) {
                    *reinterpret_cast<u32x4*>(base + g) = *reinterpret_cast<const u32x4*>(buf + g);
                } else {
                    for (uint32_t q = 0; q < 4; ++q)
                        if (g + q >= shift && g + q < end) base[g + q] = buf[g + q];
                }
            }
        }
    } else if (MODE == 4) {
        uint32_t m = 0, begin = 0, e[K];
        auto fetch = [&](uint32_t s, uint32_t& mm, uint32_t& bb, uint32_t (&x)[K]) {
            mm = 0;
            if (s >= SEGS) return;
            mm = cnt[s];
            bb = off[s];
            const uint32_t* src = slab + (size_t)s * STRIDE;
            const int keff = (mm + 63) >> 6, rem = (int)mm - lane;
#pragma unroll
            for (int j = 0; j < K; ++j) if (j < keff) x[j] = (j * 64 < rem) ? src[j * 64 + lane] : 0u;
        };
        fetch(seg, m, begin, e);
        while (m) {
            uint32_t c[K], cm = m, cb = begin;
#pragma unroll
            for (int j = 0; j < K; ++j) c[j] = e[j];
            seg += nw;
            fetch(seg, m, begin, e);
            const int keff = (cm + 63) >> 6, rem = (int)cm - lane;
#pragma unroll
            for (int j = 0; j < K; ++j) if (j < keff && j * 64 < rem) out[cb + j * 64 + lane] = c[j];
        }
    }
}

__global__ void dirty_kernel(uint32_t* slab, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) slab[i] = (uint32_t)i * 2654435761u;
}

int main()
{
    std::vector<uint32_t> cnt(SEGS), off(SEGS);
    std::mt19937 rng(1);
    std::normal_distribution<double> nd(1024.0, 32.0);
    size_t total = 0;
    for (int s = 0; s < SEGS; ++s) {
        int c = (int)nd(rng);
        c = c < 1 ? 1 : (c > STRIDE ? STRIDE : c);
        cnt[s] = c; off[s] = (uint32_t)total; total += c;
    }
    uint32_t *slab, *out, *dcnt, *doff;
    HK(hipMalloc(&slab, (size_t)SEGS * STRIDE * 4)); HK(hipMalloc(&out, total * 4 + 64)); HK(hipMalloc(&dcnt, SEGS * 4)); HK(hipMalloc(&doff, SEGS * 4));
    HK(hipMemcpy(dcnt, cnt.data(), SEGS * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(doff, off.data(), SEGS * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
    auto run = [&](const char* name, auto launch) {
        for (int dirty = 0; dirty < 2; ++dirty) {
            float best = 1e9, sum = 0;
            for (int r = 0; r < 12; ++r) {
                if (dirty) hipLaunchKernelGGL(dirty_kernel, dim3(4096), dim3(256), 0, 0, slab, (size_t)SEGS * STRIDE);
                HK(hipEventRecord(a)); launch(); HK(hipEventRecord(b)); HK(hipEventSynchronize(b));
                float ms; HK(hipEventElapsedTime(&ms, a, b));
                if (r >= 2) { sum += ms; best = ms < best ? ms : best; }
            }
            printf("  %-58s %s  avg %7.1f us  best %7.1f us  %7.0f GB/s read+write\n", name, dirty ? "behind a writer" : "clean          ", sum / 10 * 1e3, best * 1e3, 2.0 * total * 4 / (sum / 10) / 1e6);
        }
    };
    printf("%d segments, %zu keys\n", SEGS, total);
    run("mode 0 wave per segment, dword (8 waves/WG)", [&] { hipLaunchKernelGGL((finish_probe<8, 0>), dim3(SEGS / 8), dim3(512), 0, 0, slab, out, doff, dcnt); });
    run("mode 0 wave per segment, dword (4 waves/WG)", [&] { hipLaunchKernelGGL((finish_probe<4, 0>), dim3(SEGS / 4), dim3(256), 0, 0, slab, out, doff, dcnt); });
    run("mode 0 wave per segment, dword (1 wave/WG)", [&] { hipLaunchKernelGGL((finish_probe<1, 0>), dim3(SEGS), dim3(64), 0, 0, slab, out, doff, dcnt); });
    run("mode 1 dwordx4 loads, dword stores", [&] { hipLaunchKernelGGL((finish_probe<8, 1>), dim3(SEGS / 8), dim3(512), 0, 0, slab, out, doff, dcnt); });
    run("mode 2 dwordx4 loads, LDS, aligned dwordx4 stores", [&] { hipLaunchKernelGGL((finish_probe<8, 2>), dim3(SEGS / 8), dim3(512), 0, 0, slab, out, doff, dcnt); });
    run("mode 2 (4 waves/WG)", [&] { hipLaunchKernelGGL((finish_probe<4, 2>), dim3(SEGS / 4), dim3(256), 0, 0, slab, out, doff, dcnt); });
    run("mode 3 persistent 1024 x 8 waves, dword", [&] { hipLaunchKernelGGL((finish_probe<8, 3>), dim3(1024), dim3(512), 0, 0, slab, out, doff, dcnt); });
    run("mode 3 persistent 2048 x 4 waves, dword", [&] { hipLaunchKernelGGL((finish_probe<4, 3>), dim3(2048), dim3(256), 0, 0, slab, out, doff, dcnt); });
    run("mode 4 persistent 1024 x 8 waves, prefetch", [&] { hipLaunchKernelGGL((finish_probe<8, 4>), dim3(1024), dim3(512), 0, 0, slab, out, doff, dcnt); });
    run("mode 4 persistent 2048 x 4 waves, prefetch", [&] { hipLaunchKernelGGL((finish_probe<4, 4>), dim3(2048), dim3(256), 0, 0, slab, out, doff, dcnt); });
    HK(hipDeviceSynchronize());
    return 0;
}
