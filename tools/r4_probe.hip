// tools/r4_probe.hip -- round 4 diagnostic (not product): the large keys-only sort of 64 Mi u32 keys, launch by launch, with the
// kernels of the library's headers and their round-4 candidates side by side in ONE process on ONE box.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/r4_probe tools/r4_probe.hip
//   tools/r4_probe [n] [iters]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../oclradixsort_amd/csrc/persist_kernels.hpp"
#include "../oclradixsort_amd/csrc/finish16_kernels.hpp"

#define CK(x)                                                                                      \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));    \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)

using namespace adlhip;

__global__ void gen_keys(uint32_t* k, uint32_t n, uint64_t seed)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        uint64_t z = seed * 0x9E3779B97F4A7C15ull + (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        k[i] = (uint32_t)(z >> 32);
    }
}
// out[0] = inversions, out[1..2] = sum (lo, hi) of the keys, out[3] = xor
__global__ void check_sorted(const uint32_t* k, uint32_t n, unsigned long long* out)
{
    unsigned long long inv = 0, sum = 0, x = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t v = k[i];
        if (i + 1 < n && v > k[i + 1]) ++inv;
        sum += v;
        x ^= (unsigned long long)v * 0x9E3779B97F4A7C15ull;
    }
    atomicAdd(out + 0, inv);
    atomicAdd(out + 1, sum);
    atomicXor(out + 2, x);
}

// event pairs around every launch, recorded WITHOUT synchronising (the launches of a configuration run back to back, as in bench.py:
// the chip holds a lower clock under sustained load than between synchronised launches, and the LDS-bound finish shows it); all
// pairs are read at report()
// ---- histogram / read variants (the literal "HBM-read roofline" of the metric) ----------------------------------------------------
// MODE 0: xor-reduce only (read ceiling); 1: one ds_add per key on the wave's own 256 counters; 2: as 1 with the keys of a 16-byte
// load counted in one packed pass (two keys per ... no: plain).  NTL: non-temporal loads.  U: 16-byte loads in flight per lane.
template <int MODE, bool NTL, int U>
__global__ __launch_bounds__(256) void hist_variant_kernel(const uint4* __restrict__ src, uint32_t nvec, uint32_t* __restrict__ table,
                                                          int start_bit)
{
    __shared__ uint32_t hist[4 * 256];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 4 * 256; i += 256) hist[i] = 0u;
    __syncthreads();
    uint32_t* my = hist + w * 256;
    // contiguous chunk per workgroup, U loads in flight per lane
    const uint32_t per_wg = (nvec + gridDim.x - 1) / gridDim.x;
    const uint32_t begin = blockIdx.x * per_wg, end = min(nvec, begin + per_wg);
    uint32_t acc = 0u;
    auto ld = [&](uint32_t i) -> uint4 {
        if (NTL) { const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src) + i); return uint4{v.x, v.y, v.z, v.w}; }
        return src[i];
    };
    auto use = [&](const uint4& v) {
        if (MODE == 0) { acc ^= v.x ^ v.y ^ v.z ^ v.w; return; }
        __hip_atomic_fetch_add(&my[(v.x >> start_bit) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&my[(v.y >> start_bit) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&my[(v.z >> start_bit) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&my[(v.w >> start_bit) & 255u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    uint32_t i = begin + tid;
    if (i + (U - 1) * 256 < end) {
        uint4 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = ld(i + u * 256);
        i += U * 256;
        for (; i + (U - 1) * 256 < end; i += U * 256) {
            uint4 b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) b[u] = ld(i + u * 256);
#pragma unroll
            for (int u = 0; u < U; ++u) use(a[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = b[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) use(a[u]);
    }
    for (; i < end; i += 256) use(ld(i));
    __syncthreads();
    if (MODE == 0) { if (acc == 0x9e3779b9u) table[0] = acc; return; }
    table[(size_t)blockIdx.x * 256 + tid] = hist[tid] + hist[256 + tid] + hist[512 + tid] + hist[768 + tid];
}

struct Timer {
    struct Rec { std::string name; hipEvent_t e0, e1; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    size_t used = 0;
    hipEvent_t get()
    {
        if (used == pool.size()) { hipEvent_t e; CK(hipEventCreate(&e)); pool.push_back(e); }
        return pool[used++];
    }
    template <typename F>
    void run(const char* name, bool timed, F&& f)
    {
        if (!timed) { f(); return; }
        Rec r{name, get(), get()};
        CK(hipEventRecord(r.e0, 0));
        f();
        CK(hipEventRecord(r.e1, 0));
        recs.push_back(r);
    }
    void report(const char* title)
    {
        CK(hipDeviceSynchronize());
        std::vector<std::pair<std::string, std::pair<float, int>>> acc;
        for (auto& r : recs) {
            float ms = 0;
            CK(hipEventElapsedTime(&ms, r.e0, r.e1));
            bool found = false;
            for (auto& a : acc) if (a.first == r.name) { a.second.first += ms; a.second.second++; found = true; }
            if (!found) acc.push_back({r.name, {ms, 1}});
        }
        if (getenv("R4_SEQ")) {   // per-launch durations of one launch kind, in order (us)
            printf("  seq %s:", getenv("R4_SEQ"));
            for (auto& r : recs) if (r.name == getenv("R4_SEQ")) { float ms = 0; CK(hipEventElapsedTime(&ms, r.e0, r.e1)); printf(" %.0f", ms * 1000); }
            printf("\n");
        }
        printf("  [%s]", title);
        float sum = 0;
        for (auto& a : acc) { printf(" %s=%.4f", a.first.c_str(), a.second.first / a.second.second); sum += a.second.first / a.second.second; }
        printf(" | sum=%.4f\n", sum);
        recs.clear();
        used = 0;
    }
};

template <typename KERN>
static void ensure_lds(KERN k, size_t bytes)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? (size_t)atoll(argv[1]) : (size_t)64 << 20;
    const int iters = argc > 2 ? atoi(argv[2]) : 10;
    const char* only = argc > 3 ? argv[3] : "";
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs; n = %zu\n", prop.name, cus, n);

    typedef uint32_t E;
    const uint32_t stride_a = (uint32_t)(((n / 256 + (n / 256) / 2 + 4096) + 63) / 64 * 64);
    const uint32_t stride_b = 1536, slots = 65536;
    // NB key buffers taken in turns, each restored three sorts before its turn: a sort reads keys that are cold in every cache and
    // writes its result over them (as bench.py's steps do; ONE buffer restored right before its sort flatters pass 1 and the finish)
    constexpr int NB = 6;
    E *keys, *kbuf[NB], *orig, *slab_a, *tmp;
    uint16_t* slab_b;
    for (int i = 0; i < NB; ++i) CK(hipMalloc(&kbuf[i], n * 4));
    keys = kbuf[0];
    CK(hipMalloc(&orig, n * 4));
    CK(hipMalloc(&tmp, n * 4));
    CK(hipMalloc(&slab_a, (size_t)256 * stride_a * 4));
    CK(hipMalloc(&slab_b, (size_t)slots * stride_b * 2));
    uint32_t *ctl, *seg_cnt, *seg_off, *mode, *ctable, *fault, *host_mode;
    CK(hipMalloc(&ctl, (8192 + 65536 + 64) * 4));
    CK(hipMemset(ctl, 0, (8192 + 65536 + 64) * 4));
    uint32_t* cur_a = ctl;
    uint32_t* cur_b = ctl + 8192;
    uint32_t* flag = ctl + 8192 + 65536;
    uint32_t* done = flag + 1;
    uint32_t* bar = flag + 2;
    uint32_t* sample = flag + 8;
    {
        const uint32_t init[4] = {0u, 0u, ~0u, ~0u};
        CK(hipMemcpy(sample, init, 16, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&seg_cnt, 65536 * 4));
    CK(hipMalloc(&seg_off, 65537 * 4));
    CK(hipMalloc(&mode, 256));
    CK(hipMemset(mode, 0, 256));
    CK(hipMalloc(&ctable, 256 * 256 * 4 + 1024));
    CK(hipMalloc(&fault, 256));
    CK(hipMemset(fault, 0, 256));
    CK(hipHostMalloc(&host_mode, 64));
    unsigned long long* chk;
    CK(hipMalloc(&chk, 64));

    hipLaunchKernelGGL(gen_keys, dim3(4096), dim3(256), 0, 0, orig, (uint32_t)n, 123ull);
    CK(hipMemset(chk, 0, 64));
    hipLaunchKernelGGL(check_sorted, dim3(4096), dim3(256), 0, 0, orig, (uint32_t)n, chk);
    unsigned long long want[4];
    CK(hipMemcpy(want, chk, 32, hipMemcpyDeviceToHost));

    using CT = TileCfg<E, 8, 512, 32>;
    auto old1 = msd_bucket_scatter_kernel<E, 512, 32, 1>;
    auto old2 = msd_bucket_scatter_kernel<E, 512, 32, 2>;
    ensure_lds(old1, CT::LDS_BYTES);
    ensure_lds(old2, CT::LDS_BYTES);
    using CC = TileCfg<E, 8, 256, 16>;
    auto offk = msd2_offsets_kernel<E>;
    ensure_lds(offk, CC::LDS_BYTES);
    auto oldf = wave_segment_sort_kernel<E, 24, 4, 2, 2, uint16_t, false, false>;
    const size_t oldf_lds = 4 * (4 * 64 * 24 + 1024);
    ensure_lds(oldf, oldf_lds);

    BucketPass<E> pa;
    memset(&pa, 0, sizeof(pa));
    pa.src = keys; pa.dst = slab_a; pa.cursors = cur_a; pa.cursor_shift = 5; pa.src_count_shift = 0; pa.flag = flag;
    pa.src_counts = nullptr; pa.n = (uint32_t)n; pa.src_stride = 0; pa.tiles_per_bucket = 1; pa.dst_stride = stride_a;
    pa.dst_total = 256u * stride_a; pa.start_bit = 24; pa.zero_me = nullptr; pa.sample = sample; pa.which_digit = 1; pa.dst16 = 0;
    pa.seg_shift = 8;
    BucketPass<E> pb = pa;
    pb.src = slab_a; pb.dst = reinterpret_cast<E*>(slab_b); pb.cursors = cur_b; pb.cursor_shift = 0; pb.src_count_shift = 5;
    pb.src_counts = cur_a; pb.src_stride = stride_a; pb.tiles_per_bucket = (stride_a + 16383) / 16384; pb.dst_stride = stride_b;
    pb.dst_total = slots * stride_b; pb.start_bit = 16; pb.which_digit = 2; pb.dst16 = 1;

    Timer T;
    // one configuration = (pass kernels, finish); every iteration restores the input outside the timed launches
    auto chain = [&](const char* title, auto&& pass1, auto&& pass2, auto&& finish) {
        if (only[0] && !strstr(title, only)) return;
        for (int i = 0; i < 3; ++i) CK(hipMemcpyAsync(kbuf[i], orig, n * 4, hipMemcpyDeviceToDevice, 0));
        for (int it = 0; it < iters + 3; ++it) {
            const bool timed = it >= 3;
            keys = kbuf[it % NB];
            pa.src = keys;
            CK(hipMemcpyAsync(kbuf[(it + 3) % NB], orig, n * 4, hipMemcpyDeviceToDevice, 0));
            T.run("sample", timed, [&] {
                hipLaunchKernelGGL(msd2_sample_kernel<E>, dim3(kSampleWGs), dim3(64), 0, 0, (const E*)keys, (uint32_t)n, sample, bar, fault);
            });
            T.run("pass1", timed, pass1);
            T.run("pass2", timed, pass2);
            T.run("offsets", timed, [&] {
                hipLaunchKernelGGL(offk, dim3(256), dim3(256), CC::LDS_BYTES, 0, cur_a, cur_b, flag, done, bar, seg_cnt, seg_off, mode,
                                   host_mode, (uint32_t)n, sample, keys, tmp, ctable, fault, 32, 0u);
            });
            T.run("finish", timed, finish);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemset(chk, 0, 64));
        hipLaunchKernelGGL(check_sorted, dim3(4096), dim3(256), 0, 0, keys, (uint32_t)n, chk);
        unsigned long long got[4];
        CK(hipMemcpy(got, chk, 32, hipMemcpyDeviceToHost));
        uint32_t hm[4], flt[4];
        CK(hipMemcpy(hm, mode, 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(flt, fault, 16, hipMemcpyDeviceToHost));
        const bool ok = got[0] == 0 && got[1] == want[1] && got[2] == want[2];
        printf("%-34s %s (inversions %llu, mode %u, fault %x %x)\n", title, ok ? "OK" : "WRONG", got[0], hm[0], flt[0], flt[1]);
        T.report(title);
    };

    auto p1_old = [&] { hipLaunchKernelGGL(old1, dim3((uint32_t)((n + 16383) / 16384)), dim3(512), CT::LDS_BYTES, 0, pa); };
    auto p2_old = [&] { hipLaunchKernelGGL(old2, dim3(256 * pb.tiles_per_bucket), dim3(512), CT::LDS_BYTES, 0, pb); };
    auto f_old = [&] {
        hipLaunchKernelGGL(oldf, dim3(65536 / 4), dim3(256), oldf_lds, 0, reinterpret_cast<const E*>(slab_b), keys, seg_off, 65536u, 16u,
                           fault, seg_cnt, stride_b, mode, mode + kDynLowBits, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                           (const uint32_t*)nullptr, 8u);
    };
    if (strstr(only, "hist")) {
        uint32_t* table;
        CK(hipMalloc(&table, (size_t)8192 * 256 * 4));
        for (int i = 0; i < NB; ++i) CK(hipMemcpyAsync(kbuf[i], orig, n * 4, hipMemcpyDeviceToDevice, 0));
        int turn = 0;
        auto bench_hist = [&](const char* name, auto&& launch) {
            for (int it = 0; it < iters + 3; ++it) T.run(name, it >= 3, [&] { launch(kbuf[(turn++) % NB]); });
        };
        auto libk = radix_count_kernel<uint32_t, 8, 256>;
        bench_hist("lib_count_2048", [&](E* k) { hipLaunchKernelGGL(libk, dim3(2048), dim3(256), 0, 0, (const E*)k, table, (uint32_t)n, 2048, 24, 32768u, 0); });
#define HV(MODE_, NTL_, U_, G_)                                                                                                  \
        bench_hist("m" #MODE_ "_nt" #NTL_ "_u" #U_ "_g" #G_, [&](E* k) {                                                         \
            hipLaunchKernelGGL((hist_variant_kernel<MODE_, NTL_, U_>), dim3(G_), dim3(256), 0, 0, (const uint4*)k, (uint32_t)(n / 4), table, 24); });
        HV(0, false, 4, 2048) HV(0, true, 4, 2048) HV(0, true, 8, 2048) HV(0, true, 4, 4096) HV(0, true, 8, 1024) HV(0, true, 2, 4096)
        HV(1, false, 4, 2048) HV(1, true, 4, 2048) HV(1, true, 8, 2048) HV(1, true, 4, 4096) HV(1, true, 8, 1024) HV(1, true, 2, 4096)
        T.report("hist 64Mi u32 (256 MiB read per launch; cold buffers in turns)");
        return 0;
    }
    chain("old/old/old", p1_old, p2_old, f_old);

#define PERSIST(NT_, K_, WGS_, L1, S1, L2, S2, FIN)                                                                              \
    {                                                                                                                            \
        using PC = PersistCfg<E, NT_, K_>;                                                                                       \
        auto k1 = msd_scatter_persist_kernel<E, NT_, K_, 1, (WGS_) * (NT_) / 256, false, L1, S1>;                                \
        auto k2 = msd_scatter_persist_kernel<E, NT_, K_, 2, (WGS_) * (NT_) / 256, true, L2, S2>;                                 \
        ensure_lds(k1, PC::LDS_BYTES);                                                                                           \
        ensure_lds(k2, PC::LDS_BYTES);                                                                                           \
        const uint32_t grid = (uint32_t)(WGS_) * (uint32_t)cus;                                                                  \
        auto p1 = [&] { hipLaunchKernelGGL(k1, dim3(grid), dim3(NT_), PC::LDS_BYTES, 0, pa); };                                  \
        auto p2 = [&] { hipLaunchKernelGGL(k2, dim3(grid), dim3(NT_), PC::LDS_BYTES, 0, pb); };                                  \
        char title[96];                                                                                                          \
        snprintf(title, sizeof title, "persist %dx%d wg/cu=%d aux %d%d/%d%d %s", NT_, K_, WGS_, L1, S1, L2, S2, #FIN);           \
        chain(title, p1, p2, FIN);                                                                                               \
    }
#define FIN16(NAME, WAVES_, NTL_, NTS_, ALG_)                                                                                    \
    auto kf_##NAME = wave_finish16_kernel<12, WAVES_, NTL_, NTS_, ALG_>;                                                         \
    const size_t lds_##NAME = (size_t)WAVES_ * Finish16Cfg<12>::PER_WAVE;                                                        \
    ensure_lds(kf_##NAME, lds_##NAME);                                                                                           \
    auto NAME = [&] {                                                                                                            \
        hipLaunchKernelGGL(kf_##NAME, dim3(65536 / WAVES_), dim3(64 * WAVES_), lds_##NAME, 0, (const uint16_t*)slab_b, keys, seg_off,  \
                           seg_cnt, stride_b, mode, mode + kDynLowBits, 65536u, fault);                                          \
    };
    FIN16(f16_00_a0, 4, false, false, 0)
    FIN16(f16_10_a0, 4, true, false, 0)
    FIN16(f16_01_a0, 4, false, true, 0)
    FIN16(f16_11_a0, 4, true, true, 0)
    FIN16(f16_00_a1, 4, false, false, 1)
    FIN16(f16_10_a1, 4, true, false, 1)
    FIN16(f16_11_a1, 4, true, true, 1)
    FIN16(f16_10_a1_w8, 8, true, false, 1)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f_old)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_00_a0)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_10_a0)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_01_a0)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_11_a0)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_00_a1)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_10_a1)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_11_a1)
    PERSIST(512, 32, 2, 2, 0, 2, 0, f16_10_a1_w8)
    PERSIST(512, 32, 2, 0, 0, 0, 0, f16_00_a0)
    return 0;
}
