// Diagnostic (not product): what does the SHAPE of the write-out of a one-sweep pass cost on this MI355X?
// Same bytes and same destinations as a real 8-bit pass over random keys (4096 tiles of 16 Ki keys, 16 chains,
// per-(tile, digit) run lengths drawn from real random digits, so runs start and end at arbitrary 4-byte
// offsets), but no ranking / look-back: the tile-sorted LDS image is synthesised.  Only the mapping
// lanes -> destination addresses of the write-out varies:
//   mode 0  today's kernel: thread t stores tile positions t, t + NT, ... (a wave instruction = 64 consecutive
//           tile positions = pieces of ~2 runs, cut wherever the window ends)
//   mode 1  windows aligned to 128-byte lines of the DESTINATION: a half-wave per (digit, line)
//   mode 2  windows aligned to 256 bytes of the destination: a wave per (digit, 256-byte block)
//   mode 3  windows aligned to 64 bytes: a quarter-wave per (digit, 64-byte sector)
//   mode 4  hand-over emulation: every tile writes whole 128-byte lines only; the head line of a run is
//           completed with the predecessor tile's trailing keys (read from a hand-over area, sc1), the
//           run's own trailing partial line goes to the tile's hand-over slot (sc1) instead of the destination
//   mode 5  16 bytes per lane: a lane owns four consecutive tile positions; when they belong to one run (15 of 16 lanes
//           at 64-key runs) it stores them with one dwordx4 (4-byte aligned), otherwise with four dword stores
//   mode 6  8 bytes per lane: two consecutive tile positions, one dwordx2 when they share a run
//   hipcc --offload-arch=gfx950 -O3 -o tools/writeout_probe tools/writeout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int CHAINS = 16, BINS = 256;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
#define DPP(ctrl, rm) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rm, 0xf, false)
    v += DPP(0x111, 0xf); v += DPP(0x112, 0xf); v += DPP(0x114, 0xf); v += DPP(0x118, 0xf);
    v += DPP(0x142, 0xa); v += DPP(0x143, 0xc);
#undef DPP
    return v;
}

template <int NT, int K>
__global__ __launch_bounds__(NT) void writeout_probe(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                     const uint32_t* __restrict__ gstart, const uint32_t* __restrict__ cnts,
                                                     uint32_t* __restrict__ handover, uint32_t tiles_per_chain, uint32_t n, int mode)
{
    constexpr int TILE = NT * K;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    uint32_t* s_elems = lds;                       // [TILE]
    uint32_t* s_goff = lds + TILE;                 // [256] global start - tile offset
    uint32_t* s_s = s_goff + BINS;                 // [256] global start
    uint32_t* s_e = s_s + BINS;                    // [256] global end
    uint32_t* s_ws = s_e + BINS;                   // [256] first window of the digit
    uint32_t* s_wsum = s_ws + BINS;                // [32]
    uint8_t* s_wd = reinterpret_cast<uint8_t*>(s_wsum + 32);   // [<= TILE/16 + 256] digit of window k
    const uint32_t t = blockIdx.x, c = t % CHAINS, i = t / CHAINS;
    const uint32_t row = c * tiles_per_chain + i;   // tiles of a chain are consecutive rows
    const int tid = (int)threadIdx.x, w = tid >> 6, lane = tid & 63;
    // the read side of the pass: wave-striped dword loads of the tile
    const uint32_t* p = src + (size_t)row * TILE + w * 64 * K + lane;
    uint32_t e[K];
#pragma unroll
    for (int j = 0; j < K; ++j) e[j] = p[j * 64];
    uint32_t acc = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) acc ^= e[j];

    // per-digit geometry
    uint32_t cnt = 0, gs = 0;
    if (tid < BINS) { cnt = cnts[row * BINS + tid]; gs = gstart[row * BINS + tid]; }
    auto block_excl = [&](uint32_t v, uint32_t* total) -> uint32_t {   // valid for tid < 256 (4 waves)
        const uint32_t inc = wave_incl_scan(v);
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();
        uint32_t off = 0, tot = 0;
        for (int k = 0; k < 4; ++k) { const uint32_t x = s_wsum[k]; if (k < w) off += x; tot += x; }
        __syncthreads();
        if (total) *total = tot;
        return off + inc - v;
    };
    const uint32_t toff = block_excl(tid < BINS ? cnt : 0u, nullptr);
    const int W = mode == 2 ? 64 : (mode == 3 ? 16 : 32);
    uint32_t nw = 0;
    if (tid < BINS && cnt) {
        if (mode == 4) nw = (gs + cnt) / 32 - gs / 32;             // whole lines only; the tail line is handed over
        else nw = (gs + cnt - 1) / W - gs / W + 1;
    }
    uint32_t nwin = 0;
    const uint32_t wstart = block_excl(tid < BINS ? nw : 0u, &nwin);
    if (tid < BINS) {
        s_goff[tid] = gs - toff;
        s_s[tid] = gs;
        s_e[tid] = gs + cnt;
        s_ws[tid] = wstart;
        for (uint32_t k = 0; k < cnt; ++k) s_elems[toff + k] = (acc & 0xffffff00u) | (uint32_t)tid;   // key with digit tid
        for (uint32_t k = 0; k < nw; ++k) s_wd[wstart + k] = (uint8_t)tid;
    }
    __syncthreads();

    if (mode == 0) {
#pragma unroll 8
        for (int j = 0; j < K; ++j) {
            const uint32_t pos = (uint32_t)(tid + j * NT);
            const uint32_t v = s_elems[pos];
            const uint32_t g = s_goff[v & 255u] + pos;
            if (g < n) dst[g] = v;
        }
    } else if (mode == 5) {
#pragma unroll 4
        for (int j = 0; j < K / 4; ++j) {
            const uint32_t pos = (uint32_t)(tid + j * NT) * 4u;
            const uint4 v = *reinterpret_cast<const uint4*>(s_elems + pos);
            const uint32_t g0 = s_goff[v.x & 255u] + pos;
            if ((v.x & 255u) == (v.w & 255u)) {
                if (g0 + 3u < n) __builtin_memcpy(dst + g0, &v, 16);   // one 16-byte store, 4-byte aligned
            } else {
                if (g0 < n) dst[g0] = v.x;
                const uint32_t g1 = s_goff[v.y & 255u] + pos + 1u; if (g1 < n) dst[g1] = v.y;
                const uint32_t g2 = s_goff[v.z & 255u] + pos + 2u; if (g2 < n) dst[g2] = v.z;
                const uint32_t g3 = s_goff[v.w & 255u] + pos + 3u; if (g3 < n) dst[g3] = v.w;
            }
        }
    } else if (mode == 6) {
#pragma unroll 8
        for (int j = 0; j < K / 2; ++j) {
            const uint32_t pos = (uint32_t)(tid + j * NT) * 2u;
            const uint2 v = *reinterpret_cast<const uint2*>(s_elems + pos);
            const uint32_t g0 = s_goff[v.x & 255u] + pos;
            if ((v.x & 255u) == (v.y & 255u)) {
                if (g0 + 1u < n) __builtin_memcpy(dst + g0, &v, 8);
            } else {
                if (g0 < n) dst[g0] = v.x;
                const uint32_t g1 = s_goff[v.y & 255u] + pos + 1u; if (g1 < n) dst[g1] = v.y;
            }
        }
    } else if (mode <= 3) {
        const uint32_t gid = (uint32_t)tid / (uint32_t)W, l = (uint32_t)tid % (uint32_t)W, ng = (uint32_t)NT / (uint32_t)W;
#pragma unroll 4
        for (uint32_t k = gid; k < nwin; k += ng) {
            const uint32_t d = s_wd[k];
            const uint32_t s = s_s[d], en = s_e[d];
            const uint32_t g = (s / W + (k - s_ws[d])) * W + l;
            if (g >= s && g < en && g < n) dst[g] = s_elems[g - s_goff[d]];
        }
    } else {
        const uint32_t gid = (uint32_t)tid / 32u, l = (uint32_t)tid % 32u, ng = (uint32_t)NT / 32u;
        const uint32_t* hin = handover + (size_t)(c * 64u + ((i + 63u) & 63u)) * (BINS * 32);   // predecessor's slot
        uint32_t* hout = handover + (size_t)(c * 64u + (i & 63u)) * (BINS * 32);
        const bool first = i == 0, last = i + 1 == tiles_per_chain;
#pragma unroll 4
        for (uint32_t k = gid; k < nwin; k += ng) {
            const uint32_t d = s_wd[k];
            const uint32_t s = s_s[d];
            const uint32_t g = (s / 32u + (k - s_ws[d])) * 32u + l;
            uint32_t v;
            if (g >= s) v = s_elems[g - s_goff[d]];
            else if (!first) v = __hip_atomic_load(hin + d * 32u + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // predecessor's trailing keys
            else continue;
            if (g < n) dst[g] = v;
        }
        // trailing partial lines: one half-wave per digit
        for (uint32_t d = gid; d < BINS; d += ng) {
            const uint32_t s = s_s[d], en = s_e[d];
            const uint32_t g = (en / 32u) * 32u + l;
            if (g >= s && g < en) {
                const uint32_t v = s_elems[g - s_goff[d]];
                if (last) { if (g < n) dst[g] = v; }
                else __hip_atomic_store(hout + d * 32u + l, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t rng() { uint64_t x = (rng_state += 0x9E3779B97F4A7C15ull); x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

template <int NT, int K>
void run(size_t n, size_t lds_kib)
{
    constexpr int TILE = NT * K;
    const uint32_t tiles = (uint32_t)(n / TILE), tpc = tiles / CHAINS;
    std::vector<uint32_t> cnt((size_t)tiles * BINS, 0), gs((size_t)tiles * BINS);
    for (uint32_t r = 0; r < tiles; ++r)
        for (int k = 0; k < TILE; k += 8) { uint64_t x = rng(); for (int b = 0; b < 8; ++b) cnt[(size_t)r * BINS + ((x >> (8 * b)) & 255)]++; }
    // regions ordered (digit, chain); rows r = c * tpc + i
    std::vector<uint64_t> tot((size_t)BINS * CHAINS, 0);
    for (uint32_t r = 0; r < tiles; ++r) for (int d = 0; d < BINS; ++d) tot[(size_t)d * CHAINS + r / tpc] += cnt[(size_t)r * BINS + d];
    uint64_t run_ = 0;
    std::vector<uint64_t> base((size_t)BINS * CHAINS);
    for (size_t k = 0; k < tot.size(); ++k) { base[k] = run_; run_ += tot[k]; }
    for (uint32_t r = 0; r < tiles; ++r) for (int d = 0; d < BINS; ++d) { uint64_t& b = base[(size_t)d * CHAINS + r / tpc]; gs[(size_t)r * BINS + d] = (uint32_t)b; b += cnt[(size_t)r * BINS + d]; }
    uint32_t *a, *b, *dg, *dc, *ho;
    HK(hipMalloc(&a, n * 4 + 256)); HK(hipMalloc(&b, n * 4 + 256));
    HK(hipMalloc(&dg, gs.size() * 4)); HK(hipMalloc(&dc, cnt.size() * 4));
    HK(hipMalloc(&ho, (size_t)CHAINS * 64 * BINS * 32 * 4));
    HK(hipMemset(a, 1, n * 4)); HK(hipMemset(b, 2, n * 4)); HK(hipMemset(ho, 3, (size_t)CHAINS * 64 * BINS * 32 * 4));
    HK(hipMemcpy(dg, gs.data(), gs.size() * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(dc, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice));
    auto kern = writeout_probe<NT, K>;
    const size_t lds = lds_kib * 1024;
    HK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    printf("tile %d x %d = %d keys, LDS %zu KiB per workgroup -> %d workgroup(s) per CU; n = %zu keys, %u tiles\n", NT, K, TILE, lds_kib, (int)(160 / lds_kib), n, tiles);
    const char* names[] = {"0 position-major (today)", "1 aligned 128-B lines (half-wave)", "2 aligned 256-B blocks (wave)", "3 aligned 64-B sectors (quarter-wave)", "4 hand-over: whole lines only", "5 16 bytes per lane where a run allows", "6 8 bytes per lane where a run allows"};
    for (int mode : {0, 5, 6, 0, 5, 6}) {
        for (int k = 0; k < 3; ++k) { kern<<<tiles, NT, lds>>>(a, b, dg, dc, ho, tpc, (uint32_t)n, mode); kern<<<tiles, NT, lds>>>(b, a, dg, dc, ho, tpc, (uint32_t)n, mode); }
        HK(hipEventRecord(e0));
        const int reps = 10;
        for (int k = 0; k < reps; ++k) { kern<<<tiles, NT, lds>>>(a, b, dg, dc, ho, tpc, (uint32_t)n, mode); kern<<<tiles, NT, lds>>>(b, a, dg, dc, ho, tpc, (uint32_t)n, mode); }
        HK(hipEventRecord(e1)); HK(hipEventSynchronize(e1));
        float ms; HK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / (2 * reps);
        printf("  mode %-40s %7.1f us  %7.1f GB/s read+write\n", names[mode], us, 2.0 * n * 4 / us / 1e3);
        fflush(stdout);
    }
    HK(hipFree(a)); HK(hipFree(b)); HK(hipFree(dg)); HK(hipFree(dc)); HK(hipFree(ho));
}

int main(int argc, char** argv)
{
    const size_t n = argc > 1 ? strtoull(argv[1], 0, 0) : (size_t)1 << 26;
    run<512, 32>(n, 76);     // today's tile, two workgroups per CU
    return 0;
}
