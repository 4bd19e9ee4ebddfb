// adlhip.hip -- implementation of include/adlhip.h: the HIP back-end behind the Adl/Pprims facade.
// Replaces Adl/CL/AdlCL.inl (device, buffers, copies, map/unmap), Adl/CL/AdlKernelUtilsCL.inl
// (kernel launch + per-launch profiling) and the GPU branches of Tahoe/ParallelPrimitives/Pprims.cpp
// (pass drivers).  gfx950 only; kernels are compiled ahead of time into this shared object.
#include "../../include/adlhip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <utility>
#include <string>
#include <vector>

#include "radix_kernels.hpp"
#include "onesweep_kernels.hpp"
#include "hybrid_kernels.hpp"
#include "persist_kernels.hpp"
#include "finish16_kernels.hpp"
#include "soa_wide_kernels.hpp"

// The large sort's kernels and the per-digit passes' are instantiated in kernels_finish.hip / kernels_passes.hip / kernels_perdigit.hip (translation units of
// their own, compiled beside this one); here they are only declared.  -DADLHIP_SINGLE_TU builds everything in this file (what tools/gen_large_kernels.py reads the list from).
#ifndef ADLHIP_SINGLE_TU
#define X(...) extern template __global__ __VA_ARGS__;
#include "finish_kernels.inc"
#include "wavefinish_kernels.inc"
#include "passes_kernels.inc"
#include "perdigit_kernels.inc"
#undef X
#endif

namespace {

thread_local char g_err[512] = "";

int fail(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (getenv("ADLHIP_VERBOSE")) fprintf(stderr, "[adlhip] error: %s\n", g_err);
    return ADLHIP_FAILURE;
}

#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct ProfEntry {
    uint64_t launches = 0;
    double total_ms = 0.0;
};
struct PendingProf {
    const char* name;
    hipEvent_t e0, e1;
};
struct Staging {
    void* hptr;
    size_t bytes;      // bytes mapped
    hipEvent_t done;   // null while mapped; set at unmap
    size_t capacity;   // bytes of pinned memory behind hptr (>= bytes when it came from the pool)
};
struct PinnedBlock {
    void* hptr;
    size_t capacity;
};
constexpr size_t kPinnedPoolMax = size_t(512) << 20;   // pinned staging kept for re-use per device handle

}  // namespace

struct adlhip_event {
    hipEvent_t ev;
};

struct adlhip_device {
    int idx = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    hipDeviceProp_t prop;
    uint64_t used_bytes = 0;
    // knobs
    int sort_algo = -1;       // -1 automatic by size, 0 onesweep, 1 three-kernel pass
    int digit_bits = 8;       // 8 or 4
    int profile = 0;
    int tile_variant = -1;    // index into kVariants; -1 = best known per element size
    int rank_mode = 1;        // 1 = lane-ordered DS atomic ranking (needs lds_ordered), 0 = ballot match
    int lds_ordered = 0;      // result of the device self-test at creation
    int resident_wgs_device = 0;
    int resident_wgs = 0;     // workgroups of <= 80 KiB LDS / 512 threads that are certainly resident at once (2 per CU); the paths
                              // whose kernels hold a grid-wide barrier over 256 workgroups are taken only when this is >= 256
    int mid_path = 1;         // 16 Ki < n <= 2 Mi: MSD pass + LDS finish (three launches) instead of per-digit passes
    int mid_skip = 0;         // eligible sorts still to be sent down the per-digit passes after a skewed input (see mid_eligible)
    int mid2_skip = 0;        // keys-only sorts still to take the three-launch form after a slab overflow (see mid_sort_keys)
    int mid_backoff = 32, mid2_backoff = 64;
    int bin_finish = 1;       // "sort.binfinish": the large keys-only sort finishes its segments with one counting pass + compares
                              // (1: u64 keys, 2: u32 keys too, 0: the wave-per-segment LSD finish)
    int persist = 1;          // "sort.persist": the cursor passes of the large sort as persistent, prefetching kernels + the 16-bit finish
    int msd2_path = 1;        // "sort.msd2": the large sort (msd2_sort for keys, msd2s_sort for pairs); 2 = forced (tests)
    int net_lookback = 1;                   // "sort.net_lookback": the large sort's safety net runs look-back passes (0: count-scan-scatter passes)
    int partition_lookback = 1;             // "partition.lookback": the MSB partition as one look-back pass where it pays (0: always three kernels)
    int dict_path = 1;                      // "sort.dict": the large sort's safety net first tries the counting sort for keys that take at most
                                            // 256 values (dict_kernels.hpp); 0 = off
    adlhip::DictBlock* d_dict = nullptr;    // its dictionary and counters (handle-owned; rebuilt by every net that uses them)
    uint32_t* d_msd2 = nullptr;   // the large sort's handle-owned words, allocated with the handle and idle between sorts: cursors of
                                  // pass 1 (256, one 128-byte line each) and pass 2 (65536), overflow flag, done counter, the safety
                                  // net's barrier counter, the four sample words
    // profiling
    std::vector<PendingProf> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, ProfEntry> prof;
    std::vector<std::string> prof_order;
    // map/unmap staging; released staging blocks are pooled: pinning memory costs ~1 ms per 4 MiB, and the
    // reference's test maps every buffer two or three times (UnitTest/main.cpp:118-139)
    std::vector<Staging> staging;
    std::vector<PinnedBlock> pinned_pool;
    size_t pinned_pool_bytes = 0;
    // device-side fault words ([0] live, [1] sticky: onesweep_kernels.hpp raise_fault), checked at sync and by
    // adlhip_fault_check; [8] is the self-test's result slot
    uint32_t* d_fault = nullptr;
    uint32_t* h_fault = nullptr;   // pinned: [0..1] filled by adlhip_sync, [4] by the last adlhip_fault_check snapshot
    hipEvent_t fault_snap = nullptr;   // recorded behind the last snapshot copy; null = none pending
    uint32_t* d_mid_hist = nullptr;    // [16][4][256] slice histograms of the mid-size sort + 512 words of bucket cursors / flags
                                       // of its keys-only form (hybrid_kernels.hpp SegSlab): zero between sorts
};

namespace {

int bind(adlhip_device* d)
{
    if (!d) return fail("null device handle");
    HIPCHK(hipSetDevice(d->idx));
    return ADLHIP_SUCCESS;
}

hipEvent_t take_event(adlhip_device* d)
{
    if (!d->event_pool.empty()) {
        hipEvent_t e = d->event_pool.back();
        d->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// Launch wrapper: optional hipEvent bracket per launch ("profile" = 1), error check after.
inline int trace_level()
{
    static const int level = getenv("ADLHIP_TRACE") ? atoi(getenv("ADLHIP_TRACE")) : 0;
    return level;
}

template <typename F>
int launch(adlhip_device* d, const char* name, F&& f)
{
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (d->profile) {
        e0 = take_event(d);
        e1 = take_event(d);
        if (!e0 || !e1) return fail("hipEventCreate failed");
        HIPCHK(hipEventRecord(e0, d->stream));
    }
    // ADLHIP_TRACE=1 (debugging aid): name every launch on stderr and wait for it, so that the last line before a GPU fault
    // names the kernel that raised it
    // (ADLHIP_TRACE=2: names only, nothing waits)
    static const int trace = trace_level();
    if (trace) fprintf(stderr, "[adlhip] launch %s\n", name), fflush(stderr);
    f();
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("launch of %s failed: %s", name, hipGetErrorString(e));
    if (trace == 1 && (e = hipStreamSynchronize(d->stream)) != hipSuccess) return fail("%s: %s", name, hipGetErrorString(e));
    if (d->profile) {
        HIPCHK(hipEventRecord(e1, d->stream));
        d->pending.push_back({name, e0, e1});
    }
    return ADLHIP_SUCCESS;
}

int fold_profile(adlhip_device* d)
{
    if (d->pending.empty()) return ADLHIP_SUCCESS;
    HIPCHK(hipStreamSynchronize(d->stream));
    for (auto& p : d->pending) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p.e0, p.e1));
        auto it = d->prof.find(p.name);
        if (it == d->prof.end()) {
            d->prof_order.push_back(p.name);
            it = d->prof.emplace(p.name, ProfEntry()).first;
        }
        it->second.launches++;
        it->second.total_ms += ms;
        d->event_pool.push_back(p.e0);
        d->event_pool.push_back(p.e1);
    }
    d->pending.clear();
    return ADLHIP_SUCCESS;
}

void reap_staging(adlhip_device* d, bool all_done)
{
    for (size_t i = 0; i < d->staging.size();) {
        Staging& s = d->staging[i];
        if (s.done && (all_done || hipEventQuery(s.done) == hipSuccess)) {
            hipEventDestroy(s.done);
            if (d->pinned_pool_bytes + s.capacity <= kPinnedPoolMax) {
                d->pinned_pool.push_back({s.hptr, s.capacity});
                d->pinned_pool_bytes += s.capacity;
            } else {
                hipHostFree(s.hptr);
            }
            d->staging[i] = d->staging.back();
            d->staging.pop_back();
        } else {
            ++i;
        }
    }
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- sort configuration -------------------------------------------------------------------

// Tile geometry variants (threads per workgroup NT x elements per thread K), selectable at run time
// through "sort.tile".  The small digit table of the three-kernel pass is sized for the smallest tile (kMinTile);
// the status rows of the one-sweep pass are sized for the CURRENT knobs (a sort call re-checks the size, so changing the
// variant never invalidates a caller's work buffer.
struct TileVariant {
    int nt, k;
};
constexpr TileVariant kVariants[] = {{256, 16}, {512, 16}, {1024, 16}, {512, 8}, {1024, 8}, {256, 32}, {512, 32}};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));
constexpr uint32_t kMinTile = 4096;
constexpr int kWgsPerCu = 8;   // three-kernel pass: workgroups per CU that each own a run of tiles

struct PassPlan {
    int start_bit;
    int nbits;
};

std::vector<PassPlan> plan_passes(int sort_bits, int digit_bits)
{
    std::vector<PassPlan> p;
    int sb = 0;
    while (sb < sort_bits) {
        // 7-bit digits only where the key width allows 7,...,7(,4); every other width falls back to 8-bit digits
        // (and up to 32 sorted bits: a 7-bit digit starting at bit 28 would straddle the two dwords of a 64-bit key,
        // which digit_of() rules out)
        const bool seven = digit_bits == 7 && sort_bits % 7 % 4 == 0 && sort_bits <= 32;
        int nb = ((digit_bits == 8 || (digit_bits == 7 && !seven)) && sort_bits - sb >= 8) ? 8 : 4;
        if (seven && sort_bits - sb >= 7) nb = 7;
        p.push_back({sb, nb});
        sb += nb;
    }
    return p;
}

struct Geometry {
    uint32_t tile, num_tiles, tiles_per_wg, n_wgs;
};

Geometry geometry(const adlhip_device* d, size_t n, uint32_t tile)
{
    Geometry g;
    g.tile = tile;
    g.num_tiles = (uint32_t)((n + tile - 1) / tile);
    const uint32_t max_wgs = (uint32_t)d->prop.multiProcessorCount * kWgsPerCu;
    g.tiles_per_wg = (g.num_tiles + max_wgs - 1) / max_wgs;
    if (g.tiles_per_wg == 0) g.tiles_per_wg = 1;
    g.n_wgs = (g.num_tiles + g.tiles_per_wg - 1) / g.tiles_per_wg;
    return g;
}

// "sort.tile" = -1 (default): the measured best per element size on MI355X
//   4-byte elements: 512 x 32 (16 Ki keys, 73 KiB of LDS, two workgroups per CU)
//   8-byte elements: 1024 x 16 (16 Ki elements, 145 KiB of LDS, one workgroup per CU)
//   up to 8 MiB of data: 256 x 16 (4 Ki elements) so that there are enough tiles to occupy 256 CUs;
//   4-byte elements up to 24 MiB: 512 x 16
int effective_variant(const adlhip_device* d, size_t elem_bytes, size_t n)
{
    if (d->digit_bits == 7) return elem_bytes == 4 ? 6 : 2;   // 7-bit digits are instantiated for the default large tiles only
    if (d->tile_variant >= 0) return d->tile_variant;
    if (n * elem_bytes <= (size_t(8) << 20)) return 0;
    // 4-byte keys between 8 and 24 MiB (three-kernel passes): 512 x 16 (8 Ki keys) -- 4Mi keys 75 vs 79 us with 512 x 32
    if (elem_bytes == 4 && n * elem_bytes < (size_t(24) << 20)) return 1;
    return elem_bytes == 4 ? 6 : 2;
}
uint32_t current_tile(const adlhip_device* d, size_t elem_bytes, size_t n)
{
    const TileVariant v = kVariants[effective_variant(d, elem_bytes, n)];
    return (uint32_t)(v.nt * v.k);
}

constexpr size_t kMaxElems = 0xFFF00000ull;   // 32-bit element indices inside the kernels

// work buffer layout (three-kernel pass): [table 256 x n_wgs u32][totals 256 u32]
size_t table_bytes(const adlhip_device* d, size_t n, uint32_t tile)
{
    return align_up((size_t)256 * geometry(d, n, tile).n_wgs * 4, 256);
}
size_t work_bytes_three_kernel(const adlhip_device* d, size_t n) { return table_bytes(d, n, kMinTile) + 256 * 4; }

// Kernels with more than 64 KiB of dynamic LDS need the limit raised once per function.
// Kernels that need more than 64 KiB of dynamic LDS must be told so once (per device).  Distinct handles may be driven
// from distinct host threads, so the bookkeeping is locked.
template <typename KernelT>
int ensure_lds(KernelT kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return ADLHIP_SUCCESS;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> raised;   // (kernel, device) -> bytes
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = raised[std::make_pair((const void*)kernel, dev)];
    if (have >= bytes) return ADLHIP_SUCCESS;
    HIPCHK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    have = bytes;
    return ADLHIP_SUCCESS;
}

// Interned kernel names (the profiler keeps the pointer until the events are folded).
const char* intern(const std::string& s)
{
    static std::mutex mu;
    static std::map<std::string, std::string*> pool;
    std::lock_guard<std::mutex> lock(mu);
    auto it = pool.find(s);
    if (it == pool.end()) it = pool.emplace(s, new std::string(s)).first;
    return it->second->c_str();
}

// ---- what a sort ping-pongs between ------------------------------------------------------------------
// AosBuf<E>: one array of E.  SoaBuf: u32 keys + u32 values in separate arrays (pairs travel through the
// kernels as u64 {key low, value high}; histogram / count kernels read the key array only).
template <typename E>
struct AosBuf {
    typedef adlhip::AosIO<E> IO;
    typedef E key_t;
    static constexpr size_t kElemBytes = sizeof(E);
    static constexpr bool kSoa = false;
    E* p;
    const key_t* keys() const { return p; }
    bool same(const AosBuf& o) const { return p == o.p; }
    static IO io(const AosBuf& src, const AosBuf& dst) { return IO{src.p, dst.p}; }
    int copy_from(const AosBuf& src, size_t n, hipStream_t st) const
    {
        HIPCHK(hipMemcpyAsync(p, src.p, n * sizeof(E), hipMemcpyDeviceToDevice, st));
        return ADLHIP_SUCCESS;
    }
    static const char* tag() { return sizeof(E) == 4 ? "_u32" : "_e64"; }
};
struct SoaBuf {
    typedef adlhip::SoaIO IO;
    typedef uint32_t key_t;
    static constexpr size_t kElemBytes = 8;
    static constexpr bool kSoa = true;
    uint32_t* k;
    uint32_t* v;
    const key_t* keys() const { return k; }
    bool same(const SoaBuf& o) const { return k == o.k; }
    static IO io(const SoaBuf& src, const SoaBuf& dst) { return IO{src.k, src.v, dst.k, dst.v}; }
    int copy_from(const SoaBuf& src, size_t n, hipStream_t st) const
    {
        HIPCHK(hipMemcpyAsync(k, src.k, n * 4, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(v, src.v, n * 4, hipMemcpyDeviceToDevice, st));
        return ADLHIP_SUCCESS;
    }
    static const char* tag() { return "_soa"; }
};

template <typename Buf, int NBITS>
const char* kernel_name(const char* stem)
{
    return intern(std::string(stem) + Buf::tag() + (NBITS == 8 ? "_8b" : (NBITS == 7 ? "_7b" : "_4b")));
}

// tile variant: SoA pairs are instantiated for two geometries only (256x16 for small inputs, 1024x16 otherwise)
template <typename Buf>
int buf_variant(const adlhip_device* d, size_t n)
{
    const int v = effective_variant(d, Buf::kElemBytes, n);
    if (Buf::kSoa) return v == 0 ? 0 : 2;
    return v;
}
template <typename Buf>
uint32_t buf_tile(const adlhip_device* d, size_t n)
{
    const TileVariant v = kVariants[buf_variant<Buf>(d, n)];
    return (uint32_t)(v.nt * v.k);
}

#define ADLHIP_DISPATCH_TILE(FN, Buf, NBITS, ...)                                                     \
    if constexpr (NBITS == 7) {   /* 7-bit digits: the default tiles only */                          \
        if (Buf::kElemBytes == 4 && !Buf::kSoa)                                                       \
            return d->rank_mode ? FN<Buf, NBITS, 512, 32, 1>(__VA_ARGS__) : FN<Buf, NBITS, 512, 32, 0>(__VA_ARGS__); \
        return d->rank_mode ? FN<Buf, NBITS, 1024, 16, 1>(__VA_ARGS__) : FN<Buf, NBITS, 1024, 16, 0>(__VA_ARGS__);   \
    }                                                                                                 \
    if (Buf::kSoa) {                                                                                  \
        switch (buf_variant<Buf>(d, n) * 2 + (d->rank_mode ? 1 : 0)) {                                \
        case 0: return FN<Buf, NBITS, 256, 16, 0>(__VA_ARGS__);                                       \
        case 1: return FN<Buf, NBITS, 256, 16, 1>(__VA_ARGS__);                                       \
        case 4: return FN<Buf, NBITS, 1024, 16, 0>(__VA_ARGS__);                                      \
        default: return FN<Buf, NBITS, 1024, 16, 1>(__VA_ARGS__);                                     \
        }                                                                                             \
    }                                                                                                 \
    switch (buf_variant<Buf>(d, n) * 2 + (d->rank_mode ? 1 : 0)) {                                    \
    case 0: return FN<Buf, NBITS, 256, 16, 0>(__VA_ARGS__);                                           \
    case 1: return FN<Buf, NBITS, 256, 16, 1>(__VA_ARGS__);                                           \
    case 2: return FN<Buf, NBITS, 512, 16, 0>(__VA_ARGS__);                                           \
    case 3: return FN<Buf, NBITS, 512, 16, 1>(__VA_ARGS__);                                           \
    case 4: return FN<Buf, NBITS, 1024, 16, 0>(__VA_ARGS__);                                          \
    case 5: return FN<Buf, NBITS, 1024, 16, 1>(__VA_ARGS__);                                          \
    case 6: return FN<Buf, NBITS, 512, 8, 0>(__VA_ARGS__);                                            \
    case 7: return FN<Buf, NBITS, 512, 8, 1>(__VA_ARGS__);                                            \
    case 8: return FN<Buf, NBITS, 1024, 8, 0>(__VA_ARGS__);                                           \
    case 9: return FN<Buf, NBITS, 1024, 8, 1>(__VA_ARGS__);                                           \
    case 10: return FN<Buf, NBITS, 256, 32, 0>(__VA_ARGS__);                                          \
    case 11: return FN<Buf, NBITS, 256, 32, 1>(__VA_ARGS__);                                          \
    case 12: return FN<Buf, NBITS, 512, 32, 0>(__VA_ARGS__);                                          \
    case 13: return FN<Buf, NBITS, 512, 32, 1>(__VA_ARGS__);                                          \
    default: return fail("bad tile variant %d", d->tile_variant);                                     \
    }

// ---- three-kernel pass: count -> table scan -> sort+scatter --------------------------------------

template <typename Buf, int NBITS, int NT, int K, int RANK>
int launch_scatter(adlhip_device* d, const Buf& src, const Buf& dst, const uint32_t* table, const uint32_t* totals,
                   size_t n, const Geometry& g, int start_bit)
{
    if (Buf::kSoa && !((NT == 256 && K == 16) || (NT == 1024 && K == 16))) return fail("internal: SoA tile");
    typedef typename Buf::IO IO;
    using C = adlhip::TileCfg<typename IO::elem_t, NBITS, NT, K>;
    auto kern = adlhip::radix_scatter_kernel<IO, NBITS, NT, K, RANK>;
    if (ensure_lds(kern, C::LDS_BYTES)) return ADLHIP_FAILURE;
    const IO io = Buf::io(src, dst);
    return launch(d, kernel_name<Buf, NBITS>("scatter"), [&] {
        hipLaunchKernelGGL(kern, dim3(g.n_wgs), dim3(NT), C::LDS_BYTES, d->stream, io, table, totals, (uint32_t)n,
                           (int)g.n_wgs, start_bit, g.tiles_per_wg, g.num_tiles);
    });
}

template <typename Buf, int NBITS>
int dispatch_scatter(adlhip_device* d, const Buf& src, const Buf& dst, const uint32_t* table, const uint32_t* totals,
                     size_t n, const Geometry& g, int start_bit)
{
    ADLHIP_DISPATCH_TILE(launch_scatter, Buf, NBITS, d, src, dst, table, totals, n, g, start_bit)
}

// three-kernel pass with at most this many workgroups: the count kernel leaves its raw counts workgroup-major and the
// scatter kernel sums its column itself (16 or 64 independent coalesced loads per thread), so the table-scan launch is
// dropped -- 12 -> 8 dependent launches per 32-bit sort.  Same-session A/B (ADLHIP_FUSED_SCAN_MAX = 0 / 16 / 64):
// 32 Ki keys 35.8 -> 26.5 us; 64 Ki + 1 .. 128 Ki keys 38.3 -> 36.0 / 37.2 -> 34.5 us; 200000 .. 256 Ki keys level.
constexpr uint32_t kScanInScatterMaxWgs = 64;

template <typename Buf, int NBITS>
int three_kernel_pass(adlhip_device* d, const Buf& src, const Buf& dst, void* work, size_t n, int start_bit,
                      bool need_totals = false)
{
    typedef typename Buf::key_t key_t;
    constexpr int kCountNT = 256;
    const Geometry g = geometry(d, n, buf_tile<Buf>(d, n));
    uint32_t* table = reinterpret_cast<uint32_t*>(work);
    uint32_t* totals = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(work) + table_bytes(d, n, kMinTile));
    const uint32_t elems_per_wg = g.tiles_per_wg * g.tile;
    static const uint32_t fused_max = getenv("ADLHIP_FUSED_SCAN_MAX") ? (uint32_t)atoi(getenv("ADLHIP_FUSED_SCAN_MAX")) : kScanInScatterMaxWgs;
    const bool fused = !need_totals && g.n_wgs <= std::min<uint32_t>(fused_max, 64u);
    int rc = launch(d, kernel_name<Buf, NBITS>("count"), [&] {
        hipLaunchKernelGGL((adlhip::radix_count_kernel<key_t, NBITS, kCountNT>), dim3(g.n_wgs), dim3(kCountNT), 0, d->stream,
                           src.keys(), table, (uint32_t)n, (int)g.n_wgs, start_bit, elems_per_wg, fused ? 1 : 0);
    });
    if (rc) return rc;
    if (fused) return dispatch_scatter<Buf, NBITS>(d, src, dst, table, nullptr, n, g, start_bit);
    rc = launch(d, "scan_table", [&] {
        hipLaunchKernelGGL((adlhip::radix_scan_table_kernel<256>), dim3(1 << NBITS), dim3(256), 0, d->stream, table,
                           totals, (int)g.n_wgs);
    });
    if (rc) return rc;
    return dispatch_scatter<Buf, NBITS>(d, src, dst, table, totals, n, g, start_bit);
}

// ---- onesweep --------------------------------------------------------------------------------

// work buffer layout (onesweep):
//   [ctrl   : per pass 16 chain tickets, one 128-byte line each (2 KiB); 16 passes = 32 KiB]
//   [tables : 16 x PassTable]
//   [joint  : joint histograms of all passes (<= 16 passes x 16 chains x 256 bins)]
//   [partial: hist_wgs x total_bins u32]
//   [status : passes x status_rows x 256 u32]
struct OnesweepLayout {
    size_t off_ctrl, off_tables, off_joint, off_part, off_status, total;
    uint32_t hist_wgs;
};

constexpr uint32_t kMaxJointBins = 8u * 16u * 256u;   // 64-bit keys, eight 8-bit passes (16 x 4-bit passes need less)

uint32_t hist_wgs_for(const adlhip_device* d, size_t n)
{
    const uint32_t cap = (uint32_t)d->prop.multiProcessorCount;   // one 1024-thread workgroup per CU (2 or 4 measured slower)
    uint32_t w = (uint32_t)((n + adlhip::kHistChunk - 1) / adlhip::kHistChunk);
    if (w > cap) w = cap;
    return w ? w : 1;
}

// status rows of one pass: one per tile; every chain may add one partial tile
size_t status_rows(size_t n, uint32_t tile) { return (n + tile - 1) / tile + adlhip::kChains + 1; }

// most passes a sort of `key_bits`-bit keys can need with the current digit width (8,8,8,4 style plans included)
int max_passes_for(const adlhip_device* d, int key_bits)
{
    if (d->digit_bits == 7) return key_bits / 7 + 1;
    return d->digit_bits == 8 ? (key_bits + 7) / 8 : key_bits / 4;
}

OnesweepLayout onesweep_layout(const adlhip_device* d, size_t n, int max_passes, uint32_t tile)
{
    OnesweepLayout L;
    L.hist_wgs = hist_wgs_for(d, n);
    L.off_ctrl = 0;
    L.off_tables = (size_t)adlhip::kTicketVecs * 16;
    L.off_joint = L.off_tables + sizeof(adlhip::PassTable) * adlhip::kMaxPasses;
    L.off_part = align_up(L.off_joint + (size_t)kMaxJointBins * 4, 256);
    L.off_status = align_up(L.off_part + (size_t)L.hist_wgs * kMaxJointBins * 4, 256);
    L.total = L.off_status + (size_t)max_passes * status_rows(n, tile) * 256 * 4;
    return L;
}

template <typename Buf, int NBITS, int NT, int K, int RANK>
int launch_onesweep(adlhip_device* d, const Buf& src, const Buf& dst, const adlhip::PassTable* table, uint32_t* status,
                    uint32_t* tickets, size_t n, int start_bit, const uint32_t* dyn_start_bit = nullptr)
{
    if (Buf::kSoa && !((NT == 256 && K == 16) || (NT == 1024 && K == 16))) return fail("internal: SoA tile");
    typedef typename Buf::IO IO;
    using C = adlhip::TileCfg<typename IO::elem_t, NBITS, NT, K>;
    auto kern = adlhip::onesweep_chain_kernel<IO, NBITS, NT, K, RANK>;
    if (ensure_lds(kern, C::LDS_BYTES)) return ADLHIP_FAILURE;
    const size_t rows = status_rows(n, (uint32_t)C::TILE);
    const uint32_t grid = (uint32_t)((n + C::TILE - 1) / C::TILE) + adlhip::kChains;   // upper bound on the tile count
    const uint32_t status_bytes = (uint32_t)(rows * C::BINS * 4u);
    const IO io = Buf::io(src, dst);
    return launch(d, kernel_name<Buf, NBITS>("onesweep"), [&] {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), C::LDS_BYTES, d->stream, io, table, status, status_bytes, tickets,
                           d->d_fault, (uint32_t)n, start_bit, dyn_start_bit);
    });
}

template <typename Buf, int NBITS>
int dispatch_onesweep(adlhip_device* d, const Buf& src, const Buf& dst, const adlhip::PassTable* table, uint32_t* status,
                      uint32_t* tickets, size_t n, int start_bit)
{
    ADLHIP_DISPATCH_TILE(launch_onesweep, Buf, NBITS, d, src, dst, table, status, tickets, n, start_bit)
}

// copy_back = false / totals_at != nullptr: the one-pass partition (partition_top_byte): the result stays in `tmp`, and the digit
// totals of pass 0 (its joint histogram folded over the 16 chains) are left in the work buffer at *totals_at
template <typename Buf>
int onesweep_sort(adlhip_device* d, Buf data, Buf tmp, void* work, size_t n, const std::vector<PassPlan>& plan, bool copy_back = true,
                  uint32_t** totals_at = nullptr)
{
    typedef typename Buf::key_t key_t;
    const int P = (int)plan.size();
    const uint32_t tile = buf_tile<Buf>(d, n);
    const OnesweepLayout L = onesweep_layout(d, n, max_passes_for(d, Buf::kElemBytes == 4 ? 32 : 64), tile);
    char* wb = reinterpret_cast<char*>(work);
    uint32_t* ctrl = reinterpret_cast<uint32_t*>(wb + L.off_ctrl);
    adlhip::PassTable* tables = reinterpret_cast<adlhip::PassTable*>(wb + L.off_tables);
    uint32_t* joint = reinterpret_cast<uint32_t*>(wb + L.off_joint);
    uint32_t* part = reinterpret_cast<uint32_t*>(wb + L.off_part);
    uint32_t* status = reinterpret_cast<uint32_t*>(wb + L.off_status);
    const size_t rows = status_rows(n, tile);

    // the tickets and every status word of the passes we run (status is contiguous) are zeroed by the histogram kernel
    const size_t status_vecs = (size_t)P * rows * 256 * 4 / 16;

    adlhip::PassDesc desc;
    desc.num_passes = P;
    uint32_t total_bins = 0;
    for (int i = 0; i < P; ++i) {
        desc.start_bit[i] = (uint8_t)plan[i].start_bit;
        desc.nbits[i] = (uint8_t)plan[i].nbits;
        total_bins += adlhip::joint_bins(plan[i].nbits);
    }
    if (total_bins > kMaxJointBins) return fail("internal: joint histogram of %u bins exceeds %u", total_bins, kMaxJointBins);
    // histogram workgroups nest inside the 16 input slices that are pass 0's chains
    const uint32_t per_wg = (uint32_t)align_up((n + L.hist_wgs - 1) / L.hist_wgs, 1024);
    const uint32_t wgs = (uint32_t)((n + per_wg - 1) / per_wg);
    const uint32_t wgs_per_slice = (wgs + adlhip::kChains - 1) / adlhip::kChains;
    const uint32_t slice0 = per_wg * wgs_per_slice;
    const size_t hist_lds = (size_t)total_bins * 4;
    int rc = ADLHIP_FAILURE;
    const key_t* hist_src = data.keys();
    auto run_hist = [&](auto kern) -> int {
        if (ensure_lds(kern, hist_lds)) return ADLHIP_FAILURE;
        return launch(d, sizeof(key_t) == 4 ? "os_hist_u32" : "os_hist_e64", [&] {
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(adlhip::kHistNT), hist_lds, d->stream, hist_src, part, (uint32_t)n, per_wg,
                               slice0, desc, total_bins, reinterpret_cast<adlhip::u32x4*>(ctrl),
                               reinterpret_cast<adlhip::u32x4*>(status), status_vecs, d->d_fault);
        });
    };
    switch (P) {   // pass count is a template parameter of the histogram kernel (descriptors stay in SGPRs)
#define ADLHIP_HIST_CASE(N) case N: rc = run_hist(adlhip::onesweep_hist_kernel<key_t, N>); break;
        ADLHIP_HIST_CASE(1) ADLHIP_HIST_CASE(2) ADLHIP_HIST_CASE(3) ADLHIP_HIST_CASE(4) ADLHIP_HIST_CASE(5)
        ADLHIP_HIST_CASE(6) ADLHIP_HIST_CASE(7) ADLHIP_HIST_CASE(8) ADLHIP_HIST_CASE(9) ADLHIP_HIST_CASE(10)
        ADLHIP_HIST_CASE(11) ADLHIP_HIST_CASE(12) ADLHIP_HIST_CASE(13) ADLHIP_HIST_CASE(14) ADLHIP_HIST_CASE(15)
        ADLHIP_HIST_CASE(16)
#undef ADLHIP_HIST_CASE
        default: return fail("internal: %d passes", P);
    }
    if (rc) return rc;
    rc = launch(d, "os_hist_reduce", [&] {
        hipLaunchKernelGGL(adlhip::onesweep_hist_reduce_kernel, dim3((total_bins + 255) / 256), dim3(1024), 0, d->stream,
                           part, joint, wgs, total_bins);
    });
    if (rc) return rc;
    rc = launch(d, "os_tables", [&] {
        hipLaunchKernelGGL(adlhip::onesweep_tables_kernel, dim3(P), dim3(256), 0, d->stream, joint, tables, desc, tile);
    });
    if (rc) return rc;

    Buf src = data;
    Buf dst = tmp;
    for (int i = 0; i < P; ++i) {
        uint32_t* st = status + (size_t)i * rows * 256;
        uint32_t* tk = ctrl + i * adlhip::kChains * adlhip::kTicketStride;
        rc = (plan[i].nbits == 8)   ? dispatch_onesweep<Buf, 8>(d, src, dst, tables + i, st, tk, n, plan[i].start_bit)
             : (plan[i].nbits == 7) ? dispatch_onesweep<Buf, 7>(d, src, dst, tables + i, st, tk, n, plan[i].start_bit)
                                    : dispatch_onesweep<Buf, 4>(d, src, dst, tables + i, st, tk, n, plan[i].start_bit);
        if (rc) return rc;
        std::swap(src, dst);
    }
    if (totals_at) {   // `part` is free once the reduce kernel has run
        rc = launch(d, "os_fold_totals", [&] {
            hipLaunchKernelGGL(adlhip::fold_joint_kernel, dim3(1), dim3(256), 0, d->stream, (const uint32_t*)joint, part);
        });
        if (rc) return rc;
        *totals_at = part;
    }
    if (copy_back && !src.same(data)) return data.copy_from(src, n, d->stream);
    return ADLHIP_SUCCESS;
}

template <typename Buf>
int three_kernel_sort(adlhip_device* d, Buf data, Buf tmp, void* work, size_t n, const std::vector<PassPlan>& plan)
{
    Buf src = data;
    Buf dst = tmp;
    for (const PassPlan& p : plan) {
        int rc = (p.nbits == 8) ? three_kernel_pass<Buf, 8>(d, src, dst, work, n, p.start_bit)
                                : three_kernel_pass<Buf, 4>(d, src, dst, work, n, p.start_bit);
        if (rc) return rc;
        std::swap(src, dst);   // Pprims.cpp:397
    }
    // odd number of passes: result sits in the scratch buffer -> copy back (Pprims.cpp:400-403)
    if (!src.same(data)) return data.copy_from(src, n, d->stream);
    return ADLHIP_SUCCESS;
}

// ---- small inputs: the whole sort in one workgroup ---------------------------------------------------
constexpr size_t kSmallMax = 16384;

template <typename E, int NT, int K, int RANK>
int launch_small(adlhip_device* d, E* data, size_t n, const adlhip::SmallPlan& sp)
{
    auto kern = adlhip::small_sort_kernel<E, NT, K, RANK>;
    const size_t lds = sizeof(E) * NT * K + (size_t)(NT / 64) * 256 * 4 + 128;
    if (ensure_lds(kern, lds)) return ADLHIP_FAILURE;
    return launch(d, sizeof(E) == 4 ? "small_sort_u32" : "small_sort_e64", [&] {
        hipLaunchKernelGGL(kern, dim3(1), dim3(NT), lds, d->stream, data, (uint32_t)n, sp);
    });
}

template <typename E>
int small_sort(adlhip_device* d, E* data, size_t n, const std::vector<PassPlan>& plan)
{
    adlhip::SmallPlan sp;
    sp.num_passes = (int)plan.size();
    for (size_t i = 0; i < plan.size(); ++i) {
        sp.start_bit[i] = (uint8_t)plan[i].start_bit;
        sp.nbits[i] = (uint8_t)plan[i].nbits;
    }
    if (n <= 4096) return d->rank_mode ? launch_small<E, 256, 16, 1>(d, data, n, sp) : launch_small<E, 256, 16, 0>(d, data, n, sp);
    return d->rank_mode ? launch_small<E, 1024, 16, 1>(d, data, n, sp) : launch_small<E, 1024, 16, 0>(d, data, n, sp);
}

// ---- segments finished in LDS (adlhip_segment_sort; pass 3 of the mid-size sort) -----------------------------------
template <typename E, int NT, int K, int LBITS>
int launch_segment_sort(adlhip_device* d, const E* in, E* out, const uint32_t* seg_start, size_t num_segments, int low_bits,
                        const uint32_t* dyn, const adlhip::MidCoop& coop = adlhip::MidCoop{nullptr, nullptr, nullptr, 0u},
                        const adlhip::SegSlab& slab = adlhip::SegSlab{nullptr, 0u, nullptr, nullptr})
{
    auto kern = adlhip::segment_sort_kernel<E, NT, K, LBITS>;
    const size_t own = sizeof(E) * NT * K + (size_t)(NT / 64) * (1u << LBITS) * 6 + 64;
    const size_t lds = std::max(own, (size_t)adlhip::TileCfg<E, 8, NT, K>::LDS_BYTES);   // the through-memory path's carve
    if (ensure_lds(kern, lds)) return ADLHIP_FAILURE;
    // as many workgroups as are resident at once (LDS- or wave-limited), each loops over segments
    const size_t per_cu = std::max<size_t>(1, std::min<size_t>((size_t)160 * 1024 / lds, (size_t)32 / (NT / 64)));
    // slab form (slab.state != nullptr): exactly one workgroup per bucket -- the kernel reads its bucket's count and offset once;
    // workgroups that are not resident yet simply follow (mid_eligible keeps the form off devices that hold fewer than 256)
    const uint32_t grid = slab.state ? (uint32_t)num_segments
                                     : (uint32_t)std::min<size_t>(num_segments, per_cu * (size_t)d->prop.multiProcessorCount);
    return launch(d, sizeof(E) == 4 ? "segment_sort_u32" : "segment_sort_e64", [&] {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, d->stream, in, out, seg_start, (uint32_t)num_segments, (uint32_t)low_bits,
                           dyn, d->d_fault, coop, slab);
    });
}

template <typename E, int K, typename S = E, bool SOA = false, int RANK = 1>
int launch_wave_segment_sort(adlhip_device* d, const E* in, E* out, const uint32_t* seg_start, size_t num_segments, int low_bits,
                             const uint32_t* seg_cnt = nullptr, uint32_t in_stride = 0, const uint32_t* gate = nullptr,
                             const uint32_t* dyn_low_bits = nullptr, uint32_t* out_vals = nullptr, const uint32_t* list = nullptr,
                             const uint32_t* list_cnt = nullptr, uint32_t seg_shift = 8)
{
    // a wave's LDS: its tile + 256 counters; waves per workgroup so that a workgroup takes at most ~48 KiB (three per CU)
    constexpr size_t per_wave = sizeof(E) * 64 * K + 256 * 4;
    constexpr int WAVES = per_wave <= 6144 ? 8   /* eight waves with the 7 KiB of the 1536-element tile measured slower */ : per_wave <= 12288 ? 4 : per_wave <= 24576 ? 2 : 1;
    constexpr int STEP = K <= 40 ? 2 : 4;        // row-count bodies: every 2 rows, every 4 for the largest tile
    constexpr int RMIN = K <= 24 ? 2 : K / 2;    // the tiles beyond 24 rows exist for segments that need them
    const size_t lds = (size_t)WAVES * per_wave;
    if (list) {
        if constexpr (!SOA) {
            auto kern = adlhip::wave_segment_sort_kernel<E, K, WAVES, STEP, RMIN, S, SOA, true>;
            if (ensure_lds(kern, lds)) return ADLHIP_FAILURE;
            // list form (the segments the binning finish handed over -- usually none): a small grid that takes them in turns
            const size_t slots = std::min<size_t>(num_segments, (size_t)4 * d->prop.multiProcessorCount);
            const uint32_t grid = (uint32_t)((slots + WAVES - 1) / WAVES);
            return launch(d, sizeof(E) == 4 ? "segment_sort_listed_u32" : "segment_sort_listed_e64", [&] {
                hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, d->stream, in, out, seg_start, (uint32_t)num_segments,
                                   (uint32_t)low_bits, d->d_fault, seg_cnt, in_stride, gate, dyn_low_bits, out_vals, list, list_cnt, seg_shift);
            });
        } else {
            return fail("internal: no list form for SoA");
        }
    }
    auto kern = adlhip::wave_segment_sort_kernel<E, K, WAVES, STEP, RMIN, S, SOA, false, RANK>;
    if (ensure_lds(kern, lds)) return ADLHIP_FAILURE;
    const uint32_t grid = (uint32_t)((num_segments + WAVES - 1) / WAVES);
    return launch(d, sizeof(E) == 4 ? "segment_sort_wave_u32" : "segment_sort_wave_e64", [&] {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, d->stream, in, out, seg_start, (uint32_t)num_segments,
                           (uint32_t)low_bits, d->d_fault, seg_cnt, in_stride, gate, dyn_low_bits, out_vals, list, list_cnt, seg_shift);
    });
}

// the binning finish of the large keys-only sort (hybrid_kernels.hpp bin_segment_sort_kernel): one workgroup per segment slab
template <typename E, typename S, int NT, int K, int BITS>
int launch_bin_segment_sort(adlhip_device* d, const S* in, E* out, const uint32_t* seg_off, const uint32_t* seg_cnt, uint32_t in_stride,
                            uint32_t* mode, uint32_t* hard_list, uint32_t seg_shift = 8)
{
    auto kern = adlhip::bin_segment_sort_kernel<E, S, NT, K, BITS>;
    const size_t lds = align_up(sizeof(S) * NT * K, 16) + ((size_t)(1 << BITS) / 2) * 4 + (2 * NT / 64 + 4) * 4;
    if (ensure_lds(kern, lds)) return ADLHIP_FAILURE;
    // one workgroup per segment.  (The kernel can loop -- a grid of resident workgroups that take segments in turns and request the
    // next one's keys early -- but measured slower: 256 Mi u64 keys 1.02 vs 0.89 ms, 64 Mi 0.33 vs 0.26; a workgroup that leaves
    // hands its LDS to the next one while its stores drain.)
    const uint32_t grid = 256u << seg_shift;
    return launch(d, sizeof(E) == 4 ? "segment_sort_bin_u32" : "segment_sort_bin_u64", [&] {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, d->stream, in, out, seg_off, seg_cnt, in_stride, grid, (const uint32_t*)mode,
                           mode + adlhip::kDynHardCnt, hard_list, d->d_fault, seg_shift);
    });
}

// the workgroup-per-segment finish of the large sort beyond 280 Mi u32 keys (hybrid_kernels.hpp wg_segment_sort_kernel)
template <typename E, typename S, int NT, int K>
int launch_wg_segment_sort(adlhip_device* d, const S* in, E* out, const uint32_t* seg_off, const uint32_t* seg_cnt, uint32_t in_stride,
                           const uint32_t* mode, size_t slots)
{
    auto kern = adlhip::wg_segment_sort_kernel<E, S, NT, K>;
    const size_t lds = sizeof(S) * NT * K + (size_t)(NT / 64) * 256 * 6;
    if (ensure_lds(kern, lds)) return ADLHIP_FAILURE;
    return launch(d, "segment_sort_wg_u32", [&] {
        hipLaunchKernelGGL(kern, dim3((uint32_t)slots), dim3(NT), lds, d->stream, in, out, seg_off, seg_cnt, in_stride, mode,
                           mode + adlhip::kDynLowBits, d->d_fault);
    });
}

// largest segment the finishing kernel takes for this element size and number of low bits
size_t segment_capacity(size_t elem_bytes, int low_bits)
{
    if (elem_bytes == 4) return low_bits <= 24 ? 16384 : 8192;
    return low_bits <= 24 ? 8192 : 4096;
}

template <typename E>
int segment_sort(adlhip_device* d, const E* in, E* out, const uint32_t* seg_start, size_t num_segments, size_t max_segment,
                 int low_bits, const uint32_t* dyn)
{
    // small segments, at most 16 bits: one wave per segment
    static const bool no_wave = getenv("ADLHIP_SEGSORT_NO_WAVE") != nullptr;
    if (!no_wave && low_bits <= 16 && dyn == nullptr && max_segment <= 1280)
        return launch_wave_segment_sort<E, 20>(d, in, out, seg_start, num_segments, low_bits);
    // tile by the caller's bound on the segment size; the digit width of the local passes by what fits beside the tile
    if (sizeof(E) == 4) {
        if (max_segment <= 4096) return launch_segment_sort<E, 256, 16, 9>(d, in, out, seg_start, num_segments, low_bits, dyn);
        if (max_segment <= 8192) return launch_segment_sort<E, 512, 16, 9>(d, in, out, seg_start, num_segments, low_bits, dyn);
        if (max_segment <= 16384 && low_bits <= 24) return launch_segment_sort<E, 512, 32, 8>(d, in, out, seg_start, num_segments, low_bits, dyn);
    } else {
        if (max_segment <= 4096) return launch_segment_sort<E, 256, 16, 9>(d, in, out, seg_start, num_segments, low_bits, dyn);
        if (max_segment <= 8192 && low_bits <= 24) return launch_segment_sort<E, 512, 16, 8>(d, in, out, seg_start, num_segments, low_bits, dyn);
    }
    return fail("segment sort: segments of up to %zu elements with %d key bits exceed the LDS tile (%zu)", max_segment, low_bits,
                segment_capacity(sizeof(E), low_bits));
}

// ---- mid-size sort: byte histograms + tables -> one MSD pass -> buckets finished in LDS (hybrid_kernels.hpp) ----------
constexpr size_t kMidMaxU32 = size_t(2) << 20;
constexpr size_t kMidMaxE64 = size_t(1) << 20;
constexpr uint32_t kMidTile = 4096;   // pass 2 runs on 256 x 16 tiles

struct MidLayout {
    size_t off_tickets, off_table, off_seg, off_dyn, off_status, off_coop, off_slab, total;
    uint32_t wgs, rows, coop_wgs, stride;
};

MidLayout mid_layout(size_t n)
{
    MidLayout L;
    L.wgs = (uint32_t)((n + adlhip::kMidChunk - 1) / adlhip::kMidChunk);
    L.rows = (uint32_t)((n + kMidTile - 1) / kMidTile) + adlhip::kChains + 1;
    L.off_tickets = 0;
    L.off_table = (size_t)adlhip::kChains * adlhip::kTicketStride * 4;
    L.off_seg = L.off_table + sizeof(adlhip::PassTable);
    L.off_dyn = align_up(L.off_seg + 257 * 4, 16);
    L.off_status = align_up(L.off_dyn + sizeof(adlhip::MidDyn), 256);
    // cooperative LSD kernel (safety net for skewed keys): bucket-major table [256][wgs] + 256 totals
    L.coop_wgs = 256;   // = the grid of pass 3 (one workgroup per bucket)
    L.off_coop = align_up(L.off_status + (size_t)L.rows * 256 * 4, 256);
    // keys-only form: 256 bucket slabs of `stride` elements (= the LDS tile of pass 2: at least twice the mean bucket)
    L.stride = n <= (size_t(512) << 10) ? 4096u : (n <= (size_t(1) << 20) ? 8192u : 16384u);
    L.off_slab = align_up(L.off_coop + (size_t)256 * L.coop_wgs * 4 + 256 * 4, 256);
    L.total = L.off_slab + (size_t)256 * L.stride * 4;
    return L;
}

// u32 keys take the two-launch mid-size sort from here -- below the one-workgroup sort's limit kSmallMax: 10000 keys 16.5 -> 15.6 us,
// 12288 17.9 -> 15.5, 16384 19.3 -> 15.4; from 4096 it loses (6000 keys 13.6 -> 15.1).  profiles/r3_small_mid_ab.txt; A/B: ADLHIP_MID_MIN
size_t mid_min_u32()
{
    static const size_t v = getenv("ADLHIP_MID_MIN") ? (size_t)atoll(getenv("ADLHIP_MID_MIN")) : 8192;
    return v;
}

bool mid_eligible(const adlhip_device* d, size_t elem_bytes, size_t n, int sort_bits, int max_bits)
{
    if (d->resident_wgs < 256) return false;   // the finish's one workgroup per bucket doubles as a 256-workgroup cooperative sort
    return d->sort_algo < 0 && d->mid_path && max_bits == 32 && sort_bits == 32 && d->rank_mode == 1 && d->digit_bits == 8 &&
           d->tile_variant < 0 && n > (elem_bytes == 4 ? mid_min_u32() : kSmallMax) && n <= (elem_bytes == 4 ? kMidMaxU32 : kMidMaxE64);
}

// Which form an eligible sort takes: 2 = two launches (u32 keys only), 3 = three launches, 0 = the per-digit passes.
// Both mid-size forms fall back to a cooperative LSD sort when the keys do not fit their buckets -- correct, but 3-5x the cost
// of the per-digit passes (profiles/r2_mid_size_distributions_every_sort_through_mid_path.txt) -- and report what happened into
// pinned memory (hybrid_kernels.hpp: mid_prep_kernel, SegSlab::host_mode).  The host reads the latest reports WITHOUT
// synchronising (they may be a sort or two old) and steers by them:
//   two-launch form overflowed, top byte constant  -> this handle's next 64 eligible sorts take the three-launch form
//                                                     (it picks the byte that varies); 512, 4096 after repeats
//   two-launch form overflowed, several buckets    -> ... and the next 32 (256, 2048, 4096) take the per-digit passes
//   three-launch form fell back                    -> the next 32 (256, ...) take the per-digit passes
// A success resets the respective count.  Only speed depends on any of this; "sort.mid" = 2 / 3 force a form, 0 = off.
int choose_mid_form(adlhip_device* d, bool keys_only)
{
    if (d->mid_path == 2) return keys_only ? 2 : 3;
    if (d->mid_path == 3) return 3;
    if (keys_only) {
        if (d->mid2_skip > 0) {
            --d->mid2_skip;
        } else {
            const uint32_t report = d->h_fault[10];
            d->h_fault[10] = 0u;
            if (report >= 2u) {
                d->mid2_skip = d->mid2_backoff;
                d->mid2_backoff = std::min(d->mid2_backoff * 8, 4096);
                if (report == 2u) {
                    d->mid_skip = std::max(d->mid_skip, d->mid_backoff);
                    d->mid_backoff = std::min(d->mid_backoff * 8, 4096);
                }
            } else {
                if (report == 1u) d->mid2_backoff = 64;
                return 2;
            }
        }
    }
    if (d->mid_skip > 0) {
        --d->mid_skip;
        return 0;
    }
    const uint32_t report = d->h_fault[9];
    d->h_fault[9] = 0u;
    if (report == 2u) {
        d->mid_skip = d->mid_backoff;
        d->mid_backoff = std::min(d->mid_backoff * 8, 4096);
        return 0;
    }
    if (report == 1u) d->mid_backoff = 32;
    return 3;
}

template <typename E>
int mid_sort(adlhip_device* d, E* data, E* tmp, void* work, size_t n)
{
    const MidLayout L = mid_layout(n);
    char* wb = reinterpret_cast<char*>(work);
    uint32_t* tickets = reinterpret_cast<uint32_t*>(wb + L.off_tickets);
    adlhip::PassTable* table = reinterpret_cast<adlhip::PassTable*>(wb + L.off_table);
    uint32_t* seg_start = reinterpret_cast<uint32_t*>(wb + L.off_seg);
    adlhip::MidDyn* dyn = reinterpret_cast<adlhip::MidDyn*>(wb + L.off_dyn);
    uint32_t* status = reinterpret_cast<uint32_t*>(wb + L.off_status);
    const uint32_t status_vecs = (uint32_t)((size_t)L.rows * 256 * 4 / 16);
    // buckets average n / 256 elements; the LDS tile of pass 3 holds at least twice that, and an input whose largest
    // bucket still does not fit goes to the cooperative LSD kernel instead (decided on the device by mid_prep_kernel)
    const size_t cap = std::min(std::max<size_t>(4096, 2 * ((n + 255) / 256)), segment_capacity(sizeof(E), 24));
    int rc = launch(d, sizeof(E) == 4 ? "mid_prep_u32" : "mid_prep_e64", [&] {
        hipLaunchKernelGGL((adlhip::mid_prep_kernel<E>), dim3(L.wgs), dim3(adlhip::kMidPrepNT), 0, d->stream, (const E*)data, (uint32_t)n,
                           kMidTile, d->d_mid_hist, table, seg_start, dyn, reinterpret_cast<adlhip::u32x4*>(tickets),
                           reinterpret_cast<adlhip::u32x4*>(status), status_vecs, d->d_fault + 12, d->d_fault, (uint32_t)cap,
                           d->h_fault + 9);
    });
    if (rc) return rc;
    rc = launch_onesweep<AosBuf<E>, 8, 256, 16, 1>(d, AosBuf<E>{data}, AosBuf<E>{tmp}, table, status, tickets, n, 24, &dyn->start_bit);
    if (rc) return rc;
    // pass 3: 8-bit local digits (the bytes below the MSD byte); its workgroups double as the cooperative LSD sort that takes
    // over when mid_prep_kernel found a bucket beyond `cap` (the grid is 256 workgroups either way)
    uint32_t* ctable = reinterpret_cast<uint32_t*>(wb + L.off_coop);
    const adlhip::MidCoop coop{ctable, ctable + (size_t)256 * 256, d->d_fault + 13, (uint32_t)n};
    const uint32_t* dd = &dyn->start_bit;
    if (sizeof(E) == 4) {
        if (cap <= 4096) return launch_segment_sort<E, 256, 16, 8>(d, tmp, data, seg_start, 256, 24, dd, coop);
        if (cap <= 8192) return launch_segment_sort<E, 512, 16, 8>(d, tmp, data, seg_start, 256, 24, dd, coop);
        return launch_segment_sort<E, 512, 32, 8>(d, tmp, data, seg_start, 256, 24, dd, coop);
    }
    if (cap <= 4096) return launch_segment_sort<E, 256, 16, 8>(d, tmp, data, seg_start, 256, 24, dd, coop);
    return launch_segment_sort<E, 512, 16, 8>(d, tmp, data, seg_start, 256, 24, dd, coop);
}

// Keys-only form (u32 keys: the pass on the top byte need not be stable, hybrid_kernels.hpp msd_bucket_scatter_kernel):
// TWO launches.  Its buckets are fixed to the top byte, so keys that share it (anything below 2^24) or are otherwise skewed
// overflow a slab; pass 2 then runs the cooperative LSD sort (correct, slow) and reports it (choose_mid_form).
int mid_sort_keys(adlhip_device* d, uint32_t* data, uint32_t* tmp, void* work, size_t n)
{
    typedef uint32_t E;
    const MidLayout L = mid_layout(n);
    char* wb = reinterpret_cast<char*>(work);
    E* slab = reinterpret_cast<E*>(wb + L.off_slab);
    uint32_t* state = d->d_mid_hist + 16 * 1024;
    using CA = adlhip::TileCfg<E, 8, 256, 16>;
    auto ka = adlhip::msd_bucket_scatter_kernel<E, 256, 16, 0>;
    if (ensure_lds(ka, CA::LDS_BYTES)) return ADLHIP_FAILURE;
    const uint32_t tiles = (uint32_t)((n + kMidTile - 1) / kMidTile);
    adlhip::BucketPass<E> pa;
    pa.src = data; pa.dst = slab; pa.cursors = state; pa.cursor_shift = 5; pa.src_count_shift = 0; pa.flag = state + 8192;
    pa.src_counts = nullptr; pa.n = (uint32_t)n; pa.src_stride = 0; pa.tiles_per_bucket = 1; pa.dst_stride = L.stride;
    pa.dst_total = 256u * L.stride; pa.start_bit = 24; pa.zero_me = state + 8194;
    pa.sample = nullptr; pa.which_digit = 0; pa.dst16 = 0;
    pa.place = nullptr; pa.status_a = nullptr; pa.pieces = 0; pa.rows_per_chain_a = 0; pa.slice = 0; pa.seg_shift = 8;
    int rc = launch(d, "mid_bucket_scatter_u32", [&] {
        hipLaunchKernelGGL(ka, dim3(tiles), dim3(256), CA::LDS_BYTES, d->stream, pa);
    });
    if (rc) return rc;
    uint32_t* ctable = reinterpret_cast<uint32_t*>(wb + L.off_coop);
    const adlhip::MidCoop coop{ctable, ctable + (size_t)256 * 256, state + 8194, (uint32_t)n};
    const adlhip::SegSlab sl{state, L.stride, d->h_fault + 10, tmp};
    if (L.stride <= 4096) return launch_segment_sort<E, 256, 16, 8>(d, slab, data, nullptr, 256, 24, nullptr, coop, sl);
    if (L.stride <= 8192) return launch_segment_sort<E, 512, 16, 8>(d, slab, data, nullptr, 256, 24, nullptr, coop, sl);
    return launch_segment_sort<E, 512, 32, 8>(d, slab, data, nullptr, 256, 24, nullptr, coop, sl);
}

// ---- large keys-only sort: two unstable MSD passes with bucket cursors + LDS finish (hybrid_kernels.hpp "sort.msd2") ------
// u32 keys and u64 keys (equal keys are indistinguishable, so the passes need not be stable); pairs keep the stable paths.
constexpr size_t kMsd2Min = kSmallMax;                             // elements; the path works from here ("sort.msd2" >= 2) ...
constexpr size_t kMsd2AutoMin = size_t(2) << 20;                   // ... and is chosen above the mid-size sort's range
                                                                   // (profiles/r2_msd2_size_curve.txt: 2.5 Mi keys 58 vs 70 us)
// up to 280 Mi keys a segment (mean n / 65536 = 4480, + 7.5 sd <= 5120) fits the 80 rows one wave holds; beyond, the finish takes one
// workgroup per segment (wg_segment_sort_kernel: tiles of 8192 ... 20480 keys) up to a mean of 17408 + 7.5 sd
constexpr size_t kMsd2MaxU32 = size_t(1088) << 20;
constexpr size_t kMsd2MaxU64 = (size_t(1) << 28) + (size_t(1) << 22);   // mean segment 4160
// the wave-per-segment finish's tiles: 64 * 20, 64 * 40, 64 * 80 elements

// from here the 16-bit second slab of whole u32 keys always fits the caller's n-element scratch array (1.5 x the mean + 72 per
// segment <= 2 x the mean); one fixed threshold keeps the work requirement piecewise monotone in n (sort_work_bytes' edges)
constexpr size_t kSlabInTmpMin = size_t(16) << 20;

struct Msd2Layout {
    size_t off_mode, off_cnt, off_off, off_hard, off_coop, off_slab_a, off_slab_b, total;
    uint32_t stride_a, stride_b, tier_b, tiles_per_bucket, seg_shift, slots;
    bool slab_b_in_tmp;   // the second slab fits the caller's n-element scratch array (u32 keys from ~40 Mi keys: 16-bit elements,
                          // 1.5 x the mean per segment = 0.75 n * 4 bytes); the safety net, which needs that array as its partner,
                          // only ever runs when the slabs' contents are void
};

// slab of a segment = the smallest of the finish's tiles that holds its mean + 7.5 standard deviations of a uniform key
// distribution (64 Mi keys: 1024 + 240 = 1264, slab 1536; a segment beyond its slab sends the sort to the safety net, it is never
// wrong; at 7.5 sd that is one sort in 10^8)
// slab and finish tile of the first tier: 64 * 24 elements.  With 64 * 20 = 1280 (mean 1024 at 64 Mi keys + 7.5 sd) keys whose
// density varied by 20 % over the key range already went to the safety net; 1536 takes ~45 % and costs nothing measurable
// (finish 118.0 vs 118.3 us at 64 Mi keys: five workgroups of four waves per CU instead of three of eight)
constexpr uint32_t kMsd2Stride0 = 1536;
// tile of the finish for n elements: 1280 / 1536 / 2560 / 5120 (what a wave -- or a workgroup of the binning finish -- holds),
// 8192 ... 20480 for the workgroup-per-segment finish
// fine: whole u32 keys with a 16-bit second slab -- from 2560 keys up their finish is the workgroup kernel, which has tiles of 3072
// and 4096 keys too (a half-empty 5120-key tile costs its 20 predicated rows all the same: 144 Mi keys 0.35 ms for the finish)
uint32_t msd2_tier_b(size_t n, uint32_t slots = 65536, bool fine = false)
{
    const size_t mean = (n + slots - 1) / slots;
    size_t sd = 1;
    while (sd * sd < mean) ++sd;
    const size_t need = mean + (15 * sd + 1) / 2;
    if (fine && need > 2560 && need <= 4096) return need <= 3072 ? 3072u : 4096u;
    // (the tiers' bounds on the mean stay as they were; up to a mean of ~640 the 1280-element tile already leaves 1.5 x)
    if (need <= 5120) return need <= 832 ? 1280u : need <= 1280 ? kMsd2Stride0 : need <= 2560 ? 2560u : 5120u;
    // (u32 keys beyond 280 Mi only: the workgroup-per-segment finish, 512 threads x 16 / 24 rows, 1024 x 16 / 20)
    return need <= 8192 ? 8192u : need <= 12288 ? 12288u : need <= 16384 ? 16384u : 20480u;
}
// elements between two segment slabs: the mean + 50 % (or + 7.5 sd where that is more), at most the finish's tile.  (Round 2 spaced
// the slabs by the tile whatever n was: 168 MB of second slab for 2 Mi keys.)
uint32_t msd2_stride_b(size_t n, uint32_t slots = 65536, bool fine = false, bool lean = false)
{
    const size_t mean = (n + slots - 1) / slots;
    size_t sd = 1;
    while (sd * sd < mean) ++sd;
    // lean (level 2 of adlhip_radix_sort_scratch_bytes_for, stable form): the mean + 7.5 sd only -- what evenly spread keys need
    const size_t want = align_up(std::max(lean ? mean : mean + mean / 2, mean + (15 * sd + 1) / 2) + 8, 64);
    return (uint32_t)std::min<size_t>(want, msd2_tier_b(n, slots, fine));
}

// Width w of the second digit of the cursor form (hybrid_kernels.hpp slot_to_segment): 8 from 12 Mi elements up; below, as many
// bits as leave segments of about a thousand elements -- 256 << w slots instead of 65536.  The finish then sorts 8 - w bits more
// (u32 keys: more than 16, so the second slab holds whole keys), which costs less than launching 65536 waves for a few dozen
// keys each (profiles/r3_narrow_second_digit.txt).
// Measured (cursor form, 65536 segments | narrow): u32 keys 2.5 Mi 71 | 51 us, 4 Mi 81 | 58, 8 Mi 104 | 91, 16 Mi 140 | 138, 24 Mi
// 177 | 199; u64 keys (binning finish on the ~1000-key segments) 1.5 Mi 117 | 47 us, 4 Mi 154 | 79, 16 Mi 289 | 221, 24 Mi 397 | 346.
// bin_finish: the segments go to the binning finish (whole u64 keys, cursor form), which likes them a thousand keys long at any n;
// the LSD finish pays for every extra bit with LDS passes, so there the narrow digit stops at 16 Mi elements.
uint32_t msd2_seg_shift(size_t n, bool bin_finish)
{
    static const long long env = getenv("ADLHIP_SEGSHIFT_MAXN") ? atoll(getenv("ADLHIP_SEGSHIFT_MAXN")) : -1;   // A/B: 0 = never
    const size_t max_n = env >= 0 ? (size_t)env : (bin_finish ? size_t(1) << 40 : size_t(16) << 20);
    if (n >= max_n) return 8u;
    uint32_t w = 2u;
    while (w < 8u && (n >> (8u + w)) > 1024u) ++w;
    return w;
}


// head-room of a first-pass bucket slab over the mean bucket, in per cent: 50 at full speed; the lean work size
// (adlhip_radix_sort_scratch_bytes_for, level 2) takes 12 -- 64 Mi u32 keys then need 306 MB of work instead of 451, and keys whose
// density varies by more than ~10 % over the key range are sorted by the safety net inside the sort
constexpr int kFullHeadroomPct = 50, kLeanHeadroomPct = 12;

// Workgroups of the kernels that host the safety net (512 threads, one tile of the one-sweep pass's size in LDS: two per CU).
// All of them must be resident at once -- the net's phases are separated by grid-wide barriers -- and the first 256 serve the
// 256 buckets in the kernel's ordinary role.
constexpr uint32_t kNetWgsMax = 512;
constexpr size_t kNetTableBytes = (size_t)256 * kNetWgsMax * 4 + 1024;   // [256][workgroups] digit table + 256 totals
uint32_t net_wgs(const adlhip_device* d) { return d->resident_wgs >= (int)kNetWgsMax ? kNetWgsMax : 256u; }
// two handle-owned counters: nets run, and of those, sorted by counting (net_sort; "stat.net_runs" / "stat.net_counting")
// d_msd2: 256 first-pass cursors (one line each) + 65536 second-pass cursors + 64 words (flag, done, barrier, sample words, the
// first kernel's repeat count and arrival word, net counters)
constexpr size_t kMsd2Words = 8192 + 65536 + 64;
uint32_t* net_stats(const adlhip_device* d) { return d->d_msd2 + 8192 + 65536 + 16; }

// Repeats the sort's first kernel must count among its samples (hybrid_kernels.hpp sample_accumulate: 16 waves, each comparing
// its own 128 samples) to call the keys too repetitive for the slabs: D distinct values, each with n / D copies, repeat
// 16 (128 - D (1 - e^(-128/D))) times; taken at D = n / 3072 (twice what a segment slab of ~1500 holds), with 20 % off for the
// sample's noise, never below 4 (keys without repeats show none: 3e-5 expected for random 32-bit keys).  At 64 Mi keys: 4, which
// 4096 values exceed eight times over and 30 000 values (2000 copies each: no fit either) reach every other time.
uint32_t sample_dup_threshold(size_t n)
{
    const double S = 128.0, D = std::max(1.0, (double)n / 3072.0);
    return (uint32_t)std::max(4.0, 0.8 * 16.0 * (S - D * (1.0 - std::exp(-S / D))));
}

// Scratch of the net's look-back passes (hybrid_kernels.hpp coop_onesweep_sort), carved out of `region` -- the first slab area, whose
// contents are void when the net runs.  tables == nullptr when it does not fit (small inputs) or the tile status words' 30-bit
// counts would not do (n >= 2^30): the net then runs its count-scan-scatter passes.
adlhip::OsNet net_onesweep_args(const adlhip_device* d, char* region, size_t region_bytes, size_t n, int passes, uint32_t tile)
{
    adlhip::OsNet os{};
    if (!region || n >= (size_t(1) << 30) || n < (size_t(1) << 16) || !d->net_lookback) return os;
    const uint32_t total_bins = (uint32_t)passes * ((uint32_t)adlhip::kChains << 8);
    uint32_t hw = (uint32_t)std::min<size_t>(net_wgs(d), (n + adlhip::kHistChunk - 1) / adlhip::kHistChunk);
    const uint32_t per_wg = (uint32_t)align_up((n + hw - 1) / hw, 1024);
    const uint32_t wgs = (uint32_t)((n + per_wg - 1) / per_wg);
    const uint32_t wgs_per_slice = (wgs + adlhip::kChains - 1) / adlhip::kChains;
    const size_t rows = status_rows(n, tile);
    const size_t off_tables = align_up((size_t)adlhip::kTicketVecs * 16, 256);
    const size_t off_joint = align_up(off_tables + sizeof(adlhip::PassTable) * adlhip::kMaxPasses, 256);
    const size_t off_part = align_up(off_joint + (size_t)total_bins * 4, 256);
    const size_t off_status = align_up(off_part + (size_t)wgs * total_bins * 4, 256);
    const size_t total = off_status + (size_t)passes * rows * 256 * 4;
    if (total > region_bytes || (size_t)passes * rows * 256 * 4 >= (size_t(1) << 31)) return os;
    os.ctrl = reinterpret_cast<adlhip::u32x4*>(region);
    os.tables = reinterpret_cast<adlhip::PassTable*>(region + off_tables);
    os.joint = reinterpret_cast<uint32_t*>(region + off_joint);
    os.part = reinterpret_cast<uint32_t*>(region + off_part);
    os.status = reinterpret_cast<uint32_t*>(region + off_status);
    os.rows = (uint32_t)rows;
    os.hist_wgs = wgs;
    os.per_wg = per_wg;
    os.slice0 = per_wg * wgs_per_slice;
    return os;
}

Msd2Layout msd2_layout(size_t n, size_t elem_bytes, int headroom_pct = kFullHeadroomPct)
{
    Msd2Layout L;
    const uint32_t tile = elem_bytes == 4 ? 16384u : 8192u;   // TileCfg<E, 8, 512, 32 | 16>
    // mean bucket + 50 % + 4096: the head-room of the segment slabs below (1536 for a mean of 1024), so that keys whose density
    // varies by up to ~45 % over the key range stay on this path (with + 3 % any mild skew went to the safety net)
    static const int env_pct = getenv("ADLHIP_SLAB_A_HEADROOM_PCT") ? atoi(getenv("ADLHIP_SLAB_A_HEADROOM_PCT")) : -1;   // A/B only
    if (headroom_pct == kFullHeadroomPct && env_pct >= 0) headroom_pct = env_pct;
    L.stride_a = (uint32_t)align_up(n / 256 + (n / 256) * (size_t)headroom_pct / 100 + 4096, 64);
    L.seg_shift = msd2_seg_shift(n, elem_bytes == 8);   // the cursor form sorts whole keys: 8-byte elements = u64 keys
    L.slots = 256u << L.seg_shift;
    const bool fine = elem_bytes == 4 && L.seg_shift == 8;   // whole u32 keys, 16-bit second slab
    L.stride_b = msd2_stride_b(n, L.slots, fine);
    L.tier_b = msd2_tier_b(n, L.slots, fine);
    L.tiles_per_bucket = (L.stride_a + tile - 1) / tile;
    L.off_mode = 0;
    L.off_cnt = L.off_mode + 256;
    L.off_off = L.off_cnt + 65536 * 4;
    L.off_hard = align_up(L.off_off + 65537 * 4, 256);                       // segments the binning finish hands to the LSD finish
    L.off_coop = L.off_hard + 65536 * 4;                                     // safety net: table [256][256] + 256 totals
    L.off_slab_a = align_up(L.off_coop + kNetTableBytes, 256);
    L.off_slab_b = align_up(L.off_slab_a + (size_t)256 * L.stride_a * elem_bytes, 256);
    // u32 keys: 16-bit second slab (when the finish has 16 bits to sort: w = 8)
    const size_t slab_b_bytes = (size_t)L.slots * L.stride_b * (elem_bytes == 4 && L.seg_shift == 8 ? 2 : elem_bytes);
    L.slab_b_in_tmp = elem_bytes == 4 && L.seg_shift == 8 && n >= kSlabInTmpMin && slab_b_bytes <= n * elem_bytes;
    L.total = L.off_slab_b + (L.slab_b_in_tmp ? 0 : slab_b_bytes);
    return L;
}

template <typename E, typename S, bool SOA = false>
int launch_large_finish(adlhip_device* d, const E* slab_b, E* out, uint32_t* out_vals, uint32_t* seg_off, uint32_t* seg_cnt, uint32_t stride_b,
                        uint32_t tier_b, uint32_t* mode, uint32_t* hard, int low_bits_max, bool bin, uint32_t seg_shift = 8);
bool use_bin_finish(const adlhip_device* d, size_t elem_bytes, bool key64, size_t n, bool whole_keys);

template <typename E>
int msd2_sort(adlhip_device* d, E* data, E* tmp, void* work, size_t n, int headroom_pct = kFullHeadroomPct)
{
    constexpr int K = sizeof(E) == 4 ? 32 : 16;
    constexpr int KEY_BITS = 8 * (int)sizeof(E);
    constexpr bool k32 = sizeof(E) == 4;   // profile names are kept by pointer: literals only
    // d_msd2 (allocated at device creation): cursors of both passes + flag + done counter + sample words, idle values between sorts
    uint32_t* cur_a = d->d_msd2;            // 256 cursors, one 128-byte line each
    uint32_t* cur_b = d->d_msd2 + 8192;     // 65536 cursors, packed (16 atomics each per sort)
    uint32_t* flag = d->d_msd2 + 8192 + 65536;
    uint32_t* done = flag + 1;
    uint32_t* sample = flag + 8;            // or lo, or hi, and lo, and hi (hybrid_kernels.hpp msd2_placement)
    const Msd2Layout L = msd2_layout(n, sizeof(E), headroom_pct);
    char* wb = reinterpret_cast<char*>(work);
    uint32_t* mode = reinterpret_cast<uint32_t*>(wb + L.off_mode);
    uint32_t* seg_cnt = reinterpret_cast<uint32_t*>(wb + L.off_cnt);
    uint32_t* seg_off = reinterpret_cast<uint32_t*>(wb + L.off_off);
    E* slab_a = reinterpret_cast<E*>(wb + L.off_slab_a);
    E* slab_b = L.slab_b_in_tmp ? tmp : reinterpret_cast<E*>(wb + L.off_slab_b);
    using CT = adlhip::TileCfg<E, 8, 512, K>;
    auto kern = adlhip::msd_bucket_scatter_kernel<E, 512, K, 1>;
    auto kern2 = adlhip::msd_bucket_scatter_kernel<E, 512, K, 2>;
    if (ensure_lds(kern, CT::LDS_BYTES) || ensure_lds(kern2, CT::LDS_BYTES)) return ADLHIP_FAILURE;
    // where the two digits sit is chosen on the device from a sample of the keys (hybrid_kernels.hpp msd2_sample_kernel).  (Folding
    // the sample into pass 1 -- every workgroup ORs and ANDs the same 1024 keys -- saves this launch and costs more than it saves:
    // 1024 strided keys are 1024 cache lines, twice a tile's own; pass 1 at 64 Mi keys 0.150 -> 0.166 ms, nothing gained at 4 Mi.)
    int rc = launch(d, "msd2_sample", [&] {
        hipLaunchKernelGGL(adlhip::msd2_sample_kernel<E>, dim3(adlhip::kSampleWGs), dim3(64), 0, d->stream, (const E*)data, (uint32_t)n,
                           sample, flag + 2, d->d_fault, flag, sample_dup_threshold(n));
    });
    if (rc) return rc;
    adlhip::BucketPass<E> pa;   // pass 1: the input, first digit -> 256 bucket slabs
    pa.src = data; pa.dst = slab_a; pa.cursors = cur_a; pa.cursor_shift = 5; pa.src_count_shift = 0; pa.flag = flag;
    pa.src_counts = nullptr; pa.n = (uint32_t)n;
    pa.src_stride = 0; pa.tiles_per_bucket = 1; pa.dst_stride = L.stride_a; pa.dst_total = 256u * L.stride_a; pa.start_bit = KEY_BITS - 8;
    pa.zero_me = nullptr; pa.sample = sample; pa.which_digit = 1; pa.dst16 = 0;
    pa.place = nullptr; pa.status_a = nullptr; pa.pieces = 0; pa.rows_per_chain_a = 0; pa.slice = 0; pa.seg_shift = 8;
    const uint32_t tiles_a = (uint32_t)((n + CT::TILE - 1) / CT::TILE);
    // round 4: the same pass as a persistent kernel -- 2 workgroups per CU take the tiles in turns, the next tile's keys are in
    // flight (non-temporal 16-byte buffer loads) while the current one is ranked, and no wait ever covers a tile's stores
    // (persist_kernels.hpp).  Buffer addressing: the slabs must lie within 4 GiB of their base.
    const bool slab16 = sizeof(E) == 4 && L.seg_shift == 8;
    const bool persist = d->persist && (size_t)256 * L.stride_a * sizeof(E) < (size_t(1) << 32) - (1u << 20) &&
                         (size_t)L.slots * L.stride_b * (slab16 ? 2 : sizeof(E)) < (size_t(1) << 32) - (1u << 20);
    using PC = adlhip::PersistCfg<E, 512, K>;
    const uint32_t pgrid = (uint32_t)(2 * d->prop.multiProcessorCount + 7) / 8 * 8;   // a multiple of 8: workgroup % 8 = XCD
    if (persist) {
        auto pk = adlhip::msd_scatter_persist_kernel<E, 512, K, 1, 4, false, 2, 0>;
        if (ensure_lds(pk, PC::LDS_BYTES)) return ADLHIP_FAILURE;
        rc = launch(d, k32 ? "msd2_pass1_u32" : "msd2_pass1_u64", [&] { hipLaunchKernelGGL(pk, dim3(pgrid), dim3(512), PC::LDS_BYTES, d->stream, pa); });
    } else {
        rc = launch(d, k32 ? "msd2_pass1_u32" : "msd2_pass1_u64", [&] { hipLaunchKernelGGL(kern, dim3(tiles_a), dim3(512), CT::LDS_BYTES, d->stream, pa); });
    }
    if (rc) return rc;
    adlhip::BucketPass<E> pb = pa;   // pass 2: every bucket, second digit -> 65536 segment slabs
    pb.src = slab_a; pb.dst = slab_b; pb.cursors = cur_b; pb.cursor_shift = 0; pb.src_count_shift = 5; pb.flag = flag;
    pb.src_counts = cur_a; pb.n = (uint32_t)n;
    pb.src_stride = L.stride_a; pb.tiles_per_bucket = L.tiles_per_bucket; pb.dst_stride = L.stride_b;
    pb.dst_total = L.slots * L.stride_b; pb.start_bit = KEY_BITS - 16; pb.zero_me = nullptr;
    pb.sample = sample; pb.which_digit = 2; pb.seg_shift = L.seg_shift;
    pb.dst16 = slab16 ? 1 : 0;   // u32 keys: the second slab holds the low 16 bits only
    if (persist) {
        auto pk16 = adlhip::msd_scatter_persist_kernel<E, 512, K, 2, 4, true, 2, 0>;
        auto pkE = adlhip::msd_scatter_persist_kernel<E, 512, K, 2, 4, false, 2, 0>;
        if (ensure_lds(pk16, PC::LDS_BYTES) || ensure_lds(pkE, PC::LDS_BYTES)) return ADLHIP_FAILURE;
        rc = launch(d, k32 ? "msd2_pass2_u32" : "msd2_pass2_u64", [&] {
            if (slab16) hipLaunchKernelGGL(pk16, dim3(pgrid), dim3(512), PC::LDS_BYTES, d->stream, pb);
            else hipLaunchKernelGGL(pkE, dim3(pgrid), dim3(512), PC::LDS_BYTES, d->stream, pb);
        });
    } else {
        rc = launch(d, k32 ? "msd2_pass2_u32" : "msd2_pass2_u64", [&] {
            hipLaunchKernelGGL(kern2, dim3(256 * L.tiles_per_bucket), dim3(512), CT::LDS_BYTES, d->stream, pb);
        });
    }
    if (rc) return rc;
    // segment sizes and offsets -- and, when a run did not fit its slab, the safety net: the same 256 resident workgroups sort
    // the untouched input with the cooperative LSD sort (hybrid_kernels.hpp coop_lsd_sort); the finish then returns at once
    uint32_t* bar = done + 1;
    uint32_t* ctable = reinterpret_cast<uint32_t*>(wb + L.off_coop);
    using CC = adlhip::TileCfg<E, 8, 512, K>;   // the net's tile: the one-sweep pass's own (16 Ki keys / 8 Ki 8-byte elements)
    auto ko = adlhip::msd2_offsets_kernel<E, 512, K>;
    if (ensure_lds(ko, CC::LDS_BYTES)) return ADLHIP_FAILURE;
    rc = launch(d, "msd2_offsets", [&] {
        hipLaunchKernelGGL(ko, dim3(net_wgs(d)), dim3(512), CC::LDS_BYTES, d->stream, cur_a, cur_b, flag, done, bar,
                           seg_cnt, seg_off, mode, (uint32_t)n, sample, data, tmp, ctable, d->d_fault, KEY_BITS,
                           8u - L.seg_shift, d->dict_path ? d->d_dict : nullptr, net_stats(d),
                           net_onesweep_args(d, wb + L.off_slab_a, (size_t)256 * L.stride_a * sizeof(E), n, 4, CC::TILE));
    });
    if (rc) return rc;
    // the finish sorts the bits below the second digit (the offsets kernel has published how many)
    using S = typename std::conditional<sizeof(E) == 4, uint16_t, E>::type;   // what pass 2 wrote (w = 8)
    uint32_t* hard = reinterpret_cast<uint32_t*>(wb + L.off_hard);
    const int low_max = KEY_BITS - 8 - (int)L.seg_shift;
    // (whole u64 keys: the binning finish wants segments of ~384 keys and more -- 24 Mi keys in 65536 segments, or any n with a
    // narrow second digit, which leaves segments of about a thousand keys)
    const bool bin = use_bin_finish(d, sizeof(E), sizeof(E) == 8, L.seg_shift < 8 ? (size_t(24) << 20) : n, true);
    if (sizeof(E) == 4 && !slab16)
        return launch_large_finish<E, E>(d, slab_b, data, nullptr, seg_off, seg_cnt, L.stride_b, L.tier_b, mode, hard, low_max, bin, L.seg_shift);
    return launch_large_finish<E, S>(d, slab_b, data, nullptr, seg_off, seg_cnt, L.stride_b, L.tier_b, mode, hard, low_max, bin, L.seg_shift);
}

// ---- the same for {key, value} pairs, STABLE: look-back instead of cursors (hybrid_kernels.hpp msd_lookback_scatter_kernel) --
constexpr size_t kMsd2sAutoMin = size_t(1) << 20;   // pairs; measured with the narrow second digit: 1.5 Mi pairs 55 vs 89 us (three-kernel
                                                    // passes), 4 Mi 89 vs 105, 8 Mi 135 vs 183 (profiles/r3_narrow_second_digit.txt)
constexpr size_t kMsd2sMax = (size_t(1) << 28) + (size_t(1) << 22);
// tile of the look-back passes: TileCfg<uint64_t, 8, 512, 16> = 8192 pairs, TileCfg<uint32_t, 8, 512, 32> = 16384 keys
constexpr uint32_t msd2s_tile(size_t elem_bytes) { return elem_bytes == 8 ? 8192u : 16384u; }

struct Msd2sLayout {
    size_t off_mode, off_place, off_cnt, off_off, off_hard, off_coop, off_tickets, off_status_a, off_status_b, off_slab_a, off_slab_b, total;
    uint32_t pieces, slice, rows_a, rows_b, stride_a, stride_b, tier_b, ticket_words, seg_shift, slots;
    size_t status_bytes_a, status_bytes_b;
    bool slab_b_in_tmp;
};

// slab16: whole u32 keys -- the second slab holds their low 16 bits and, from 16 Mi keys, sits in the caller's n-element scratch
// array (see Msd2Layout::slab_b_in_tmp)
// lean: the sub-slabs of the first pass and the segment slabs keep statistical head-room only (mean + 8 sd / + 7.5 sd instead of
// + 50 %): 64 Mi pairs 1.25 GB of work instead of 1.7.  Evenly spread keys run at the same speed; keys whose density varies by more
// than a few per cent between neighbouring parts of their range take the safety net.
Msd2sLayout msd2s_layout(size_t n, size_t elem_bytes = 8, bool slab16 = false, bool lean = false)
{
    Msd2sLayout L;
    const uint32_t kMsd2sTile = msd2s_tile(elem_bytes);
    // pieces = chains of pass A = sub-slabs per bucket.  16 = twice the number of XCDs: workgroup i runs on XCD i % 8 and takes
    // chain i % 16, so a chain's tiles stay on ONE XCD -- its status rows and the abutting runs of consecutive tiles meet in
    // that XCD's L2.  Chain counts that break this measured slower although they make pass B's tiles fuller (64 Mi pairs,
    // pass A / pass B: 16 chains 0.252 / 0.309 ms, 12 chains 0.282 / 0.325, 18 chains 0.303 / 0.324).  The code below keeps
    // the choice open (any count up to 24 works).
    double best = 1e30;
    for (uint32_t p : {16u}) {
        const size_t slice = align_up((n + p - 1) / p, kMsd2sTile);
        const size_t mean = slice / 256;
        size_t sd = 1;
        while (sd * sd < mean) ++sd;
        const size_t stride = align_up(mean + (lean ? 0 : mean / 2) + 8 * sd + 64, 64);   // + 50 %: as much skew as the segment slabs take
        const double waste = (double)((stride + kMsd2sTile - 1) / kMsd2sTile) * kMsd2sTile / (double)mean;
        if (waste < best - 1e-9) {
            best = waste;
            L.pieces = p;
            L.slice = (uint32_t)slice;
            L.stride_a = (uint32_t)stride;
        }
    }
    L.rows_a = L.slice / kMsd2sTile;
    // the most tiles a bucket can have: pass B's tiles run across the sub-slabs, and a sub-slab holds at most its stride
    L.rows_b = (uint32_t)(((size_t)L.pieces * L.stride_a + kMsd2sTile - 1) / kMsd2sTile);
    // (whole u32 keys keep 65536 segments: their second slab holds 16-bit keys only while the finish has 16 bits to sort)
    L.seg_shift = slab16 ? 8u : msd2_seg_shift(n, false);
    L.slots = 256u << L.seg_shift;
    L.stride_b = msd2_stride_b(n, L.slots, slab16, lean);
    L.tier_b = msd2_tier_b(n, L.slots, slab16);
    L.ticket_words = (32 + 256) * adlhip::kTicketStride;
    L.status_bytes_a = (size_t)L.pieces * L.rows_a * 1024;
    L.status_bytes_b = (size_t)256 * L.rows_b * 1024;
    L.off_mode = 0;
    L.off_place = 128;
    L.off_cnt = 256;
    L.off_off = L.off_cnt + 65536 * 4;
    L.off_hard = align_up(L.off_off + 65537 * 4, 256);
    L.off_coop = L.off_hard + 65536 * 4;
    L.off_tickets = align_up(L.off_coop + kNetTableBytes, 256);
    // the regions are sized by bounds that do not depend on `pieces` and grow with n, so that the scratch for n suffices
    // for every smaller n (adlhip_radix_sort_scratch_bytes): sub-slabs of a bucket together <= n/256 + 16 * (head-room),
    // rows of pass A <= n/tile + 16, rows of a bucket in pass B <= its sub-slabs / tile + 16
    size_t sdb = 1;
    while (sdb * sdb * 4096 < n) ++sdb;   // >= sd of every choice (a sub-slab's mean is at most n / 4096 + 32)
    const size_t bucket_bound = n / 256 + (lean ? 0 : n / 512) + 24 * (32 + 16 + 8 * (sdb + 1) + 128);
    const size_t rows_a_bound = n / kMsd2sTile + 24;
    const size_t rows_b_bound = bucket_bound / kMsd2sTile + 25;
    L.off_status_a = align_up(L.off_tickets + (size_t)L.ticket_words * 4, 256);
    L.off_status_b = L.off_status_a + rows_a_bound * 1024;
    L.off_slab_a = align_up(L.off_status_b + 256 * rows_b_bound * 1024, 256);
    L.off_slab_b = align_up(L.off_slab_a + 256 * bucket_bound * elem_bytes, 256);
    const size_t slab_b_bytes = (size_t)L.slots * L.stride_b * (slab16 ? 2 : elem_bytes);
    L.slab_b_in_tmp = slab16 && n >= kSlabInTmpMin && slab_b_bytes <= n * elem_bytes;
    L.total = L.off_slab_b + (L.slab_b_in_tmp ? 0 : slab_b_bytes);
    if ((size_t)L.pieces * L.rows_a > rows_a_bound || L.rows_b > rows_b_bound || (size_t)L.pieces * L.stride_a > bucket_bound)
        L.total = 0;   // cannot happen; msd2s_sort refuses
    return L;
}

// Which sorts take the large sort, and in which form.  keys: the element is the key (u32 / u64 keys), else {key, value} pairs.
//   whole u32 keys from 96 Mi keys               -> hybrid: stable first pass + cursor-placed second pass (msd2s_sort, hybrid)
//   whole u64 keys from 48 Mi keys               -> stable passes (msd2s_sort) + binning finish
//   whole keys below                            -> cursor passes (msd2_sort: smaller fixed cost)
//   pairs; partial sortBits (>= 16)             -> stable passes (msd2s_sort)
// "sort.msd2" = 3 / 4 / 5 force the stable / cursor / hybrid form where it applies (tests, A/B measurements).
enum LargeForm { kLargeNone = 0, kLargeCursor, kLargeStable, kLargeHybrid };
LargeForm large_sort_form(const adlhip_device* d, size_t elem_bytes, bool keys, size_t n, int sort_bits, int max_bits)
{
    if (!(d->sort_algo < 0 && d->msd2_path && d->digit_bits == 8 && d->tile_variant < 0)) return kLargeNone;
    if (d->resident_wgs < 256) return kLargeNone;   // the safety net's grid barrier spans 256 workgroups
    if (sort_bits < 16) return kLargeNone;          // two 8-bit digits must fit inside the sorted bits
    const bool forced = d->msd2_path >= 2;
    const bool whole = sort_bits == max_bits;
    // "sort.rank" = 0 (no reliance on the lane order of colliding DS atomics): pairs keep the stable form -- its passes and its
    // wave-per-segment finish have ballot-ranked variants (tiles up to 2560 pairs: n <= 96 Mi) --, keys take the per-digit passes
    if (d->rank_mode != 1 && (keys || n > (size_t(96) << 20) || msd2_tier_b(n, 256u << msd2_seg_shift(n, false)) > 2560u)) return kLargeNone;
    if (!keys) return n > (forced ? kMsd2Min : kMsd2sAutoMin) && n <= kMsd2sMax ? kLargeStable : kLargeNone;
    // u64 keys have no mid-size sort (it serves 32-bit keys), and with the narrow second digit and the binning finish the large sort
    // beats their per-digit passes from the one-workgroup sort's limit up: 100 K keys 78 -> 32 us, 1 Mi 117 -> 47, 2 Mi 172 -> 52
    // (profiles/r3_small_sizes_large_sort_vs_auto.txt)
    if (n <= (forced || elem_bytes == 8 ? kMsd2Min : kMsd2AutoMin)) return kLargeNone;
    const size_t cursor_max = elem_bytes == 4 ? kMsd2MaxU32 : kMsd2MaxU64;
    if (!whole || d->msd2_path == 3) return n <= kMsd2sMax ? kLargeStable : kLargeNone;
    if (d->msd2_path == 4) return n <= cursor_max ? kLargeCursor : kLargeNone;
    if (d->msd2_path == 5) return n <= kMsd2sMax ? kLargeHybrid : kLargeNone;
    // profiles/r3_forms_by_size.txt -- u32 keys: cursor | hybrid 64 Mi 0.424 | 0.423 ms, 128 Mi 0.782 | 0.751, 256 Mi 1.700 | 1.620
    // (16 Mi: 0.140 | 0.159); u64 keys: cursor | stable 16 Mi 0.294 | 0.312, 64 Mi 0.846 | 0.803, 256 Mi 3.18 | 2.99 (hybrid 2.96)
    // round 4 (profiles/r4_forms_by_size.txt, persistent cursor passes): u32 keys cursor | hybrid 64 Mi 0.355 | 0.420 ms, 96 Mi 0.559 | 0.585,
    // 128 Mi 0.688 | 0.743, 256 Mi 1.496 | 1.436 -- the cursor-placed first pass loses ground with n (4.3 TB/s at 64 Mi, 3.4 at 256 Mi)
    const size_t hybrid_min = d->persist ? (size_t(192) << 20) : (size_t(96) << 20);
    if (elem_bytes == 4 && n >= hybrid_min && n <= kMsd2sMax) return kLargeHybrid;
    if (elem_bytes == 8 && n >= (size_t(48) << 20) && n <= kMsd2sMax) return kLargeStable;
    return n <= cursor_max ? kLargeCursor : kLargeNone;
}

// The finish of either large sort: the binning finish (whole u64 keys from ~24 Mi up: one counting pass + compares, then the
// LSD finish's list form for what it handed over) or the wave-per-segment LSD finish.  S = what the second slab holds.
template <typename E, typename S, bool SOA>
int launch_large_finish(adlhip_device* d, const E* slab_b, E* out, uint32_t* out_vals, uint32_t* seg_off, uint32_t* seg_cnt, uint32_t stride_b,
                        uint32_t tier_b, uint32_t* mode, uint32_t* hard, int low_bits_max, bool bin, uint32_t seg_shift)
{
    // tier_b = the tile that holds a segment (msd2_tier_b); stride_b <= tier_b = the spacing of the segment slabs;
    // 256 << seg_shift slots (hybrid_kernels.hpp slot_to_segment)
    const uint32_t* lowb = mode + adlhip::kDynLowBits;
    const size_t slots = (size_t)256 << seg_shift;
    // segments beyond a wave's tile: whole u32 keys above 280 Mi keys (16-bit second slab, 65536 segments).  The 5120-key tier
    // (about 140 Mi ... 288 Mi keys) goes the same way: four waves with 20 rows per thread beat the one wave that holds 80
    // (256 Mi keys: finish 0.64 -> 0.48 ms, 144 Mi: 0.80 -> 0.35); below, the wave per segment stays ahead (64 Mi keys: 0.125 vs
    // 0.129 ms, 128 Mi: 0.218 vs 0.235).  Workgroup sizes by measurement (profiles/r3_wg_finish.txt): 5120 keys 256 x 20 | 512 x 10
    // = 0.478 | 0.606 ms at 256 Mi keys; 8192: 256 x 32 | 512 x 16 = 0.672 | 0.731 at 384 Mi; 12288: 512 x 24 | 256 x 48 = 1.047 |
    // 1.089 at 512 Mi; 16384 keys with 1024 x 16 took 2.37 ms against 1.63 with 512 x 32 at 768 Mi
    constexpr bool wg_kind = !SOA && sizeof(E) == 4 && sizeof(S) == 2;
    // round 4: 16-bit keys stay 16 bits wide in the finish (finish16_kernels.hpp): tiles of 1280 / 1536 / 2560 keys, one wave each
    if constexpr (wg_kind) {
        if (d->persist && !bin && seg_shift == 8 && tier_b <= 2560 && (stride_b & 1u) == 0u) {
            const uint16_t* sb = reinterpret_cast<const uint16_t*>(slab_b);
#define ADLHIP_F16(R2_, WAVES_)                                                                                                      \
    {                                                                                                                                \
        auto kf = adlhip::wave_finish16_kernel<R2_, WAVES_, true, true, 1>;                                                                   \
        const size_t lds = (size_t)WAVES_ * adlhip::Finish16Cfg<R2_>::PER_WAVE;                                                      \
        if (ensure_lds(kf, lds)) return ADLHIP_FAILURE;                                                                              \
        return launch(d, "segment_sort_wave_u32", [&] {                                                                              \
            hipLaunchKernelGGL(kf, dim3((uint32_t)((slots + WAVES_ - 1) / WAVES_)), dim3(64 * WAVES_), lds, d->stream, sb, out, seg_off,       \
                               seg_cnt, stride_b, mode, lowb, (uint32_t)slots, d->d_fault);                                          \
        });                                                                                                                          \
    }
            if (tier_b <= 1280) ADLHIP_F16(10, 4)
            if (tier_b <= 1536) ADLHIP_F16(12, 4)
            ADLHIP_F16(20, 4)
#undef ADLHIP_F16
        }
    }
    // A/B knobs: ADLHIP_WG_MIN_TIER = smallest tile that takes the workgroup finish, ADLHIP_WG_NT = 0 (default choice) / 128 / 256 / 512
    static const uint32_t wg_min = getenv("ADLHIP_WG_MIN_TIER") ? (uint32_t)atoi(getenv("ADLHIP_WG_MIN_TIER")) : 3072u;
    static const int wg_nt = getenv("ADLHIP_WG_NT") ? atoi(getenv("ADLHIP_WG_NT")) : 0;
    if (tier_b > 5120 || (wg_kind && tier_b >= wg_min && seg_shift == 8)) {
        if constexpr (wg_kind) {
            const S* sb = reinterpret_cast<const S*>(slab_b);
            if (seg_shift != 8) return fail("internal: the workgroup-per-segment finish takes 65536 segments");
#define ADLHIP_WG(NT_, K_) return launch_wg_segment_sort<E, S, NT_, K_>(d, sb, out, seg_off, seg_cnt, stride_b, mode, slots)
            switch (tier_b) {
            case 1280: if (wg_nt == 256) ADLHIP_WG(256, 5); ADLHIP_WG(128, 10);
            case 1536: if (wg_nt == 256) ADLHIP_WG(256, 6); ADLHIP_WG(128, 12);
            case 2560: if (wg_nt == 128) ADLHIP_WG(128, 20); ADLHIP_WG(256, 10);
            case 3072: ADLHIP_WG(256, 12);
            case 4096: ADLHIP_WG(256, 16);
            case 5120: if (wg_nt == 512) ADLHIP_WG(512, 10); ADLHIP_WG(256, 20);
            case 8192: if (wg_nt == 512) ADLHIP_WG(512, 16); ADLHIP_WG(256, 32);
            case 12288: if (wg_nt == 256) ADLHIP_WG(256, 48); ADLHIP_WG(512, 24);
            case 16384: ADLHIP_WG(512, 32);
            default: ADLHIP_WG(512, 40);
            }
#undef ADLHIP_WG
        } else {
            return fail("internal: no finish for segments of %u elements of this kind", tier_b);
        }
    }
    if constexpr (!SOA) {
        if (bin) {
            const S* sb = reinterpret_cast<const S*>(slab_b);
            int rc;
            if (tier_b <= kMsd2Stride0) rc = launch_bin_segment_sort<E, S, 256, 6, 11>(d, sb, out, seg_off, seg_cnt, stride_b, mode, hard, seg_shift);
            else if (tier_b == 2560) rc = launch_bin_segment_sort<E, S, 256, 10, 12>(d, sb, out, seg_off, seg_cnt, stride_b, mode, hard, seg_shift);
            else rc = launch_bin_segment_sort<E, S, 512, 10, 12>(d, sb, out, seg_off, seg_cnt, stride_b, mode, hard, seg_shift);
            if (rc) return rc;
            const uint32_t* hc = mode + adlhip::kDynHardCnt;
            if (tier_b <= kMsd2Stride0) return launch_wave_segment_sort<E, kMsd2Stride0 / 64, S>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, nullptr, hard, hc, seg_shift);
            if (tier_b == 2560) return launch_wave_segment_sort<E, 40, S>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, nullptr, hard, hc, seg_shift);
            return launch_wave_segment_sort<E, 80, S>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, nullptr, hard, hc, seg_shift);
        }
    }
    if constexpr (sizeof(E) == 8 && sizeof(S) == 8) {
        if (d->rank_mode == 0) {   // "sort.rank" = 0 (pairs; large_sort_form keeps n within these tiles): the ballot-ranked finish
            if (tier_b == 1280) return launch_wave_segment_sort<E, 20, S, SOA, 0>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, out_vals, nullptr, nullptr, seg_shift);
            if (tier_b == kMsd2Stride0) return launch_wave_segment_sort<E, kMsd2Stride0 / 64, S, SOA, 0>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, out_vals, nullptr, nullptr, seg_shift);
            if (tier_b == 2560) return launch_wave_segment_sort<E, 40, S, SOA, 0>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, out_vals, nullptr, nullptr, seg_shift);
            return fail("internal: no ballot-ranked finish for segments of %u elements", tier_b);
        }
    }
    if (tier_b == 1280) return launch_wave_segment_sort<E, 20, S, SOA>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, out_vals, nullptr, nullptr, seg_shift);
    if (tier_b == kMsd2Stride0) return launch_wave_segment_sort<E, kMsd2Stride0 / 64, S, SOA>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, out_vals, nullptr, nullptr, seg_shift);
    if (tier_b == 2560) return launch_wave_segment_sort<E, 40, S, SOA>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, out_vals, nullptr, nullptr, seg_shift);
    return launch_wave_segment_sort<E, 80, S, SOA>(d, slab_b, out, seg_off, slots, low_bits_max, seg_cnt, stride_b, mode, lowb, out_vals, nullptr, nullptr, seg_shift);
}

// whole u64 keys: the binning finish pays from a mean segment of ~384 keys (16 Mi keys: 0.144 vs 0.148 ms, 4 Mi: 0.116 vs 0.098,
// 64 Mi: 0.263 vs 0.406, profiles/r3_bin_finish.txt)
bool use_bin_finish(const adlhip_device* d, size_t elem_bytes, bool key64, size_t n, bool whole_keys)
{
    if (!whole_keys || !d->bin_finish) return false;
    if (d->bin_finish == 2) return elem_bytes == 4 || key64;   // forced (tests, A/B): any size, u32 keys too
    return elem_bytes == 8 && key64 && n >= (size_t(24) << 20);
}

// The stable large sort.  E / KEY64: uint64_t / false = {key, value} pairs (AoS: data / tmp; SoA: soa_keys != nullptr, the input and
// output are the two u32 arrays, data / tmp are unused -- the safety net packs the input into the first slab area and sorts it
// there against the second); uint64_t / true = u64 keys; uint32_t / false = u32 keys (whole keys: the second slab holds their low
// 16 bits, as in the cursor form).  sort_bits < key bits (Pprims.cpp:357, a multiple of 4, at least 16 here): only the low sort_bits
// bits of a key count -- the digits are placed inside them, everything is stable, so keys that agree there keep their input order
// as the reference's LSD passes would leave them.
// hybrid (whole keys only -- equal keys are indistinguishable, so the second pass need not be stable): pass A as above, then the
// CURSOR-placed second pass of the keys-only form over buckets made of pass A's sub-slabs (msd_bucket_scatter_kernel, PASS = 3).
// Measured per pass at 64 Mi u32 keys: first pass 0.138 ms by look-back vs 0.147-0.159 with cursors (the abutting runs of a
// chain's consecutive tiles meet in one XCD's L2: 1.05 x instead of 1.21 x bytes written), second pass 0.124 with cursors vs
// 0.172 by look-back (profiles/r3_first_ab_cursor_vs_lookback_u32.txt).
template <typename E, bool KEY64>
int msd2s_sort(adlhip_device* d, E* data, E* tmp, void* work, size_t n, int sort_bits, uint32_t* soa_keys = nullptr,
               uint32_t* soa_vals = nullptr, bool hybrid = false, bool lean = false)
{
    constexpr int K = sizeof(E) == 8 ? 16 : 32;
    constexpr bool k32 = sizeof(E) == 4;
    constexpr int KEY_BITS = KEY64 ? 64 : 32;
    const bool whole = sort_bits == KEY_BITS;
    uint32_t* flag = d->d_msd2 + 8192 + 65536;
    uint32_t* done = flag + 1;
    uint32_t* bar = flag + 2;
    const Msd2sLayout L = msd2s_layout(n, sizeof(E), k32 && whole, lean);
    if (L.total == 0) return fail("internal: layout bounds of the stable large sort");
    char* wb = reinterpret_cast<char*>(work);
    uint32_t* mode = reinterpret_cast<uint32_t*>(wb + L.off_mode);
    adlhip::StablePlace* place = reinterpret_cast<adlhip::StablePlace*>(wb + L.off_place);
    uint32_t* seg_cnt = reinterpret_cast<uint32_t*>(wb + L.off_cnt);
    uint32_t* seg_off = reinterpret_cast<uint32_t*>(wb + L.off_off);
    uint32_t* hard = reinterpret_cast<uint32_t*>(wb + L.off_hard);
    uint32_t* tickets = reinterpret_cast<uint32_t*>(wb + L.off_tickets);
    uint32_t* status_a = reinterpret_cast<uint32_t*>(wb + L.off_status_a);
    uint32_t* status_b = reinterpret_cast<uint32_t*>(wb + L.off_status_b);
    E* slab_a = reinterpret_cast<E*>(wb + L.off_slab_a);
    E* slab_b = L.slab_b_in_tmp ? tmp : reinterpret_cast<E*>(wb + L.off_slab_b);
    using CT = adlhip::TileCfg<E, 8, 512, K>;
    static_assert(CT::TILE == (int)msd2s_tile(sizeof(E)), "layout and kernel agree on the tile");
    auto kern = adlhip::msd_lookback_scatter_kernel<E, 512, K, KEY64>;
    if constexpr (sizeof(E) == 8 && !KEY64) {   // pairs with "sort.rank" = 0: ballot ranking in the passes (and in the finish, below)
        if (d->rank_mode == 0) kern = adlhip::msd_lookback_scatter_kernel<E, 512, K, KEY64, 0>;
    }
    if (ensure_lds(kern, CT::LDS_BYTES)) return ADLHIP_FAILURE;
    // status rows of both passes: zero (one memset; the rows are contiguous).  Hybrid: pass A's only.
    HIPCHK(hipMemsetAsync(status_a, 0, hybrid ? L.status_bytes_a : (L.off_status_b - L.off_status_a) + L.status_bytes_b, d->stream));
    const uint32_t dup_thr = sample_dup_threshold(n);
    int rc = launch(d, "msd2s_prep", [&] {
        if (soa_keys)
            hipLaunchKernelGGL((adlhip::msd2s_prep_kernel<uint32_t, false>), dim3(adlhip::kSampleWGs), dim3(64), 0, d->stream, (const uint32_t*)soa_keys,
                               (uint32_t)n, place, tickets, L.ticket_words, bar, d->d_fault, (uint32_t)sort_bits, flag, dup_thr);
        else
            hipLaunchKernelGGL((adlhip::msd2s_prep_kernel<E, KEY64>), dim3(adlhip::kSampleWGs), dim3(64), 0, d->stream, (const E*)data, (uint32_t)n, place,
                               tickets, L.ticket_words, bar, d->d_fault, (uint32_t)sort_bits, flag, dup_thr);
    });
    if (rc) return rc;
    adlhip::LookbackPass<E> pa;
    pa.src = data; pa.dst = slab_a; pa.status = status_a; pa.status_bytes = (uint32_t)L.status_bytes_a; pa.tickets = tickets;
    pa.flag = flag; pa.fault = d->d_fault; pa.place = place; pa.which_digit = 1; pa.n = (uint32_t)n; pa.chains = L.pieces; pa.pieces = L.pieces;
    pa.rows_per_chain = L.rows_a; pa.slice = L.slice; pa.src_stride = 0; pa.status_a = nullptr; pa.rows_per_chain_a = 0;
    pa.dst_stride = L.stride_a; pa.dst_total = 256u * L.pieces * L.stride_a;
    pa.soa_keys = soa_keys; pa.soa_vals = soa_vals; pa.dst16 = 0; pa.seg_shift = 8;
    rc = launch(d, k32 ? "msd2s_pass1_u32" : KEY64 ? "msd2s_pass1_u64" : soa_keys ? "msd2s_pass1_soa" : "msd2s_pass1_kv32", [&] {
        hipLaunchKernelGGL(kern, dim3(L.pieces * L.rows_a), dim3(512), CT::LDS_BYTES, d->stream, pa);
    });
    if (rc) return rc;
    const bool slab16 = k32 && whole;   // whole u32 keys: a segment's keys share everything above their low 16 bits
    uint32_t* cur_b = nullptr;
    if constexpr (sizeof(E) == 4 || KEY64) {
        if (hybrid) {
            if (!whole || soa_keys) return fail("internal: the hybrid form sorts whole keys only");
            cur_b = d->d_msd2 + 8192;   // 65536 cursors [bucket][digit], zero between sorts (the offsets kernel clears them)
            auto kern2 = adlhip::msd_bucket_scatter_kernel<E, 512, K, 3>;
            if (ensure_lds(kern2, CT::LDS_BYTES)) return ADLHIP_FAILURE;
            adlhip::BucketPass<E> pc;
            pc.src = slab_a; pc.dst = slab_b; pc.cursors = cur_b; pc.cursor_shift = 0; pc.src_count_shift = 0; pc.flag = flag;
            pc.src_counts = nullptr; pc.n = (uint32_t)n; pc.src_stride = L.stride_a; pc.tiles_per_bucket = L.rows_b;
            pc.dst_stride = L.stride_b; pc.dst_total = L.slots * L.stride_b; pc.start_bit = 0; pc.zero_me = nullptr; pc.sample = nullptr;
            pc.which_digit = 2; pc.dst16 = slab16 ? 1 : 0;
            pc.place = place; pc.status_a = status_a; pc.pieces = L.pieces; pc.rows_per_chain_a = L.rows_a; pc.slice = L.slice;
            pc.seg_shift = L.seg_shift;
            rc = launch(d, k32 ? "msd2h_pass2_u32" : "msd2h_pass2_u64", [&] {
                hipLaunchKernelGGL(kern2, dim3(256 * L.rows_b), dim3(512), CT::LDS_BYTES, d->stream, pc);
            });
            if (rc) return rc;
        }
    }
    adlhip::LookbackPass<E> pb = pa;
    pb.src = slab_a; pb.dst = slab_b; pb.status = status_b; pb.status_bytes = (uint32_t)L.status_bytes_b;
    pb.tickets = tickets + 32 * adlhip::kTicketStride; pb.which_digit = 2; pb.chains = 256; pb.rows_per_chain = L.rows_b;
    pb.src_stride = L.stride_a; pb.status_a = status_a; pb.rows_per_chain_a = L.rows_a; pb.dst_stride = L.stride_b;
    pb.dst_total = L.slots * L.stride_b; pb.seg_shift = L.seg_shift;
    pb.soa_keys = nullptr; pb.soa_vals = nullptr;
    pb.dst16 = slab16 ? 1 : 0;
    if (!cur_b) {
        rc = launch(d, k32 ? "msd2s_pass2_u32" : KEY64 ? "msd2s_pass2_u64" : soa_keys ? "msd2s_pass2_soa" : "msd2s_pass2_kv32", [&] {
            hipLaunchKernelGGL(kern, dim3(256 * L.rows_b), dim3(512), CT::LDS_BYTES, d->stream, pb);
        });
        if (rc) return rc;
    }
    uint32_t* ctable = reinterpret_cast<uint32_t*>(wb + L.off_coop);
    constexpr int NK = sizeof(E) == 4 ? 32 : 16;
    // whole-key sorts: keys of few values are sorted by counting (equal keys are interchangeable), pairs with such keys by ONE
    // stable pass on the key's rank among the values (dict_kernels.hpp)
    adlhip::DictBlock* net_dict = whole && (!soa_keys || sizeof(E) == 8) && d->dict_path ? d->d_dict : nullptr;
    using CC = adlhip::TileCfg<E, 8, 512, NK>;   // the safety net's tile (it runs in the offsets kernel when a run did not fit)
    // the net's look-back passes: whole keys; SoA input (packed into the first slab area) and sorts on part of the key run its
    // count-scan-scatter passes
    constexpr int NP = KEY64 ? 8 : 4;   // key bytes: passes of the net's look-back sort (run four at a time)
    auto ko = adlhip::msd2s_offsets_kernel<E, CT::TILE, 512, NK, 1, NP>;
    if constexpr (sizeof(E) == 8 && !KEY64) {
        if (d->rank_mode == 0) ko = adlhip::msd2s_offsets_kernel<E, CT::TILE, 512, NK, 0, NP>;   // the net ranks by ballots too
    }
    if (ensure_lds(ko, CC::LDS_BYTES)) return ADLHIP_FAILURE;
    const adlhip::OsNet os = net_onesweep_args(d, (NP && whole && !soa_keys) ? wb + L.off_slab_a : nullptr, L.off_slab_b - L.off_slab_a, n, 4, CC::TILE);
    rc = launch(d, "msd2s_offsets", [&] {
        hipLaunchKernelGGL(ko, dim3(net_wgs(d)), dim3(512), CC::LDS_BYTES, d->stream,
                           (const uint32_t*)status_a, L.rows_a, L.slice, L.pieces, (const uint32_t*)status_b, L.rows_b, L.stride_a, flag,
                           done, bar, seg_cnt, seg_off, mode, (uint32_t)n, (const adlhip::StablePlace*)place,
                           soa_keys ? slab_a : data, soa_keys ? slab_b : tmp, ctable, d->d_fault, soa_keys, soa_vals, cur_b, 8u - L.seg_shift,
                           net_dict, net_stats(d), os);
    });
    if (rc) return rc;
    const int low_max = sort_bits - 8 - (int)L.seg_shift;
    const bool bin = use_bin_finish(d, sizeof(E), KEY64, L.seg_shift < 8 ? (size_t(24) << 20) : n, whole);
    if constexpr (k32) {
        if (slab16) return launch_large_finish<E, uint16_t>(d, slab_b, data, nullptr, seg_off, seg_cnt, L.stride_b, L.tier_b, mode, hard, low_max, bin, L.seg_shift);
        return launch_large_finish<E, E>(d, slab_b, data, nullptr, seg_off, seg_cnt, L.stride_b, L.tier_b, mode, hard, low_max, false, L.seg_shift);
    } else {
        if (soa_keys)   // the finish writes keys and values to their own arrays
            return launch_large_finish<E, E, true>(d, slab_b, reinterpret_cast<E*>(soa_keys), soa_vals, seg_off, seg_cnt, L.stride_b, L.tier_b, mode,
                                                   hard, low_max, false, L.seg_shift);
        return launch_large_finish<E, E>(d, slab_b, data, nullptr, seg_off, seg_cnt, L.stride_b, L.tier_b, mode, hard, low_max, bin, L.seg_shift);
    }
}

// what the large sort's form for this sort needs (0: the sort does not take the large sort)
size_t large_work_bytes(LargeForm form, size_t elem_bytes, size_t n, bool whole)
{
    if (form == kLargeCursor) return msd2_layout(n, elem_bytes).total;
    if (form == kLargeStable || form == kLargeHybrid) return msd2s_layout(n, elem_bytes, elem_bytes == 4 && whole).total;
    return 0;
}

// Work bytes with which a sort of n elements on sort_bits bits runs at full speed (level 1), or runs at all (level 0: the
// per-digit three-kernel passes -- the reference's own contract: a table of a few KiB beside the n-element partner array,
// Pprims.cpp:332-337), or keeps the large sort for whole u32 / u64 keys with 12 % instead of 50 % of head-room in the first
// slabs (level 2, "lean": 64 Mi u32 keys 307 MB instead of 451).  With anything in between, every path checks its own need and
// the sort takes the fastest one that fits.
size_t sort_work_bytes_at(const adlhip_device* d, int elem_kind, size_t n, int sort_bits, int level)
{
    const size_t a = work_bytes_three_kernel(d, n);
    if (level == 0) return a;
    size_t b;
    if (elem_kind == ADLHIP_ELEM_U32) b = onesweep_layout(d, n, max_passes_for(d, 32), buf_tile<AosBuf<uint32_t>>(d, n)).total;
    else if (elem_kind == ADLHIP_ELEM_SOA32) b = onesweep_layout(d, n, max_passes_for(d, 64), buf_tile<SoaBuf>(d, n)).total;
    else b = onesweep_layout(d, n, max_passes_for(d, 64), buf_tile<AosBuf<uint64_t>>(d, n)).total;   // as onesweep_sort<Buf>()
    const size_t c = n <= kMidMaxU32 ? mid_layout(n).total : mid_layout(kMidMaxU32).total;   // mid-size sort (monotone in n)
    // the large sort: every form this (kind, n, sort_bits) can take, whatever "sort.msd2" says now (the knob may change between
    // the query and the sort)
    size_t e = 0;
    const size_t eb = elem_kind == ADLHIP_ELEM_U32 ? 4 : 8;
    const int max_bits = elem_kind == ADLHIP_ELEM_U64 ? 64 : 32;
    const bool whole = sort_bits == max_bits;
    const bool keys = elem_kind == ADLHIP_ELEM_U32 || elem_kind == ADLHIP_ELEM_U64;
    if (n > kMsd2Min && sort_bits >= 16) {
        if (level == 2) {   // lean: whole keys keep the cursor form with little head-room, pairs and partial sorts the stable form
            if (keys && whole && n <= (eb == 4 ? kMsd2MaxU32 : kMsd2MaxU64)) e = msd2_layout(n, eb, kLeanHeadroomPct).total;
            else if (n <= kMsd2sMax) e = msd2s_layout(n, eb, false, true).total;
        } else {
            if (n <= kMsd2sMax) e = msd2s_layout(n, eb, eb == 4 && whole).total;
            if (keys && whole && n <= (eb == 4 ? kMsd2MaxU32 : kMsd2MaxU64)) e = std::max(e, msd2_layout(n, eb).total);
        }
    }
    return std::max(std::max(a, b), std::max(c, e));
}

// Work bytes that suffice for EVERY n' <= n with the current knobs: the requirement of one n is not monotone (a smaller
// input selects a smaller tile, which needs more status rows; from 16 Mi keys the second slab of whole u32 keys moves into the
// partner array), so a caller that sizes its scratch once for its largest batch must get the maximum over the sizes at which
// such a choice changes.  Changing "sort.tile", "sort.digit_bits" or "sort.algo" afterwards can raise the requirement.
size_t sort_work_bytes(const adlhip_device* d, int elem_kind, size_t n, int sort_bits, int level)
{
    const size_t esz = (elem_kind == ADLHIP_ELEM_U32) ? 4 : 8;
    size_t need = sort_work_bytes_at(d, elem_kind, n, sort_bits, level);
    for (size_t edge : {(size_t(8) << 20) / esz, (size_t(24) << 20) / esz - 1, kSlabInTmpMin - 1, kMsd2sMax, kMsd2MaxU32})
        if (edge < n) need = std::max(need, sort_work_bytes_at(d, elem_kind, edge, sort_bits, level));
    return need;
}

// choose the path (shared by the AoS and SoA entry points)
template <typename Buf>
int run_sort(adlhip_device* d, Buf data, Buf tmp, void* work, size_t work_bytes, size_t n, const std::vector<PassPlan>& plan)
{
    const bool seven = !plan.empty() && plan[0].nbits == 7;   // 7-bit digits exist in the one-sweep pass only
    {   // a work buffer sized by the reference's contract (level 0) holds the three-kernel pass's table and nothing else
        const size_t os = onesweep_layout(d, n, max_passes_for(d, Buf::kElemBytes == 4 && !Buf::kSoa ? 32 : 64), buf_tile<Buf>(d, n)).total;
        if (work_bytes < os) {
            if (seven || d->sort_algo == 0)
                return fail("work buffer too small for the one-sweep path the knobs ask for: %zu < %zu (adlhip_radix_sort_scratch_bytes)",
                            work_bytes, os);
            return three_kernel_sort<Buf>(d, data, tmp, work, n, plan);
        }
    }
    if (d->sort_algo < 0 && !seven) {   // automatic choice by size (profiles/r1_ncurve.txt)
        // the one-sweep path has more fixed cost (histogram, tables) and wins from ~24 MiB of data (fresh random keys,
        // profiles/r1_ncurve.txt: 4Mi u32 keys 78 vs 88 us, 8Mi 124 vs 117 us, 16Mi 224 vs 174 us)
        // (AoS pairs: from ~36 MiB; 4 Mi pairs 0.103 vs 0.125 ms, 5 Mi pairs 0.149 vs 0.127 ms)
        const size_t edge = (Buf::kElemBytes == 8 && !Buf::kSoa) ? (size_t(36) << 20) : (size_t(24) << 20);
        if (n * Buf::kElemBytes < edge) return three_kernel_sort<Buf>(d, data, tmp, work, n, plan);
    }
    // tile status words carry 30-bit counts: beyond 2^30 elements use the table-based pass
    if (seven && n >= (size_t(1) << 30)) return fail("sort.digit_bits = 7 supports fewer than 2^30 elements");
    if (!seven && (d->sort_algo == 1 || n >= (size_t(1) << 30))) return three_kernel_sort<Buf>(d, data, tmp, work, n, plan);
    return onesweep_sort<Buf>(d, data, tmp, work, n, plan);
}

int check_sort_args(adlhip_device* d, int elem_kind, const void* a, const void* b, const void* work, size_t work_bytes,
                    size_t n, int sort_bits, int max_bits)
{
    if (sort_bits < 4 || sort_bits > max_bits || (sort_bits & 3))   // Pprims.cpp:330
        return fail("sort_bits must be a multiple of 4 in [4,%d], got %d", max_bits, sort_bits);
    if (n > kMaxElems) return fail("n = %zu exceeds the supported maximum %zu", n, (size_t)kMaxElems);
    if (n == 0) return ADLHIP_SUCCESS;
    if (!a || !b || !work) return fail("null buffer passed to radix sort");
    if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15u)
        return fail("sort buffers must be 16-byte aligned");
    const size_t need = sort_work_bytes(d, elem_kind, n, sort_bits, 0);
    if (work_bytes < need)
        return fail("work buffer too small: %zu < %zu (the minimum; adlhip_radix_sort_scratch_bytes gives the size for full speed)",
                    work_bytes, need);
    return ADLHIP_SUCCESS;
}

template <typename E>
int sort_entry(adlhip_device* d, int elem_kind, E* data, E* tmp, void* work, size_t work_bytes, size_t n,
               int sort_bits, int max_bits)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (check_sort_args(d, elem_kind, data, tmp, work, work_bytes, n, sort_bits, max_bits)) return ADLHIP_FAILURE;
    if (n == 0) return ADLHIP_SUCCESS;
    if (trace_level())
        fprintf(stderr, "[adlhip] sort kind %d n %zu bits %d data %p tmp %p work %p + %zu\n", elem_kind, n, sort_bits, (void*)data, (void*)tmp,
                work, work_bytes);
    const std::vector<PassPlan> plan = plan_passes(sort_bits, d->digit_bits);
    // one workgroup, one launch -- except u32 keys just below its limit, which the two-launch mid-size sort does faster (while
    // the mid-size sort's own reports do not keep them off it, choose_mid_form)
    const bool small = d->sort_algo < 0 && n <= kSmallMax;
    const bool small_mid = small && sizeof(E) == 4 && sort_bits == max_bits && n > mid_min_u32() && (d->mid_path == 2 || d->mid2_skip == 0);
    if (small && !small_mid) return small_sort<E>(d, data, n, plan);
    if (mid_eligible(d, sizeof(E), n, sort_bits, max_bits) && mid_layout(n).total <= work_bytes) {
        const int form = choose_mid_form(d, sizeof(E) == 4);
        if constexpr (sizeof(E) == 4) {
            if (form == 2) return mid_sort_keys(d, data, tmp, work, n);   // two launches
        }
        if (small) return small_sort<E>(d, data, n, plan);                // (a small input never takes the slower forms)
        if (form == 3) return mid_sort<E>(d, data, tmp, work, n);         // three launches
    }
    if (small) return small_sort<E>(d, data, n, plan);
    const bool keys = (int)sizeof(E) * 8 == max_bits;
    LargeForm form = large_sort_form(d, sizeof(E), keys, n, sort_bits, max_bits);
    int headroom = kFullHeadroomPct;
    bool lean_stable = false;
    if (form != kLargeNone && large_work_bytes(form, sizeof(E), n, sort_bits == max_bits) > work_bytes) {
        // the slabs do not fit the caller's work buffer: whole keys may still fit the cursor form (its second slab is smaller),
        // at full head-room or at the lean one (level 2 of adlhip_radix_sort_scratch_bytes_for)
        const bool whole = sort_bits == max_bits;
        const bool cursor_ok = keys && whole && n <= (sizeof(E) == 4 ? kMsd2MaxU32 : kMsd2MaxU64);
        if (cursor_ok && large_work_bytes(kLargeCursor, sizeof(E), n, true) <= work_bytes) {
            form = kLargeCursor;
        } else if (cursor_ok && msd2_layout(n, sizeof(E), kLeanHeadroomPct).total <= work_bytes) {
            form = kLargeCursor;
            headroom = kLeanHeadroomPct;
        } else if (n <= kMsd2sMax && msd2s_layout(n, sizeof(E), false, true).total <= work_bytes) {
            form = kLargeStable;   // pairs, partial sorts: the stable form with statistical head-room only
            lean_stable = true;
        } else {
            form = kLargeNone;
        }
    }
    // The large sort takes every input it is eligible for by size, bits and scratch -- whatever the keys are like: runs that do not fit
    // their slabs are detected on the device and the sort's own offsets kernel then sorts the untouched input (net_sort: counting
    // sort for few distinct values, else LSD passes with grid barriers).  Nothing is reported to the host and nothing is remembered:
    // a handle's first sort of an input takes the time its hundredth does.
    if (form != kLargeNone) {
        if (form == kLargeCursor) return msd2_sort<E>(d, data, tmp, work, n, headroom);
        const bool hybrid = form == kLargeHybrid;
        if constexpr (sizeof(E) == 4) {
            // (lean: whole u32 keys would have taken the cursor form; here the keys are sorted on part of their bits)
            return msd2s_sort<E, false>(d, data, tmp, work, n, sort_bits, nullptr, nullptr, hybrid, lean_stable);
        } else {
            if (keys) return msd2s_sort<E, true>(d, data, tmp, work, n, sort_bits, nullptr, nullptr, hybrid, lean_stable);
            return msd2s_sort<E, false>(d, data, tmp, work, n, sort_bits, nullptr, nullptr, false, lean_stable);
        }
    }
    return run_sort<AosBuf<E>>(d, AosBuf<E>{data}, AosBuf<E>{tmp}, work, work_bytes, n, plan);
}

// ---- MSB partition (multi-GPU send side) -----------------------------------------------------------
// E = uint32_t (keys) or uint64_t ({key, value} pairs: the key is the low dword, so its top byte is bits 24..31)

// One three-kernel pass on the top byte: out = in, stably ordered by bits 24..31; *totals_at (device pointer into the
// work buffer) = the 256 digit totals.
template <typename E>
int partition_top_byte(adlhip_device* d, const E* in, E* out, void* work, size_t work_bytes, size_t n, uint32_t** totals_at)
{
    if (n > kMaxElems) return fail("n too large");
    if (!in || !out) return fail("null buffer");
    const size_t need = work_bytes_three_kernel(d, n);
    if (work_bytes < need || !work) return fail("work buffer too small: %zu < %zu", work_bytes, need);
    if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) return fail("buffers must be 16-byte aligned");
    // From 24 MiB of data, and with a work buffer of the sort's full-speed size: ONE look-back pass on the top byte -- the one-sweep
    // path's histogram, tables and chain kernel with a one-pass plan -- instead of count -> scan -> scatter: the keys are read twice
    // instead of three times and there is no table scan (64 Mi keys: 0.26 -> ~0.19 ms).  Both are stable: the same output, bit for bit.
    if (d->sort_algo != 1 && d->digit_bits == 8 && d->partition_lookback && n * sizeof(E) >= (size_t(24) << 20) && n < (size_t(1) << 30) &&
        onesweep_layout(d, n, max_passes_for(d, sizeof(E) == 4 ? 32 : 64), buf_tile<AosBuf<E>>(d, n)).total <= work_bytes) {
        const std::vector<PassPlan> plan{PassPlan{24, 8}};
        return onesweep_sort<AosBuf<E>>(d, AosBuf<E>{const_cast<E*>(in)}, AosBuf<E>{out}, work, n, plan, /*copy_back=*/false, totals_at);
    }
    int rc = three_kernel_pass<AosBuf<E>, 8>(d, AosBuf<E>{const_cast<E*>(in)}, AosBuf<E>{out}, work, n, 24, /*need_totals=*/true);
    if (rc) return rc;
    *totals_at = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(work) + table_bytes(d, n, kMinTile));
    return ADLHIP_SUCCESS;
}

template <typename E>
int partition_msb(adlhip_device* d, const E* in, E* out, uint32_t* counts, void* work, size_t work_bytes, size_t n,
                         int num_buckets)
{
    if (bind(d)) return ADLHIP_FAILURE;
    int lg = 0;
    while ((1 << lg) < num_buckets) ++lg;
    if (num_buckets < 1 || num_buckets > 256 || (1 << lg) != num_buckets)
        return fail("num_buckets must be a power of two in [1,256], got %d", num_buckets);
    if (!counts) return fail("null counts pointer");
    if (n > kMaxElems) return fail("n too large");
    if (n == 0 || num_buckets == 1) {
        HIPCHK(hipMemsetAsync(counts, 0, 4 * (size_t)num_buckets, d->stream));
        if (n) {
            if (!in || !out) return fail("null buffer");
            HIPCHK(hipMemcpyAsync(out, in, n * sizeof(E), hipMemcpyDeviceToDevice, d->stream));
            uint32_t nn = (uint32_t)n;
            // counts[0] = n  (stream-ordered fill of one word)
            int rc = launch(d, "fill_u32", [&] {
                hipLaunchKernelGGL(adlhip::fill_u32_kernel, dim3(1), dim3(256), 0, d->stream, counts, nn, (size_t)1);
            });
            if (rc) return rc;
        }
        return ADLHIP_SUCCESS;
    }
    // one three-kernel pass on the top byte (the top `lg` bits decide the bucket; ordering by the
    // whole top byte refines buckets without mixing them), then fold the 256 digit totals into buckets
    uint32_t* totals = nullptr;
    int rc = partition_top_byte<E>(d, in, out, work, work_bytes, n, &totals);
    if (rc) return rc;
    return launch(d, "fold_buckets", [&] {
        hipLaunchKernelGGL(adlhip::fold_buckets_kernel, dim3(1), dim3(256), 0, d->stream, (const uint32_t*)totals, counts, num_buckets);
    });
}

// the same pass with the 256 top-byte totals handed to the caller (who chooses balanced splitters from them)
template <typename E>
int partition_top_byte_entry(adlhip_device* d, const E* in, E* out, uint32_t* totals256, void* work, size_t work_bytes, size_t n)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!totals256) return fail("null totals pointer");
    if (n == 0) {
        HIPCHK(hipMemsetAsync(totals256, 0, 256 * 4, d->stream));
        return ADLHIP_SUCCESS;
    }
    uint32_t* totals = nullptr;
    int rc = partition_top_byte<E>(d, in, out, work, work_bytes, n, &totals);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(totals256, totals, 256 * 4, hipMemcpyDeviceToDevice, d->stream));
    return ADLHIP_SUCCESS;
}

// ---- SoA key-value sort with 4- / 8-byte keys and 4- / 8- / 16-byte values (soa_wide_kernels.hpp) -----------------------------
struct SoaWideLayout {
    size_t off_pairs_a, off_pairs_b, off_kv, kv_bytes, total;
};
SoaWideLayout soa_wide_layout(const adlhip_device* d, size_t n)
{
    SoaWideLayout L;
    L.off_pairs_a = 0;
    L.off_pairs_b = align_up(n * 8, 256);
    L.off_kv = 2 * L.off_pairs_b;
    L.kv_bytes = sort_work_bytes(d, ADLHIP_ELEM_KV32, n, 32, 1);   // (whole keys need the least of the large sort's forms; partial
    L.kv_bytes = std::max(L.kv_bytes, sort_work_bytes(d, ADLHIP_ELEM_KV32, n, 28, 1));   // sorts the most: size for either)
    L.total = L.off_kv + L.kv_bytes;
    return L;
}

struct V16 {
    uint32_t x, y, z, w;
} __attribute__((aligned(16)));

template <typename K, typename V>
int soa_wide_sort(adlhip_device* d, K* keys, V* vals, K* tmp_keys, V* tmp_vals, void* work, size_t n, int sort_bits)
{
    const SoaWideLayout L = soa_wide_layout(d, n);
    char* w = static_cast<char*>(work);
    uint64_t* pa = reinterpret_cast<uint64_t*>(w + L.off_pairs_a);
    uint64_t* pb = reinterpret_cast<uint64_t*>(w + L.off_pairs_b);
    void* kv = w + L.off_kv;
    const uint32_t nn = (uint32_t)n;
    const uint32_t wgs = (uint32_t)std::min<size_t>((n + adlhip::kSoaNT - 1) / adlhip::kSoaNT, (size_t)d->prop.multiProcessorCount * 16);
    int rc = launch(d, sizeof(K) == 4 ? "soa_pack_index_k32" : "soa_pack_index_k64", [&] {
        hipLaunchKernelGGL(adlhip::soa_pack_index_kernel<K>, dim3(wgs), dim3(adlhip::kSoaNT), 0, d->stream, (const K*)keys, pa, nn);
    });
    if (rc) return rc;
    // the stable pair sort of Pprims::radixSort(Buffer<uint2>) (Pprims.h:38) on {low key dword, index}
    rc = sort_entry<uint64_t>(d, ADLHIP_ELEM_KV32, pa, pb, kv, L.kv_bytes, n, std::min(sort_bits, 32), 32);
    if (rc) return rc;
    const uint64_t* sorted = pa;
    if constexpr (sizeof(K) == 8) {
        if (sort_bits > 32) {   // second 32-bit digit; the sort is stable, so equal high dwords keep the order of their low dwords
            rc = launch(d, "soa_repack_high", [&] {
                hipLaunchKernelGGL(adlhip::soa_repack_high_kernel, dim3(wgs), dim3(adlhip::kSoaNT), 0, d->stream, (const uint64_t*)keys,
                                   (const uint64_t*)pa, pb, nn);
            });
            if (rc) return rc;
            rc = sort_entry<uint64_t>(d, ADLHIP_ELEM_KV32, pb, pa, kv, L.kv_bytes, n, sort_bits - 32, 32);
            if (rc) return rc;
            sorted = pb;
        }
    }
    // u32 keys are the pairs' own low dwords and go straight to the caller's array; everything that is gathered lands in the
    // partner arrays first (the gather reads its source out of order) and is copied back (Pprims.cpp:298-301 copies back too)
    K* kout = sizeof(K) == 4 ? keys : tmp_keys;
    rc = launch(d, "soa_gather", [&] {
        hipLaunchKernelGGL((adlhip::soa_gather_kernel<K, V>), dim3(wgs), dim3(adlhip::kSoaNT), 0, d->stream, sorted, (const K*)keys, kout,
                           (const V*)vals, tmp_vals, nn);
    });
    if (rc) return rc;
    if (sizeof(K) == 8) HIPCHK(hipMemcpyAsync(keys, tmp_keys, n * sizeof(K), hipMemcpyDeviceToDevice, d->stream));
    HIPCHK(hipMemcpyAsync(vals, tmp_vals, n * sizeof(V), hipMemcpyDeviceToDevice, d->stream));
    return ADLHIP_SUCCESS;
}

int soa_check_widths(int key_bytes, int value_bytes)
{
    if (key_bytes != 4 && key_bytes != 8) return fail("key_bytes must be 4 or 8, got %d", key_bytes);
    if (value_bytes != 4 && value_bytes != 8 && value_bytes != 16) return fail("value_bytes must be 4, 8 or 16, got %d", value_bytes);
    return ADLHIP_SUCCESS;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* adlhip_version(void) { return "adlhip 0.2 (gfx950)"; }
const char* adlhip_last_error(void) { return g_err; }
// internal (not in include/adlhip.h): lets the library's other translation unit (sharded.cpp) set the calling thread's
// error text
void adlhip_set_last_error(const char* text) { snprintf(g_err, sizeof(g_err), "%s", text ? text : ""); }

int adlhip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// 256-thread and 1024-thread workgroups; blocks the calling thread until the result is back
static int run_lds_order_selftest(adlhip_device* d, int workgroups, uint32_t* mismatches)
{
    static uint32_t salt = 0;
    uint32_t* slot = d->d_fault + 8;
    HIPCHK(hipMemsetAsync(slot, 0, 4, d->stream));
    hipLaunchKernelGGL(adlhip::lds_order_selftest_kernel, dim3(workgroups), dim3(256), 0, d->stream, slot, ++salt);
    hipLaunchKernelGGL(adlhip::lds_order_selftest_kernel, dim3(workgroups), dim3(1024), 0, d->stream, slot, ++salt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(d->h_fault + 8, slot, 4, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    *mismatches = d->h_fault[8];
    return ADLHIP_SUCCESS;
}

int adlhip_selftest_lds_order(adlhip_device* d, int workgroups, uint32_t* mismatches)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!mismatches) return fail("null out pointer");
    if (workgroups < 1 || workgroups > 65536) return fail("selftest: workgroups must be in [1,65536]");
    return run_lds_order_selftest(d, workgroups, mismatches);
}

int adlhip_selftest_probe_positions(adlhip_device* d, size_t n, uint32_t* max_index, uint32_t* out_of_cell)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!max_index || !out_of_cell) return fail("null out pointer");
    if (n < 16384 || n > kMaxElems) return fail("selftest: n must be in [16384, %zu]", (size_t)kMaxElems);
    // two words of the dictionary block's count[] (idle between sorts: zero) serve as the result
    uint32_t* res = d->d_dict->count;
    HIPCHK(hipMemsetAsync(res, 0, 8, d->stream));
    int rc = launch(d, "probe_positions_selftest", [&] {
        hipLaunchKernelGGL(adlhip::probe_positions_selftest_kernel, dim3(1), dim3(1024), 0, d->stream, (uint32_t)n, res);
    });
    if (rc) return rc;
    uint32_t h[2] = {0u, 0u};
    HIPCHK(hipMemcpyAsync(h, res, 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipMemsetAsync(res, 0, 8, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    *max_index = h[0];
    *out_of_cell = h[1];
    return ADLHIP_SUCCESS;
}

static int create_common(int device_idx, void* stream, bool own, adlhip_device** out)
{
    if (!out) return fail("null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail("no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_idx < 0) device_idx = 0;
    if (device_idx >= n) device_idx = n - 1;   // AdlCL.inl:244 clamps the same way
    adlhip_device* d = new adlhip_device();
    d->idx = device_idx;
    if (hipSetDevice(device_idx) != hipSuccess || hipGetDeviceProperties(&d->prop, device_idx) != hipSuccess) {
        delete d;
        return fail("cannot bind HIP device %d", device_idx);
    }
    if (strncmp(d->prop.gcnArchName, "gfx950", 6) != 0 && !getenv("ADLHIP_ALLOW_ANY_ARCH")) {
        std::string arch = d->prop.gcnArchName;
        delete d;
        return fail("adlhip is built for gfx950 only; device %d is %s", device_idx, arch.c_str());
    }
    d->own_stream = own;
    if (own) {
        if (hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
            delete d;
            return fail("hipStreamCreate failed");
        }
    } else {
        d->stream = reinterpret_cast<hipStream_t>(stream);
    }
    if (hipMalloc(&d->d_fault, 64) != hipSuccess || hipHostMalloc(&d->h_fault, 64) != hipSuccess ||
        hipMemsetAsync(d->d_fault, 0, 64, d->stream) != hipSuccess) {
        if (d->d_fault) hipFree(d->d_fault);
        if (d->h_fault) hipHostFree(d->h_fault);
        if (own && d->stream) hipStreamDestroy(d->stream);
        delete d;
        return fail("cannot allocate the fault word");
    }
    memset(d->h_fault, 0, 64);
    if (hipMalloc(&d->d_mid_hist, (16 * 1024 + 8192 + 64) * 4) != hipSuccess ||
        hipMemsetAsync(d->d_mid_hist, 0, (16 * 1024 + 8192 + 64) * 4, d->stream) != hipSuccess) {
        if (d->d_mid_hist) hipFree(d->d_mid_hist);
        hipFree(d->d_fault);
        hipHostFree(d->h_fault);
        if (own && d->stream) hipStreamDestroy(d->stream);
        delete d;
        return fail("cannot allocate the mid-size sort's histogram area");
    }
    // the large sort's handle-owned words (cursors, flags: 270 KB), zero between sorts -- allocated here rather than by a handle's
    // first large sort
    if (hipMalloc(&d->d_msd2, kMsd2Words * 4) != hipSuccess ||
        hipMemsetAsync(d->d_msd2, 0, kMsd2Words * 4, d->stream) != hipSuccess ||
        hipMemsetAsync(d->d_msd2 + 8192 + 65536 + 10, 0xff, 8, d->stream) != hipSuccess) {   // the sample's AND words: all ones when idle
        if (d->d_msd2) hipFree(d->d_msd2);
        hipFree(d->d_mid_hist);
        hipFree(d->d_fault);
        hipHostFree(d->h_fault);
        if (own && d->stream) hipStreamDestroy(d->stream);
        delete d;
        return fail("cannot allocate the large sort's cursors");
    }
    // (the larger dictionary of dict_big_kernels.hpp lies behind the small one: the kernels find it at d_dict + 1)
    static_assert(sizeof(adlhip::DictBlock) % 16 == 0, "the second block starts aligned");
    if (hipMalloc(&d->d_dict, sizeof(adlhip::DictBlock) + sizeof(adlhip::BigDictBlock)) != hipSuccess ||
        hipMemsetAsync(d->d_dict, 0, sizeof(adlhip::DictBlock) + sizeof(adlhip::BigDictBlock), d->stream) != hipSuccess) {
        if (d->d_dict) hipFree(d->d_dict);
        hipFree(d->d_msd2);
        hipFree(d->d_mid_hist);
        hipFree(d->d_fault);
        hipHostFree(d->h_fault);
        if (own && d->stream) hipStreamDestroy(d->stream);
        delete d;
        return fail("cannot allocate the counting sort's dictionary");
    }
    if (trace_level())
        fprintf(stderr, "[adlhip] handle %p: d_fault %p h_fault %p d_mid_hist %p d_msd2 %p d_dict %p\n", (void*)d, (void*)d->d_fault,
                (void*)d->h_fault, (void*)d->d_mid_hist, (void*)d->d_msd2, (void*)d->d_dict);
    {   // self-test: are returning DS atomics lane-ordered on this device?  (enables "sort.rank" = 1)
        uint32_t mism = 1;
        d->lds_ordered = (run_lds_order_selftest(d, 64, &mism) == ADLHIP_SUCCESS && mism == 0) ? 1 : 0;
        d->rank_mode = d->lds_ordered;
    }
    {   // how many workgroups of the kernels that hold a grid-wide barrier (the safety nets of the mid-size and large sorts, 256
        // workgroups each) this device keeps resident at once: asked of the runtime for the two largest of them
        int a = 0, b = 0;
        using CS = adlhip::TileCfg<uint32_t, 8, 512, 32>;
        using CO = adlhip::TileCfg<uint32_t, 8, 512, 32>;
        const size_t lds_s = std::max<size_t>(sizeof(uint32_t) * 512 * 32 + (size_t)8 * 256 * 6 + 64, CS::LDS_BYTES);
        auto ks = adlhip::segment_sort_kernel<uint32_t, 512, 32, 8>;
        auto ko = adlhip::msd2_offsets_kernel<uint32_t, 512, 32>;   // (the net's look-back passes)
        // (the most LDS of all: the pairs' tile + the dictionary values of coop_dict_pair_sort)
        using CP = adlhip::TileCfg<uint64_t, 8, 512, 16>;
        auto kp = adlhip::msd2s_offsets_kernel<uint64_t, CP::TILE, 512, 16, 1, 4>;
        int c = 0;
        if (ensure_lds(ks, lds_s) == ADLHIP_SUCCESS && ensure_lds(ko, CO::LDS_BYTES) == ADLHIP_SUCCESS &&
            ensure_lds(kp, CP::LDS_BYTES) == ADLHIP_SUCCESS &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, reinterpret_cast<const void*>(ks), 512, lds_s) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, reinterpret_cast<const void*>(ko), 512, CO::LDS_BYTES) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, reinterpret_cast<const void*>(kp), 512, CP::LDS_BYTES) == hipSuccess)
            d->resident_wgs = std::min(std::min(a, b), c) * d->prop.multiProcessorCount;
        else
            d->resident_wgs = d->prop.multiProcessorCount;   // one per CU at least
        d->resident_wgs_device = d->resident_wgs;
        (void)hipGetLastError();
    }
    if (const char* a = getenv("ADLHIP_SORT_ALGO")) { int v = atoi(a); if (v >= -1 && v <= 1) d->sort_algo = v; }
    if (const char* t = getenv("ADLHIP_SORT_TILE")) { int v = atoi(t); if (v >= -1 && v < kNumVariants) d->tile_variant = v; }
    if (const char* r = getenv("ADLHIP_SORT_RANK")) d->rank_mode = (atoi(r) && d->lds_ordered) ? 1 : 0;
    if (const char* m = getenv("ADLHIP_SORT_MID")) d->mid_path = atoi(m) ? 1 : 0;
    if (const char* m = getenv("ADLHIP_SORT_MSD2")) { int v = atoi(m); if (v >= 0 && v <= 2) d->msd2_path = v; }
    if (const char* b = getenv("ADLHIP_DIGIT_BITS")) { int v = atoi(b); d->digit_bits = (v == 4 || v == 7) ? v : 8; }
    *out = d;
    return ADLHIP_SUCCESS;
}

int adlhip_device_create(int device_idx, adlhip_device** out) { return create_common(device_idx, nullptr, true, out); }

int adlhip_device_create_on_stream(int device_idx, void* hip_stream, adlhip_device** out)
{
    return create_common(device_idx, hip_stream, false, out);
}

int adlhip_device_destroy(adlhip_device* d)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (d->used_bytes != 0)   // Adl.inl:102 ADLASSERT( getUsedMemory() == 0 )
        return fail("device still has %llu live bytes; free every buffer first", (unsigned long long)d->used_bytes);
    hipStreamSynchronize(d->stream);
    reap_staging(d, true);
    for (auto& s : d->staging) hipHostFree(s.hptr);
    for (auto& b : d->pinned_pool) hipHostFree(b.hptr);
    for (auto& p : d->pending) { hipEventDestroy(p.e0); hipEventDestroy(p.e1); }
    for (auto e : d->event_pool) hipEventDestroy(e);
    if (d->fault_snap) hipEventDestroy(d->fault_snap);
    hipFree(d->d_mid_hist);
    if (d->d_msd2) hipFree(d->d_msd2);
    if (d->d_dict) hipFree(d->d_dict);
    hipFree(d->d_fault);
    hipHostFree(d->h_fault);
    if (d->own_stream) hipStreamDestroy(d->stream);
    delete d;
    return ADLHIP_SUCCESS;
}

int adlhip_device_info(adlhip_device* d, adlhip_info* out)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!out) return fail("null info pointer");
    memset(out, 0, sizeof(*out));
    out->compute_units = d->prop.multiProcessorCount;
    out->wavefront_size = d->prop.warpSize;
    out->lds_bytes_per_cu = (int32_t)d->prop.maxSharedMemoryPerMultiProcessor;
    out->clock_khz = d->prop.clockRate;
    out->total_mem_bytes = d->prop.totalGlobalMem;
    out->max_alloc_bytes = d->prop.totalGlobalMem;
    snprintf(out->name, sizeof(out->name), "%s", d->prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", d->prop.gcnArchName);
    snprintf(out->vendor, sizeof(out->vendor), "Advanced Micro Devices, Inc.");
    return ADLHIP_SUCCESS;
}

uint64_t adlhip_used_bytes(adlhip_device* d) { return d ? d->used_bytes : 0; }

void* adlhip_stream(adlhip_device* d) { return d ? (void*)d->stream : nullptr; }

static int report_fault(adlhip_device* d, uint32_t code)
{
    hipMemsetAsync(d->d_fault, 0, 8, d->stream);
    if (code & 0x40000u)
        return fail("device-side fault 0x%x (a segment exceeded the LDS tile of the finishing pass); results are invalid", code);
    return fail("device-side fault 0x%x (look-back wait exceeded its bound); results are invalid", code);
}

int adlhip_sync(adlhip_device* d)
{
    if (bind(d)) return ADLHIP_FAILURE;
    // pick up the device-side fault words together with the drain
    HIPCHK(hipMemcpyAsync(d->h_fault, d->d_fault, 8, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipStreamSynchronize(d->stream));
    reap_staging(d, true);
    uint32_t code = d->h_fault[1];
    if (d->fault_snap) {   // a snapshot taken before this drain has landed too
        code |= d->h_fault[4];
        d->h_fault[4] = 0;
        hipEventDestroy(d->fault_snap);
        d->fault_snap = nullptr;
    }
    if (code != 0) {
        d->h_fault[0] = d->h_fault[1] = 0;
        return report_fault(d, code);
    }
    return ADLHIP_SUCCESS;
}

int adlhip_fault_check(adlhip_device* d)
{
    if (bind(d)) return ADLHIP_FAILURE;
    uint32_t code = 0;
    if (d->fault_snap) {
        if (hipEventQuery(d->fault_snap) != hipSuccess) return ADLHIP_SUCCESS;   // the last snapshot is still in flight
        code = d->h_fault[4];
        d->h_fault[4] = 0;
    } else {
        HIPCHK(hipEventCreateWithFlags(&d->fault_snap, hipEventDisableTiming));
    }
    if (code != 0) report_fault(d, code);   // clears the device words BEFORE the next snapshot is taken: reported once
    HIPCHK(hipMemcpyAsync(d->h_fault + 4, d->d_fault + 1, 4, hipMemcpyDeviceToHost, d->stream));
    HIPCHK(hipEventRecord(d->fault_snap, d->stream));
    return code != 0 ? ADLHIP_FAILURE : ADLHIP_SUCCESS;
}

int adlhip_flush(adlhip_device* d) { return bind(d); }

int adlhip_malloc(adlhip_device* d, size_t bytes, void** dptr)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!dptr) return fail("null out pointer");
    *dptr = nullptr;
    if (bytes == 0) return ADLHIP_SUCCESS;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e != hipSuccess) {   // AdlCL.inl:390-406: log, leave m_ptr = 0
        *dptr = nullptr;
        return fail("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    d->used_bytes += bytes;
    if (trace_level()) fprintf(stderr, "[adlhip] malloc %p + %zu\n", *dptr, bytes);
    return ADLHIP_SUCCESS;
}

int adlhip_free(adlhip_device* d, void* dptr, size_t bytes)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!dptr) return ADLHIP_SUCCESS;
    if (trace_level()) fprintf(stderr, "[adlhip] free %p + %zu\n", dptr, bytes);
    HIPCHK(hipStreamSynchronize(d->stream));   // queued work may still use it
    HIPCHK(hipFree(dptr));
    d->used_bytes -= std::min<uint64_t>(bytes, d->used_bytes);
    return ADLHIP_SUCCESS;
}

int adlhip_memcpy_h2d(adlhip_device* d, void* dst, const void* src, size_t bytes)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (bytes == 0) return ADLHIP_SUCCESS;
    if (trace_level()) fprintf(stderr, "[adlhip] h2d %p <- %p + %zu\n", dst, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->stream));
    return ADLHIP_SUCCESS;
}

int adlhip_memcpy_d2h(adlhip_device* d, void* dst, const void* src, size_t bytes)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (bytes == 0) return ADLHIP_SUCCESS;
    if (trace_level()) fprintf(stderr, "[adlhip] d2h %p <- %p + %zu\n", dst, src, bytes);
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, d->stream));
    return ADLHIP_SUCCESS;
}

int adlhip_memcpy_d2d(adlhip_device* d, void* dst, const void* src, size_t bytes)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (bytes == 0) return ADLHIP_SUCCESS;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, d->stream));
    return ADLHIP_SUCCESS;
}

int adlhip_memset(adlhip_device* d, void* dptr, int byte_value, size_t bytes)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (bytes == 0) return ADLHIP_SUCCESS;
    HIPCHK(hipMemsetAsync(dptr, byte_value, bytes, d->stream));
    return ADLHIP_SUCCESS;
}

int adlhip_fill_u32(adlhip_device* d, void* dptr, uint32_t pattern, size_t count)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (count == 0) return ADLHIP_SUCCESS;
    const int grid = (int)std::min<size_t>((count + 255) / 256, (size_t)d->prop.multiProcessorCount * 8);
    return launch(d, "fill_u32", [&] {
        hipLaunchKernelGGL(adlhip::fill_u32_kernel, dim3(grid), dim3(256), 0, d->stream, (uint32_t*)dptr, pattern, count);
    });
}

int adlhip_fill_pattern(adlhip_device* d, void* dptr, const void* pattern, size_t pattern_bytes, size_t count)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (pattern_bytes != 4 && pattern_bytes != 8 && pattern_bytes != 16) return fail("fill: pattern of %zu bytes (4, 8 or 16)", pattern_bytes);
    if (count == 0) return ADLHIP_SUCCESS;
    if (!dptr || !pattern) return fail("fill: null pointer");
    if (reinterpret_cast<uintptr_t>(dptr) % pattern_bytes) return fail("fill: destination not aligned to the pattern size");
    if (pattern_bytes == 4) {
        uint32_t v;
        memcpy(&v, pattern, 4);
        return adlhip_fill_u32(d, dptr, v, count);
    }
    uint4 p16;
    uint2 p8 = make_uint2(0u, 0u);
    uint2* tail = nullptr;
    uint4* dst = reinterpret_cast<uint4*>(dptr);
    size_t vecs = count;
    if (pattern_bytes == 16) {
        memcpy(&p16, pattern, 16);
    } else {
        memcpy(&p8, pattern, 8);
        p16 = make_uint4(p8.x, p8.y, p8.x, p8.y);
        uint2* d8 = reinterpret_cast<uint2*>(dptr);
        size_t c = count;
        if (reinterpret_cast<uintptr_t>(dptr) & 8u) {   // leading odd element: written through the tail slot of a first launch
            int rc = launch(d, "fill_pattern", [&] {
                hipLaunchKernelGGL(adlhip::fill_pattern16_kernel, dim3(1), dim3(256), 0, d->stream, (uint4*)nullptr, p16, (size_t)0, d8, p8);
            });
            if (rc) return rc;
            ++d8;
            if (--c == 0) return ADLHIP_SUCCESS;
        }
        dst = reinterpret_cast<uint4*>(d8);
        vecs = c / 2;
        if (c & 1) tail = d8 + (c - 1);
    }
    const int grid = (int)std::max<size_t>(1, std::min<size_t>((vecs + 255) / 256, (size_t)d->prop.multiProcessorCount * 8));
    return launch(d, "fill_pattern", [&] {
        hipLaunchKernelGGL(adlhip::fill_pattern16_kernel, dim3(grid), dim3(256), 0, d->stream, dst, p16, vecs, tail, p8);
    });
}

int adlhip_map(adlhip_device* d, void* dptr, size_t bytes, void** hptr)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!hptr) return fail("null out pointer");
    *hptr = nullptr;
    reap_staging(d, false);
    if (bytes == 0) return ADLHIP_SUCCESS;
    void* h = nullptr;
    size_t cap = bytes;
    for (size_t i = 0; i < d->pinned_pool.size(); ++i) {   // smallest pooled block that fits without wasting more than half
        const PinnedBlock b = d->pinned_pool[i];
        if (b.capacity >= bytes && b.capacity <= 2 * bytes + 4096 && (!h || b.capacity < cap)) {
            h = b.hptr;
            cap = b.capacity;
        }
    }
    if (h) {
        for (size_t i = 0; i < d->pinned_pool.size(); ++i)
            if (d->pinned_pool[i].hptr == h) {
                d->pinned_pool[i] = d->pinned_pool.back();
                d->pinned_pool.pop_back();
                break;
            }
        d->pinned_pool_bytes -= cap;
    } else {
        HIPCHK(hipHostMalloc(&h, bytes));
    }
    hipError_t e = hipMemcpyAsync(h, dptr, bytes, hipMemcpyDeviceToHost, d->stream);
    if (e != hipSuccess) {
        hipHostFree(h);
        return fail("map: device->host copy failed: %s", hipGetErrorString(e));
    }
    d->staging.push_back({h, bytes, nullptr, cap});
    *hptr = h;
    return ADLHIP_SUCCESS;
}

int adlhip_unmap(adlhip_device* d, void* dptr, void* hptr, size_t bytes)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!hptr) return ADLHIP_SUCCESS;
    for (auto& s : d->staging) {
        if (s.hptr == hptr && !s.done) {
            if (bytes > s.bytes) return fail("unmap: %zu bytes exceeds the mapped %zu", bytes, s.bytes);
            if (bytes == 0) bytes = s.bytes;   // 0 = the whole mapping (Buffer::returnHostPtr carries no size)
            HIPCHK(hipMemcpyAsync(dptr, hptr, bytes, hipMemcpyHostToDevice, d->stream));
            hipEvent_t ev;
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ev, d->stream));
            s.done = ev;
            return ADLHIP_SUCCESS;
        }
    }
    return fail("unmap: pointer %p was not returned by adlhip_map", hptr);
}

// ---- sort ---------------------------------------------------------------------------------------

int adlhip_radix_sort_scratch_bytes_for(adlhip_device* d, int elem_kind, size_t n, int sort_bits, int level, size_t* tmp_bytes,
                                        size_t* work_bytes)
{
    if (!d) return fail("null device handle");
    if (elem_kind < ADLHIP_ELEM_U32 || elem_kind > ADLHIP_ELEM_SOA32) return fail("bad element kind %d", elem_kind);
    const int max_bits = elem_kind == ADLHIP_ELEM_U64 ? 64 : 32;
    if (sort_bits < 4 || sort_bits > max_bits || (sort_bits & 3)) return fail("sort_bits must be a multiple of 4 in [4,%d], got %d", max_bits, sort_bits);
    if (level < 0 || level > 2) return fail("level must be 0 (minimum), 1 (full speed) or 2 (lean), got %d", level);
    const size_t esz = (elem_kind == ADLHIP_ELEM_U32 || elem_kind == ADLHIP_ELEM_SOA32) ? 4 : 8;   // SoA: per array
    if (tmp_bytes) *tmp_bytes = align_up(n * esz, 256);
    if (work_bytes) *work_bytes = sort_work_bytes(d, elem_kind, n, sort_bits, level);
    return ADLHIP_SUCCESS;
}

int adlhip_radix_sort_scratch_bytes(adlhip_device* d, int elem_kind, size_t n, size_t* tmp_bytes, size_t* work_bytes)
{
    return adlhip_radix_sort_scratch_bytes_for(d, elem_kind, n, elem_kind == ADLHIP_ELEM_U64 ? 64 : 32, 1, tmp_bytes, work_bytes);
}

int adlhip_radix_sort_u32(adlhip_device* d, uint32_t* keys, uint32_t* tmp, void* work, size_t work_bytes, size_t n, int sort_bits)
{
    return sort_entry<uint32_t>(d, ADLHIP_ELEM_U32, keys, tmp, work, work_bytes, n, sort_bits, 32);
}

int adlhip_radix_sort_kv32(adlhip_device* d, void* pairs, void* tmp, void* work, size_t work_bytes, size_t n, int sort_bits)
{
    return sort_entry<uint64_t>(d, ADLHIP_ELEM_KV32, (uint64_t*)pairs, (uint64_t*)tmp, work, work_bytes, n, sort_bits, 32);
}

int adlhip_radix_sort_u64(adlhip_device* d, uint64_t* keys, uint64_t* tmp, void* work, size_t work_bytes, size_t n, int sort_bits)
{
    return sort_entry<uint64_t>(d, ADLHIP_ELEM_U64, keys, tmp, work, work_bytes, n, sort_bits, 64);
}

int adlhip_radix_sort_soa32(adlhip_device* d, uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals,
                            void* work, size_t work_bytes, size_t n, int sort_bits)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (check_sort_args(d, ADLHIP_ELEM_SOA32, keys, tmp_keys, work, work_bytes, n, sort_bits, 32)) return ADLHIP_FAILURE;
    if (n == 0) return ADLHIP_SUCCESS;
    if (!vals || !tmp_vals) return fail("null value buffer passed to radix sort");
    if ((reinterpret_cast<uintptr_t>(vals) | reinterpret_cast<uintptr_t>(tmp_vals)) & 15u)
        return fail("sort buffers must be 16-byte aligned");
    if (large_sort_form(d, 8, false, n, sort_bits, 32) != kLargeNone) {
        if (large_work_bytes(kLargeStable, 8, n, sort_bits == 32) <= work_bytes)
            return msd2s_sort<uint64_t, false>(d, nullptr, nullptr, work, n, sort_bits, keys, vals);
        if (msd2s_layout(n, 8, false, true).total <= work_bytes)   // level 2, lean
            return msd2s_sort<uint64_t, false>(d, nullptr, nullptr, work, n, sort_bits, keys, vals, false, true);
    }
    const std::vector<PassPlan> plan = plan_passes(sort_bits, d->digit_bits);
    return run_sort<SoaBuf>(d, SoaBuf{keys, vals}, SoaBuf{tmp_keys, tmp_vals}, work, work_bytes, n, plan);
}

int adlhip_radix_sort_soa_scratch_bytes(adlhip_device* d, int key_bytes, int value_bytes, size_t n, int sort_bits, size_t* tmp_keys_bytes,
                                        size_t* tmp_vals_bytes, size_t* work_bytes)
{
    if (!d) return fail("null device handle");
    if (soa_check_widths(key_bytes, value_bytes)) return ADLHIP_FAILURE;
    if (sort_bits < 4 || sort_bits > 8 * key_bytes || (sort_bits & 3)) return fail("sort_bits must be a multiple of 4 in [4,%d], got %d", 8 * key_bytes, sort_bits);
    if (tmp_keys_bytes) *tmp_keys_bytes = align_up(n * (size_t)key_bytes, 256);
    if (tmp_vals_bytes) *tmp_vals_bytes = align_up(n * (size_t)value_bytes, 256);
    if (work_bytes) {
        if (key_bytes == 4 && value_bytes == 4) *work_bytes = sort_work_bytes(d, ADLHIP_ELEM_SOA32, n, sort_bits, 1);
        else *work_bytes = soa_wide_layout(d, n).total;
    }
    return ADLHIP_SUCCESS;
}

int adlhip_radix_sort_soa(adlhip_device* d, void* keys, int key_bytes, void* vals, int value_bytes, void* tmp_keys, void* tmp_vals,
                          void* work, size_t work_bytes, size_t n, int sort_bits)
{
    if (soa_check_widths(key_bytes, value_bytes)) return ADLHIP_FAILURE;
    if (key_bytes == 4 && value_bytes == 4)
        return adlhip_radix_sort_soa32(d, (uint32_t*)keys, (uint32_t*)vals, (uint32_t*)tmp_keys, (uint32_t*)tmp_vals, work, work_bytes, n, sort_bits);
    if (bind(d)) return ADLHIP_FAILURE;
    if (sort_bits < 4 || sort_bits > 8 * key_bytes || (sort_bits & 3))   // Pprims.cpp:330
        return fail("sort_bits must be a multiple of 4 in [4,%d], got %d", 8 * key_bytes, sort_bits);
    if (n > kMaxElems || n >= (size_t(1) << 32)) return fail("n = %zu exceeds the supported maximum", n);
    if (n == 0) return ADLHIP_SUCCESS;
    if (!keys || !vals || !tmp_vals || !work || (key_bytes == 8 && !tmp_keys)) return fail("null buffer passed to radix sort");
    if ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(vals) | reinterpret_cast<uintptr_t>(tmp_keys) |
         reinterpret_cast<uintptr_t>(tmp_vals) | reinterpret_cast<uintptr_t>(work)) & 15u)
        return fail("sort buffers must be 16-byte aligned");
    const size_t need = soa_wide_layout(d, n).total;
    if (work_bytes < need) return fail("work buffer too small: %zu < %zu (adlhip_radix_sort_soa_scratch_bytes)", work_bytes, need);
#define ADLHIP_SOA(K_, V_) return soa_wide_sort<K_, V_>(d, (K_*)keys, (V_*)vals, (K_*)tmp_keys, (V_*)tmp_vals, work, n, sort_bits)
    if (key_bytes == 4) {
        if (value_bytes == 8) ADLHIP_SOA(uint32_t, uint64_t);
        ADLHIP_SOA(uint32_t, V16);
    }
    if (value_bytes == 4) ADLHIP_SOA(uint64_t, uint32_t);
    if (value_bytes == 8) ADLHIP_SOA(uint64_t, uint64_t);
    ADLHIP_SOA(uint64_t, V16);
#undef ADLHIP_SOA
}

int adlhip_segment_sort(adlhip_device* d, int elem_kind, void* data, const uint32_t* seg_start, size_t num_segments,
                        size_t max_segment, int low_bits)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (elem_kind != ADLHIP_ELEM_U32 && elem_kind != ADLHIP_ELEM_KV32) return fail("segment sort: u32 keys or {key, value} pairs only");
    if (low_bits < 1 || low_bits > 27) return fail("segment sort: low_bits must be in [1,27], got %d", low_bits);
    if (num_segments == 0) return ADLHIP_SUCCESS;
    if (!data || !seg_start) return fail("null buffer passed to segment sort");
    if (num_segments > 0x7fffffffull) return fail("segment sort: too many segments");
    if (!d->lds_ordered) return fail("segment sort needs lane-ordered DS atomics; the device self-test failed");
    if (elem_kind == ADLHIP_ELEM_U32)
        return segment_sort<uint32_t>(d, static_cast<const uint32_t*>(data), static_cast<uint32_t*>(data), seg_start, num_segments,
                                      max_segment, low_bits, nullptr);
    return segment_sort<uint64_t>(d, static_cast<const uint64_t*>(data), static_cast<uint64_t*>(data), seg_start, num_segments,
                                  max_segment, low_bits, nullptr);
}

// ---- scan ---------------------------------------------------------------------------------------

int adlhip_scan_scratch_bytes(adlhip_device* d, size_t n, size_t* work_bytes)
{
    if (!d) return fail("null device handle");
    const size_t blocks = (n + adlhip::kScanTile - 1) / adlhip::kScanTile;
    if (work_bytes) *work_bytes = align_up((blocks + 2) * 4, 256);
    return ADLHIP_SUCCESS;
}

int adlhip_exclusive_scan_u32(adlhip_device* d, uint32_t* dst, const uint32_t* src, void* work, size_t work_bytes,
                              size_t n, uint32_t* h_sum)
{
    if (bind(d)) return ADLHIP_FAILURE;
    size_t need = 0;
    adlhip_scan_scratch_bytes(d, n, &need);
    if (work_bytes < need || !work) return fail("scan work buffer too small: %zu < %zu", work_bytes, need);
    uint32_t* partial = reinterpret_cast<uint32_t*>(work);
    if (n == 0) {
        if (h_sum) *h_sum = 0;
        return ADLHIP_SUCCESS;
    }
    if (!dst || !src) return fail("null buffer passed to scan");
    const size_t blocks = (n + adlhip::kScanTile - 1) / adlhip::kScanTile;
    if (blocks > 0x7fffffffull) return fail("scan: n too large");
    uint32_t* d_total = partial + blocks;   // grand total lives behind the block sums
    int rc;
    if (blocks <= 8) {
        rc = launch(d, "scan_single", [&] {
            hipLaunchKernelGGL(adlhip::scan_single_kernel, dim3(1), dim3(adlhip::kScanNT), 0, d->stream, src, dst, n, d_total);
        });
        if (rc) return rc;
    } else {
        rc = launch(d, "scan_reduce", [&] {
            hipLaunchKernelGGL(adlhip::scan_reduce_kernel, dim3((uint32_t)blocks), dim3(adlhip::kScanNT), 0, d->stream, src, partial, n);
        });
        if (rc) return rc;
        rc = launch(d, "scan_partials", [&] {
            hipLaunchKernelGGL(adlhip::scan_single_kernel, dim3(1), dim3(adlhip::kScanNT), 0, d->stream,
                               (const uint32_t*)partial, partial, blocks, d_total);
        });
        if (rc) return rc;
        rc = launch(d, "scan_apply", [&] {
            hipLaunchKernelGGL(adlhip::scan_apply_kernel, dim3((uint32_t)blocks), dim3(adlhip::kScanNT), 0, d->stream,
                               src, dst, (const uint32_t*)partial, n);
        });
        if (rc) return rc;
    }
    if (h_sum) HIPCHK(hipMemcpyAsync(h_sum, d_total, 4, hipMemcpyDeviceToHost, d->stream));   // Pprims.cpp:164-167
    return ADLHIP_SUCCESS;
}

// ---- MSB partition (multi-GPU send side) -----------------------------------------------------------

int adlhip_partition_msb_u32(adlhip_device* d, const uint32_t* in, uint32_t* out, uint32_t* counts, void* work,
                             size_t work_bytes, size_t n, int num_buckets)
{
    return partition_msb<uint32_t>(d, in, out, counts, work, work_bytes, n, num_buckets);
}

int adlhip_partition_msb_kv32(adlhip_device* d, const void* in, void* out, uint32_t* counts, void* work, size_t work_bytes,
                              size_t n, int num_buckets)
{
    return partition_msb<uint64_t>(d, static_cast<const uint64_t*>(in), static_cast<uint64_t*>(out), counts, work, work_bytes, n,
                                   num_buckets);
}

int adlhip_partition_top_byte_u32(adlhip_device* d, const uint32_t* in, uint32_t* out, uint32_t* totals256, void* work,
                                  size_t work_bytes, size_t n)
{
    return partition_top_byte_entry<uint32_t>(d, in, out, totals256, work, work_bytes, n);
}

int adlhip_partition_top_byte_kv32(adlhip_device* d, const void* in, void* out, uint32_t* totals256, void* work,
                                   size_t work_bytes, size_t n)
{
    return partition_top_byte_entry<uint64_t>(d, static_cast<const uint64_t*>(in), static_cast<uint64_t*>(out), totals256, work,
                                              work_bytes, n);
}

// ---- synthetic inputs ---------------------------------------------------------------------------

int adlhip_generate_keys(adlhip_device* d, int elem_kind, void* dptr, size_t n, uint64_t seed, uint64_t first_index)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (elem_kind < ADLHIP_ELEM_U32 || elem_kind > ADLHIP_ELEM_U64) return fail("bad element kind %d", elem_kind);
    if (n == 0) return ADLHIP_SUCCESS;
    if (!dptr) return fail("null buffer");
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull + first_index;
    const int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)d->prop.multiProcessorCount * 16);
    return launch(d, "generate_keys", [&] {
        hipLaunchKernelGGL(adlhip::generate_keys_kernel, dim3(grid), dim3(256), 0, d->stream, dptr, n, base, first_index, elem_kind);
    });
}

// ---- knobs ------------------------------------------------------------------------------------

int adlhip_set_param(adlhip_device* d, const char* name, int value)
{
    if (!d || !name) return fail("null argument");
    if (!strcmp(name, "sort.algo")) {
        if (value < -1 || value > 1) return fail("sort.algo must be -1, 0 or 1");
        d->sort_algo = value;
    } else if (!strcmp(name, "sort.digit_bits")) {
        if (value != 4 && value != 7 && value != 8) return fail("sort.digit_bits must be 4, 7 or 8");
        d->digit_bits = value;
    } else if (!strcmp(name, "sort.tile")) {
        if (value < -1 || value >= kNumVariants) return fail("sort.tile must be in [-1,%d)", kNumVariants);
        d->tile_variant = value;

    } else if (!strcmp(name, "sort.mid")) {
        if (value < 0 || value > 3) return fail("sort.mid must be 0 (off), 1 (on), 2 (keys: always the two-launch form) or 3 (always the three-launch form)");
        d->mid_path = value;
    } else if (!strcmp(name, "sort.msd2")) {
        if (value < 0 || value > 5)
            return fail("sort.msd2 must be 0 (off), 1 (on), 2 (always, from 1 Mi elements; forms by size), 3 (always the stable passes), "
                        "4 (always the cursor passes for whole keys) or 5 (always the hybrid form for whole keys)");
        d->msd2_path = value;
    } else if (!strcmp(name, "sort.persist")) {
        if (value != 0 && value != 1) return fail("sort.persist must be 0 (one tile per workgroup, round 3) or 1 (persistent prefetching passes + 16-bit finish)");
        d->persist = value;
    } else if (!strcmp(name, "sort.net_lookback")) {
        d->net_lookback = value ? 1 : 0;
    } else if (!strcmp(name, "partition.lookback")) {
        d->partition_lookback = value ? 1 : 0;
    } else if (!strcmp(name, "sort.dict")) {
        if (value != 0 && value != 1) return fail("sort.dict must be 0 (off) or 1 (counting sort for keys that take few distinct values)");
        d->dict_path = value;
    } else if (!strcmp(name, "sort.binfinish")) {
        if (value < 0 || value > 2) return fail("sort.binfinish must be 0 (LSD finish), 1 (binning finish for whole u64 keys from 24 Mi keys up) or 2 (always, u32 keys too)");
        d->bin_finish = value;
    } else if (!strcmp(name, "debug.resident_wgs")) {
        // what the paths with a grid-wide barrier (the safety nets) and the one-workgroup-per-bucket finish may count on;
        // 0 = ask the device again.  Tests use it to stand in for a small partition.
        if (value < 0) return fail("debug.resident_wgs must be >= 0");
        d->resident_wgs = value ? value : d->resident_wgs_device;
    } else if (!strcmp(name, "sort.rank")) {
        if (value != 0 && value != 1) return fail("sort.rank must be 0 or 1");
        if (value == 1 && !d->lds_ordered) return fail("sort.rank = 1 needs lane-ordered DS atomics; the device self-test failed");
        d->rank_mode = value;
    } else if (!strcmp(name, "profile")) {
        if (bind(d)) return ADLHIP_FAILURE;
        if (!value && fold_profile(d)) return ADLHIP_FAILURE;
        d->profile = value ? 1 : 0;
    } else {
        return fail("unknown parameter '%s'", name);
    }
    return ADLHIP_SUCCESS;
}

int adlhip_get_param(adlhip_device* d, const char* name, int* value)
{
    if (!d || !name || !value) return fail("null argument");
    if (!strcmp(name, "sort.algo")) *value = d->sort_algo;
    else if (!strcmp(name, "sort.digit_bits")) *value = d->digit_bits;
    else if (!strcmp(name, "sort.tile")) *value = d->tile_variant;
    else if (!strcmp(name, "sort.rank")) *value = d->rank_mode;
    else if (!strcmp(name, "sort.mid")) *value = d->mid_path;
    else if (!strcmp(name, "sort.msd2")) *value = d->msd2_path;
    else if (!strcmp(name, "sort.binfinish")) *value = d->bin_finish;
    else if (!strcmp(name, "sort.persist")) *value = d->persist;
    else if (!strcmp(name, "sort.dict")) *value = d->dict_path;
    else if (!strcmp(name, "partition.lookback")) *value = d->partition_lookback;
    else if (!strcmp(name, "sort.net_lookback")) *value = d->net_lookback;
    else if (!strcmp(name, "debug.resident_wgs")) *value = d->resident_wgs;
    else if (!strcmp(name, "stat.net_runs") || !strcmp(name, "stat.net_counting")) {
        // how often the large sort's safety net has run on this handle, and how often it sorted by counting (waits for the stream)
        uint32_t v[2] = {0u, 0u};
        HIPCHK(hipMemcpyAsync(v, net_stats(d), 8, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        *value = (int)v[name[9] == 'c' ? 1 : 0];
    }
    else if (!strncmp(name, "debug.net_stamp", 15) && name[15] >= '0' && name[15] <= '9' && !name[16]) {
        // diagnostic: when workgroup 0 of the last net reached its k-th phase boundary, in 10-ns ticks (low 31 bits; tools/net_phases.py)
        uint32_t v = 0u;
        HIPCHK(hipMemcpyAsync(&v, net_stats(d) + 4 + (name[15] - '0'), 4, hipMemcpyDeviceToHost, d->stream));
        HIPCHK(hipStreamSynchronize(d->stream));
        *value = (int)(v & 0x7fffffffu);
    }
    else if (!strcmp(name, "sort.lds_ordered")) *value = d->lds_ordered;
    else if (!strcmp(name, "profile")) *value = d->profile;
    else return fail("unknown parameter '%s'", name);
    return ADLHIP_SUCCESS;
}

// ---- events / profiling ---------------------------------------------------------------------------

int adlhip_event_create(adlhip_device* d, adlhip_event** out)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!out) return fail("null out pointer");
    adlhip_event* e = new adlhip_event();
    if (hipEventCreate(&e->ev) != hipSuccess) {
        delete e;
        return fail("hipEventCreate failed");
    }
    *out = e;
    return ADLHIP_SUCCESS;
}

int adlhip_event_record(adlhip_device* d, adlhip_event* ev)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!ev) return fail("null event");
    HIPCHK(hipEventRecord(ev->ev, d->stream));
    return ADLHIP_SUCCESS;
}

int adlhip_event_elapsed_ms(adlhip_device* d, adlhip_event* a, adlhip_event* b, float* ms)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!a || !b || !ms) return fail("null argument");
    HIPCHK(hipEventSynchronize(b->ev));
    HIPCHK(hipEventElapsedTime(ms, a->ev, b->ev));
    return ADLHIP_SUCCESS;
}

int adlhip_event_synchronize(adlhip_device* d, adlhip_event* ev)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!ev) return fail("null event");
    HIPCHK(hipEventSynchronize(ev->ev));   // (an event that was never recorded is complete)
    return ADLHIP_SUCCESS;
}

int adlhip_event_query(adlhip_device* d, adlhip_event* ev, int* done)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!ev || !done) return fail("null argument");
    const hipError_t e = hipEventQuery(ev->ev);
    if (e != hipSuccess && e != hipErrorNotReady) return fail("hipEventQuery failed: %s", hipGetErrorString(e));
    (void)hipGetLastError();   // hipErrorNotReady is an answer, not an error to leave behind
    *done = e == hipSuccess ? 1 : 0;
    return ADLHIP_SUCCESS;
}

int adlhip_event_destroy(adlhip_device* d, adlhip_event* ev)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!ev) return ADLHIP_SUCCESS;
    hipEventDestroy(ev->ev);
    delete ev;
    return ADLHIP_SUCCESS;
}

int adlhip_profile_reset(adlhip_device* d)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (fold_profile(d)) return ADLHIP_FAILURE;
    d->prof.clear();
    d->prof_order.clear();
    return ADLHIP_SUCCESS;
}

int adlhip_profile_count(adlhip_device* d)
{
    if (bind(d)) return -1;
    if (fold_profile(d)) return -1;
    return (int)d->prof_order.size();
}

int adlhip_profile_get(adlhip_device* d, int i, char name_out[64], uint64_t* launches, double* total_ms)
{
    if (!d) return fail("null device handle");
    if (i < 0 || i >= (int)d->prof_order.size()) return fail("profile index %d out of range", i);
    const std::string& nm = d->prof_order[i];
    const ProfEntry& e = d->prof[nm];
    if (name_out) snprintf(name_out, 64, "%s", nm.c_str());
    if (launches) *launches = e.launches;
    if (total_ms) *total_ms = e.total_ms;
    return ADLHIP_SUCCESS;
}

#ifdef ADLHIP_STAMPS
// Diagnostic build only: point the phase-stamp side buffer at caller-provided device memory ([tiles][16] u64).
int adlhip_debug_set_stamp_buffer(adlhip_device* d, void* dptr)
{
    if (bind(d)) return ADLHIP_FAILURE;
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(adlhip::g_stamp_buf), &dptr, sizeof(void*)));
    return ADLHIP_SUCCESS;
}
#endif

int adlhip_profile_write_csv(adlhip_device* d, const char* path)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (!path) return fail("null path");
    if (fold_profile(d)) return ADLHIP_FAILURE;
    FILE* probe = fopen(path, "r");
    const bool fresh = probe == nullptr;
    if (probe) fclose(probe);
    FILE* f = fopen(path, "a");
    if (!f) return fail("cannot open %s for appending", path);
    if (fresh) fprintf(f, "\"kernel\",\"launches\",\"total_ms\",\"avg_ms\"\n");
    for (const std::string& nm : d->prof_order) {
        const ProfEntry& e = d->prof[nm];
        fprintf(f, "\"%s\",%llu,%.6f,%.6f\n", nm.c_str(), (unsigned long long)e.launches, e.total_ms,
                e.launches ? e.total_ms / (double)e.launches : 0.0);
    }
    fclose(f);
    return ADLHIP_SUCCESS;
}

// ---- probes ---------------------------------------------------------------------------------------

int adlhip_probe_copy(adlhip_device* d, void* dst, const void* src, size_t bytes)
{
    if (bind(d)) return ADLHIP_FAILURE;
    const size_t nvec = bytes / 16;
    if (nvec == 0) return ADLHIP_SUCCESS;
    const int grid = (int)std::min<size_t>((nvec + 255) / 256, (size_t)d->prop.multiProcessorCount * 8);
    return launch(d, "probe_copy", [&] {
        hipLaunchKernelGGL(adlhip::probe_copy_kernel, dim3(grid), dim3(256), 0, d->stream, (uint4*)dst, (const uint4*)src, nvec);
    });
}

// hints: bit 0 = non-temporal loads, bit 1 = non-temporal stores; grid_per_cu workgroups of 256 threads per CU (0 = 8)
int adlhip_probe_copy_ex(adlhip_device* d, void* dst, const void* src, size_t bytes, int hints, int grid_per_cu)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (hints < 0 || hints > 3 || grid_per_cu < 0 || grid_per_cu > 64) return fail("bad probe variant");
    const size_t nvec = bytes / 16;
    if (nvec == 0) return ADLHIP_SUCCESS;
    const int grid = (int)std::min<size_t>((nvec + 255) / 256, (size_t)d->prop.multiProcessorCount * (grid_per_cu ? grid_per_cu : 8));
    return launch(d, "probe_copy", [&] {
        switch (hints) {
        case 0: hipLaunchKernelGGL((adlhip::probe_copy_hint_kernel<false, false>), dim3(grid), dim3(256), 0, d->stream, (uint4*)dst, (const uint4*)src, nvec); break;
        case 1: hipLaunchKernelGGL((adlhip::probe_copy_hint_kernel<true, false>), dim3(grid), dim3(256), 0, d->stream, (uint4*)dst, (const uint4*)src, nvec); break;
        case 2: hipLaunchKernelGGL((adlhip::probe_copy_hint_kernel<false, true>), dim3(grid), dim3(256), 0, d->stream, (uint4*)dst, (const uint4*)src, nvec); break;
        default: hipLaunchKernelGGL((adlhip::probe_copy_hint_kernel<true, true>), dim3(grid), dim3(256), 0, d->stream, (uint4*)dst, (const uint4*)src, nvec); break;
        }
    });
}

int adlhip_probe_read_ex(adlhip_device* d, const void* src, size_t bytes, void* sink8, int hints, int grid_per_cu)
{
    if (bind(d)) return ADLHIP_FAILURE;
    if (hints < 0 || hints > 1 || grid_per_cu < 0 || grid_per_cu > 64) return fail("bad probe variant");
    const size_t nvec = bytes / 16;
    if (nvec == 0) return ADLHIP_SUCCESS;
    const int grid = (int)std::min<size_t>((nvec + 255) / 256, (size_t)d->prop.multiProcessorCount * (grid_per_cu ? grid_per_cu : 8));
    return launch(d, "probe_read", [&] {
        if (hints) hipLaunchKernelGGL((adlhip::probe_read_hint_kernel<true>), dim3(grid), dim3(256), 0, d->stream, (const uint4*)src, nvec, (unsigned long long*)sink8);
        else hipLaunchKernelGGL((adlhip::probe_read_hint_kernel<false>), dim3(grid), dim3(256), 0, d->stream, (const uint4*)src, nvec, (unsigned long long*)sink8);
    });
}

int adlhip_probe_read(adlhip_device* d, const void* src, size_t bytes, void* sink8)
{
    if (bind(d)) return ADLHIP_FAILURE;
    const size_t nvec = bytes / 16;
    if (nvec == 0) return ADLHIP_SUCCESS;
    const int grid = (int)std::min<size_t>((nvec + 255) / 256, (size_t)d->prop.multiProcessorCount * 8);
    return launch(d, "probe_read", [&] {
        hipLaunchKernelGGL(adlhip::probe_read_kernel, dim3(grid), dim3(256), 0, d->stream, (const uint4*)src, nvec,
                           (unsigned long long*)sink8);
    });
}

}  // extern "C"
