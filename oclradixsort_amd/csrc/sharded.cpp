// sharded.cpp -- the multi-GPU sharded sort behind the C ABI (include/adlhip.h, "sharded sort" section): ONE host
// process drives G devices, one adlhip_device each, and the MSB-bucket exchange runs over RCCL point-to-point
// (grouped ncclSend / ncclRecv: every pair of GPUs of an MI355X node has its own xGMI link, so the G-1 links of a
// GPU carry traffic at once).  No reference counterpart: the reference drives exactly one device
// (Adl/Adl.h:90-94); its API language is C++ (Tahoe/ParallelPrimitives/Pprims.h:35-41), which is why this lives
// behind the C ABI and not only in the Python host layer (oclradixsort_amd/dist.py runs the same steps with one
// process per GPU over torch.distributed).
//
// Written against the PUBLIC C ABI only (partition, sort, copies, sync), so it cannot depend on back-end
// internals.  RCCL is loaded lazily with dlopen: libadlhip.so itself has no link-time dependency on it and a
// single-GPU user never loads it.
#include "../../include/adlhip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

extern "C" void adlhip_set_last_error(const char* text);   // adlhip.hip: the calling thread's error text

namespace {

int gfail(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    adlhip_set_last_error(buf);
    return ADLHIP_FAILURE;
}

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl* rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (r.lib) {
#define SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name))
            SYM(CommInitAll, "ncclCommInitAll");
            SYM(CommDestroy, "ncclCommDestroy");
            SYM(GroupStart, "ncclGroupStart");
            SYM(GroupEnd, "ncclGroupEnd");
            SYM(Send, "ncclSend");
            SYM(Recv, "ncclRecv");
            SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
            if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv || !r.GetErrorString) {
                dlclose(r.lib);
                r.lib = nullptr;
            }
        }
    }
    return r.lib ? &r : nullptr;
}

struct RankState {
    adlhip_device* dev = nullptr;
    int device_idx = 0;
    void* part = nullptr;       // partitioned copy of the rank's shard
    size_t part_bytes = 0;
    void* tmp = nullptr;        // ping-pong partner of the local sort
    size_t tmp_bytes = 0;
    void* work = nullptr;
    size_t work_bytes = 0;
    uint32_t* d_totals = nullptr;   // 256 top-byte totals
    uint32_t* h_totals = nullptr;   // pinned
};

}  // namespace

struct adlhip_group {
    int G = 0;
    std::vector<RankState> rank;
    std::vector<ncclComm_t> comm;
    std::vector<int> bounds;   // G + 1 top-byte boundaries of the last sort
};

namespace {

int grow(adlhip_device* d, void** p, size_t* have, size_t need)
{
    if (*have >= need) return ADLHIP_SUCCESS;
    if (*p && adlhip_free(d, *p, *have)) return ADLHIP_FAILURE;
    *p = nullptr;
    *have = 0;
    need += need / 8;
    if (adlhip_malloc(d, need, p)) return ADLHIP_FAILURE;
    *have = need;
    return ADLHIP_SUCCESS;
}

// The same rule as oclradixsort_amd/dist.py choose_splitters: every cut is the byte boundary whose cumulative
// count is nearest to g * total / G (ties: the boundary after the byte); cuts never decrease.
void choose_splitters(const uint64_t (&tot)[256], int G, std::vector<int>& bounds)
{
    uint64_t cum[256];
    uint64_t run = 0;
    for (int b = 0; b < 256; ++b) { run += tot[b]; cum[b] = run; }
    bounds.assign((size_t)G + 1, 0);
    bounds[(size_t)G] = 256;
    int prev = 0;
    for (int g = 1; g < G; ++g) {
        const uint64_t target = run * (uint64_t)g / (uint64_t)G;
        int idx = 0;
        while (idx < 255 && cum[idx] < target) ++idx;            // first byte with cum >= target
        const uint64_t hi = cum[idx], lo = idx > 0 ? cum[idx - 1] : 0;
        int cut = (hi - target) <= (target - lo) ? idx + 1 : idx;
        if (cut < prev) cut = prev;
        bounds[(size_t)g] = cut;
        prev = cut;
    }
}

template <bool PAIRS>
int sharded_sort(adlhip_group* g, void* const* in, const size_t* n_in, void* const* out, const size_t* out_capacity,
                 size_t* n_out)
{
    if (!g || !in || !n_in || !out || !out_capacity || !n_out) return gfail("sharded sort: null argument");
    Rccl* nc = rccl();
    if (!nc) return gfail("sharded sort: RCCL (librccl.so) could not be loaded");
    const int G = g->G;
    const size_t esz = PAIRS ? 8 : 4;
    const int kind = PAIRS ? ADLHIP_ELEM_KV32 : ADLHIP_ELEM_U32;
    // 1. local stable partition by the top byte + the 256 totals, on every device (asynchronous)
    for (int r = 0; r < G; ++r) {
        RankState& s = g->rank[(size_t)r];
        size_t tb = 0, wb = 0;
        if (adlhip_radix_sort_scratch_bytes(s.dev, kind, n_in[r], &tb, &wb)) return ADLHIP_FAILURE;
        if (grow(s.dev, &s.part, &s.part_bytes, n_in[r] * esz + 256)) return ADLHIP_FAILURE;
        if (grow(s.dev, &s.work, &s.work_bytes, wb)) return ADLHIP_FAILURE;
        if (n_in[r] && !in[r]) return gfail("sharded sort: null shard %d", r);
        int rc = PAIRS ? adlhip_partition_top_byte_kv32(s.dev, in[r], s.part, s.d_totals, s.work, s.work_bytes, n_in[r])
                       : adlhip_partition_top_byte_u32(s.dev, static_cast<const uint32_t*>(in[r]), static_cast<uint32_t*>(s.part),
                                                       s.d_totals, s.work, s.work_bytes, n_in[r]);
        if (rc) return rc;
        if (adlhip_memcpy_d2h(s.dev, s.h_totals, s.d_totals, 256 * 4)) return ADLHIP_FAILURE;
    }
    // 2. the one host synchronisation: all totals are on the host -> global histogram, splitters, count matrix
    for (int r = 0; r < G; ++r)
        if (adlhip_sync(g->rank[(size_t)r].dev)) return ADLHIP_FAILURE;
    uint64_t glob[256];
    memset(glob, 0, sizeof(glob));
    for (int r = 0; r < G; ++r)
        for (int b = 0; b < 256; ++b) glob[b] += g->rank[(size_t)r].h_totals[b];
    choose_splitters(glob, G, g->bounds);
    std::vector<size_t> cnt((size_t)G * (size_t)G, 0);   // cnt[r * G + p]: rank r sends this many elements to rank p
    for (int r = 0; r < G; ++r)
        for (int p = 0; p < G; ++p) {
            size_t c = 0;
            for (int b = g->bounds[(size_t)p]; b < g->bounds[(size_t)p + 1]; ++b) c += g->rank[(size_t)r].h_totals[b];
            cnt[(size_t)r * G + p] = c;
        }
    bool fits = true;
    for (int p = 0; p < G; ++p) {
        size_t tot = 0;
        for (int r = 0; r < G; ++r) tot += cnt[(size_t)r * G + p];
        n_out[p] = tot;
        if (tot > out_capacity[p]) fits = false;
        if (tot && !out[p]) return gfail("sharded sort: null output buffer %d", p);
    }
    if (!fits) return gfail("sharded sort: an output buffer is too small (the required sizes are in n_out)");
    // 3. the exchange: rank r's segment p goes to rank p; segments land in source-rank order (keeps pairs stable)
    ncclResult_t st = nc->GroupStart();
    if (st != ncclSuccess) return gfail("ncclGroupStart: %s", nc->GetErrorString(st));
    const ncclDataType_t dt = PAIRS ? ncclUint64 : ncclUint32;
    for (int r = 0; r < G && st == ncclSuccess; ++r) {
        RankState& s = g->rank[(size_t)r];
        hipSetDevice(s.device_idx);
        hipStream_t stream = reinterpret_cast<hipStream_t>(adlhip_stream(s.dev));
        size_t send_off = 0, recv_off = 0;
        for (int p = 0; p < G && st == ncclSuccess; ++p) {
            const size_t sc = cnt[(size_t)r * G + p];
            const size_t rc = cnt[(size_t)p * G + r];
            if (sc) st = nc->Send(static_cast<const char*>(s.part) + send_off * esz, sc, dt, p, g->comm[(size_t)r], stream);
            if (rc && st == ncclSuccess) st = nc->Recv(static_cast<char*>(out[r]) + recv_off * esz, rc, dt, p, g->comm[(size_t)r], stream);
            send_off += sc;
            recv_off += rc;
        }
    }
    ncclResult_t en = nc->GroupEnd();
    if (st != ncclSuccess) return gfail("ncclSend/ncclRecv: %s", nc->GetErrorString(st));
    if (en != ncclSuccess) return gfail("ncclGroupEnd: %s", nc->GetErrorString(en));
    // 4. local sort of what arrived, stream-ordered behind the receives
    for (int p = 0; p < G; ++p) {
        RankState& s = g->rank[(size_t)p];
        if (n_out[p] == 0) continue;
        size_t tb = 0, wb = 0;
        if (adlhip_radix_sort_scratch_bytes(s.dev, kind, n_out[p], &tb, &wb)) return ADLHIP_FAILURE;
        if (s.tmp_bytes < tb || s.work_bytes < wb) {
            if (adlhip_sync(s.dev)) return ADLHIP_FAILURE;   // growing frees memory queued work may use
            if (grow(s.dev, &s.tmp, &s.tmp_bytes, tb)) return ADLHIP_FAILURE;
            if (grow(s.dev, &s.work, &s.work_bytes, wb)) return ADLHIP_FAILURE;
        }
        int rc = PAIRS ? adlhip_radix_sort_kv32(s.dev, out[p], s.tmp, s.work, s.work_bytes, n_out[p], 32)
                       : adlhip_radix_sort_u32(s.dev, static_cast<uint32_t*>(out[p]), static_cast<uint32_t*>(s.tmp), s.work,
                                               s.work_bytes, n_out[p], 32);
        if (rc) return rc;
    }
    return ADLHIP_SUCCESS;
}

}  // namespace

extern "C" {

int adlhip_group_create(const int* device_indices, int num_devices, adlhip_group** out)
{
    if (!out) return gfail("null out pointer");
    *out = nullptr;
    if (num_devices < 1 || num_devices > 256) return gfail("group: between 1 and 256 devices, got %d", num_devices);
    const int have = adlhip_device_count();
    if (have <= 0) return gfail("no HIP device available");
    std::vector<int> devs((size_t)num_devices);
    for (int r = 0; r < num_devices; ++r) {
        devs[(size_t)r] = device_indices ? device_indices[r] : r;
        if (devs[(size_t)r] < 0 || devs[(size_t)r] >= have) return gfail("group: device index %d out of range (%d devices)", devs[(size_t)r], have);
        for (int q = 0; q < r; ++q)
            if (devs[(size_t)q] == devs[(size_t)r]) return gfail("group: device %d listed twice (one rank per GPU)", devs[(size_t)r]);
    }
    Rccl* nc = rccl();
    if (!nc) return gfail("group: RCCL (librccl.so) could not be loaded");
    adlhip_group* g = new adlhip_group();
    g->G = num_devices;
    g->rank.resize((size_t)num_devices);
    g->comm.assign((size_t)num_devices, nullptr);
    for (int r = 0; r < num_devices; ++r) {
        RankState& s = g->rank[(size_t)r];
        s.device_idx = devs[(size_t)r];
        // every step records what it acquired before the next one runs, so that the destroy below accounts for exactly what
        // exists (it refuses a device whose live bytes it cannot explain) and the first error's text survives it
        void* t = nullptr;
        bool ok = adlhip_device_create(s.device_idx, &s.dev) == ADLHIP_SUCCESS;
        if (ok) {
            ok = adlhip_malloc(s.dev, 256 * 4, &t) == ADLHIP_SUCCESS;
            if (ok) s.d_totals = static_cast<uint32_t*>(t);
        }
        if (ok && hipHostMalloc(reinterpret_cast<void**>(&s.h_totals), 256 * 4) != hipSuccess) {
            s.h_totals = nullptr;
            gfail("group: hipHostMalloc of the totals staging failed");
            ok = false;
        }
        if (!ok) {
            const std::string why = adlhip_last_error();
            adlhip_group_destroy(g);
            return gfail("%s", why.c_str());
        }
    }
    ncclResult_t st = nc->CommInitAll(g->comm.data(), num_devices, devs.data());
    if (st != ncclSuccess) {
        g->comm.assign((size_t)num_devices, nullptr);
        adlhip_group_destroy(g);
        return gfail("ncclCommInitAll: %s", nc->GetErrorString(st));
    }
    *out = g;
    return ADLHIP_SUCCESS;
}

int adlhip_group_destroy(adlhip_group* g)
{
    if (!g) return ADLHIP_SUCCESS;
    Rccl* nc = rccl();
    int rc = ADLHIP_SUCCESS;
    // like DeviceUtils::deallocate (Adl/Adl.inl:100-105): refuse, leaving everything alive, while the caller still
    // holds memory of one of the group's devices
    for (size_t r = 0; r < g->rank.size(); ++r) {
        const RankState& s = g->rank[r];
        if (!s.dev) continue;
        const uint64_t own = s.part_bytes + s.tmp_bytes + s.work_bytes + (s.d_totals ? 256 * 4 : 0);
        if (adlhip_used_bytes(s.dev) != own)
            return gfail("group: device of rank %zu still has %llu live bytes; free every buffer first", r,
                         (unsigned long long)(adlhip_used_bytes(s.dev) - own));
    }
    for (RankState& s : g->rank)
        if (s.dev) adlhip_sync(s.dev);
    for (ncclComm_t c : g->comm)
        if (c && nc) nc->CommDestroy(c);
    for (RankState& s : g->rank) {
        if (!s.dev) continue;
        if (s.part) adlhip_free(s.dev, s.part, s.part_bytes);
        if (s.tmp) adlhip_free(s.dev, s.tmp, s.tmp_bytes);
        if (s.work) adlhip_free(s.dev, s.work, s.work_bytes);
        if (s.d_totals) adlhip_free(s.dev, s.d_totals, 256 * 4);
        if (s.h_totals) hipHostFree(s.h_totals);
        if (adlhip_device_destroy(s.dev)) rc = ADLHIP_FAILURE;   // fails while the caller still holds buffers of this device
    }
    delete g;
    return rc;
}

int adlhip_group_size(adlhip_group* g) { return g ? g->G : 0; }

adlhip_device* adlhip_group_device(adlhip_group* g, int rank)
{
    if (!g || rank < 0 || rank >= g->G) return nullptr;
    return g->rank[(size_t)rank].dev;
}

int adlhip_group_last_bounds(adlhip_group* g, int* bounds_out)
{
    if (!g || !bounds_out) return gfail("null argument");
    if (g->bounds.size() != (size_t)g->G + 1) return gfail("no sharded sort has run on this group yet");
    for (size_t i = 0; i < g->bounds.size(); ++i) bounds_out[i] = g->bounds[i];
    return ADLHIP_SUCCESS;
}

int adlhip_sharded_sort_u32(adlhip_group* g, uint32_t* const* d_shards_in, const size_t* n_in, uint32_t* const* d_out,
                            const size_t* out_capacity, size_t* n_out)
{
    return sharded_sort<false>(g, reinterpret_cast<void* const*>(d_shards_in), n_in, reinterpret_cast<void* const*>(d_out),
                               out_capacity, n_out);
}

int adlhip_sharded_sort_kv32(adlhip_group* g, void* const* d_shards_in, const size_t* n_in, void* const* d_out,
                             const size_t* out_capacity, size_t* n_out)
{
    return sharded_sort<true>(g, d_shards_in, n_in, d_out, out_capacity, n_out);
}

}  // extern "C"
