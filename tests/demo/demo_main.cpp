// Demo harness: the three checks of the reference's unit test (UnitTest/main.cpp:40-56, 88-213 --
// Demo.Sort32, Demo.SortKeyValue, Demo.Scan over the sizes 1K..1024K with srand(123) data), re-created
// without gtest on top of the facade headers, plus the cases the reference cannot run:
//   --host     run the sorts on an Adl TYPE_HOST device (BASELINE config #1: the CPU path; the shipped test
//              hard-codes TYPE_CL, UnitTest/main.cpp:98)
//   --gpus G   additionally run Demo.ShardedSort: G shards on G devices through Tahoe::ShardedSort (one process, RCCL
//              exchange), checked against std::sort / std::stable_sort of the concatenated shards
//   default    TYPE_CL = the MI355X HIP back-end; Demo.Scan includes 1024K, where the reference gives up
//              (README.md:73-74, Pprims.cpp:134-138)
// The ground truth is computed here with the C++ standard library (std::sort / std::stable_sort / running
// sum), so this binary depends on nothing under oracle/.
// Exit status = number of failed checks.
#include <Adl/Adl.h>
#include <Tahoe/Algorithm/Sort/RadixSort.h>
#include <Tahoe/ParallelPrimitives/Pprims.h>
#include <Tahoe/ParallelPrimitives/uArray.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <Tahoe/ParallelPrimitives/ShardedSort.h>
#include <vector>

using namespace adl;
using namespace Tahoe;

char adl::s_cacheDirectory[128] = "cache";

namespace {

// the reference's data recipe (UnitTest/main.cpp:76-86)
template <typename T>
T demoRandom(const T& lo, const T& hi)
{
    const double r = std::min((double)RAND_MAX - 1, (double)rand()) / RAND_MAX;
    const T range = hi - lo;
    return (T)(lo + r * range);
}

int g_failed = 0;
void check(bool ok, const char* what, int n)
{
    if (!ok) {
        ++g_failed;
        printf("  FAILED: %s at %d elems\n", what, n);
    }
}

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void demoSort32(Device* d, Pprims& p)
{
    printf("[ RUN      ] Demo.Sort32\n");
    const double t0 = now_ms();
    for (int n = 1024; n < 2 * 1024 * 1024; n *= 2) {
        printf("test %6.1fK elems\n", n / 1024.f);
        srand(123);
        Buffer<u32> gpu(d, n);
        std::vector<u32> cpu(n);
        u32* h = gpu.getHostPtr(n);
        DeviceUtils::waitForCompletion(d);
        for (int i = 0; i < n; ++i) h[i] = cpu[i] = demoRandom(0u, 0xffffffffu);
        gpu.returnHostPtr(h);
        DeviceUtils::waitForCompletion(d);

        p.radixSort(d, gpu, n);
        std::sort(cpu.begin(), cpu.end());

        h = gpu.getHostPtr(n);
        DeviceUtils::waitForCompletion(d);
        check(memcmp(h, cpu.data(), sizeof(u32) * n) == 0, "Sort32 differs from std::sort", n);
        gpu.returnHostPtr(h);
        DeviceUtils::waitForCompletion(d);
    }
    printf("[       %s ] Demo.Sort32 (%.0f ms)\n", g_failed ? "FAIL" : "OK", now_ms() - t0);
}

void demoSortKeyValue(Device* d, Pprims& p)
{
    printf("[ RUN      ] Demo.SortKeyValue\n");
    const double t0 = now_ms();
    const int before = g_failed;
    for (int n = 1024; n < 2 * 1024 * 1024; n *= 2) {
        n += 13;   // the reference bumps the size inside the doubling loop (main.cpp:144): 1037, 2087, ...
        printf("test %6.1fK elems\n", n / 1024.f);
        srand(123);
        Buffer<SortData> gpu(d, n);
        uArray<SortData> cpu(n);
        SortData* h = gpu.getHostPtr(n);
        DeviceUtils::waitForCompletion(d);
        for (int i = 0; i < n; ++i) h[i] = cpu[i] = SortData(demoRandom(0u, 0xffffffffu), (u32)i);
        gpu.returnHostPtr(h);
        DeviceUtils::waitForCompletion(d);

        p.radixSort(d, *(Buffer<uint2>*)&gpu, n);
        std::stable_sort(cpu.begin(), cpu.begin() + n);   // by key, ties keep input order

        h = gpu.getHostPtr(n);
        DeviceUtils::waitForCompletion(d);
        bool ok = true;
        for (int i = 0; i < n; ++i) ok &= (h[i].m_key == cpu[i].m_key) && (h[i].m_value == cpu[i].m_value);
        check(ok, "SortKeyValue differs from std::stable_sort", n);
        gpu.returnHostPtr(h);
        DeviceUtils::waitForCompletion(d);
    }
    printf("[       %s ] Demo.SortKeyValue (%.0f ms)\n", g_failed > before ? "FAIL" : "OK", now_ms() - t0);
}

void demoScan(Device* d, Pprims& p)
{
    printf("[ RUN      ] Demo.Scan\n");
    const double t0 = now_ms();
    const int before = g_failed;
    for (int n = 1024; n < 2 * 1024 * 1024; n *= 2) {
        printf("test %6.1fK elems\n", n / 1024.f);
        srand(123);
        Buffer<int> gpu(d, n);
        Buffer<int> gpuRes(d, n);
        std::vector<int> cpu(n);
        int* h = gpu.getHostPtr(n);
        DeviceUtils::waitForCompletion(d);
        for (int i = 0; i < n; ++i) h[i] = cpu[i] = demoRandom(0, 0xf);
        gpu.returnHostPtr(h);
        DeviceUtils::waitForCompletion(d);

        u32 total = 0xdeadbeef;
        p.scan(d, gpuRes, gpu, n, &total);

        h = gpuRes.getHostPtr(n);
        DeviceUtils::waitForCompletion(d);
        int ans = 0;
        bool fail = false;
        for (int i = 0; i < n; ++i) {   // the reference's own check (main.cpp:193-199)
            fail |= (h[i] != ans);
            ans += cpu[i];
        }
        check(!fail, "Scan differs from the running sum", n);
        check(total == (u32)ans, "Scan grand total", n);
        gpuRes.returnHostPtr(h);
        DeviceUtils::waitForCompletion(d);
    }
    printf("[       %s ] Demo.Scan (%.0f ms)\n", g_failed > before ? "FAIL" : "OK", now_ms() - t0);
}

// Pprims::fill / copy (commented out in the reference, Pprims.cpp:31-120): int, u32 and float4 elements, odd counts,
// and everything past n untouched.
void demoFillCopy(Device* d, Pprims& p)
{
    printf("[ RUN      ] Demo.FillCopy\n");
    const double t0 = now_ms();
    const int before = g_failed;
    const int sizes[] = {1, 2, 3, 255, 1024, 100003, 1 << 20};
    for (int n : sizes) {
        const int cap = n + 5;
        Buffer<int> bi(d, cap), ci(d, cap);
        Buffer<u32> bu(d, cap);
        Buffer<float4> bf(d, cap), cf(d, cap);
        bi.clear(); ci.clear(); bu.clear(); bf.clear(); cf.clear();
        float4 pat;
        pat.x = 1.5f; pat.y = -2.25f; pat.z = 3.f; pat.w = (float)n;
        p.fill(d, bi, -7 - n, n);
        p.fill(d, bu, 0xC0FFEE00u + (u32)n, n);
        p.fill(d, bf, pat, n);
        p.copy(d, ci, bi, n);
        p.copy(d, cf, bf, n);
        int* hi = ci.getHostPtr(cap);
        u32* hu = bu.getHostPtr(cap);
        float4* hf = cf.getHostPtr(cap);
        DeviceUtils::waitForCompletion(d);
        bool ok = true;
        for (int i = 0; i < cap; ++i) {
            const bool in = i < n;
            ok &= hi[i] == (in ? -7 - n : 0);
            ok &= hu[i] == (in ? 0xC0FFEE00u + (u32)n : 0u);
            ok &= hf[i].x == (in ? pat.x : 0.f) && hf[i].y == (in ? pat.y : 0.f) && hf[i].z == (in ? pat.z : 0.f) &&
                  hf[i].w == (in ? pat.w : 0.f);
        }
        check(ok, "fill/copy result", n);
        ci.returnHostPtr(hi); bu.returnHostPtr(hu); cf.returnHostPtr(hf);
        DeviceUtils::waitForCompletion(d);
    }
    printf("[       %s ] Demo.FillCopy (%.0f ms)\n", g_failed > before ? "FAIL" : "OK", now_ms() - t0);
}

// Demo.SortWideValues: separate key and value buffers, u32 keys + u64 values and u64 keys + u64 values, value =
// {hash of the key, source index}: sorted by key, every value beside its key, equal keys in source order (stable).
void demoSortWideValues(Device* d, Pprims& p)
{
    printf("[ RUN      ] Demo.SortWideValues\n");
    const double t0 = now_ms();
    const int before = g_failed;
    const int sizes[] = {1, 255, 4097, 100003, (1 << 21) + 17};
    for (int n : sizes) {
        for (int wide_keys = 0; wide_keys < 2; ++wide_keys) {
            std::vector<u64> k64(n), v(n);
            std::vector<u32> k32(n);
            srand(123 + n);
            for (int i = 0; i < n; ++i) {
                const u32 r = ((u32)rand() << 16) ^ (u32)rand();
                k32[i] = r & 0x00ffffffu;                                   // duplicates: stability is visible
                k64[i] = ((u64)(r & 0xffu) << 40) | (u64)(r >> 20);         // ties in either dword
                const u64 key = wide_keys ? k64[i] : (u64)k32[i];
                v[i] = ((key * 0x9E3779B1ull) << 32) | (u64)(u32)i;
            }
            Buffer<u64> vb(d, n);
            vb.write(v.data(), n);
            std::vector<u64> gk(n), gv(n);
            if (wide_keys) {
                Buffer<u64> kb(d, n);
                kb.write(k64.data(), n);
                p.radixSort(d, kb, vb, n);
                kb.read(gk.data(), n);
                vb.read(gv.data(), n);
                DeviceUtils::waitForCompletion(d);
            } else {
                Buffer<u32> kb(d, n);
                kb.write(k32.data(), n);
                p.radixSort(d, kb, vb, n);
                std::vector<u32> g32(n);
                kb.read(g32.data(), n);
                vb.read(gv.data(), n);
                DeviceUtils::waitForCompletion(d);
                for (int i = 0; i < n; ++i) gk[i] = g32[i];
            }
            bool ok = true;
            for (int i = 0; i < n; ++i) {
                ok &= (gv[i] >> 32) == ((gk[i] * 0x9E3779B1ull) & 0xffffffffull);            // the value still belongs to its key
                const u32 src = (u32)gv[i];
                ok &= src < (u32)n && (wide_keys ? k64[src] : (u64)k32[src]) == gk[i];          // ... and came from that position
                if (i) ok &= gk[i - 1] < gk[i] || (gk[i - 1] == gk[i] && (u32)gv[i - 1] < src); // sorted, stable
            }
            check(ok, wide_keys ? "u64 keys + u64 values" : "u32 keys + u64 values", n);
        }
    }
    printf("[       %s ] Demo.SortWideValues (%.0f ms)\n", g_failed > before ? "FAIL" : "OK", now_ms() - t0);
}

// Demo.BufferUtils (Adl.h:224-248) and SyncObject (AdlKernel.h:45-54): a host buffer seen from the device and back, a device
// buffer seen from the host, in-place variants, and an event recorded behind a copy.
void demoBufferUtils(Device* d)
{
    printf("[ RUN      ] Demo.BufferUtils\n");
    const double t0 = now_ms();
    const int before = g_failed;
    Device* host = DeviceUtils::allocate(TYPE_HOST);
    const int n = 100003;
    {
        HostBuffer<u32> h(host, n);
        for (int i = 0; i < n; ++i) h[i] = (u32)i * 2654435761u;
        // host buffer -> device (copied), sorted there, copied back by unmap
        Buffer<u32>* onDev = BufferUtils::map<TYPE_CL, true>(d, (const Buffer<u32>*)&h);
        check(onDev != (Buffer<u32>*)&h && onDev->getType() == TYPE_CL && (int)onDev->getSize() == n, "map to the device", n);
        {
            Pprims p;
            p.radixSort(d, *onDev, n);
        }
        BufferUtils::unmap<true>(onDev, (const Buffer<u32>*)&h);
        bool ok = true;
        for (int i = 1; i < n; ++i) ok &= h[i - 1] <= h[i];
        check(ok, "sorted through map / unmap", n);
        // device buffer seen from the host: a view of its mapping; writes go back at unmap
        Buffer<u32> dev(d, n);
        dev.write(h.begin(), n);
        Buffer<u32>* onHost = BufferUtils::map<TYPE_HOST, true>(host, (const Buffer<u32>*)&dev);
        DeviceUtils::waitForCompletion(d);
        ok = onHost->getType() == TYPE_HOST;
        for (int i = 0; i < n; ++i) ok &= onHost->m_ptr[i] == h[i];
        onHost->m_ptr[7] = 0xdeadbeefu;
        BufferUtils::unmap<true>(onHost, (const Buffer<u32>*)&dev);
        DeviceUtils::waitForCompletion(d);
        u32 seven = 0;
        SyncObject ev(d);
        check(DeviceUtils::isComplete(&ev), "an event that was never recorded is complete", 0);
        dev.read(&seven, 1, 7, &ev);
        DeviceUtils::waitForCompletion(&ev);                       // waits for the copy only
        check(ok && seven == 0xdeadbeefu && DeviceUtils::isComplete(&ev), "host view of a device buffer + SyncObject", n);
        // same type: map hands back the buffer itself; in-place variants use the caller's buffer
        check(BufferUtils::map<TYPE_CL, true>(d, (const Buffer<u32>*)&dev) == &dev, "map of a native buffer is the buffer", n);
        Buffer<u32> scratch(d, n);
        Buffer<u32>* in = BufferUtils::mapInplace<TYPE_CL, true>(d, &scratch, (const Buffer<u32>*)&h);
        check(in == &scratch, "mapInplace uses the caller's buffer", n);
        scratch.fill((void*)&seven, 4);
        BufferUtils::unmapInplace<true>(in, (const Buffer<u32>*)&h);
        ok = true;
        for (int i = 0; i < n; ++i) ok &= h[i] == 0xdeadbeefu;
        check(ok, "unmapInplace copies back", n);
    }
    DeviceUtils::deallocate(host);
    printf("[       %s ] Demo.BufferUtils (%.0f ms)\n", g_failed > before ? "FAIL" : "OK", now_ms() - t0);
}

// Demo.ShardedSort: G shards of the reference's random data (one srand seed per shard), keys and {key, value} pairs,
// sorted across G devices; every rank's slice is downloaded and the concatenation compared with the host's sort.
void demoShardedSort(int G)
{
    printf("[ RUN      ] Demo.ShardedSort (%d device%s)\n", G, G == 1 ? "" : "s");
    const double t0 = now_ms();
    const int before = g_failed;
    {
        ShardedSort s(G);
        if (adl_assert_failures() == 0) {
            for (int n = 1024; n <= 1024 * 1024; n *= 32) {
                std::vector<Buffer<u32>*> in(G), out(G);
                std::vector<Buffer<uint2>*> kin(G), kout(G);
                std::vector<size_t> nIn(G), nOut(G), knOut(G);
                std::vector<u32> all;
                std::vector<SortData> kall;
                for (int r = 0; r < G; ++r) {
                    srand(123u + (unsigned)r);
                    const int m = n + 13 * r;                       // ragged shards
                    std::vector<u32> k(m);
                    std::vector<SortData> kv(m);
                    for (int i = 0; i < m; ++i) {
                        k[i] = demoRandom<u32>(0u, 0xffffffffu) >> (r & 1 ? 3 : 0);   // odd ranks skewed low
                        kv[i].m_key = k[i] & 0xfff000ffu;
                        kv[i].m_value = (u32)(all.size() + i);
                    }
                    nIn[r] = (size_t)m;
                    in[r] = new Buffer<u32>(s.getDevice(r), m);
                    out[r] = new Buffer<u32>(s.getDevice(r), (u64)2 * G * (n + 13 * G));
                    kin[r] = new Buffer<uint2>(s.getDevice(r), m);
                    kout[r] = new Buffer<uint2>(s.getDevice(r), (u64)2 * G * (n + 13 * G));
                    in[r]->write(&k[0], m);
                    kin[r]->write((const uint2*)&kv[0], m);
                    all.insert(all.end(), k.begin(), k.end());
                    kall.insert(kall.end(), kv.begin(), kv.end());
                }
                s.waitForCompletion();
                s.radixSort(&in[0], &nIn[0], &out[0], &nOut[0]);
                s.radixSort(&kin[0], &nIn[0], &kout[0], &knOut[0]);
                s.waitForCompletion();
                std::vector<u32> got;
                std::vector<SortData> kgot;
                for (int r = 0; r < G; ++r) {
                    std::vector<u32> part(nOut[r] ? nOut[r] : 1);
                    std::vector<SortData> kpart(knOut[r] ? knOut[r] : 1);
                    if (nOut[r]) out[r]->read(&part[0], (int)nOut[r]);
                    if (knOut[r]) kout[r]->read((uint2*)&kpart[0], (int)knOut[r]);
                    s.waitForCompletion();
                    got.insert(got.end(), part.begin(), part.begin() + nOut[r]);
                    kgot.insert(kgot.end(), kpart.begin(), kpart.begin() + knOut[r]);
                }
                std::sort(all.begin(), all.end());
                std::stable_sort(kall.begin(), kall.end(), [](const SortData& a, const SortData& b) { return a.m_key < b.m_key; });
                check(got == all, "ShardedSort keys", n);
                bool same = kgot.size() == kall.size();
                for (size_t i = 0; same && i < kall.size(); ++i) same = kgot[i].m_key == kall[i].m_key && kgot[i].m_value == kall[i].m_value;
                check(same, "ShardedSort key-value (stable)", n);
                printf("test %.1fK per device x %d\n", n / 1024.f, G);
                for (int r = 0; r < G; ++r) { delete in[r]; delete out[r]; delete kin[r]; delete kout[r]; }
            }
        }
    }
    printf("[       %s ] Demo.ShardedSort (%.0f ms)\n", g_failed + adl_assert_failures() > before ? "FAIL" : "OK", now_ms() - t0);
}

}  // namespace

int main(int argc, char** argv)
{
    bool host = false;
    int gpus = 0;
    for (int i = 1; i < argc; ++i) {
        host |= !strcmp(argv[i], "--host");
        if (!strcmp(argv[i], "--gpus") && i + 1 < argc) gpus = atoi(argv[++i]);
    }

    DeviceUtils::Config cfg;
    cfg.m_type = host ? DeviceUtils::Config::DEVICE_CPU : DeviceUtils::Config::DEVICE_GPU;
    Device* d = DeviceUtils::allocate(host ? TYPE_HOST : TYPE_CL, cfg);
    if (adl_assert_failures()) {
        printf("cannot open the device\n");
        return 100;
    }
    char name[128];
    d->getDeviceName(name);
    printf("device: %s (%s path)\n", name, host ? "Adl/Host CPU" : "HIP");
    {
        Pprims p;
        demoSort32(d, p);
        demoSortKeyValue(d, p);
        demoFillCopy(d, p);
        if (!host) demoScan(d, p);   // scan has no host path in the reference either (Pprims.cpp:124-127)
        if (!host) demoSortWideValues(d, p);
    }
    if (!host) demoBufferUtils(d);
    DeviceUtils::deallocate(d);
    if (!host && gpus > 0) demoShardedSort(gpus);
    g_failed += adl_assert_failures();
    printf("%s: %d failed checks\n", g_failed ? "FAILED" : "PASSED", g_failed);
    return g_failed;
}
