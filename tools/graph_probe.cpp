// Diagnostic (not product): is the mid-size regime (32 Ki .. 4 Mi keys, 12 dependent launches) bound by the host's
// launch rate or by the device's kernel-to-kernel boundaries?  Times `reps` back-to-back sorts launched eagerly and
// replayed from a captured hipGraph.
//   hipcc -O2 -std=c++17 -I include tools/graph_probe.cpp -o tools/graph_probe -L oclradixsort_amd/lib -ladlhip -Wl,-rpath,'$ORIGIN/../oclradixsort_amd/lib'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "adlhip.h"
#define CK(x) do { if ((x) != 0) { fprintf(stderr, "fail %s: %s\n", #x, adlhip_last_error()); exit(1); } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "hip fail %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    adlhip_device* d;
    CK(adlhip_device_create(0, &d));
    hipStream_t s = (hipStream_t)adlhip_stream(d);
    for (int algo : {1, 0}) {
        CK(adlhip_set_param(d, "sort.algo", algo));
        for (size_t n : {(size_t)32768, (size_t)262144, (size_t)1048576, (size_t)4194304, (size_t)16777216}) {
            size_t tb, wb;
            CK(adlhip_radix_sort_scratch_bytes(d, ADLHIP_ELEM_U32, n, &tb, &wb));
            void *keys, *tmp, *work;
            CK(adlhip_malloc(d, n * 4, &keys)); CK(adlhip_malloc(d, tb, &tmp)); CK(adlhip_malloc(d, wb, &work));
            CK(adlhip_generate_keys(d, ADLHIP_ELEM_U32, keys, n, 1, 0));
            for (int i = 0; i < 3; ++i) CK(adlhip_radix_sort_u32(d, (uint32_t*)keys, (uint32_t*)tmp, work, wb, n, 32));
            CK(adlhip_sync(d));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) CK(adlhip_radix_sort_u32(d, (uint32_t*)keys, (uint32_t*)tmp, work, wb, n, 32));
            auto t1 = std::chrono::steady_clock::now();
            CK(adlhip_sync(d));
            auto t2 = std::chrono::steady_clock::now();
            const double issue_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
            const double eager_us = std::chrono::duration<double, std::micro>(t2 - t0).count() / reps;
            hipGraph_t g; hipGraphExec_t ge;
            HK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            CK(adlhip_radix_sort_u32(d, (uint32_t*)keys, (uint32_t*)tmp, work, wb, n, 32));
            HK(hipStreamEndCapture(s, &g));
            HK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 3; ++i) HK(hipGraphLaunch(ge, s));
            HK(hipStreamSynchronize(s));
            t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; ++i) HK(hipGraphLaunch(ge, s));
            HK(hipStreamSynchronize(s));
            t2 = std::chrono::steady_clock::now();
            const double graph_us = std::chrono::duration<double, std::micro>(t2 - t0).count() / reps;
            printf("algo %d n %9zu: eager %7.1f us/sort (host issue %6.1f us)   graph replay %7.1f us/sort\n", algo, n, eager_us,
                   issue_us, graph_us);
            HK(hipGraphExecDestroy(ge)); HK(hipGraphDestroy(g));
            CK(adlhip_free(d, keys, n * 4)); CK(adlhip_free(d, tmp, tb)); CK(adlhip_free(d, work, wb));
        }
    }
    CK(adlhip_device_destroy(d));
    return 0;
}
