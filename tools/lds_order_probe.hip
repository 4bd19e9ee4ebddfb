// Micro-probe (diagnostic, not product): does a returning LDS atomic add issued by all 64 lanes of one
// wave-instruction to conflicting addresses resolve in ascending lane order on gfx950?
// If yes, rank = ds_add_rtn(&cnt[digit], 1) would be a stable in-wave ranking in ONE DS instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void probe(const unsigned* digits, unsigned* out, int items, int bins)
{
    extern __shared__ unsigned cnt[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned* c = cnt + w * bins;
    for (int b = lane; b < bins; b += 64) c[b] = 0;
    for (int j = 0; j < items; ++j) {
        size_t idx = ((size_t)blockIdx.x * (blockDim.x >> 6) + w) * items * 64 + (size_t)j * 64 + lane;
        unsigned d = digits[idx] % bins;
        out[idx] = atomicAdd(&c[d], 1u);
    }
}
int main()
{
    const int blocks = 512, threads = 256, items = 16;
    int bad_total = 0;
    for (int bins : {1, 2, 16, 256}) {
        size_t n = (size_t)blocks * threads * items;
        std::vector<unsigned> h(n), o(n);
        srand(bins);
        for (auto& x : h) x = rand();
        unsigned *d, *r;
        hipMalloc(&d, n * 4); hipMalloc(&r, n * 4);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        probe<<<blocks, threads, (threads / 64) * bins * 4>>>(d, r, items, bins);
        hipMemcpy(o.data(), r, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t wv = 0; wv < n / (64 * items); ++wv) {
            std::vector<unsigned> c(bins, 0);
            for (int j = 0; j < items; ++j)
                for (int l = 0; l < 64; ++l) {
                    size_t idx = wv * items * 64 + (size_t)j * 64 + l;
                    unsigned dd = h[idx] % bins;
                    if (o[idx] != c[dd]) ++bad;
                    c[dd]++;
                }
        }
        printf("lds_order_probe bins=%d: %zu of %zu ranks differ from ascending-lane order\n", bins, bad, n);
        bad_total += bad != 0;
        hipFree(d); hipFree(r);
    }
    printf("lds_order_probe verdict: %s\n", bad_total ? "NOT lane-ordered" : "lane-ordered in every trial");
    return 0;
}
