// CPU radix sort of the facade: the TYPE_HOST path of Pprims::radixSort and a directly callable
// Tahoe::RadixSort::sort, with the reference's contract (Tahoe/Algorithm/Sort/RadixSort.cpp:10-104):
// ascending, stable, in place, single-threaded, 8-bit digits over the 32-bit key.
// Own implementation: one counting pass builds all four digit histograms, passes whose digit is constant
// over the whole input are skipped, and the ping-pong ends in the caller's array by construction.
#include <Tahoe/Algorithm/Sort/RadixSort.h>

#include <string.h>

namespace Tahoe {

namespace {

inline u32 keyOf(const u32& v) { return v; }
inline u32 keyOf(const SortData& v) { return v.m_key; }

template <typename T>
void lsdSort(T* data, int n)
{
    if (n <= 1) return;
    enum { PASSES = 32 / RadixSort::BITS_PER_PASS, BINS = RadixSort::NUM_TABLES };
    size_t hist[PASSES][BINS];
    memset(hist, 0, sizeof(hist));
    for (int i = 0; i < n; ++i) {
        const u32 k = keyOf(data[i]);
        for (int p = 0; p < PASSES; ++p) hist[p][(k >> (p * RadixSort::BITS_PER_PASS)) & (BINS - 1)]++;
    }
    T* work = new T[n];
    T* src = data;
    T* dst = work;
    for (int p = 0; p < PASSES; ++p) {
        size_t* h = hist[p];
        bool trivial = false;
        for (int b = 0; b < BINS; ++b)
            if (h[b] == (size_t)n) { trivial = true; break; }   // every key has the same digit: order unchanged
        if (trivial) continue;
        size_t sum = 0;
        for (int b = 0; b < BINS; ++b) {
            const size_t c = h[b];
            h[b] = sum;
            sum += c;
        }
        const int shift = p * RadixSort::BITS_PER_PASS;
        for (int i = 0; i < n; ++i) dst[h[(keyOf(src[i]) >> shift) & (BINS - 1)]++] = src[i];
        T* t = src; src = dst; dst = t;
    }
    if (src != data) memcpy(data, src, sizeof(T) * (size_t)n);
    delete[] work;
}

}  // namespace

void RadixSort::sort(SortData* data, int n) { lsdSort(data, n); }
void RadixSort::sort(u32* data, int n) { lsdSort(data, n); }

}  // namespace Tahoe
