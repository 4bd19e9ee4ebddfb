// Tahoe/Algorithm/Sort/RadixSort.h -- the CPU radix sort of the facade: the path Pprims::radixSort
// takes for an Adl TYPE_HOST device (reference: Tahoe/Algorithm/Sort/RadixSort.h:10-44, used at
// Pprims.cpp:202-212 and :306-316) and that a caller can invoke directly as the reference's test does
// (UnitTest/main.cpp:128, :158).  It is reachable ONLY through an explicitly created TYPE_HOST device
// or a direct call; GPU devices never fall back to it.
#pragma once
#include <Tahoe/Math/Math.h>

namespace Tahoe {

struct SortData {
    union {
        u32 m_key;
        struct { u16 m_key16[2]; };
    };
    u32 m_value;

    SortData() {}
    SortData(u32 key, u32 value) : m_key(key), m_value(value) {}
    friend bool operator<(const SortData& a, const SortData& b) { return a.m_key < b.m_key; }
};

class RadixSort {
public:
    enum { BITS_PER_PASS = 8, NUM_TABLES = (1 << BITS_PER_PASS) };
    static void sort(SortData* data, int n);   // stable, by m_key, ascending
    static void sort(u32* data, int n);        // ascending
};

}  // namespace Tahoe
