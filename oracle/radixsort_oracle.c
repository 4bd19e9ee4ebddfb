/*
 * oracle/radixsort_oracle.c -- CPU restatement of the reference's CPU radix sort and
 * of the test's sequential exclusive scan.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (libadlhip.so and the
 * include/ facade) never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks every function below against
 *   (a) the golden vectors in tests/golden/ that were produced by the reference's own
 *       Tahoe::RadixSort::sort compiled from /root/reference (oracle/Makefile ->
 *       oracle/_ref/libref.so, generator tests/golden/make_golden.py), and
 *   (b) oracle/_ref/libref.so directly on seeded inputs whenever that file is present.
 *
 * What the reference does (citations relative to /root/reference):
 *   Tahoe/Algorithm/Sort/RadixSort.cpp:58-104   RadixSort::sort(u32*, int)
 *   Tahoe/Algorithm/Sort/RadixSort.cpp:10-56    RadixSort::sort(SortData*, int)
 *   Tahoe/Algorithm/Sort/RadixSort.h:10-27      SortData {u32 m_key; u32 m_value;}
 *   UnitTest/main.cpp:193-199                   running-sum check of Pprims::scan
 * Stable LSD radix sort, 8 bits per pass, passes over startBit = 0,8,16,24; each pass
 * zeroes a 256-entry table, counts digits, exclusive-scans the table, then distributes
 * dst[table[d] + counter[d]++] = src[i] in input order, and swaps src/dst.  After the
 * even number of passes the result is back in the caller's array.
 *
 * The restatement keeps that structure (count, exclusive scan, in-order distribute,
 * ping-pong) but folds table[]+counter[] into one running cursor per digit and is
 * generic over the pass count so that the 64-bit-key configuration (BASELINE config #5,
 * which has no reference API) is the same algorithm run for 8 passes.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_BITS_PER_PASS 8               /* RadixSort.h:40  BITS_PER_PASS = 8 */
#define ORACLE_NUM_TABLES (1 << ORACLE_BITS_PER_PASS) /* RadixSort.h:41 */

/* One LSD pass over 32-bit elements.  RadixSort.cpp:68-98. */
static void pass_u32(const uint32_t *src, uint32_t *dst, size_t n, int start_bit)
{
    size_t cursor[ORACLE_NUM_TABLES];
    memset(cursor, 0, sizeof(cursor));                       /* :70-73 */
    for (size_t i = 0; i < n; i++)                           /* :75-79 */
        cursor[(src[i] >> start_bit) & (ORACLE_NUM_TABLES - 1)]++;
    size_t sum = 0;                                          /* :82-89 */
    for (int d = 0; d < ORACLE_NUM_TABLES; d++) {
        size_t c = cursor[d];
        cursor[d] = sum;
        sum += c;
    }
    for (size_t i = 0; i < n; i++) {                         /* :92-98 */
        uint32_t d = (src[i] >> start_bit) & (ORACLE_NUM_TABLES - 1);
        dst[cursor[d]++] = src[i];
    }
}

/* One LSD pass over 64-bit elements whose sort key is `key_bits` wide starting at bit 0
 * of the little-endian 8-byte element.  For SortData the key is the low dword
 * (RadixSort.h:12-17, m_key first), for u64 keys it is the whole element. */
static void pass_e64(const uint64_t *src, uint64_t *dst, size_t n, int start_bit)
{
    size_t cursor[ORACLE_NUM_TABLES];
    memset(cursor, 0, sizeof(cursor));                       /* RadixSort.cpp:22-25 */
    for (size_t i = 0; i < n; i++)                           /* :27-31 */
        cursor[(src[i] >> start_bit) & (ORACLE_NUM_TABLES - 1)]++;
    size_t sum = 0;                                          /* :34-41 */
    for (int d = 0; d < ORACLE_NUM_TABLES; d++) {
        size_t c = cursor[d];
        cursor[d] = sum;
        sum += c;
    }
    for (size_t i = 0; i < n; i++) {                         /* :44-50 */
        uint32_t d = (uint32_t)(src[i] >> start_bit) & (ORACLE_NUM_TABLES - 1);
        dst[cursor[d]++] = src[i];
    }
}

/* RadixSort::sort(u32*, int)  -- RadixSort.cpp:58-104.  In place, ascending, all 32 bits.
 * Returns 0, or 1 if the temporary could not be allocated (the reference's new[] throws). */
int oracle_radix_sort_u32(uint32_t *data, size_t n)
{
    if (n == 0) return 0;
    uint32_t *work = (uint32_t *)malloc(n * sizeof(uint32_t)); /* :60 */
    if (!work) return 1;
    uint32_t *src = data, *dst = work;                        /* :65-66 */
    for (int start_bit = 0; start_bit < 32; start_bit += ORACLE_BITS_PER_PASS) { /* :68 */
        pass_u32(src, dst, n, start_bit);
        uint32_t *t = src; src = dst; dst = t;                /* :100 swap2 */
    }
    free(work);                                               /* :103 */
    return 0;
}

/* RadixSort::sort(SortData*, int) -- RadixSort.cpp:10-56.  Elements are {u32 key, u32 value}
 * pairs (8 bytes, key first); sorted by key, stable, so equal keys keep input order. */
int oracle_radix_sort_kv32(uint64_t *pairs, size_t n)
{
    if (n == 0) return 0;
    uint64_t *work = (uint64_t *)malloc(n * sizeof(uint64_t)); /* :12 */
    if (!work) return 1;
    uint64_t *src = pairs, *dst = work;
    for (int start_bit = 0; start_bit < 32; start_bit += ORACLE_BITS_PER_PASS) { /* :20 */
        pass_e64(src, dst, n, start_bit);
        uint64_t *t = src; src = dst; dst = t;                /* :52 */
    }
    free(work);
    return 0;
}

/* 64-bit keys: the same algorithm run over all 8 bytes (BASELINE config #5; no reference API,
 * the output of a stable total-order sort is unique so this equals std::sort on u64). */
int oracle_radix_sort_u64(uint64_t *keys, size_t n)
{
    if (n == 0) return 0;
    uint64_t *work = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!work) return 1;
    uint64_t *src = keys, *dst = work;
    for (int start_bit = 0; start_bit < 64; start_bit += ORACLE_BITS_PER_PASS) {
        pass_e64(src, dst, n, start_bit);
        uint64_t *t = src; src = dst; dst = t;
    }
    free(work);
    return 0;
}

/* What the GPU branch of Pprims::radixSort computes when sortBits < 32
 * (Tahoe/ParallelPrimitives/Pprims.cpp:357, :330): the elements ordered, stably with respect
 * to input order, by their low `sort_bits` key bits only.  The CPU branch of the reference
 * refuses sortBits != 32 (Pprims.cpp:308), so this is derived from the GPU branch's contract:
 * 4-bit LSD passes for ib = 0,4,..,sortBits-4, each stable.  Restated here as stable LSD passes
 * of up to 8 bits over the same bit range, which yields the identical (unique) ordering. */
int oracle_radix_sort_u32_bits(uint32_t *data, size_t n, int sort_bits)
{
    if (n == 0 || sort_bits <= 0) return 0;
    uint32_t *work = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (!work) return 1;
    uint32_t *src = data, *dst = work;
    for (int start_bit = 0; start_bit < sort_bits; start_bit += 4) {
        /* 4-bit digit pass: mask the digit to 4 bits by shifting a copy -- done by a
         * dedicated loop so the digit width matches Pprims.h:31 R32SORT_BITS_PER_PASS. */
        size_t cursor[16];
        memset(cursor, 0, sizeof(cursor));
        for (size_t i = 0; i < n; i++) cursor[(src[i] >> start_bit) & 15u]++;
        size_t sum = 0;
        for (int d = 0; d < 16; d++) { size_t c = cursor[d]; cursor[d] = sum; sum += c; }
        for (size_t i = 0; i < n; i++) dst[cursor[(src[i] >> start_bit) & 15u]++] = src[i];
        uint32_t *t = src; src = dst; dst = t;
    }
    if (src != data) memcpy(data, src, n * sizeof(uint32_t)); /* Pprims.cpp:400-403 copy-back */
    free(work);
    return 0;
}

int oracle_radix_sort_e64_bits(uint64_t *data, size_t n, int sort_bits)
{
    if (n == 0 || sort_bits <= 0) return 0;
    uint64_t *work = (uint64_t *)malloc(n * sizeof(uint64_t));
    if (!work) return 1;
    uint64_t *src = data, *dst = work;
    for (int start_bit = 0; start_bit < sort_bits; start_bit += 4) {
        size_t cursor[16];
        memset(cursor, 0, sizeof(cursor));
        for (size_t i = 0; i < n; i++) cursor[(src[i] >> start_bit) & 15u]++;
        size_t sum = 0;
        for (int d = 0; d < 16; d++) { size_t c = cursor[d]; cursor[d] = sum; sum += c; }
        for (size_t i = 0; i < n; i++) dst[cursor[(src[i] >> start_bit) & 15u]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != data) memcpy(data, src, n * sizeof(uint64_t)); /* Pprims.cpp:298-301 */
    free(work);
    return 0;
}

/* Key-value sort on SEPARATE key and value arrays with keys of 4 or 8 bytes and values of any width: the contract of
 * the reference's SoA kernel (Tahoe/ClKernels/RadixSortKeyValueKernels.cl:354-509: gSrc / gSrcVal -> gDst / gDstVal, one stable
 * 4-bit pass per call) run for ib = 0,4,..,sortBits-4 as Pprims.cpp:357 does: keys ascending by their low sort_bits bits, equal
 * ones in input order, every value beside its key.  Restated as the same stable 4-bit LSD passes over an index permutation
 * (the values' width then does not enter the passes), applied to both arrays at the end.
 * For 4-byte keys and values this must equal oracle_radix_sort_kv32 on the packed pairs (tests/test_oracle.py pins it there). */
int oracle_radix_sort_soa(void *keys, int key_bytes, void *vals, int value_bytes, size_t n, int sort_bits)
{
    if ((key_bytes != 4 && key_bytes != 8) || value_bytes <= 0 || sort_bits < 0 || sort_bits > 8 * key_bytes) return 2;
    if (n == 0 || sort_bits == 0) return 0;
    uint64_t *k = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint32_t *a = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *b = (uint32_t *)malloc(n * sizeof(uint32_t));
    unsigned char *out = (unsigned char *)malloc(n * (size_t)(key_bytes > value_bytes ? key_bytes : value_bytes));
    if (!k || !a || !b || !out) { free(k); free(a); free(b); free(out); return 1; }
    for (size_t i = 0; i < n; i++) {
        k[i] = key_bytes == 4 ? ((const uint32_t *)keys)[i] : ((const uint64_t *)keys)[i];
        a[i] = (uint32_t)i;
    }
    uint32_t *src = a, *dst = b;
    for (int start_bit = 0; start_bit < sort_bits; start_bit += 4) {
        size_t cursor[16];
        memset(cursor, 0, sizeof(cursor));
        for (size_t i = 0; i < n; i++) cursor[(k[src[i]] >> start_bit) & 15u]++;
        size_t sum = 0;
        for (int d = 0; d < 16; d++) { size_t c = cursor[d]; cursor[d] = sum; sum += c; }
        for (size_t i = 0; i < n; i++) dst[cursor[(k[src[i]] >> start_bit) & 15u]++] = src[i];
        uint32_t *t = src; src = dst; dst = t;
    }
    for (size_t i = 0; i < n; i++) memcpy(out + i * (size_t)key_bytes, (const unsigned char *)keys + (size_t)src[i] * key_bytes, (size_t)key_bytes);
    memcpy(keys, out, n * (size_t)key_bytes);
    for (size_t i = 0; i < n; i++) memcpy(out + i * (size_t)value_bytes, (const unsigned char *)vals + (size_t)src[i] * value_bytes, (size_t)value_bytes);
    memcpy(vals, out, n * (size_t)value_bytes);
    free(k); free(a); free(b); free(out);
    return 0;
}

/* Exclusive prefix sum with 32-bit wrap-around, as the test checks Pprims::scan:
 * UnitTest/main.cpp:193-199 (ans starts at 0; h[i] must equal ans; ans += cpu[i]).
 * Returns the grand total (what Pprims::scan hands back through sumOut, Pprims.cpp:164-167). */
uint32_t oracle_exclusive_scan_u32(uint32_t *dst, const uint32_t *src, size_t n)
{
    uint32_t ans = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t v = src[i];
        dst[i] = ans;
        ans += v;
    }
    return ans;
}

/* FNV-1a 64 over raw bytes: the digest the golden tables use for the large sizes
 * (offset basis 0xcbf29ce484222325, prime 0x100000001b3). */
uint64_t oracle_fnv1a64(const void *bytes, size_t nbytes)
{
    const uint8_t *p = (const uint8_t *)bytes;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < nbytes; i++) {
        h ^= p[i];
        h *= 0x100000001b3ull;
    }
    return h;
}

/* Synthetic inputs, reproducible by index:
 *   key32(i) = hi32(splitmix64(seed*0x9E3779B97F4A7C15 + i)),  key64(i) = splitmix64(...). */
static inline uint64_t splitmix64_at(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void oracle_fill_keys_u32(uint32_t *dst, size_t n, uint64_t seed, uint64_t first_index)
{
    uint64_t base = seed * 0x9E3779B97F4A7C15ull + first_index;
    for (size_t i = 0; i < n; i++) dst[i] = (uint32_t)(splitmix64_at(base + i) >> 32);
}

void oracle_fill_keys_u64(uint64_t *dst, size_t n, uint64_t seed, uint64_t first_index)
{
    uint64_t base = seed * 0x9E3779B97F4A7C15ull + first_index;
    for (size_t i = 0; i < n; i++) dst[i] = splitmix64_at(base + i);
}

/* {key32(i), value = first_index + i} pairs (value = original index makes stability checkable,
 * as UnitTest/main.cpp:152 does). */
void oracle_fill_pairs_kv32(uint64_t *dst, size_t n, uint64_t seed, uint64_t first_index)
{
    uint64_t base = seed * 0x9E3779B97F4A7C15ull + first_index;
    for (size_t i = 0; i < n; i++) {
        uint64_t key = splitmix64_at(base + i) >> 32;
        dst[i] = key | ((uint64_t)(uint32_t)(first_index + i) << 32);
    }
}

/* The Demo recipe's data generator (UnitTest/main.cpp:76-86): srand(seed) once per size, then
 * value = (T)(min + r*(max-min)) with r = min(RAND_MAX-1, rand())/RAND_MAX as a double. */
void oracle_demo_fill_u32(uint32_t *dst, size_t n, unsigned seed)
{
    srand(seed);                                              /* main.cpp:77, :109 */
    for (size_t i = 0; i < n; i++) {
        double rv = (double)rand();
        double lim = (double)RAND_MAX - 1.0;
        double r = (rv < lim ? rv : lim) / (double)RAND_MAX;  /* main.cpp:83 */
        uint32_t range = 0xffffffffu - 0u;                    /* main.cpp:84 */
        dst[i] = (uint32_t)(0u + r * range);                  /* main.cpp:85 */
    }
}

void oracle_demo_fill_kv32(uint64_t *dst, size_t n, unsigned seed)
{
    srand(seed);
    for (size_t i = 0; i < n; i++) {                          /* main.cpp:150-153 */
        double rv = (double)rand();
        double lim = (double)RAND_MAX - 1.0;
        double r = (rv < lim ? rv : lim) / (double)RAND_MAX;
        uint32_t key = (uint32_t)(0u + r * 0xffffffffu);
        dst[i] = (uint64_t)key | ((uint64_t)(uint32_t)i << 32);
    }
}

void oracle_demo_fill_scan(uint32_t *dst, size_t n, unsigned seed)
{
    srand(seed);
    for (size_t i = 0; i < n; i++) {                          /* main.cpp:181-184, getRandom(0,0xf) */
        double rv = (double)rand();
        double lim = (double)RAND_MAX - 1.0;
        double r = (rv < lim ? rv : lim) / (double)RAND_MAX;
        int range = 0xf - 0;
        dst[i] = (uint32_t)(int)(0 + r * range);
    }
}
