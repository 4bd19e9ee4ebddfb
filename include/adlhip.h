/*
 * adlhip.h -- C ABI of the MI355X (gfx950) HIP back-end that replaces the reference's OpenCL
 * back-end (the Adl/CL directory) and the device half of Tahoe::Pprims for the radix-sort / scan hot path.
 *
 * Plain C: opaque handle, raw device pointers, sizes.  No C++ or torch types cross this boundary.
 * Every function returns ADLHIP_SUCCESS (0) or ADLHIP_FAILURE (1) unless stated otherwise
 * (the reference defines the same two codes, Adl/Adl.h:22-23, but never returns them: its APIs are
 * void and fail through ADLASSERT, Tahoe/Math/Error.h:24-38).  Nothing throws across the boundary;
 * adlhip_last_error() returns the text of the calling thread's most recent failure.
 *
 * Threading: like the reference (one in-order command queue per device, Adl/CL/AdlCL.inl:303;
 * nothing re-entrant), calls on one adlhip_device must be serialised by the caller; distinct handles
 * may be driven from distinct threads.  All work is enqueued on the handle's one HIP stream and the
 * primitives return without synchronising, as the GPU branches of Pprims.cpp do.
 *
 * All citations are relative to the reference repository root.
 */
#ifndef ADLHIP_H
#define ADLHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADLHIP_SUCCESS 0 /* Adl/Adl.h:22 ADL_SUCCESS */
#define ADLHIP_FAILURE 1 /* Adl/Adl.h:23 ADL_FAILURE */

typedef struct adlhip_device adlhip_device;

/* Replaces Device::getDeviceName/getDeviceVendor/getMemSize/getMaxAllocationSize and
 * DeviceUtils::getNCUs (Adl/CL/AdlCL.inl:704-759). */
typedef struct adlhip_info {
    int32_t compute_units;      /* CL_DEVICE_MAX_COMPUTE_UNITS analogue (AdlCL.inl:704-709) */
    int32_t wavefront_size;     /* 64 on gfx950 */
    int32_t lds_bytes_per_cu;   /* sharedMemPerMultiprocessor */
    int32_t clock_khz;
    uint64_t total_mem_bytes;
    uint64_t max_alloc_bytes;
    char name[128];
    char arch[64];              /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    char vendor[32];
} adlhip_info;

/* ---- device lifetime ----------------------------------------------------------------------- */

/* DeviceUtils::getNDevices(TYPE_CL) -- Adl/Adl.inl:8-23.  Returns the count (0 if none / no driver). */
int adlhip_device_count(void);

/* DeviceUtils::allocate(TYPE_CL, Config{m_deviceIdx}) -- Adl/Adl.inl:73-98, DeviceCL::initialize
 * Adl/CL/AdlCL.inl:148-345.  device_idx beyond the last device is clamped to the last one
 * (AdlCL.inl:244).  Creates one in-order stream. */
int adlhip_device_create(int device_idx, adlhip_device** out);

/* Same, but enqueue on a caller-owned hipStream_t (e.g. torch's current stream) instead of creating
 * one.  No reference counterpart; used by the multi-GPU host code so sort and RCCL share a stream. */
int adlhip_device_create_on_stream(int device_idx, void* hip_stream, adlhip_device** out);

/* DeviceUtils::deallocate -- Adl/Adl.inl:100-105.  Like the reference (ADLASSERT(getUsedMemory()==0))
 * this FAILS, leaving the handle alive, while adlhip_used_bytes() != 0. */
int adlhip_device_destroy(adlhip_device* dev);

int adlhip_device_info(adlhip_device* dev, adlhip_info* out);

/* Device::getUsedMemory -- Adl/Adl.h:141; live bytes handed out by adlhip_malloc. */
uint64_t adlhip_used_bytes(adlhip_device* dev);

/* DeviceUtils::waitForCompletion(const Device*) -- Adl/Adl.inl:107-110 (clFinish, AdlCL.inl:567-570).
 * Also reports, as a failure, any device-side fault flag a kernel raised since the last sync. */
int adlhip_sync(adlhip_device* dev);

/* Stream-ordered fault check that never blocks, for callers that drain the stream by other means (a handle
 * made with adlhip_device_create_on_stream and synchronised through the stream's owner, e.g. torch): enqueues a
 * copy of the device's sticky fault word into pinned memory and reports -- as a failure, clearing the word -- what
 * the PREVIOUS such copy delivered once it has completed.  Call it once per batch: a look-back time-out or an
 * oversized segment of batch i surfaces at the check that follows the completion of batch i.  The live word the
 * waiters poll is cleared by the first kernel of every sort, so a failure never leaks into later sorts.
 * No reference counterpart (its errors are ADLASSERTs on the host, Tahoe/Math/Error.h:24-38). */
int adlhip_fault_check(adlhip_device* dev);

/* DeviceUtils::flush -- Adl/CL/AdlCL.inl:614-617.  HIP streams need no flush; kept for symmetry. */
int adlhip_flush(adlhip_device* dev);

/* The stream this handle enqueues on (hipStream_t as void*). */
void* adlhip_stream(adlhip_device* dev);

const char* adlhip_last_error(void);

/* ---- buffers: Buffer<T> alloc / copies / map -------------------------------------------------- */

/* Buffer<T>::allocate -> DeviceCL::allocate (Adl/CL/AdlCL.inl:356-420): device allocation, accounted
 * in the live-byte counter.  bytes == 0 yields *dptr == NULL and success. */
int adlhip_malloc(adlhip_device* dev, size_t bytes, void** dptr);
/* DeviceCL::deallocate (AdlCL.inl:422-439).  `bytes` must be the size given to adlhip_malloc. */
int adlhip_free(adlhip_device* dev, void* dptr, size_t bytes);

/* Buffer::write(host)/read(host)/write(Buffer) -- Adl/Adl.inl:273-303, AdlCL.inl:441-510.
 * Asynchronous and stream-ordered, like the non-blocking clEnqueue* calls they replace: the host
 * memory must stay valid until adlhip_sync().  Element offsets are folded into the pointers. */
int adlhip_memcpy_h2d(adlhip_device* dev, void* dst_dev, const void* src_host, size_t bytes);
int adlhip_memcpy_d2h(adlhip_device* dev, void* dst_host, const void* src_dev, size_t bytes);
int adlhip_memcpy_d2d(adlhip_device* dev, void* dst_dev, const void* src_dev, size_t bytes);

/* Buffer::clear / fill -- Adl/Adl.inl:305-315, AdlCL.inl:512-542 (byte-wise / 4-byte pattern). */
int adlhip_memset(adlhip_device* dev, void* dptr, int byte_value, size_t bytes);
int adlhip_fill_u32(adlhip_device* dev, void* dptr, uint32_t pattern, size_t count);
/* Pprims::fill(int | u32 | float4) -- Tahoe/ParallelPrimitives/Pprims.cpp:69-120 (FillIntKernel / FillU32Kernel /
 * FillF4Kernel, commented out in the reference): `count` copies of a 4-, 8- or 16-byte pattern read from host
 * memory at call time.  dptr must be aligned to pattern_bytes. */
int adlhip_fill_pattern(adlhip_device* dev, void* dptr, const void* pattern, size_t pattern_bytes, size_t count);

/* Buffer::getHostPtr / returnHostPtr -- Adl/Adl.inl:317-329, AdlCL.inl:544-565 (non-blocking
 * clEnqueueMapBuffer READ|WRITE / clEnqueueUnmapMemObject).  adlhip_map enqueues a device->pinned-host
 * copy and returns the host pointer; contents are valid after adlhip_sync().  adlhip_unmap enqueues the
 * host->device write-back and releases the staging memory once that copy has run; the device sees the
 * writes after adlhip_sync().  Exactly the call sequence of UnitTest/main.cpp:118-125.
 * adlhip_unmap: bytes = 0 writes back the whole mapping (Buffer::returnHostPtr carries no size). */
int adlhip_map(adlhip_device* dev, void* dptr, size_t bytes, void** hptr);
int adlhip_unmap(adlhip_device* dev, void* dptr, void* hptr, size_t bytes);

/* ---- the primitives: Tahoe::Pprims ----------------------------------------------------------- */

/* Element kinds the sort entry points handle. */
#define ADLHIP_ELEM_U32  0 /* Buffer<u32>   : Pprims::radixSort(..., Buffer<u32>&, ...)   Pprims.h:41 */
#define ADLHIP_ELEM_KV32 1 /* Buffer<uint2> : {x = key, y = value}                         Pprims.h:38 */
#define ADLHIP_ELEM_U64  2 /* 64-bit keys (BASELINE config #5; no reference API)                       */
#define ADLHIP_ELEM_SOA32 3 /* separate u32 key and u32 value arrays (*tmp_bytes is per array)           */

/* Scratch the caller must own, replacing Pprims' m_u32WorkBuffer[0] (ping-pong copy of the data,
 * Pprims.cpp:226-232, :332) and m_u32WorkBuffer[1] (histogram table, :229-230, :333-337).
 *   *tmp_bytes  : second data buffer, n elements
 *   *work_bytes : control scratch (digit tables / tile status words) and, for the sizes the large sort takes
 *                 ("sort.msd2": sorts of more than 1 Mi elements), its bucket and segment slabs -- about 1.5 x n elements
 *                 for whole u32 keys from 16 Mi keys up (their 16-bit second slab then lives in d_tmp; 64 Mi keys: 451 MB),
 *                 3 x n elements otherwise (64 Mi pairs: 1.7 GB; 256 Mi u64 keys: 6.1 GB).  The value
 *                 suffices for EVERY n' <= n with the current knobs (the need of a single n is not monotone:
 *                 smaller inputs use smaller tiles and so more status rows), so a caller may size its scratch
 *                 once for its largest batch; changing "sort.tile", "sort.digit_bits" or "sort.algo" later can
 *                 raise the requirement (the sort entry points re-check and fail loudly). */
int adlhip_radix_sort_scratch_bytes(adlhip_device* dev, int elem_kind, size_t n,
                                    size_t* tmp_bytes, size_t* work_bytes);

/* The same for a sort on `sort_bits` bits, at one of three levels (sizes: level 0 <= level 2 <= level 1):
 *   level 0: the minimum -- the reference's own contract (Pprims.cpp:332-337: the n-element partner array and a digit table of
 *            a few KiB).  Every sort entry point accepts a work buffer of this size; the sort then runs the per-digit
 *            three-kernel passes (64 Mi u32 keys: 75-81 instead of 206-215 Gkeys/s, profiles/r4_bench_n1.json).
 *   level 1: full speed -- what adlhip_radix_sort_scratch_bytes reports for whole keys.  A sort on fewer bits than the key has
 *            takes the stable form of the large sort, whose second slab cannot shrink to 16 bits per key, and so needs more.
 *   level 2: lean -- whole u32 / u64 keys keep the large sort (cursor form) with 12 % instead of 50 % of head-room in the
 *            first pass's bucket slabs: 64 Mi u32 keys 306 MB of work instead of 451, same speed on keys spread evenly over
 *            their range; keys whose density varies by more than ~10 % from one 256th of the range to the next go through
 *            the safety net inside the sort (correct, about 4 x slower; nothing is remembered between sorts).  Pairs, SoA and sorts
 *            on part of the key keep the stable form of the large sort with statistical head-room only (mean + 8 sd in the first
 *            pass's sub-slabs, + 7.5 sd in the segment slabs, instead of + 50 %): 64 Mi pairs 1.25 GB instead of 1.7.
 * With a work buffer between the levels, every path checks its own need and the sort takes the fastest one that fits. */
int adlhip_radix_sort_scratch_bytes_for(adlhip_device* dev, int elem_kind, size_t n, int sort_bits, int level,
                                        size_t* tmp_bytes, size_t* work_bytes);

/* Pprims::radixSort(const Device*, const Buffer<u32>& inout, int n, int sortBits=32)
 * -- Tahoe/ParallelPrimitives/Pprims.h:41, Pprims.cpp:304-406.
 * Sorts d_keys_inout[0..n) ascending by the low `sort_bits` bits of each key, stably with respect to
 * input order; the result is in d_keys_inout (odd pass counts are copied back, Pprims.cpp:400-403).
 * sort_bits: multiple of 4 in [4,32] (Pprims.cpp:330); anything else fails.  Unlike the reference
 * (n % 256 == 0, Pprims.cpp:327) any n >= 0 is accepted.  Enqueues and returns. */
int adlhip_radix_sort_u32(adlhip_device* dev, uint32_t* d_keys_inout, uint32_t* d_tmp,
                          void* d_work, size_t work_bytes, size_t n, int sort_bits);

/* Pprims::radixSort(const Device*, const Buffer<uint2>& inout, int n, int sortBits=32)
 * -- Pprims.h:38, Pprims.cpp:200-302.  Elements are 8-byte {u32 key (.x); u32 value (.y)} pairs
 * (Tahoe/Math/Math.h:175-188 == SortData, Tahoe/Algorithm/Sort/RadixSort.h:10-27); stable. */
int adlhip_radix_sort_kv32(adlhip_device* dev, void* d_pairs_inout, void* d_tmp,
                           void* d_work, size_t work_bytes, size_t n, int sort_bits);

/* Key-value sort on SEPARATE key and value arrays (structure of arrays): the layout of the reference's
 * never-launched SortAndScatterKernel(gSrc, gSrcVal, ...) (RadixSortKeyValueKernels.cl:354-509).
 * Same contract as adlhip_radix_sort_kv32: ascending by the low sort_bits key bits, stable; values follow
 * their keys.  d_tmp_keys / d_tmp_vals: n u32 each (adlhip_radix_sort_scratch_bytes with ADLHIP_ELEM_SOA32).
 * The histogram / count kernels read the key array only (half the bytes of the AoS layout). */
int adlhip_radix_sort_soa32(adlhip_device* dev, uint32_t* d_keys_inout, uint32_t* d_vals_inout,
                            uint32_t* d_tmp_keys, uint32_t* d_tmp_vals, void* d_work, size_t work_bytes,
                            size_t n, int sort_bits);

/* The same on keys of 4 or 8 bytes and values of 4, 8 or 16 bytes ("SoA key/value API ... and 64-bit values";
 * the reference's kernel, RadixSortKeyValueKernels.cl:354-509, takes `const u32* gSrc, const int* gSrcVal`).  Ascending by the
 * low sort_bits key bits (a multiple of 4 in [4, 8 * key_bytes]), stable, values follow their keys; n < 2^32.
 * (4, 4) is adlhip_radix_sort_soa32.  Every other width sorts {32 key bits, source index} pairs with the stable pair sort of
 * adlhip_radix_sort_kv32 -- once for u32 keys, twice (low dword, then high dword) for u64 keys on more than 32 bits -- and
 * fetches keys and values ONCE, at the end, from where the indices point; the value's width only costs in that gather.
 * d_tmp_keys: n keys (may be NULL for 4-byte keys), d_tmp_vals: n values; sizes and the work buffer's size from
 * adlhip_radix_sort_soa_scratch_bytes.  All buffers 16-byte aligned. */
int adlhip_radix_sort_soa_scratch_bytes(adlhip_device* dev, int key_bytes, int value_bytes, size_t n, int sort_bits,
                                        size_t* tmp_keys_bytes, size_t* tmp_vals_bytes, size_t* work_bytes);
int adlhip_radix_sort_soa(adlhip_device* dev, void* d_keys_inout, int key_bytes, void* d_vals_inout, int value_bytes,
                          void* d_tmp_keys, void* d_tmp_vals, void* d_work, size_t work_bytes, size_t n, int sort_bits);

/* 64-bit keys, ascending; sort_bits multiple of 4 in [4,64]. */
int adlhip_radix_sort_u64(adlhip_device* dev, uint64_t* d_keys_inout, uint64_t* d_tmp,
                          void* d_work, size_t work_bytes, size_t n, int sort_bits);

/* ---- segments finished in LDS (no reference counterpart) ------------------------------------- */

/* Sorts, stably and in place, every segment [d_seg_start[s], d_seg_start[s + 1]) of an array of u32 keys
 * (ADLHIP_ELEM_U32) or {key, value} pairs (ADLHIP_ELEM_KV32) by the low `low_bits` bits of its keys: one
 * workgroup per segment, the segment lives in LDS, one read and one write of global memory.  It is the
 * finishing pass of the hybrid sort ("sort.algo" = 2: two MSD passes of the kind
 * Tahoe/ClKernels/RadixSort32Kernels.cl:493-631 implements per digit, then this) and usable on its own.
 * d_seg_start: num_segments + 1 ascending element offsets in device memory.  max_segment: the caller's bound
 * on the largest segment (selects the LDS tile): at most 16384 keys / 8192 pairs for low_bits <= 24, half of
 * that up to 27 bits.  A segment that exceeds the tile is left unsorted and raises the device fault word
 * (reported by adlhip_sync / adlhip_fault_check). */
int adlhip_segment_sort(adlhip_device* dev, int elem_kind, void* d_data, const uint32_t* d_seg_start,
                        size_t num_segments, size_t max_segment, int low_bits);

/* Pprims::scan(const Device*, Buffer<int>& dst, const Buffer<int>& src, int n, u32* sumOut=0)
 * -- Pprims.h:35, Pprims.cpp:122-179.  Exclusive prefix sum, 32-bit wrap-around.  dst may equal src.
 * h_sum_or_null: when non-NULL the grand total is copied there (stream-ordered; valid after
 * adlhip_sync(), like the reference's non-blocking read at Pprims.cpp:164-167).  No n < 1,048,576
 * limit (the reference silently returns for numBlocks >= 4096, Pprims.cpp:134-138). */
int adlhip_scan_scratch_bytes(adlhip_device* dev, size_t n, size_t* work_bytes);
int adlhip_exclusive_scan_u32(adlhip_device* dev, uint32_t* d_dst, const uint32_t* d_src,
                              void* d_work, size_t work_bytes, size_t n, uint32_t* h_sum_or_null);

/* ---- multi-GPU helper: MSB-bucket partition (no reference counterpart) ----- */

/* Stable partition of n u32 keys into `num_buckets` (<= 256, power of two) contiguous segments by
 * their top log2(num_buckets) bits; d_counts_out[num_buckets] (u32) receives the segment sizes.
 * This is one radix pass on the most significant digit: the send side of the all-to-all exchange. */
int adlhip_partition_msb_u32(adlhip_device* dev, const uint32_t* d_keys_in, uint32_t* d_keys_out,
                             uint32_t* d_counts_out, void* d_work, size_t work_bytes,
                             size_t n, int num_buckets);
/* Same for n {u32 key, u32 value} pairs (the element of Pprims::radixSort(Buffer<uint2>), Pprims.h:38): partitioned by
 * the top bits of the KEY, stable, values travel with their keys. */
int adlhip_partition_msb_kv32(adlhip_device* dev, const void* d_pairs_in, void* d_pairs_out,
                              uint32_t* d_counts_out, void* d_work, size_t work_bytes,
                              size_t n, int num_buckets);

/* The same pass with the 256 top-byte totals handed out instead of folded: out = in, stably ordered by the key's
 * bits 24..31; d_totals256_out[b] (u32) = number of keys whose top byte is b.  The sharded sort all-reduces these
 * totals over the ranks and cuts the byte range into G contiguous pieces of near-equal population (balanced
 * splitters); any such cut yields G contiguous send segments of this output. */
int adlhip_partition_top_byte_u32(adlhip_device* dev, const uint32_t* d_keys_in, uint32_t* d_keys_out,
                                  uint32_t* d_totals256_out, void* d_work, size_t work_bytes, size_t n);
int adlhip_partition_top_byte_kv32(adlhip_device* dev, const void* d_pairs_in, void* d_pairs_out,
                                   uint32_t* d_totals256_out, void* d_work, size_t work_bytes, size_t n);

/* ---- sharded sort: ONE process, G devices (no reference counterpart) ------------------------
 *
 * The reference's API language is C++ (Tahoe/ParallelPrimitives/Pprims.h:35-41) and it drives one device
 * (Adl/Adl.h:90-94).  A group owns one adlhip_device per GPU and one RCCL communicator per device
 * (ncclCommInitAll; RCCL is loaded with dlopen on first use).  adlhip_sharded_sort_* sorts G shards that live
 * on the G devices into ONE global order: rank r ends up with a contiguous, ascending slice and the slices in
 * rank order are the sorted whole (pairs: stable in (source rank, position) order).  Steps: stable partition by
 * the top byte on every device (adlhip_partition_top_byte_*), ONE host synchronisation that brings the G x 256
 * totals to the host, balanced splitters (contiguous top-byte ranges of near-equal population; a rank's share
 * exceeds the mean by at most one byte value's population), one grouped ncclSend/ncclRecv exchange (every pair
 * of GPUs of a node has its own xGMI link), local sort.  The call returns with the exchange and the local sorts
 * ENQUEUED on the devices' streams: results are valid after adlhip_sync() on each device of the group.
 * Like every handle, a group must be driven by one host thread at a time. */
typedef struct adlhip_group adlhip_group;

/* device_indices: num_devices distinct HIP device indices, or NULL for 0 .. num_devices-1. */
int adlhip_group_create(const int* device_indices, int num_devices, adlhip_group** out);
/* Frees the group's scratch, communicators and device handles.  Fails (like adlhip_device_destroy,
 * Adl/Adl.inl:100-105) while the caller still holds memory allocated from one of its devices. */
int adlhip_group_destroy(adlhip_group* group);
int adlhip_group_size(adlhip_group* group);
/* The handle of rank `rank` (allocate the shards and outputs with it; NULL if out of range). */
adlhip_device* adlhip_group_device(adlhip_group* group, int rank);
/* Top-byte boundaries of the last sharded sort: bounds_out[G + 1], rank r owns top bytes [b[r], b[r+1]). */
int adlhip_group_last_bounds(adlhip_group* group, int* bounds_out);

/* d_shards_in[r]: n_in[r] keys on device r (left intact).  d_out[r]: room for out_capacity[r] keys on device r;
 * n_out[r] receives the size of rank r's slice.  If a slice does not fit, the call fails before anything is
 * exchanged and n_out holds the required sizes (1.25 x the mean + slack is enough unless one byte value dominates). */
int adlhip_sharded_sort_u32(adlhip_group* group, uint32_t* const* d_shards_in, const size_t* n_in,
                            uint32_t* const* d_out, const size_t* out_capacity, size_t* n_out);
/* Same for {u32 key, u32 value} pairs (8-byte elements, key in the low dword; Pprims.h:38). */
int adlhip_sharded_sort_kv32(adlhip_group* group, void* const* d_shards_in, const size_t* n_in,
                             void* const* d_out, const size_t* out_capacity, size_t* n_out);

/* ---- synthetic inputs: generated in place, reproducible by index ---------- */

/* key32(i) = hi32(splitmix64(seed*0x9E3779B97F4A7C15 + first_index + i)); key64 = the full 64 bits;
 * KV32 pair = {key32(i), value = (u32)(first_index + i)} (value = original index, as
 * UnitTest/main.cpp:152 does, so stability is checkable).  elem_kind: ADLHIP_ELEM_*. */
int adlhip_generate_keys(adlhip_device* dev, int elem_kind, void* dptr, size_t n, uint64_t seed,
                         uint64_t first_index);

/* ---- knobs ---------------------------------------------------------------------------------- */

/* Integer tunables, by name.  Unknown names fail.  Current names:
 *   "sort.algo"        -1 [default] = by size: n <= 16384 one workgroup does the whole sort in one launch;
 *                      n <= 2 Mi the mid-size sort ("sort.mid"), above it the large sort ("sort.msd2"); for what
 *                      those do not take (sort_bits < 16, a work buffer below level 1, sizes beyond their limits): below
 *                      24 MiB of data the three-kernel pass, from there on the one-sweep path
 *                      0 = onesweep (one sweep per digit, 16 decoupled look-back chains)
 *                      1 = three kernels per pass: count -> table scan -> sort+scatter (the
 *                          reference's pass structure, Pprims.cpp:357-398)
 *   "sort.digit_bits"  8 [default], 4 (the reference's R32SORT_BITS_PER_PASS, Pprims.h:31) or 7 (one-sweep passes only: 7,7,7,7,4)
 *   "sort.tile"        tile geometry variant (threads x elements per thread): -1 [default] = best known
 *                      per element size (512x32 for 4-byte, 1024x16 for 8-byte elements); 0 = 256x16,
 *                      1 = 512x16, 2 = 1024x16, 3 = 512x8, 4 = 1024x8, 5 = 256x32, 6 = 512x32
 *   "sort.rank"        1 [default when the device self-test passes] = in-tile ranking by lane-ordered
 *                      returning LDS atomics; 0 = ranking by 64-lane ballot match.  Mode 1 depends on a
 *                      hardware behaviour no ISA document promises (lanes of ONE returning DS atomic
 *                      instruction that hit the same address are served in ascending lane order); it is
 *                      checked at device creation and can be re-checked at any time
 *                      (adlhip_selftest_lds_order).  Mode 0 is the safe fallback: documented wave
 *                      intrinsics only.  Pairs (AoS and SoA) keep the stable large sort in mode 0 -- its passes, its
 *                      wave-per-segment finish and its safety net have ballot-ranked variants (up to 96 Mi pairs) --; keys
 *                      take the per-digit passes (~1.4 x the time).  Every LSD pass must be stable, so key-only
 *                      sorts depend on the ranking as much as key-value sorts do.
 *   "sort.lds_ordered" (read-only) result of that self-test
 *   "sort.mid"         1 [default] / 0: between 8 Ki and 2 Mi u32 keys (16 Ki and 1 Mi pairs), full 32-bit sorts take two
 *                      launches (u32 keys: MSD pass with bucket cursors, buckets finished in LDS) or three
 *                      (pairs; keys with a constant top byte: byte histograms, stable MSD pass, LDS finish)
 *                      instead of the per-digit passes.  Keys that do not fit the buckets are detected on the
 *                      device and sorted by a cooperative LSD sort inside the same launches (correct, slower);
 *                      the handle then steers later sorts of this size class by asynchronous hints (speed only;
 *                      results never depend on them).  2 / 3 force the two- / three-launch form (tests)
 *   "sort.msd2"        1 [default] / 0: sorts of 2 Mi .. 1088 Mi u32 keys, 100 K .. 260 Mi u64 keys and 1 Mi .. 260 Mi pairs on 16 or
 *                      more bits take two MSD passes into slabs of the work buffer plus one finish in LDS (six moves of
 *                      every element instead of nine); where the two digits sit is chosen on the device from a sample of
 *                      the keys (inside the low sort_bits bits).  Whole keys: runs are placed with atomic cursors (equal keys
 *                      are indistinguishable), from 192 Mi u32 / 48 Mi u64 keys the first (or both) passes by look-back;
 *                      pairs and sorts on part of the key: by look-back, stably.  The slabs give every bucket the same room:
 *                      keys whose density varies by more than ~45 % over their range, or that repeat a few values, do not fit.
 *                      That is detected on the device -- by the sort's first kernel when its 2048 sampled keys repeat
 *                      themselves (the passes then leave at once), else by the passes, which stop at their next tile -- and
 *                      the sort's own offsets kernel then sorts the untouched input: keys that take at most 256 values by
 *                      counting, pairs with such keys by one stable pass on the key's rank among them ("sort.dict"), anything
 *                      else by four (eight) LSD passes with grid-wide barriers between them (64 Mi u32 keys: 0.30-0.44 ms
 *                      and 1.0-1.1 ms instead of 0.32; 64 Mi pairs: 0.6-0.75 and 1.4-1.5 instead of 0.75;
 *                      profiles/r4_safety_net.txt).  Nothing is reported to the host and
 *                      nothing is remembered between sorts: a sort entry point never waits, and the first sort of an input
 *                      takes the time its hundredth does.  (Rounds 2-3 kept such keys off this path by a probe launch,
 *                      pinned-memory reports and a back-off counter in the handle; all of that is gone.)
 *                      2 forces the path from 1 Mi elements (tests), 3 / 4 / 5 force its stable / cursor / hybrid form
 *   "sort.binfinish"   1 [default]: whole u64 keys finish their segments by one counting pass on the top bits below the
 *                      digits + whole-key compares inside the bins (where a segment holds ~384 keys and more); 0: the
 *                      wave-per-segment LSD finish; 2: always, u32 keys too (tests, measurements)
 *   "sort.dict"        1 [default] / 0: the large sort's safety net first samples 16 Ki keys (if the sort's first kernel saw its
 *                      samples repeat often enough for that to be possible); if they take at most 256 distinct values
 *                      (whole-key sorts only) it sorts u32 / u64 keys by counting -- dictionary, one read, one write: equal
 *                      keys are interchangeable -- and {key, value} pairs by ONE stable pass on the key's rank in the
 *                      dictionary; u32 keys of up to 4096 values are counted with a larger dictionary (64 Ki samples); it
 *                      falls through to its LSD passes when a key misses the dictionary
 *   "sort.net_lookback" 1 [default] / 0: the LSD passes of the large sort's safety net on whole keys are look-back passes -- the
 *                      one-sweep path's histogram, tables and tile body, taken in turns by the net's resident workgroups, four
 *                      passes at a time (u64 keys: two rounds) -- instead of count -> scan -> scatter passes with per-workgroup
 *                      carries (which sorts on part of the key and SoA arrays always get); same result, ~25 % less time
 *   "partition.lookback" 1 [default] / 0: adlhip_partition_* on 24 MiB of data and more, with a work buffer of the sort's
 *                      full-speed size, is one look-back pass (histogram + chain kernel of the one-sweep path) instead of
 *                      count -> scan -> scatter; the same output bit for bit
 *   "stat.net_runs", "stat.net_counting" (read-only; reading waits for the stream) how often the large sort's safety net has run
 *                      on this handle, and how often it sorted by counting
 *   "debug.net_stamp0" .. "debug.net_stamp9" (read-only, diagnostic) when workgroup 0 of the handle's last safety net reached its
 *                      phase boundaries, 10-ns ticks (tools/net_phases.py)
 *   "debug.resident_wgs" workgroups the device certainly keeps resident at once (asked of the runtime at creation); the
 *                      paths whose safety nets hold a grid-wide barrier over 256 workgroups are taken only when it is
 *                      >= 256.  Setting it stands in for a small partition (tests); 0 = ask the device again
 *   "profile"          0/1: bracket every kernel launch with hipEvents (Device::toggleProfiling,
 *                          Adl/Adl.h:142, AdlKernelUtilsCL.inl:654-677) */
int adlhip_set_param(adlhip_device* dev, const char* name, int value);
int adlhip_get_param(adlhip_device* dev, const char* name, int* value);

/* ---- timing / profiling ----------------------------------------------------------------------- */

/* adl::Stopwatch (Adl/AdlStopwatch.h:60-83) on the handle's stream: start/split/stop map onto
 * hipEvent records; elapsed is device time between two recorded events. */
typedef struct adlhip_event adlhip_event;
int adlhip_event_create(adlhip_device* dev, adlhip_event** out);
int adlhip_event_record(adlhip_device* dev, adlhip_event* ev);
int adlhip_event_elapsed_ms(adlhip_device* dev, adlhip_event* start, adlhip_event* stop, float* ms);
int adlhip_event_destroy(adlhip_device* dev, adlhip_event* ev);
/* DeviceCL::waitForCompletion(const SyncObject*) / isComplete(const SyncObject*) -- Adl/CL/AdlCL.inl:572-612 (clWaitForEvents /
 * clGetEventInfo on the event a copy or launch was given): wait for / poll the point of the stream at which `ev` was last
 * recorded.  An event that was never recorded counts as complete. */
int adlhip_event_synchronize(adlhip_device* dev, adlhip_event* ev);
int adlhip_event_query(adlhip_device* dev, adlhip_event* ev, int* done_out);

/* Per-kernel launch timing collected while "profile" = 1 (replaces the per-launch CSV rows of
 * Adl/CL/AdlKernelUtilsCL.inl:664-677).  adlhip_profile_count synchronises the stream and folds the
 * pending event pairs; entry i is then readable with adlhip_profile_get. */
int adlhip_profile_reset(adlhip_device* dev);
int adlhip_profile_count(adlhip_device* dev);
int adlhip_profile_get(adlhip_device* dev, int i, char name_out[64], uint64_t* launches, double* total_ms);
/* Append the table as CSV rows "kernel",launches,total_ms,avg_ms to `path` (header written when the file is
 * new) -- the reference appends a row per launch to ProfileCL.<device>.<driver>.csv
 * (Adl/CL/AdlKernelUtilsCL.inl:664-677); here the rows are per kernel, folded since the last reset. */
int adlhip_profile_write_csv(adlhip_device* dev, const char* path);

/* ---- bandwidth probes (diagnostics for bench.py: empirical HBM ceilings) ---------------------- */
int adlhip_probe_copy(adlhip_device* dev, void* d_dst, const void* d_src, size_t bytes);
int adlhip_probe_read(adlhip_device* dev, const void* d_src, size_t bytes, void* d_sink8);
/* The same with cache-policy hints (hints bit 0 = non-temporal loads, bit 1 = non-temporal stores) and a choice of grid
 * (grid_per_cu workgroups of 256 threads per CU, 0 = 8): bench.py measures every variant on buffers that are cold in every
 * cache and reports the best one as the copy / read ceiling of the box. */
int adlhip_probe_copy_ex(adlhip_device* dev, void* d_dst, const void* d_src, size_t bytes, int hints, int grid_per_cu);
int adlhip_probe_read_ex(adlhip_device* dev, const void* d_src, size_t bytes, void* d_sink8, int hints, int grid_per_cu);

/* Re-runs the device self-test that "sort.rank" = 1 rests on (returning DS atomics of one wave instruction
 * resolve colliding lanes in ascending lane order; ranks are compared with ballot/mbcnt ranks) with `workgroups`
 * workgroups of 256 and of 1024 threads, and BLOCKS until the number of disagreements is in *mismatches
 * (0 = the property holds).  Device creation runs it once on an idle chip; stress tests call it on a second
 * handle while sorts run on the first.  No reference counterpart. */
int adlhip_selftest_lds_order(adlhip_device* dev, int workgroups, uint32_t* mismatches);

/* Self-test of the key probe's sampling (hybrid_kernels.hpp probe_sample_index): computes, on the device, the 16384 positions
 * the probe would read in an array of n elements (16384 <= n) and BLOCKS until *max_index holds the largest of them (must be
 * < n) and *out_of_cell the number of positions outside their 16384th of the array (must be 0).  A regression hook: a
 * compiler expansion of an integer remainder once sent 13 of the samples 64 MiB past a 7.7 Mi-key array.  No reference
 * counterpart. */
int adlhip_selftest_probe_positions(adlhip_device* dev, size_t n, uint32_t* max_index, uint32_t* out_of_cell);

const char* adlhip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ADLHIP_H */
