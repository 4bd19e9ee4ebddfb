// Tahoe/Math/Math.h -- the handful of scalar typedefs / helpers the sort path uses.
// Reference: Tahoe/Math/Math.h:19 (NEXTMULTIPLEOF), :53-60 (nextPowerOf2), :90-93 (u8..u64),
// :95-112 (float4, as data), :175-188 (uint2), :230-242 (min2/max2), :324-330 (swap2).  The renderer's float/matrix math in the
// reference header is out of scope (SURVEY.md section 2.1 #9b).
#pragma once
#include <stddef.h>
#include <stdlib.h>
#include <Tahoe/Math/Error.h>

#define NEXTMULTIPLEOF(num, alignment) ((((num) + (alignment)-1) / (alignment)) * (alignment))

#ifndef TH_DECLARE_ALLOCATOR
#define TH_DECLARE_ALLOCATOR(T)   /* Tahoe/Base/Memory/AllocatorBase.h:30-85: plain new/delete here */
#endif

namespace Tahoe {

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

struct uint2 { u32 x, y; };
struct int2 { int x, y; };
struct uint4 { u32 x, y, z, w; };
struct int4 { int x, y, z, w; };
struct alignas(16) float4 { float x, y, z, w; };   // Math.h:95-112 (16-byte aligned; only moved around here, never computed on)

template <typename T> inline T max2(const T& a, const T& b) { return a > b ? a : b; }
template <typename T> inline T min2(const T& a, const T& b) { return a < b ? a : b; }
template <typename T> inline void swap2(T& a, T& b) { T t = a; a = b; b = t; }

template <typename T> inline T nextPowerOf2(T n)
{
    T p = 1;
    while (p < n) p <<= 1;
    return p;
}

}  // namespace Tahoe
