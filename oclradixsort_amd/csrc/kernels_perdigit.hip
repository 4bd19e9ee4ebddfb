// kernels_perdigit.hip -- the device code of the per-digit passes (one-sweep histogram and chain kernels, three-kernel count and
// scatter, one-workgroup sort) in every tile / digit / ranking variant, instantiated here so that it compiles beside adlhip.hip
// (see kernels_large.hip).
#include <hip/hip_runtime.h>

#define ADLHIP_KERNEL static   // the headers' non-template kernels belong to adlhip.hip
#include "radix_kernels.hpp"
#include "onesweep_kernels.hpp"

#define X(...) template __global__ __VA_ARGS__;
#include "perdigit_kernels.inc"
#undef X
