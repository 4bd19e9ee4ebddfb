"""oclradixsort_amd -- MI355X-native (gfx950 HIP) radix sort / scan behind the OCLRadixSort
`Tahoe::Pprims` + `adl::` call surface.

Layout:
  csrc/        hand-written HIP kernels + the C-ABI shared library (include/adlhip.h)
  _lib.py      ctypes binding of that ABI (no fallback: missing library => error)
  adl.py       mirror of adl::DeviceUtils / Device / Buffer / Stopwatch
  pprims.py    mirror of Tahoe::Pprims (radixSort, scan)
  dist.py      multi-GPU MSB-bucket sharded sort over torch.distributed (RCCL)
"""
from ._lib import AdlHipError, LIB_PATH  # noqa: F401
from .adl import TYPE_CL, TYPE_HIP, TYPE_HOST, Buffer, Config, Device, DeviceUtils, Stopwatch  # noqa: F401
from .pprims import Pprims  # noqa: F401

__version__ = "0.1.0"
