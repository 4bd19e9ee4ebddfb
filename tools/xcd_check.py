import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, _lib
n = 1 << 26
d = DeviceUtils.allocate(); p = Pprims()
lib = _lib.load()
lib.adlhip_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
stamps = Buffer(d, ((n + 4095) // 4096 + 64) * 16, np.uint64)
d.setParam("sort.algo", 0); d.setParam("sort.tile", 6)
buf = Buffer(d, n, np.uint32); buf.generate(n, seed=5); stamps.clear()
assert lib.adlhip_debug_set_stamp_buffer(d._h, stamps.ptr()) == 0
p.radixSort(d, buf, n, 8)
s = stamps.toHost().reshape(-1, 16).astype(np.int64)
rows = np.nonzero(s[:, 11] > 0)[0]
xcc = (s[rows, 12] >> 32) & 0xf
chain = rows // 256          # 64Mi keys, pass 0: sixteen slices of exactly 256 tiles
for c in range(16):
    m = chain == c
    vals, cnt = np.unique(xcc[m], return_counts=True)
    print("chain %2d: %d tiles; XCC -> tiles %s" % (c, m.sum(), dict(zip(vals.tolist(), cnt.tolist()))))
