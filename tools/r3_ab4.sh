#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_ab4}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
echo "== new tests"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config3 or size_classes or partial_sort_bits or binning or small_partition or large or mid_size or safety or scratch" 2>&1 | tail -15 | tee $OUT/pytest_subset.txt
{
for n in 4194304 16777216 67108864 268435456; do
echo "== u64 $n: stable passes (2) vs cursor passes (4)"
$S --steps 4 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --verify | tail -1
$S --steps 4 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=4 | tail -1
done
echo "== u32 64Mi 28-bit sort"
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims(); n = 1 << 26
bufs = [Buffer(d, n, np.uint32) for _ in range(6)]
for bits in (32, 28, 24, 20):
  for mode in (2, 0):
    d.setParam("sort.msd2", mode)
    best = 1e9
    for t in range(3):
        for i, b in enumerate(bufs): b.generate(n, seed=t * 10 + i, kind=0)
        DeviceUtils.waitForCompletion(d)
        sw = Stopwatch(d); sw.start()
        for b in bufs: p.radixSort(d, b, n, bits)
        sw.stop(); best = min(best, sw.getMs() / len(bufs))
    print("bits %d msd2=%d: %.3f ms  %.1f Gkeys/s" % (bits, mode, best, n / best / 1e6), flush=True)
PY
} 2>&1 | tee $OUT/ab.txt
