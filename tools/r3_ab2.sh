#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_ab2}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
{
for n in 4194304 16777216 67108864 134217728 268435456; do
echo "== u64 $n: binning finish (1) vs LSD finish (0)"
$S --steps 4 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --param sort.binfinish=1 --verify | tail -1
$S --steps 4 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --param sort.binfinish=0 | tail -1
done
for n in 67108864 268435456; do
echo "== u32 $n: LSD finish | binning finish"
$S --steps 8 --kind u32 --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --param sort.binfinish=0 --verify | tail -1
$S --steps 8 --kind u32 --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --param sort.binfinish=2 --verify | tail -1
done
} 2>&1 | tee $OUT/ab.txt
echo "== pytest subset"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large or mid_size or safety or scratch or full_size_256m" 2>&1 | tail -5 | tee $OUT/pytest_subset.txt
