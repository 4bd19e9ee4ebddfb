#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_narrow2}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
echo "== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "large or hybrid or partial_sort_bits or binning or size_classes or config3 or scratch or safety" 2>&1 | tail -6
{
for n in 1500000 2500000 4194304 6291456 8388608 12582912; do
echo "== kv $n: three-kernel / one-sweep (0) | stable large sort, 65536 segments | narrow second digit"
$S --steps 8 --kind kv --n $n --configs=-1:8:-1:1 --param sort.msd2=0 --verify | tail -1
ADLHIP_SEGSHIFT_MAXN=0 $S --steps 8 --kind kv --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --verify | tail -1
$S --steps 8 --kind kv --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --verify | tail -1
done
for n in 33554432 41943040; do
echo "== u64 $n: cursor, 65536 segments | cursor, narrow | stable"
ADLHIP_SEGSHIFT_MAXN=0 $S --steps 6 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=4 --verify | tail -1
$S --steps 6 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=4 --verify | tail -1
$S --steps 6 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=3 --verify | tail -1
done
} 2>&1 | tee $OUT/narrow.txt
