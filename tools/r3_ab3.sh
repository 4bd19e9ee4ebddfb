#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_ab3}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
{
for n in 4194304 16777216 67108864 134217728 268435456; do
for g in 65536 0; do
echo "== u64 $n: binning finish grid=$g"
ADLHIP_BIN_GRID=$g $S --steps 4 --kind u64 --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --param sort.binfinish=1 --verify | tail -1
done; done
} 2>&1 | tee $OUT/ab.txt
