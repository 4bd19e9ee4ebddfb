#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_forms}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
KINDS=${2:-u32 u64}; SIZES=${3:-4194304 8388608 16777216 33554432 67108864 134217728 268435456}
{
for kind in $KINDS; do
for n in $SIZES; do
echo "== $kind $n: cursor (4) | hybrid (5) | stable (3)"
for m in 4 5 3; do
$S --steps 6 --kind $kind --n $n --configs=-1:8:-1:1 --param sort.msd2=$m --verify | tail -1
done; done; done
} 2>&1 | tee $OUT/forms.txt
