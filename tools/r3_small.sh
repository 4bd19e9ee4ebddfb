#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_small}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
{
for kind in u64 kv u32; do
for n in 100000 262144 524288 1048576 1500000 2097152; do
echo "== $kind $n: automatic | large sort forced (2)"
$S --steps 8 --kind $kind --n $n --configs=-1:8:-1:1 --verify | tail -1
$S --steps 8 --kind $kind --n $n --configs=-1:8:-1:1 --param sort.msd2=2 --param sort.mid=0 --verify | tail -1
done; done
} 2>&1 | tee $OUT/small.txt
