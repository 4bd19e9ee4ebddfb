#!/bin/bash
# narrow second digit (small inputs): ADLHIP_SEGSHIFT_MAXN=0 = off (65536 segments at every size)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_narrow}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
{
for kind in u32 u64; do
for n in 1500000 2500000 4194304 6291456 8388608 12582912 16777216 25165824; do
echo "== $kind $n: 65536 segments | narrow second digit"
ADLHIP_SEGSHIFT_MAXN=0 $S --steps 8 --kind $kind --n $n --configs=-1:8:-1:1 --param sort.msd2=4 --verify | tail -1
ADLHIP_SEGSHIFT_MAXN=40000000 $S --steps 8 --kind $kind --n $n --configs=-1:8:-1:1 --param sort.msd2=4 --verify | tail -1
done; done
} 2>&1 | tee $OUT/narrow.txt
