#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_misc}; mkdir -p $OUT
S="timeout -k 10 280 python tools/sweep.py"
{
echo "== pass 1 into a first slab with 50 % head-room (407 MB) | 3 % (dense, 281 MB): 64 Mi u32 keys, cursor form"
for h in 50 3 50 3; do
ADLHIP_SLAB_A_HEADROOM_PCT=$h $S --steps 10 --kind u32 --n 67108864 --configs=-1:8:-1:1 --param sort.msd2=4 --verify | tail -1
done
} 2>&1 | tee $OUT/dense_slab.txt
echo "== distributions (first / eighth sort of a fresh handle)"; timeout -k 10 600 python tools/distributions.py 2>&1 | tee $OUT/distributions.txt
