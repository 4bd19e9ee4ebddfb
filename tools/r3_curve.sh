#!/bin/bash
# automatic choice, u32 keys: size curve (fresh random keys per sort) + the full test suite
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_curve}; mkdir -p $OUT
echo "== pytest"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
echo "== curve (msd2 = 2 forced | msd2 = 0)"; timeout -k 10 600 python tools/msd2curve.py 1500000 2500000 3145728 4194304 6291456 8388608 12582912 16777216 25165824 33554432 50331648 67108864 134217728 268435456 2>&1 | tee $OUT/msd2_size_curve.txt
