#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-r3_dict}; mkdir -p $OUT
echo "== distributions (first / eighth sort of a fresh handle)"; timeout -k 10 600 python tools/distributions.py --profile 2>&1 | cut -c1-400 | tee $OUT/distributions.txt
echo "== pytest"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt
