#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python tools/sweep.py --steps 5 --verify --configs 0:8:6:1,0:8:1:1,0:8:2:1,0:8:5:1,0:8:0:1,0:4:6:1,1:8:6:1,0:8:6:0 2>&1 | tee gpurun_out/sweep_r13.txt
ADLHIP_LIB=$PWD/oclradixsort_amd/lib/libadlhip_stamps.so timeout 600 python tools/stamps.py 2>&1 | tee gpurun_out/stamps_r13.txt
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
