#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python tools/sweep.py --steps 5 --verify --configs 2:8:6:1,0:8:6:1,3:8:6:1,2:8:1:1,2:8:2:1,2:8:5:1,2:8:0:1,2:4:6:1 2>&1 | tee gpurun_out/sweep_r10.txt
ADLHIP_LIB=$PWD/oclradixsort_amd/lib/libadlhip_stamps.so timeout 600 python tools/stamps.py --configs 2:8:6:1,2:8:1:1 2>&1 | tee gpurun_out/stamps_r10.txt
