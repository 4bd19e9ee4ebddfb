#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
ADLHIP_LIB=$PWD/oclradixsort_amd/lib/libadlhip_stamps.so timeout 600 python tools/stamps.py --configs 2:8:6:1,2:8:1:1 2>&1 | tee gpurun_out/stamps_r12.txt
