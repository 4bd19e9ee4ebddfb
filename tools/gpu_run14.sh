#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python tools/sweep.py --steps 5 --verify --configs 0:8:6:1,0:8:2:1,0:4:6:1 2>&1 | tee gpurun_out/sweep_r14.txt
timeout 1200 python tools/sweep.py --steps 5 --verify --kind kv --configs 0:8:1:1,0:8:6:1,0:8:0:1,0:8:2:1,0:8:5:1,1:8:1:1 2>&1 | tee gpurun_out/sweep_r14_kv.txt
timeout 1200 python tools/sweep.py --steps 3 --verify --kind u64 --n 268435456 --configs 0:8:1:1,0:8:6:1,0:8:0:1,1:8:1:1 2>&1 | tee gpurun_out/sweep_r14_u64.txt
