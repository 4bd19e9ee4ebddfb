#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python tools/sweep.py --steps 5 --verify --configs 0:8:0:1,0:8:1:1,0:8:2:1,0:8:5:1,0:8:6:1,0:8:6:0,2:8:6:1,1:8:6:1,0:4:6:1 2>&1 | tee gpurun_out/sweep_r7.txt
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
