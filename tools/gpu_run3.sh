#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu ==" ; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== sweep u32 ==" ; timeout 1200 python tools/sweep.py --steps 5 2>&1 | tee gpurun_out/sweep_u32.txt
