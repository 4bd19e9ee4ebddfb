#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
