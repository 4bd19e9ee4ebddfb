#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_facade.py -m gpu -x -q 2>&1 | tail -8
echo "== reference unit test (unmodified) on the facade"; (cd oracle/_ref && timeout 600 ./ref_unittest_on_facade 2>&1 | tail -12) | tee gpurun_out/ref_unittest_on_facade.txt
echo "== demo"; timeout 600 ./tests/demo/demo 2>&1 | grep -E "RUN|OK|FAIL|PASS|device" | tee gpurun_out/demo_device.txt
