#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke ==" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu ==" ; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -40
echo "== bench onesweep8 ==" ; timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -1 | tee gpurun_out/bench_onesweep8.json
echo "== bench onesweep4 ==" ; timeout 600 python bench.py --steps 10 --warmup 2 --digit-bits 4 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_onesweep4.json
