#!/bin/bash
# First GPU session: probe, smoke, parity tests, bench variants, rocprof stats.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" ; rocminfo | grep -E "Marketing|gfx9|Compute Unit" | head -6
echo "== lds order probe ==" ; timeout 120 ./tools/lds_order_probe 2>&1 | tail -6
echo "== smoke ==" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu ==" ; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
echo "== bench onesweep8 ==" ; timeout 900 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 | tee gpurun_out/bench_onesweep8.json
echo "== bench three-kernel8 ==" ; timeout 600 python bench.py --steps 10 --warmup 2 --algo 1 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_three8.json
echo "== bench onesweep4 ==" ; timeout 600 python bench.py --steps 10 --warmup 2 --digit-bits 4 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_onesweep4.json
echo "== bench three-kernel4 ==" ; timeout 600 python bench.py --steps 10 --warmup 2 --algo 1 --digit-bits 4 --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_three4.json
echo "== rocprof ==" 
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify 2>&1 | tail -3
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_r1 -name "*stats*" | head; for f in $(find gpurun_out/prof_r1 -name "*kernel_stats.csv"); do head -20 $f; done
