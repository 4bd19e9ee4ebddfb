#!/bin/bash
export TMPDIR=/tmp; mkdir -p gpurun_out
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest"; timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== bench"; timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r1.json
echo "== rocprof stats"; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1; tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log | cut -c1-300
cd $GRAFT_REPO_ROOT; for f in $(find gpurun_out/prof_bench -name "*kernel_stats.csv"); do cp $f gpurun_out/bench_kernel_stats.csv; head -12 $f | cut -c1-200; done
