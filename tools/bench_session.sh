#!/bin/bash
# bench + rocprof stats of the same command (+ the bench's own PMC child runs)
export TMPDIR=/tmp
TAG=${1:-r4_bench}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
echo "== bench"; timeout -k 10 900 python bench.py 2>$OUT/bench.err | tail -1 | tee $OUT/bench_n1.json | cut -c1-600
tail -5 $OUT/bench.err
echo "== rocprofv3 --kernel-trace --stats (same command, without the PMC child runs and the CPU leg)"
cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc > $OUT/prof_stats.log 2>&1
cd $GRAFT_REPO_ROOT
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; head -12 $f | cut -c1-200; done
rm -rf $OUT/prof_stats/*/*kernel_trace.csv
