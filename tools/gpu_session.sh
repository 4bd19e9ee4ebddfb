#!/bin/bash
# One GPU-box session: parity tests, sweeps, bench, rocprof stats + PMC traffic of the bench command.
#   gpurun --timeout 1200 -- 'bash tools/gpu_session.sh [tag]'
TAG=${1:-session}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
echo "== smoke";  timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest -m gpu"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
echo "== sweep u32"; timeout -k 10 600 python tools/sweep.py --steps 10 --verify --configs 0:8:-1:1,0:7:-1:1,0:4:-1:1,1:8:-1:1,0:8:-1:0 2>&1 | tee $OUT/sweep_u32.txt
echo "== sweep kv";  timeout -k 10 600 python tools/sweep.py --steps 10 --verify --kind kv --configs 0:8:-1:1,0:8:1:1,0:8:6:1,1:8:-1:1 2>&1 | tee $OUT/sweep_kv32.txt
echo "== sweep soa"; timeout -k 10 600 python tools/sweep.py --steps 10 --verify --kind soa --configs 0:8:-1:1,1:8:-1:1 2>&1 | tee $OUT/sweep_soa32.txt
echo "== sweep u64 256Mi"; timeout -k 10 600 python tools/sweep.py --steps 3 --kind u64 --n 268435456 --configs 0:8:-1:1,0:8:6:1,1:8:-1:1 2>&1 | tee $OUT/sweep_u64.txt
echo "== distributions"; timeout -k 10 600 python tools/distributions.py 2>&1 | tee $OUT/distributions.txt
echo "== n curve"; timeout -k 10 900 python tools/ncurve.py 2>&1 | tee $OUT/ncurve.txt
echo "== scan"; timeout -k 10 300 python tools/scan_bench.py 2>&1 | tee $OUT/scan.txt
echo "== bench"; timeout -k 10 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_n1.json
echo "== bench, multi-GPU code path with one rank (NOT the N=1 benchmark)"; ADLHIP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_forcedist.json
echo "== rocprofv3 --kernel-trace --stats (same command)"
cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/prof_stats.log 2>&1
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; head -6 $f | cut -c1-180; done
echo "== PMC passes (separate runs, --kernel-trace only)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify > $OUT/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT | tee $OUT/pmc_summary.txt | grep -A3 -E "onesweep_chain|msd_bucket"
python3 tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json
