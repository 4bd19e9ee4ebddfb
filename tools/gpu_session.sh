#!/bin/bash
# One GPU-box session: parity tests, sweeps, bench, rocprof stats + PMC traffic of the bench command.
#   gpurun --timeout 1200 -- 'bash tools/gpu_session.sh [tag]'
TAG=${1:-session}; PART=${2:-all}   # part: a = tests + sweeps + curves, b = bench + profiles, all = both (two gpurun calls fit the limit better)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
if [ "$PART" != b ]; then
echo "== smoke";  timeout -k 10 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== pytest -m gpu"; timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.txt
echo "== sweep u32"; timeout -k 10 600 python tools/sweep.py --steps 10 --verify --configs=-1:8:-1:1,0:8:-1:1,0:7:-1:1,0:4:-1:1,1:8:-1:1,0:8:-1:0 2>&1 | tee $OUT/sweep_u32.txt
echo "== sweep kv";  timeout -k 10 600 python tools/sweep.py --steps 10 --verify --kind kv --configs=-1:8:-1:1,0:8:-1:1,0:8:6:1,1:8:-1:1 2>&1 | tee $OUT/sweep_kv32.txt
echo "== sweep soa"; timeout -k 10 600 python tools/sweep.py --steps 10 --verify --kind soa --configs=-1:8:-1:1,0:8:-1:1,1:8:-1:1 2>&1 | tee $OUT/sweep_soa32.txt
echo "== sweep u64 256Mi"; timeout -k 10 600 python tools/sweep.py --steps 3 --kind u64 --n 268435456 --configs=-1:8:-1:1,0:8:-1:1,1:8:-1:1 2>&1 | tee $OUT/sweep_u64.txt
echo "== distributions"; timeout -k 10 600 python tools/distributions.py 2>&1 | tee $OUT/distributions.txt
echo "== n curve"; timeout -k 10 900 python tools/ncurve.py 2>&1 | tee $OUT/ncurve.txt
echo "== large-sort size curves"; timeout -k 10 300 python tools/msd2curve.py 1500000 2500000 3145728 4194304 6291456 8388608 12582912 16777216 25165824 33554432 50331648 67108864 134217728 268435456 2>&1 | tee $OUT/msd2_size_curve.txt
for k in kv u64; do for n in 4194304 8388608 16777216 33554432 67108864 134217728; do timeout -k 10 200 python tools/sweep.py --steps 5 --kind $k --n $n --configs=-1:8:-1:1,0:8:-1:1 2>&1 | tail -2 | sed "s/^/$k n=$n  /"; done; done | tee $OUT/large_sort_kv_u64_curve.txt
echo "== large sort, keys of one eighth of the range (a rank of an 8-GPU sort)"; MSD2CURVE_SHIFT=3 MSD2CURVE_MODES=1,0 timeout -k 10 300 python tools/msd2curve.py 16777216 67108864 134217728 2>&1 | tee $OUT/msd2_rank_range.txt
echo "== multi-rank rehearsal on one GPU (code path only, host-staged collectives)"
ADLHIP_BENCH_REHEARSE=1 ADLHIP_BENCH_N=8388608 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_multirank_rehearsal.txt
echo "== scan"; timeout -k 10 300 python tools/scan_bench.py 2>&1 | tee $OUT/scan.txt
fi
if [ "$PART" = a ]; then exit 0; fi
echo "== bench"; timeout -k 10 900 python bench.py 2>&1 | tail -1 | tee $OUT/bench_n1.json
echo "== bench, multi-GPU code path with one rank (NOT the N=1 benchmark)"; ADLHIP_BENCH_FORCE_DIST=1 timeout -k 10 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee $OUT/bench_forcedist.json
echo "== rocprofv3 --kernel-trace --stats (same command)"
cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-pmc > $OUT/prof_stats.log 2>&1
for f in $(find $OUT/prof_stats -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; head -6 $f | cut -c1-180; done
echo "== PMC passes (separate runs, --kernel-trace only)"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-verify --no-other-configs --no-pmc > $OUT/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $OUT | tee $OUT/pmc_summary.txt | grep -A3 -E "onesweep_chain|msd_bucket"
python3 tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json || true
