#!/bin/bash
# Round-3 final measurements, part a (no tests): sweeps, distributions, size curves.  gpurun --timeout 1200 -- 'bash tools/r3_final_session_a.sh'
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r3_final; mkdir -p $OUT
echo "== sweep u32"; timeout -k 10 300 python tools/sweep.py --steps 10 --verify --configs=-1:8:-1:1,0:8:-1:1,0:7:-1:1,0:4:-1:1,1:8:-1:1,0:8:-1:0 2>&1 | tee $OUT/sweep_u32.txt
echo "== sweep kv";  timeout -k 10 300 python tools/sweep.py --steps 10 --verify --kind kv --configs=-1:8:-1:1,0:8:-1:1,1:8:-1:1 2>&1 | tee $OUT/sweep_kv32.txt
echo "== sweep soa"; timeout -k 10 300 python tools/sweep.py --steps 10 --verify --kind soa --configs=-1:8:-1:1,0:8:-1:1,1:8:-1:1 2>&1 | tee $OUT/sweep_soa32.txt
echo "== sweep u64 256Mi"; timeout -k 10 300 python tools/sweep.py --steps 3 --kind u64 --n 268435456 --configs=-1:8:-1:1,0:8:-1:1,1:8:-1:1 2>&1 | tee $OUT/sweep_u64.txt
echo "== distributions"; timeout -k 10 600 python tools/distributions.py 2>&1 | tee $OUT/distributions.txt
echo "== large-sort size curves"; timeout -k 10 400 python tools/msd2curve.py 1500000 2500000 3145728 4194304 6291456 8388608 12582912 16777216 25165824 33554432 50331648 67108864 100663296 134217728 150994944 201326592 268435456 402653184 536870912 805306368 1073741824 2>&1 | tee $OUT/msd2_size_curve.txt
for k in kv u64; do for n in 4194304 8388608 16777216 33554432 67108864 134217728; do timeout -k 10 200 python tools/sweep.py --steps 5 --kind $k --n $n --configs=-1:8:-1:1,0:8:-1:1 2>&1 | tail -2 | sed "s/^/$k n=$n  /"; done; done | tee $OUT/large_sort_kv_u64_curve.txt
echo "== multi-rank rehearsal on one GPU (code path only, host-staged collectives)"
ADLHIP_BENCH_REHEARSE=1 ADLHIP_BENCH_N=8388608 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 4 --steps 3 --warmup 1 2>&1 | tail -1 | tee $OUT/bench_multirank_rehearsal.txt
