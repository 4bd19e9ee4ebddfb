export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mid; mkdir -p $OUT
cd /tmp
for cfg in "1048576 0" "1048576 1" "131072 0" "131072 1" "4194304 0" "4194304 1"; do
  set -- $cfg
  rm -rf /tmp/tr_$1_$2
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$1_$2 -- python3 $GRAFT_REPO_ROOT/tools/midsize_trace.py --n $1 --algo $2 > $OUT/run_$1_$2.log 2>&1
  echo "== n=$1 algo=$2" >> $OUT/summary.txt
  tail -2 $OUT/run_$1_$2.log >> $OUT/summary.txt
  python3 $GRAFT_REPO_ROOT/tools/midsize_trace.py --analyze /tmp/tr_$1_$2 >> $OUT/summary.txt 2>&1
done
cat $OUT/summary.txt
