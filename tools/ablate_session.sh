#!/bin/bash
# diagnostic: time the one-sweep pass with pieces removed (libadlhip_ab*.so, results wrong by construction)
OUT=gpurun_out/ablate; mkdir -p $OUT
for lib in "" _ab1 _ab2 _ab3 $EXTRA_LIBS; do
  echo "== lib$lib" | tee -a $OUT/ablate.txt
  ADLHIP_LIB=$PWD/oclradixsort_amd/lib/libadlhip$lib.so timeout -k 10 300 python tools/sweep.py --steps 10 --configs 0:8:-1:1 2>&1 | tail -1 | tee -a $OUT/ablate.txt
done
