#!/bin/bash
# diagnostic: non-temporal key loads (libadlhip.so) vs plain loads (libadlhip_nt0.so, -DADLHIP_NT_LOADS=0), same session
for rep in 1 2; do
  for kind in "kv 67108864" "u64 268435456" "u64 67108864" "u32 134217728" "soa 67108864"; do
    set -- $kind
    for lib in "" "_nt0"; do
      printf "%-4s %-10s lib%-5s " $1 $2 "$lib"
      ADLHIP_LIB=$PWD/oclradixsort_amd/lib/libadlhip$lib.so python tools/sweep.py --steps 8 --n $2 --kind $1 --configs=-1:8:-1:1 2>&1 | tail -1
    done
  done
done
