#!/bin/bash
# A/B of the large sort's finish for whole u32 keys: wave per segment | workgroup per segment (wg_segment_sort_kernel), by tile
run() { timeout -k 10 200 python tools/sweep.py --steps 6 --kind u32 --n $1 --configs=-1:8:-1:1 2>&1 | tail -1 | sed "s/^/n=$1 $2  /"; }
for n in 33554432 67108864; do            # tile 1280 / 1536
  ADLHIP_WG_MIN_TIER=5120 run $n "wave"
  ADLHIP_WG_MIN_TIER=1280 ADLHIP_WG_NT=0 run $n "wg128"
  ADLHIP_WG_MIN_TIER=1280 ADLHIP_WG_NT=256 run $n "wg256"
done
for n in 100663296 134217728; do           # tile 2560
  ADLHIP_WG_MIN_TIER=5120 run $n "wave"
  ADLHIP_WG_MIN_TIER=2560 ADLHIP_WG_NT=128 run $n "wg128"
  ADLHIP_WG_MIN_TIER=2560 ADLHIP_WG_NT=0 run $n "wg256"
done
for n in 402653184 536870912; do           # tile 8192 / 12288
  ADLHIP_WG_NT=0 run $n "wg512"
  ADLHIP_WG_NT=256 run $n "wg256"
done
