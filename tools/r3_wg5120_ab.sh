#!/bin/bash
# A/B of the finish for the 5120-key tier of whole u32 keys: wave kernel (80 rows) | workgroup of 512 x 10 | workgroup of 256 x 20
for n in 150994944 184549376 218103808 251658240 268435456 293601280; do
  for v in 0 1 2; do
    ADLHIP_WG5120=$v timeout -k 10 200 python tools/sweep.py --steps 4 --kind u32 --n $n --configs=-1:8:-1:1 2>&1 | tail -1 | sed "s/^/n=$n wg5120=$v  /"
  done
done
