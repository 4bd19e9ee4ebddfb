#!/bin/bash
# diagnostic: A/B two builds of the library in ONE GPU session (box-to-box variation is larger than most effects)
#   bash tools/ab_session.sh _old,cur 0:8:-1:1 [n] [kind] [steps]
LIBS=${1:-"_old,cur"}; CFG=${2:-0:8:-1:1}; N=${3:-67108864}; KIND=${4:-u32}; STEPS=${5:-10}
for rep in 1 2 3; do
  for lib in ${LIBS//,/ }; do
    [ "$lib" = "cur" ] && lib=""
    printf "%-8s " "lib$lib"
    ADLHIP_LIB=$PWD/oclradixsort_amd/lib/libadlhip$lib.so python tools/sweep.py --steps $STEPS --n $N --kind $KIND --configs=$CFG 2>&1 | tail -1
  done
done
