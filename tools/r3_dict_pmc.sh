#!/bin/bash
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_dict_pmc}; mkdir -p $OUT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq1 -- python3 $GRAFT_REPO_ROOT/tools/dict_profile.py > $OUT/sq1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/tools/dict_profile.py > $OUT/sq2.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "dict_count" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for c in sorted(acc):
    v = sorted(acc[c])
    print("%-24s" % c, " ".join("%12.0f" % x[1] for x in v))
PY
