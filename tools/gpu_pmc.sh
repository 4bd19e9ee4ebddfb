#!/bin/bash
# PMC passes (own runs, --kernel-trace only) for one sweep config: $1 = algo:bits:tile:rank, $2 = tag
CFG=${1:-0:8:2:1}; TAG=${2:-pmc}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp
run() { # name, counters
  timeout 600 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1 -- python3 $GRAFT_REPO_ROOT/tools/sweep.py --steps 3 --configs $CFG > $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1.log 2>&1
  tail -2 $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1.log
}
run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run tcc1 "FETCH_SIZE GRBM_GUI_ACTIVE"
run tcc2 "WRITE_SIZE"
run tcc3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
run tcc4 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/$TAG | tee gpurun_out/$TAG/summary.txt
