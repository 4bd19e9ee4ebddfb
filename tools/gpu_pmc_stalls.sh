#!/bin/bash
# PMC passes (own runs, --kernel-trace only) with the memory pipeline's STALL counters for one sweep config.
#   gpurun -- 'bash tools/gpu_pmc_stalls.sh 0:8:-1:1 r2_stalls'
CFG=${1:-0:8:-1:1}; TAG=${2:-pmc_stalls}
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1 -- python3 $GRAFT_REPO_ROOT/tools/sweep.py --steps 3 --configs $CFG > $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1.log
}
run tcc_a "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
run tcc_b "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum"
run tcc_c "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum TCC_REQ_sum"
run tcp_a "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
run tcp_b "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
run tcp_c "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"
run ta "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/$TAG | tee gpurun_out/$TAG/summary.txt
