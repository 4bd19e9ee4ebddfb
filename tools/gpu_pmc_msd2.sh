#!/bin/bash
# PMC passes (own runs, --kernel-trace only) over the large sort's kernels: memory-pipeline stall counters, LDS counters, traffic.
#   gpurun -- 'bash tools/gpu_pmc_msd2.sh r3_pmc_msd2 [kind] [n] [extra sweep args]'
TAG=${1:-pmc_msd2}; KIND=${2:-u32}; N=${3:-67108864}; shift 3
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
cd /tmp
run() { # name, counters
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1 -- python3 $GRAFT_REPO_ROOT/tools/sweep.py --steps 3 --kind $KIND --n $N --configs=-1:8:-1:1 --param sort.msd2=2 "${EXTRA[@]}" > $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1.log 2>&1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/$TAG/$1.log
}
EXTRA=("$@")
run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run sq2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
run sq3 "SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE"
run tcc1 "FETCH_SIZE"
run tcc2 "WRITE_SIZE"
run tcc3 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
run tcc4 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_EA0_ATOMIC_sum"
run tcc_a "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
run tcc_b "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_IB_STALL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum"
run tcc_c "TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum"
run tcp_a "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
run tcp_b "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
run tcp_c "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"
run ta "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum"
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/$TAG | tee gpurun_out/$TAG/summary.txt | head -100
