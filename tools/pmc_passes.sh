#!/bin/bash
# PMC passes for the bench workload (each counter group in its own run, kernel-trace only).
# usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <tag> [bench args...]
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/$TAG -o $name -- \
    python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline $BENCH_ARGS > $R/gpurun_out/$TAG.$name.log 2>&1
  tail -1 $R/gpurun_out/$TAG.$name.log | cut -c1-200
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_WAVES
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run hbm FETCH_SIZE GRBM_GUI_ACTIVE
ls $R/gpurun_out/$TAG
