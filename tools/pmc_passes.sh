#!/bin/bash
# PMC passes (each counter group in its own run, kernel-trace only, bounded by `timeout`).
# usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <tag> [python script + args]
TAG=${1:-pmc}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=${@:-tools/field_probe.py 3}
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/$TAG -o $name -- \
    python3 $R/$CMD > $R/gpurun_out/$TAG.$name.log 2>&1
  echo "$name rc=$? $(grep -E '^M=' $R/gpurun_out/$TAG.$name.log | tail -1)"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_128B_sum
run hbm FETCH_SIZE GRBM_GUI_ACTIVE
ls $R/gpurun_out/$TAG | tr '\n' ' '
