#!/bin/bash
# PMC passes over the RoIAlign-3D backward kernels on configs[4] (tools/roialign_probe.py --fast).
TAG=${1:-pmc_roibwd}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/$TAG -o $name -- \
    python3 $R/tools/roialign_probe.py --fast > $R/gpurun_out/$TAG.$name.log 2>&1
  echo "$name rc=$?"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
run sq3 SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_LDS_UNALIGNED_STALL
run tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
cd $R
for k in gather_bwd sep_bwd sep_fwd; do echo "-- $k"; python3 tools/pmc_summary.py gpurun_out/$TAG $k; done
