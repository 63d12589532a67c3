#!/bin/bash
# PMC passes over the instance-stage training step (tools/train_probe.py), summarised for the table-gradient scatter.
# Each counter group runs in its own process, kernel-trace only, bounded by `timeout`.
# usage (GPU box, repo root): bash tools/pmc_train.sh <tag> [kernel-substring] [probe script, default tools/train_probe.py]
TAG=${1:-pmct}
KERNEL=${2:-k_grid_bwd}
PROBE=${3:-tools/train_probe.py}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/$TAG -o $name -- \
    python3 $R/$PROBE 12 > $R/gpurun_out/$TAG.$name.log 2>&1
  echo "$name rc=$? $(tail -1 $R/gpurun_out/$TAG.$name.log)"
}
run grbm GRBM_GUI_ACTIVE TCC_CYCLE_sum TCC_BUSY_sum
# round 6: MFMA pipe utilisation - busy cycles (per SIMD) and the kernel's own cycle count from ONE pass (tools/mfma_json.py)
run mfma GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run atom1 TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_ATOMIC_sum TCC_REQ_sum
run atom2 TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum
run atom3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
run atom4 TCC_ATOMIC_SECTORS_sum TCC_EA0_WRREQ_WRITE_ATOMIC_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_TAG_STALL_sum
run ta TA_FLAT_ATOMIC_WAVEFRONTS_sum TA_BUFFER_ATOMIC_WAVEFRONTS_sum TCC_EA0_RDREQ_sum TCC_WRITEBACK_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VALU SQ_WAVES
run hbm FETCH_SIZE WRITE_SIZE
python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG $KERNEL
