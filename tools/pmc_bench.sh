#!/bin/bash
# PMC passes over bench.py's render loop (each counter group in its own run, kernel-trace only, bounded).
# usage (GPU box, repo root): bash tools/pmc_bench.sh <tag>
TAG=${1:-pmcb}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/$TAG -o $name -- \
    python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-train-probe > $R/gpurun_out/$TAG.$name.log 2>&1
  echo "$name rc=$?"
}
run grbm GRBM_GUI_ACTIVE
# round 6: MFMA pipe utilisation - busy cycles (per SIMD) and the kernel's own cycle count from ONE pass (tools/mfma_json.py)
run mfma GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
run hbm_rd FETCH_SIZE
run hbm_wr WRITE_SIZE
run rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_HIT_sum TCC_MISS_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG k_nerf_fwd
