#!/bin/bash
# PMC passes over the fused field kernel at the headline configuration and at bound 4 (round-4 verdict item 1c).
# usage (GPU box, repo root): bash tools/pmc_bound.sh <tag>
TAG=${1:-pmcbound}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
# <config>[-sliced]: the frame path (NeRFNetwork.frame_slices) the probe runs; fused unless named
for RUN in 1:1:0:0:0 2:2:128:0:0 4:4:0:0:0 4:4:128:0:0 4:4:0:0:0-sliced 4:4:128:0:0-sliced; do
  CFG=${RUN%-sliced}
  export PROBE_MODES=fused
  [ "$RUN" != "$CFG" ] && export PROBE_MODES=sliced
  D=$R/gpurun_out/$TAG/$(echo $RUN | tr ':' '_')
  run() {
    name=$1; shift
    timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $D -o $name -- \
      python3 $R/tools/bound_field_probe.py 3 $CFG > $D.$name.log 2>&1
    echo "$CFG $name rc=$?"
  }
  mkdir -p $D
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
  run hbm FETCH_SIZE GRBM_GUI_ACTIVE
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
  echo "== $RUN"; grep -h "field" $D.tcc.log | tail -1
  python3 $R/tools/pmc_summary.py $D k_nerf_fwd
  [ "$RUN" != "$CFG" ] && python3 $R/tools/pmc_summary.py $D k_grid_fine_slices
done
python3 $R/tools/bound_traffic_json.py $R/gpurun_out/$TAG $R/gpurun_out/$TAG.bound_traffic.json > /dev/null
