#!/bin/bash
# PMC passes over the XCD-sliced frame path (pre-pass k_grid_fine_slices + k_nerf_fwd<..kPre>).
# usage (GPU box, repo root): bash tools/pmc_slices.sh <tag> [config ...]
TAG=${1:-pmcslices}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export PROBE_SLICES=${PROBE_SLICES:-1}
cd /tmp
for CFG in ${@:-4:4:0:0:0 1:1:0:0:0}; do
  D=$R/gpurun_out/$TAG/$(echo $CFG | tr ':' '_')
  run() {
    name=$1; shift
    timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $D -o $name -- \
      python3 $R/tools/bound_field_probe.py 3 $CFG > $D.$name.log 2>&1
    echo "$CFG $name rc=$?"
  }
  mkdir -p $D
  run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
  run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
  echo "== $CFG"; grep -h "field" $D.tcc.log | tail -1
  for k in k_grid_fine_slices k_nerf_fwd; do echo "-- $k"; python3 $R/tools/pmc_summary.py $D $k; done
done
