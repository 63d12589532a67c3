#!/bin/bash
# PMC passes over the lattice extraction kernel (k_nerf_fwd_dirs, 160^3, four view directions; tools/extract_dirs_probe.py):
# how many 128-byte fabric requests a voxel costs - the lattice's fine levels share no lines between neighbouring voxels,
# like a frame's at bound 4 (profiles/r05_NOTES.txt 1).
# usage (GPU box, repo root): bash tools/pmc_extract.sh <tag>
TAG=${1:-pmc_extract}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
export PROBE_DIRS=4
cd /tmp
run() {
  name=$1; shift
  timeout 240 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/$TAG -o $name -- \
    python3 $R/tools/extract_dirs_probe.py > $R/gpurun_out/$TAG.$name.log 2>&1
  echo "$name rc=$?"
}
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr
run hbm FETCH_SIZE GRBM_GUI_ACTIVE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
cd $R
grep -h "D=4" gpurun_out/$TAG.tcc.log | tail -1
python3 tools/pmc_summary.py gpurun_out/$TAG k_nerf_fwd_dirs
