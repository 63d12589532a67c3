#!/bin/bash
# Round 6: how many memory-side atomic requests does the table-gradient scatter issue per sample on a TRAINED scene?
# (profiles/r06_scatter_requests.json holds 38.1 per sample, measured on the untrained bench scene where every sample
# carries weight; behind the first surface of a trained scene most compositing weights are exactly zero and their
# contributions are skipped.)  usage (GPU box, repo root): bash tools/pmc_trained_scatter.sh <tag>
TAG=${1:-pmcts}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
python3 $R/tools/trained_step_profile.py train /tmp/room.pt 2>&1 | tail -1
for ph in nerf instance; do
  N_STEPS=40 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum GRBM_GUI_ACTIVE \
    -d $R/gpurun_out/$TAG/$ph -o atom -- python3 $R/tools/trained_step_profile.py $ph /tmp/room.pt > $R/gpurun_out/$TAG.$ph.log 2>&1
  tail -1 $R/gpurun_out/$TAG.$ph.log
  python3 $R/tools/pmc_summary.py $R/gpurun_out/$TAG/$ph k_grid_bwd
done
rm -rf $R/gpurun_out/$TAG
