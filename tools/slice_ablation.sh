#!/bin/bash
# Pre-pass ablations (profiling builds of tools/build_probe.py): kernel times of k_grid_fine_slices per variant.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PROBE_SLICES=1
cd /tmp
for v in ${VARIANTS:-default abl1 abl4 abl5 t2 t8}; do
  [ $v = default ] && unset INR_LIB_PATH || export INR_LIB_PATH=$R/tools/_probe/libinr_$v.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05h_abl -o $v -- python3 $R/tools/bound_field_probe.py 3 ${1:-4:4:0:0:0} > /dev/null 2>&1
  echo "$v $(grep -h 'k_grid_fine_slices\|k_nerf_fwd' $R/gpurun_out/r05h_abl/${v}_kernel_stats.csv | awk -F'","' '{printf "%s %.3f ms | ", substr($1,2,28), $4/1e6}')"
done
