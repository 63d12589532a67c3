#!/bin/bash
# Everything the committed profiles/ artefacts of a round come from, in one GPU call:
#   GPU test suite, the default bench line, the same command under rocprofv3 --kernel-trace --stats, the PMC passes
#   over the render loop (-> traffic json keyed by the kernel sources' sha), the PMC passes over the scatter of the
#   instance-stage training step, and kernel statistics of both training steps.
# usage (GPU box, repo root): bash tools/refresh_profiles.sh <tag>      results under gpurun_out/<tag>/
TAG=${1:-refresh}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
if [ "${SKIP_PYTEST:-0}" != "1" ]; then python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt; fi
# the PMC passes first: the three sha-keyed json files are installed under profiles/ ON THE BOX before the bench runs, so
# that the bench line of this very call carries `traffic`, `atomic_unit` and `fabric_requests` (copy them into the
# repository's profiles/ afterwards - they describe exactly these kernel sources)
bash tools/pmc_bench.sh ${TAG}_pmc > $O/pmc_bench.txt 2>&1
python tools/traffic_json.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc.hbm_rd.log $O/r06_traffic.json > /dev/null; echo "traffic rc=$?"
bash tools/pmc_train.sh ${TAG}_pmct > $O/pmc_train.txt 2>&1
NO_UPDATE=1 bash tools/pmc_train.sh ${TAG}_pmcn k_grid_bwd tools/train_nerf_probe.py > $O/pmc_train_nerf.txt 2>&1
# samples per step of the two probes' timed steps: their last output line ("train step ... ms, N samples/step, ...")
NI=$(grep -o "[0-9]* samples/step" gpurun_out/${TAG}_pmct.atom2.log | tail -1 | cut -d" " -f1)
NN=$(grep -o "[0-9]* samples/step" gpurun_out/${TAG}_pmcn.atom2.log | tail -1 | cut -d" " -f1)
python tools/scatter_requests_json.py gpurun_out/${TAG}_pmct ${NI:-209000} gpurun_out/${TAG}_pmcn ${NN:-209000} $O/r06_scatter_requests.json > /dev/null; echo "scatter json rc=$? ($NI / $NN samples per step)"
bash tools/pmc_bound.sh ${TAG}_pmcb > $O/bound_pmc.txt 2>&1; cp gpurun_out/${TAG}_pmcb.bound_traffic.json $O/r06_bound_traffic.json
# round 6: MFMA pipe utilisation of the field kernel and of the head-backward kernels (the `mfma` pass of the three PMC runs above)
python tools/mfma_json.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmct gpurun_out/${TAG}_pmcn $O/r06_mfma.json > /dev/null; echo "mfma json rc=$?"
cp $O/r06_traffic.json $O/r06_scatter_requests.json $O/r06_bound_traffic.json $O/r06_mfma.json $R/profiles/
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cp gpurun_out/bench_full_n1.json $O/bench_full_n1.json
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 20 --warmup 3 \
  > $O/bench_profiled.json 2> $O/bench_profiled.err; echo "profiled bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_inst -o inst -- python3 $R/tools/train_probe.py 40 \
  > $O/train_probe.txt 2>&1
NO_UPDATE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_nerf -o nerf -- python3 $R/tools/train_nerf_probe.py 40 \
  > $O/train_nerf_probe.txt 2>&1
cd $R
python tools/timed_launches.py $O/trace $O/bench_profiled.json > $O/bench_timed_launches.txt 2>&1
python tools/step_launches.py $O/trace_inst > $O/launches_inst.txt 2>&1
python tools/step_launches.py $O/trace_nerf > $O/launches_nerf.txt 2>&1
ls $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
# round 4: timelines of the instance step (eager, captured two-stream pipeline with and without the shaded head)
export TMPDIR=/tmp
cd /tmp
for m in eager pipe pipe_shade; do
  case $m in eager) E="EMA=1";; pipe) E="PIPE=1 SHADE=0";; pipe_shade) E="PIPE=1 SHADE=1";; esac
  env $E timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$m -o t -- python3 $R/tools/train_probe.py 40 > $O/train_probe_$m.txt 2>&1
done
cd $R
for m in eager pipe pipe_shade; do python tools/step_timeline.py $O/trace_$m k_instance_fwd 2 > $O/timeline_$m.txt 2>&1; tail -1 $O/timeline_$m.txt; done
for m in "EMA=1" "USE_GRAPH=1" "PIPE=1 SHADE=0" "PIPE=1 SHADE=1" "EMA=1" "PIPE=1 SHADE=1"; do echo "== $m: $(env $m python tools/train_probe.py 300 2>&1 | tail -1)"; done > $O/train_probe_modes.txt 2>&1
# round 4: RoIAlign-3D (PMC of the separable forward, launch-parameter sweep, both implementations) and the extraction
bash tools/pmc_roialign.sh ${TAG}_pmcr > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_pmcr sep_fwd > $O/roialign_pmc.txt 2>&1
./tools/micro/roialign_bench > $O/roialign_sweep.txt 2>&1
python tools/roialign_probe.py > $O/roialign_probe.txt 2>&1
python tools/extract_order_probe.py > $O/extract_order_probe.txt 2>&1
python -m pytest tests/test_train_twin.py -m gpu -q -s 2>&1 | grep -E "held-out|passed|failed" > $O/train_twin.txt
# round 5: the field kernel off its tuned configuration (PMC per configuration -> bound traffic json), both frame paths,
# the RoIAlign shapes (incl. the GT-mask crop call) and the extraction's direction sweep
PROBE_MODES=fused,sliced,auto python tools/bound_field_probe.py 5 1:1:0:0:0 2:2:128:0:0 4:4:0:0:0 4:4:128:0:0 8:8:128:0:0 2>&1 | grep -v amdgpu.ids > $O/frame_paths.txt
python tools/roialign_shapes_probe.py 2>&1 | grep -v amdgpu > $O/roialign_shapes.txt
python tools/extract_dirs_probe.py 2>&1 | grep -v amdgpu > $O/extract_dirs_probe.txt
bash tools/pmc_extract.sh ${TAG}_pmcx > $O/extract_pmc.txt 2>&1
./tools/micro/roialign_bench bwd > $O/roialign_bwd_sweep.txt 2>&1
./tools/micro/level_xcd_bench 19 > $O/level_xcd_bench.txt 2>&1
# round 6: atomic operand types against the memory-side unit (rates + PMC: which atomics does the L2 forward?)
./tools/micro/atomic_type_bench > $O/atomic_type_bench.txt 2>&1
bash tools/pmc_atomic_types.sh ${TAG}_pmcat > $O/atomic_type_pmc.txt 2>&1
# what travels back is capped at 64 MiB: keep the summaries, drop the raw traces and counter dumps
for t in trace trace_inst trace_nerf; do
  for f in $(find $O/$t -name "*kernel_stats.csv" 2>/dev/null); do cp $f $O/${t}_kernel_stats.csv; done
done
python tools/pmc_summary.py gpurun_out/${TAG}_pmc k_nerf_fwd > $O/bench_pmc_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_pmct k_grid_bwd > $O/grid_bwd_pmc_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_pmcn k_grid_bwd > $O/grid_bwd_pmc_nerf_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_pmct head_bwd > $O/head_bwd_pmc_summary.txt 2>&1
python tools/pmc_summary.py gpurun_out/${TAG}_pmcn head_bwd > $O/head_bwd_pmc_nerf_summary.txt 2>&1
rm -rf $O/trace $O/trace_inst $O/trace_nerf $O/trace_eager $O/trace_pipe $O/trace_pipe_shade
for d in gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmct gpurun_out/${TAG}_pmcn gpurun_out/${TAG}_pmcb gpurun_out/${TAG}_pmcr gpurun_out/${TAG}_pmcx gpurun_out/${TAG}_pmcat; do
  rm -rf $d
done
find gpurun_out -size +4M -delete
du -sm gpurun_out
