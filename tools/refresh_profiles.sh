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
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
export TMPDIR=/tmp
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 20 --warmup 3 \
  > $O/bench_profiled.json 2> $O/bench_profiled.err; echo "profiled bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_inst -o inst -- python3 $R/tools/train_probe.py 40 \
  > $O/train_probe.txt 2>&1
NO_UPDATE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_nerf -o nerf -- python3 $R/tools/train_nerf_probe.py 40 \
  > $O/train_nerf_probe.txt 2>&1
cd $R
bash tools/pmc_bench.sh ${TAG}_pmc > $O/pmc_bench.txt 2>&1
python tools/traffic_json.py gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc.hbm_rd.log $O/r03_traffic.json; echo "traffic rc=$?"
bash tools/pmc_train.sh ${TAG}_pmct > $O/pmc_train.txt 2>&1; tail -25 $O/pmc_train.txt
python tools/timed_launches.py $O/trace $O/bench_profiled.json > $O/bench_timed_launches.txt 2>&1
python tools/step_launches.py $O/trace_inst > $O/launches_inst.txt 2>&1
python tools/step_launches.py $O/trace_nerf > $O/launches_nerf.txt 2>&1
ls $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
NO_UPDATE=1 bash tools/pmc_train.sh ${TAG}_pmcn k_grid_bwd tools/train_nerf_probe.py > $O/pmc_train_nerf.txt 2>&1
python tools/scatter_requests_json.py gpurun_out/${TAG}_pmct 207610 gpurun_out/${TAG}_pmcn 207610 $O/r03_scatter_requests.json > /dev/null; echo "scatter json rc=$?"
