# Steady-state training steps of a TRAINED scene under rocprofv3 (gpurun -- bash tools/trained_step_profile.sh):
# trains the synthetic room, then profiles 200 steps of each stage from the saved state; stats -> gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/tools/trained_step_profile.py train /tmp/room.pt 2>&1 | tail -1
for ph in nerf instance; do
  python3 $R/tools/trained_step_profile.py $ph /tmp/room.pt 2>&1 | tail -1 | tee $R/gpurun_out/trained_step_${ph}.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$ph -o tsp -- python3 $R/tools/trained_step_profile.py $ph /tmp/room.pt 2>&1 | tail -1
  f=$(find /tmp/prof_$ph -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/trained_step_${ph}_kernel_stats.csv
done
