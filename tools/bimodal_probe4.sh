#!/bin/bash
# fourth part: the slow mode is per PROCESS (two bench runs of one call can differ) - many short headline runs, each
# with the XCD map of its own process beside the result.  usage: bash tools/bimodal_probe4.sh <tag> [runs]
TAG=${1:-bimodal4}; N=${2:-14}
O=gpurun_out/$TAG; mkdir -p $O
for i in $(seq 1 $N); do
  python bench.py --no-cpu-baseline --no-train-probe --no-trained-scene --steps 12 > $O/run_$i.json 2>/dev/null
  python - "$O/run_$i.json" "$i" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("clocks") or {}
x = c.get("xcd_map") or {}
print(f"run {sys.argv[2]:>2s}  frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.3f} ms  sclk {c.get('sclk_mhz_median')}  "
      f"round robin {x.get('xcd_of_block_is_block_mod_8_up_to_rotation')} rot {x.get('rotation')} per xcd {x.get('workgroups_per_xcd')} first {x.get('first_16_blocks')}")
PY
  if [ $i -eq 5 ]; then python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -k "two_ranks_stay" > $O/t.txt 2>&1; tail -1 $O/t.txt; fi
done
