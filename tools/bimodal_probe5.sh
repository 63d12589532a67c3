#!/bin/bash
# fifth part: the sequence that flipped a box in r3t (all DDP tests, then the parity tests), with the XCD map beside
# every result.  usage: bash tools/bimodal_probe5.sh <tag>
TAG=${1:-bimodal5}
O=gpurun_out/$TAG; mkdir -p $O
one() {
  python bench.py --no-cpu-baseline --no-train-probe --no-trained-scene --steps 12 > $O/$1.json 2>/dev/null
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("clocks") or {}
x = c.get("xcd_map") or {}
print(f"{sys.argv[2]:12s} frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.3f} ms  round robin {x.get('xcd_of_block_is_block_mod_8_up_to_rotation')} rot {x.get('rotation')} per xcd {x.get('workgroups_per_xcd')} first {x.get('first_16_blocks')}")
PY
}
one fresh
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu > $O/t.txt 2>&1; tail -1 $O/t.txt
one after_ddp_1
one after_ddp_2
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/t.txt 2>&1; tail -1 $O/t.txt
one after_par_1
one after_par_2
