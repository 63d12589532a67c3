#!/bin/bash
# third part: headline-only bench in a loop with the clocks / power / temperatures bench.py now reports, the suspect
# test in between.  usage: bash tools/bimodal_probe3.sh <tag>
TAG=${1:-bimodal3}
O=gpurun_out/$TAG; mkdir -p $O
one() {
  python bench.py --no-cpu-baseline --no-train-probe --no-trained-scene $2 > $O/$1.json 2>/dev/null
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("clocks") or {}
print(f"{sys.argv[2]:14s} frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.3f} ms  " + "  ".join(f"{k}={v}" for k, v in c.items() if k != "source"))
PY
}
one run_1
one run_2
one run_3_60steps "--steps 60"
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -k "two_ranks_stay" > $O/t.txt 2>&1; tail -1 $O/t.txt
one after_ddp_1
one after_ddp_2
sleep 30
one after_30s
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu -k "eight_ranks" > $O/t.txt 2>&1; tail -1 $O/t.txt
one after_8rank_1
one after_8rank_2
rocm-smi --showperflevel --showpower --showmemuse --showvoltage 2>/dev/null | grep -vE "^=|^$" | head
