#!/bin/bash
# second part of tools/bimodal_probe.sh: what is left on the box after tests/test_gpu_ddp.py (the suspect)?
TAG=${1:-bimodal2}
O=gpurun_out/$TAG; mkdir -p $O
one() {
  python bench.py --no-cpu-baseline --no-train-probe --no-trained-scene > $O/$1.json 2>/dev/null
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:24s} value {d['value']:8.1f}  frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.3f} ms")
PY
}
state() {
  echo "--- $1: processes / VRAM"
  ps -eo pid,ppid,stat,etimes,rss,cmd | grep -E "python|torch|pytest" | grep -v grep | cut -c1-150
  rocm-smi --showpids 2>/dev/null | grep -vE "^=|^$" | head -12
  rocm-smi --showmeminfo vram 2>/dev/null | grep -E "Used|Total" | tr -s ' '
}
one fresh
state fresh
for t in $(python -m pytest tests/test_gpu_ddp.py -q -m gpu --collect-only 2>/dev/null | grep "::" ); do
  python -m pytest "$t" -x -q -m gpu > $O/one_test.txt 2>&1; echo "$(tail -1 $O/one_test.txt)  <- $t"
  one "after_$(echo $t | sed 's/.*:://' | cut -c1-40)"
done
state after_ddp
sleep 20
state after_ddp_20s
