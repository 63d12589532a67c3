#!/bin/bash
# Why does the headline read 0.80 of the roofline on some calls and 0.88 on others?  Headline-only bench on a fresh
# box, right after the GPU test suite, after a pause, and again - with the clocks rocm-smi reports beside each.
# usage (GPU box, repo root): bash tools/bimodal_probe.sh <tag>
TAG=${1:-bimodal}
O=gpurun_out/$TAG; mkdir -p $O
one() {
  python bench.py --no-cpu-baseline --no-train-probe --no-trained-scene > $O/$1.json 2>/dev/null
  python - "$O/$1.json" "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:24s} value {d['value']:8.1f}  frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.3f} ms")
PY
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -3 | tr -s ' ' | tr '\n' ';'; echo
}
one fresh_1
one fresh_2
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -1 $O/pytest.txt
one after_suite_1
one after_suite_2
sleep 45
one after_pause_1
python -m pytest tests/test_gpu_ddp.py -x -q -m gpu > $O/pytest_ddp.txt 2>&1; tail -1 $O/pytest_ddp.txt
one after_ddp_tests
python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_par.txt 2>&1; tail -1 $O/pytest_par.txt
one after_parity_tests
