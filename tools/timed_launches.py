"""The headline kernel's launches in the kernel trace of the profiled default bench command, beside the launch times
bench.py measured with HIP events in the same run (profiles/r0Xz_bench_timed_launches.txt).
Sequence of fused frame-kernel launches in bench.py: W warm-up + K timed frames of the pipelined loop (the headline's
timed region -> roofline.frac / avg_launch_ms), then 4 warm-up + min(K, 60) timed frames of the one-stream loop
(-> roofline.frac_kernel_alone / avg_launch_ms_alone, the kernel with the chip to itself), then the parity / probe legs.
usage: python tools/timed_launches.py <trace dir> <bench_profiled.json> [warmup=3] [steps=20] > out.txt"""
import csv
import glob
import json
import os
import re
import sys

trace, line = sys.argv[1], sys.argv[2]
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
# the fused frame kernel: colour + table feed, no saving, fp32 table, bf16x3 MLP, not the pre-pass variant
name = re.compile(r"k_nerf_fwd<true, true, 0, false, false(, false)?>")
rows = []
for f in glob.glob(os.path.join(trace, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if name.search(r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
ms = [(e - s) / 1e6 for s, e in rows]
d = json.loads([l for l in open(line).read().strip().splitlines() if l.startswith("{")][-1])
print(f"k_nerf_fwd<true,true,0,false,false,false> in the kernel trace of the profiled default command (the --stats summary of the "
      f"same trace averages over ALL {len(ms)} launches of the process, probe legs included)")
n_one = min(steps, 60)
a, b = warmup + steps, warmup + steps + 4
print("launch  ms")
for i, t in enumerate(ms[:b + n_one + 2]):
    tag = ("   warm-up (pipelined)" if i < warmup else f"   <- timed frame {i - warmup} (pipelined loop)" if i < a else
           "   warm-up (one stream)" if i < b else f"   <- one-stream frame {i - b}" if i < b + n_one else "")
    print(f"{i:4d}  {t:.4f}{tag}")
print("...")
if len(ms) >= b + n_one:
    timed, alone = ms[warmup:a], ms[b:b + n_one]
    rf = d["roofline"]
    print(f"pipelined loop, {steps} timed launches from the trace: mean {sum(timed) / len(timed):.4f} ms;  bench.py "
          f"roofline.avg_launch_ms of the same run (HIP events, the timed region = roofline.frac): {rf.get('avg_launch_ms')} ms")
    print(f"one-stream loop, {n_one} launches from the trace: mean {sum(alone) / len(alone):.4f} ms;  bench.py "
          f"roofline.avg_launch_ms_alone (the kernel alone = roofline.frac_kernel_alone, HIP events): {rf.get('avg_launch_ms_alone')} ms")
if ms:
    print(f"mean over all {len(ms)} launches (what --stats prints): {sum(ms) / len(ms):.4f} ms")
else:
    print("no launch of the kernel in the trace (kernel name changed?)")
