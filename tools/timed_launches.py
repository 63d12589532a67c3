"""The headline kernel's launches in the kernel trace of the profiled default bench command, beside the launch time
bench.py measured with HIP events in the same run (profiles/r03z_bench_timed_launches.txt).
usage: python tools/timed_launches.py <trace dir> <bench_profiled.json> [warmup=3] [steps=20] > out.txt"""
import csv
import glob
import json
import os
import sys

trace, line = sys.argv[1], sys.argv[2]
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 3
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rows = []
for f in glob.glob(os.path.join(trace, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_nerf_fwd<true, true, 0, false, false>" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
ms = [(e - s) / 1e6 for s, e in rows]
d = json.loads(open(line).read().strip().splitlines()[-1])
print(f"k_nerf_fwd<true,true,0,false,false> in the kernel trace of the profiled default command "
      f"(profiles/r03z_bench_kernel_stats.csv is the\n--stats summary of the same trace: its average runs over ALL "
      f"{len(ms)} launches of the process - the headline's {warmup} warm-up + {steps} timed\nframes, the 25 frames of "
      f"the two-stream `pipelined` leg right behind them, the parity re-renders, the instance-render probe and the\n"
      f"72 M-sample frames of the trained-scene leg).")
print("launch  ms")
for i, t in enumerate(ms[:warmup + steps + 3]):
    tag = "   warm-up" if i < warmup else (f"   <- timed frame {i - warmup}" if i < warmup + steps else "")
    print(f"{i:4d}  {t:.4f}{tag}")
print("...")
timed = ms[warmup:warmup + steps]
print(f"mean of the {steps} timed launches from the trace: {sum(timed) / len(timed):.4f} ms;  bench.py "
      f"roofline.avg_launch_ms of the same run (HIP events): {d['roofline']['avg_launch_ms']} ms")
print(f"mean over all {len(ms)} launches (what --stats prints): {sum(ms) / len(ms):.4f} ms")
