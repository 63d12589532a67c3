"""Timeline of the LAST training step in a rocprofv3 --kernel-trace csv: start (us from the step's first launch), end,
duration, queue, name - overlapping launches (two streams / graph branches) show as interleaved intervals.
usage: python tools/step_timeline.py <trace dir> [first-kernel substring, default k_instance_fwd] [n_steps, default 1]"""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
first = sys.argv[2] if len(sys.argv) > 2 else "k_instance_fwd"
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = [r for f in files for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
s, e = idx[-1 - nst], idx[-1]
t0 = int(rows[s]["Start_Timestamp"])
busy_end, overlap = 0, 0.0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if st < busy_end:
        overlap += (min(en, busy_end) - st) / 1e3
    busy_end = max(busy_end, en)
    name = r["Kernel_Name"].replace("void ", "").replace("at::native::", "")
    print(f"{st / 1e3:8.1f} {en / 1e3:8.1f} {(en - st) / 1e3:7.1f}  q{r.get('Queue_Id', '?'):>3}  {name[:90]}")
print(f"{e - s} launches over {nst} step(s), span {(int(rows[e]['Start_Timestamp']) - t0) / 1e3 / nst:.1f} us per step, "
      f"{overlap / nst:.1f} us per step of launches running beside an earlier one")
