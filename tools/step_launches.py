"""Lists the launches of the LAST training step in a rocprofv3 --kernel-trace csv (gap before, duration, name).
usage: python tools/step_launches.py <trace dir> [first-kernel substring, default k_near_far]"""
import csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
first = sys.argv[2] if len(sys.argv) > 2 else "k_near_far"
rows = [r for f in files for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
s, e = idx[-2], idx[-1]
prev, tot = None, 0.0
for r in rows[s:e]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev) / 1e3 if prev else 0.0
    tot += (en - st) / 1e3
    name = r["Kernel_Name"].replace("void ", "").replace("at::native::", "")
    print(f"{gap:7.1f} {(en - st) / 1e3:7.1f}  grid {int(r['Grid_Size_X']):>8}  {name[:100]}")
    prev = en
print(f"{e - s} launches, kernel time {tot:.1f} us, span {(int(rows[e]['Start_Timestamp']) - int(rows[s]['Start_Timestamp'])) / 1e3:.1f} us")
