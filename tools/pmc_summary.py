"""Summarises rocprofv3 --pmc CSVs: per counter, the mean over the dispatches of one kernel of the per-dispatch total
(rows of one dispatch - one per counter dimension/instance - are summed first).
usage: pmc_summary.py <dir> [kernel-substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_nerf_fwd"
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    per = defaultdict(lambda: defaultdict(float))         # counter -> dispatch -> total
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            per[row["Counter_Name"]][row.get("Dispatch_Id", row.get("Correlation_Id", "0"))] += float(row["Counter_Value"])
    for k, disp in per.items():
        v = list(disp.values())
        print(f"{os.path.basename(f):32s} {k:44s} dispatches={len(v):4d} mean/dispatch={sum(v)/len(v):.6g} "
              f"min={min(v):.6g} max={max(v):.6g}")
