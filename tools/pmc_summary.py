"""Summarises rocprofv3 --pmc CSVs: per-kernel mean of each counter.  usage: pmc_summary.py <dir> [kernel-substr]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_nerf_fwd"
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, v in acc.items():
        # rows are per (dispatch, counter[, dimension]); sum over dimensions per dispatch
        print(f"{os.path.basename(f):32s} {k:36s} n={len(v):4d} mean={sum(v)/len(v):.6g} sum/disp~{sum(v)/3:.6g}")
