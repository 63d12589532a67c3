"""Writes profiles/<tag>_traffic.json from the PMC passes of tools/pmc_bench.sh: HBM-side bytes per sample of the fused
field kernel (FETCH_SIZE x 1024 x 2 on gfx950 - 128-byte requests tallied at 64 B, MI355X_MICROARCH.md section HBM -
and WRITE_SIZE x 1024) over the TIMED dispatches of `bench.py --steps 4 --warmup 1`, together with the sha of the
kernel sources the numbers belong to (bench.py quotes them only for that build).
usage: python tools/traffic_json.py gpurun_out/<tag> gpurun_out/<tag>.hbm_rd.log profiles/r03_traffic.json"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instance_nerf_amd import build  # noqa: E402

d, log, out = sys.argv[1:4]
line = [json.loads(l) for l in open(log) if l.startswith("{") and '"metric"' in l][-1]
steps, per_step = line["steps"], line["config"]["samples_per_step"]


def per_dispatch(name, counter):
    acc = defaultdict(float)
    for row in csv.DictReader(open(os.path.join(d, name + "_counter_collection.csv"))):
        if "k_nerf_fwd" in row["Kernel_Name"] and row["Counter_Name"] == counter:
            acc[int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    vals = [acc[k] for k in sorted(acc)]
    return vals[-steps:]                       # the timed launches (warm-up first)


rd = sum(per_dispatch("hbm_rd", "FETCH_SIZE")) * 1024 * 2
wr = sum(per_dispatch("hbm_wr", "WRITE_SIZE")) * 1024
n = per_step * steps
json.dump({"kernel": "k_nerf_fwd<true,true>", "source_sha": build.source_sha("field"),
           "workload": f"bench.py render 800x800, {steps} timed views, {n} samples",
           "source": f"{d} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_bench.sh)",
           "correction": "FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE x1",
           "read_bytes_per_sample": round(rd / n, 1), "write_bytes_per_sample": round(wr / n, 1)}, open(out, "w"), indent=2)
print(open(out).read())
