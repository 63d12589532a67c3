"""Writes profiles/r05_bound_traffic.json from the PMC passes of tools/pmc_bound.sh: per configuration of
tools/bound_field_probe.py (bound : scene scale : 1/dt_gamma : finest level : log2 rows) the fused field kernel's L1
look-ups, L2 requests, L2 misses and fabric read requests per sample, with the sha of the kernel sources they belong to
(bench.py quotes them for that build only).
usage: python tools/bound_traffic_json.py gpurun_out/<tag> profiles/r05_bound_traffic.json"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instance_nerf_amd import build  # noqa: E402

d, out = sys.argv[1:3]
res = {}
for sub in sorted(glob.glob(os.path.join(d, "*_*_*_*_*"))):
    if not os.path.isdir(sub):
        continue
    cfg = os.path.basename(sub).replace("_", ":")
    log = open(sub + ".tcc.log").read()
    m = re.search(r"M=(\d+)\s+field ([\d.]+) ms", log)
    if not m:
        continue
    M, ms = int(m.group(1)), float(m.group(2))
    # per kernel of the frame path (the sliced path: the level-major pre-pass + the field kernel on thirteen levels):
    # mean over its dispatches; the frame's figure is the sum over the kernels
    mean = defaultdict(float)
    for kern in ("k_nerf_fwd", "k_grid_fine_slices"):
        tot = defaultdict(lambda: defaultdict(float))
        for f in glob.glob(os.path.join(sub, "*counter_collection.csv")):
            for row in csv.DictReader(open(f)):
                if kern in row["Kernel_Name"]:
                    tot[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for k, v in tot.items():
            mean[k] += sum(v.values()) / len(v)
    res[cfg] = {"frame_path": "sliced" if cfg.endswith("-sliced") else "fused", "samples": M, "field_ms_under_the_profiler": ms,
                "l1_lookups_per_sample": round(mean["TCP_TOTAL_CACHE_ACCESSES_sum"] / M, 2),
                "l2_requests_per_sample": round(mean["TCP_TCC_READ_REQ_sum"] / M, 2),
                "l2_misses_per_sample": round(mean["TCC_MISS_sum"] / M, 2),
                "l2_hit_rate": round(mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"]), 4),
                "fabric_read_requests_per_sample": round(mean["TCC_EA0_RDREQ_sum"] / M, 3),
                "fabric_read_bytes_per_sample": round(mean["TCC_EA0_RDREQ_sum"] * 128 / M, 1)}
json.dump({"kernel": "k_nerf_fwd<true,true> (fused) | k_grid_fine_slices + k_nerf_fwd<.., kPre> (sliced)", "source_sha": build.source_sha("field"),
           "source": f"{d} (rocprofv3 --pmc, separate passes, tools/pmc_bound.sh; one 800x800 view per configuration)",
           "fabric_request_bytes": 128,
           "random_line_rate_of_the_fabric_g_per_s": 69.0,
           "random_line_rate_source": "tools/micro/level_xcd_bench.hip, 8+ levels of 4 MiB (profiles/r05_level_xcd_bench.txt)",
           "configs": res}, open(out, "w"), indent=2)
print(open(out).read())
