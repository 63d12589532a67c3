"""Writes profiles/<name>_mfma.json: MFMA pipe utilisation of the fused field kernel (bench.py's render loop) and of the
two head-backward kernels of the training steps, from the `mfma` PMC pass of tools/pmc_bench.sh / tools/pmc_train.sh
(GRBM_GUI_ACTIVE + SQ_VALU_MFMA_BUSY_CYCLES in one run), keyed by the sha of the kernel sources.

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 256 CUs x 4 SIMDs),   cycles = GRBM_GUI_ACTIVE / 8 XCDs

SQ_VALU_MFMA_BUSY_CYCLES ticks in CYCLES of a SIMD's matrix pipe (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC
units": cycles, not the quad-cycles of SQ_WAVE_CYCLES / SQ_BUSY_CYCLES), summed over every SIMD of the chip; `cycles` =
GRBM_GUI_ACTIVE of the dispatch divided by the number of counter instances rocprofv3 reports for it (one per XCD).
usage: python tools/mfma_json.py gpurun_out/<bench tag> gpurun_out/<instance-step tag> gpurun_out/<nerf-step tag> out.json"""
import csv
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instance_nerf_amd import build  # noqa: E402

CUS, SIMDS, XCDS = 256, 4, 8


def kernel_busy(d, sub, last=None):
    """-> dict for the kernels whose name contains `sub` in <d>/mfma_counter_collection.csv (mean over dispatches)."""
    f = os.path.join(d, "mfma_counter_collection.csv")
    tot = defaultdict(lambda: defaultdict(float))             # counter -> dispatch -> total
    inst = defaultdict(lambda: defaultdict(int))              # counter -> dispatch -> rows (counter instances)
    name = None
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            name = row["Kernel_Name"]
            k = int(row["Dispatch_Id"])
            tot[row["Counter_Name"]][k] += float(row["Counter_Value"])
            inst[row["Counter_Name"]][k] += 1
    if not tot:
        return None
    ids = sorted(tot["SQ_VALU_MFMA_BUSY_CYCLES"])
    ids = ids[-last:] if last else ids
    busy = [tot["SQ_VALU_MFMA_BUSY_CYCLES"][k] for k in ids]
    # cycles of ONE XCD: rocprofv3 on this image reports GRBM_GUI_ACTIVE as one row per dispatch that is already the SUM
    # over the chip's 8 XCDs (74.1 M for a 4.5-5 ms launch would otherwise be a 15 GHz clock; / 8 = 1.9-2.0 GHz); a
    # version that reports one row per XCD is handled by the row count
    cyc = [tot["GRBM_GUI_ACTIVE"][k] / (XCDS if inst["GRBM_GUI_ACTIVE"][k] == 1 else inst["GRBM_GUI_ACTIVE"][k]) for k in ids]
    frac = [b / (c * CUS * SIMDS) for b, c in zip(busy, cyc) if c > 0]
    return {"kernel": name[:60], "dispatches": len(ids), "mfma_busy": round(sum(frac) / len(frac), 4),
            "mfma_busy_min": round(min(frac), 4), "mfma_busy_max": round(max(frac), 4),
            "mfma_busy_cycles_per_dispatch": round(sum(busy) / len(busy)),
            "gpu_cycles_per_xcd_per_dispatch": round(sum(cyc) / len(cyc)),
            "grbm_instances": int(max(inst["GRBM_GUI_ACTIVE"][k] for k in ids))}


d_bench, d_inst, d_nerf, out = sys.argv[1:5]
rec = {"source_sha": build.source_sha("field"),
       "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); BUSY counts cycles per SIMD",
       "source": f"{d_bench}, {d_inst}, {d_nerf} (rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ..., tools/pmc_bench.sh / pmc_train.sh)"}
rec["k_nerf_fwd"] = kernel_busy(d_bench, "k_nerf_fwd")
for d, sub, key in ((d_inst, "k_instance_head_bwd", "k_instance_head_bwd"), (d_nerf, "k_nerf_head_bwd", "k_nerf_head_bwd"),
                    (d_inst, "k_instance_fwd", "k_instance_fwd_train"), (d_inst, "k_grid_bwd", "k_grid_bwd")):
    try:
        r = kernel_busy(d, sub, last=12)
        if r is not None:
            rec[key] = r
    except (OSError, KeyError, ZeroDivisionError) as e:
        print(f"{key}: {type(e).__name__}: {e}", file=sys.stderr)
json.dump(rec, open(out, "w"), indent=2)
print(open(out).read())
