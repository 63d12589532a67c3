"""Writes profiles/<name>.json: memory-side atomic requests per sample of the table-gradient scatter (k_grid_bwd) in the
two training stages, from the PMC passes of tools/pmc_train.sh (TCP_TCC_ATOMIC_WITHOUT_RET_REQ, mean per dispatch) and
the probes' samples per step, together with the sha of the kernel sources they belong to.
usage: python tools/scatter_requests_json.py gpurun_out/<tag_inst> <samples_inst> gpurun_out/<tag_nerf> <samples_nerf> out.json"""
import csv, json, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from instance_nerf_amd import build  # noqa: E402


def per_dispatch(d):
    acc = defaultdict(float)
    for row in csv.DictReader(open(os.path.join(d, "atom2_counter_collection.csv"))):
        if "k_grid_bwd" in row["Kernel_Name"] and row["Counter_Name"] == "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum":
            acc[int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    v = [acc[k] for k in sorted(acc)][-12:]            # the probe's timed steps (steady state)
    return sum(v) / len(v)


d_i, n_i, d_n, n_n, out = sys.argv[1], float(sys.argv[2]), sys.argv[3], float(sys.argv[4]), sys.argv[5]
r_i, r_n = per_dispatch(d_i), per_dispatch(d_n)
json.dump({"kernel": "k_grid_bwd", "source_sha": build.source_sha("scatter"),
           "counter": "TCP_TCC_ATOMIC_WITHOUT_RET_REQ (== TCC_EA0_WRREQ_ATOMIC_DRAM: every request goes to the memory side)",
           "instance_stage": {"requests_per_step": round(r_i), "samples_per_step": n_i, "requests_per_sample": round(r_i / n_i, 2)},
           "nerf_stage": {"requests_per_step": round(r_n), "samples_per_step": n_n, "requests_per_sample": round(r_n / n_n, 2)},
           # the probes run the product's default - fp32 atomics - unless INR_FX_GRAD=1 (the opt-in int32 form)
           "scatter_form": "int32 (fixed point)" if os.environ.get("INR_FX_GRAD", "0") == "1" else "fp32 atomics",
           "unit_rate_requests_per_s": 26.9e9 if os.environ.get("INR_FX_GRAD", "0") == "1" else 21.0e9,
           "unit_rate_source": "tools/micro/atomic_type_bench.hip, profiles/r06_atomic_type_bench.txt: 16-byte requests into a 49 MB "
                               "table, 20.97 G/s fp32 adds, 26.9 G/s int32 adds (all forwarded to the memory side)"},
          open(out, "w"), indent=2)
print(open(out).read())
