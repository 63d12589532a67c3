#!/bin/bash
# Round 6: where do the atomics of each operand type execute?  tools/micro/atomic_type_bench under rocprofv3 PMC (one
# counter group per process, kernel-trace only), summarised per kernel instantiation: atomics that reached the L2
# (TCC_ATOMIC) against atomics the L2 forwarded to the memory side (TCC_EA0_WRREQ_ATOMIC_DRAM / TCC_EA0_ATOMIC).
# usage (GPU box, repo root): bash tools/pmc_atomic_types.sh <tag>
TAG=${1:-pmcat}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $R/gpurun_out/$TAG -o $name -- \
    $R/tools/micro/atomic_type_bench > $R/gpurun_out/$TAG.$name.log 2>&1
  echo "$name rc=$?"
}
run atom TCC_ATOMIC_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_sum
run req TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCC_REQ_sum TCC_EA0_WRREQ_64B_sum
python3 - $R/gpurun_out/$TAG <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
names = {"0": "f32 add (agent)", "7": "f32 add (workgroup)", "1": "u32 add (agent)", "8": "u32 add (workgroup)", "6": "u32 add, returning",
         "2": "u64 add", "5": "f64 add", "3": "pk_add_bf16", "4": "pk_add_f16"}
per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))       # kernel -> counter -> dispatch -> total
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_atomic" in k:
            per[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
print(f"{'kernel':44s} {'TCC_ATOMIC':>12s} {'EA_ATOMIC_DRAM':>15s} {'EA_ATOMIC':>12s} {'EA_WRREQ':>12s} {'TCP no-ret':>12s} {'TCP ret':>10s}  forwarded")
for k in sorted(per):
    c = {n: sum(v.values()) / max(len(v), 1) for n, v in per[k].items()}
    t = k[k.index("<") + 1:k.index(">")].replace(" ", "").split(",")
    label = f"{names.get(t[0], t[0])}, {'XCD-owned 4 MiB' if t[1] == '1' else 'whole table'}"
    a = c.get("TCC_ATOMIC_sum", 0)
    print(f"{label:44s} {a:12.4g} {c.get('TCC_EA0_WRREQ_ATOMIC_DRAM_sum', 0):15.4g} {c.get('TCC_EA0_ATOMIC_sum', 0):12.4g} "
          f"{c.get('TCC_EA0_WRREQ_sum', 0):12.4g} {c.get('TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum', 0):12.4g} "
          f"{c.get('TCP_TCC_ATOMIC_WITH_RET_REQ_sum', 0):10.4g}  {c.get('TCC_EA0_WRREQ_ATOMIC_DRAM_sum', 0) / a if a else float('nan'):.3f}")
PY
