#!/bin/bash
# MFMA utilisation of K7 (k_gemm_ts) at waveguide size from hardware counters (one rocprofv3 --pmc pass, no tracing flags):
#   bash scripts/pmc_mfma.sh <outdir>
set -u
out=${1:-gpurun_out/pmc_mfma}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$root/$out/run" -o p -- python "$root/scripts/kernel_bench.py" wep --reps 3 > "$root/$out/run.log" 2>&1
cd "$root" && python - "$out" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "run", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if name.startswith("k_gemm_ts") or name.startswith("k_orth") or name.startswith("k_vc"):
            acc[name + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
for k, e in out.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("SQ_BUSY_CYCLES"):
        e["mfma_busy_over_sq_busy"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        e["mfma_busy_cycles_per_gpu_cycle"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["GRBM_GUI_ACTIVE"]
        # busy cycles of all 1024 SIMDs against the cycles the launch took (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
        e["mfma_utilisation"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * e["GRBM_GUI_ACTIVE"] / 8.0)
json.dump(out, open(os.path.join(d, "mfma_counters.json"), "w"), indent=1)
for k, e in out.items():
    print(k[:50], {c: (round(v, 3) if v < 100 else int(v)) for c, v in e.items()})
PY
