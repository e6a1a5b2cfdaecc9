#!/bin/bash
# HBM traffic of the kernels of scripts/kernel_bench.py from hardware counters: one rocprofv3 pass per counter
# (never combined with tracing flags), summarised by scripts/pmc_summary.py.
#   bash scripts/pmc_collect.sh <gun|wep|lu> <outdir>
set -u
which=${1:-gun}; out=${2:-gpurun_out/pmc2}
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d "$root/$out/${which}_$c" -o p -- \
      python "$root/scripts/kernel_bench.py" $which --reps 3 > "$root/$out/${which}_$c.log" 2>&1
done
cd "$root" && python scripts/pmc_summary.py "$out" $which
