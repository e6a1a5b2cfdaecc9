#!/bin/bash
# per-kernel times of one K6 shape: bash scripts/diag/orth_prof.sh <rows:k[:n]> ...   (through gpurun, from the repo root)
root=$(pwd)
for sh in "$@"; do
  ORTH_SHAPES=$sh scripts/prof_stats.sh op_$sh python $root/scripts/diag/orth_shapes.py > /dev/null 2>&1
  echo "== $sh"; grep "^rows" gpurun_out/op_$sh/cmd.log
  python - gpurun_out/op_$sh/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'orth' in n:
        print("  %-40s calls %5s avg %7.1f us min %7.1f max %7.1f"%(n[:40],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3))
PY
done
