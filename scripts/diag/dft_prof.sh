#!/bin/bash
# per-kernel times of nep_wep_sylv_solve for the DFT variants: bash scripts/diag/dft_prof.sh 0 42 22   (through gpurun, from the repo root)
root=$(pwd)
for v in "$@"; do
  NEP_WEP_DFT_SYM=$v scripts/prof_stats.sh dp_$v python $root/scripts/diag/dft_sym_bench.py > /dev/null 2>&1
  echo "== NEP_WEP_DFT_SYM=$v"
  python - gpurun_out/dp_$v/kernel_stats.csv <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'dft' in n or 'tridiag' in n:
        print("  %-48s calls %5s avg %7.1f us min %7.1f max %7.1f"%(n[:48],r['Calls'],float(r['AverageNs'])/1e3,float(r['MinNs'])/1e3,float(r['MaxNs'])/1e3))
PY
done
