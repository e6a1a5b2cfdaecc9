#!/bin/bash
# regenerates the rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun):
#   r2_orth_kernel_stats.csv     python bench.py --only orth      (the loop `roofline` times: fixed shape, AverageNs comparable)
#   r2_k5_kernel_stats.csv       python bench.py --only k5        (`roofline_k5`)
#   r2_mlincomb_kernel_stats.csv python bench.py --only mlincomb  (`roofline_compute_Mlincomb`)
#   r2_iar_kernel_stats.csv      13 full iar runs (config C2)
set -u
root=$(pwd)
for what in orth k5 mlincomb; do
  scripts/prof_stats.sh r2p/$what python $root/bench.py --only $what --reps 50
  cp gpurun_out/r2p/$what/kernel_stats.csv gpurun_out/r2p/r2_${what}_kernel_stats.csv
  tail -n +1 gpurun_out/r2p/$what/cmd.log | grep "^{" > gpurun_out/r2p/r2_${what}_bench_line.json
done
scripts/prof_stats.sh r2p/iar python $root/scripts/diag/iar_lag.py 9
cp gpurun_out/r2p/iar/kernel_stats.csv gpurun_out/r2p/r2_iar_kernel_stats.csv
