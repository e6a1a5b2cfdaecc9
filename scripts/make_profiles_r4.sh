#!/bin/bash
# regenerates the round-4 rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun, from the repo root):
#   r4_orth_kernel_stats.csv      python bench.py --only orth      (the loop `roofline` times: fixed shape, AverageNs comparable)
#   r4_k5_kernel_stats.csv        python bench.py --only k5        (`roofline_k5`)
#   r4_mlincomb_kernel_stats.csv  python bench.py --only mlincomb  (`roofline_compute_Mlincomb`)
#   r4_iar_kernel_stats.csv       9 full iar runs (config C2, eig(H_k) on the device) + r4_iar_host_eig_kernel_stats.csv (NEP_IAR_EIG=host)
#   r4_iar_trace_k6.json          per-dispatch K6 analysis of the last of 6 runs;  r4_iar_eig_timeline.txt: steps / eig batches of that run
#   r4_c5step_kernel_stats.csv    python bench.py --only c5step;  r4_c5_kernel_stats.csv: two full C5 runs
#   r4_hess_eig_bench.jsonl       scripts/hess_eig_bench.py (device eigen-decomposition against LAPACK: accuracy, kernel times)
#   r4_iar_steps_and_setup.txt    launch sequence of steps 5 / 50 / 90 and of a call's set-up from the same trace (scripts/diag/trace_steps.py)
#   r4_c4_kernel_stats.csv, r4_c3_kernel_stats.csv, r4_lufac_kernel_stats.csv   contour_beyn / nleigs runs (scripts/diag/c4_runs.py, c3_runs.py), device LU check
#   pmc2/r4_gun_traffic.json      separate --pmc FETCH_SIZE / WRITE_SIZE passes of scripts/kernel_bench.py gun (K6 kernels changed this round)
set -u
root=$(pwd)
out=gpurun_out/r4p
mkdir -p $out
for what in orth k5 mlincomb; do
  scripts/prof_stats.sh r4p/$what python $root/bench.py --only $what --reps 50
  cp $out/$what/kernel_stats.csv $out/r4_${what}_kernel_stats.csv
  grep "^{" $out/$what/cmd.log > $out/r4_${what}_bench_line.json
done
scripts/prof_stats.sh r4p/c5step python $root/bench.py --only c5step
cp $out/c5step/kernel_stats.csv $out/r4_c5step_kernel_stats.csv
grep "^{" $out/c5step/cmd.log > $out/r4_c5step_bench_line.json
scripts/prof_stats.sh r4p/c5 python $root/scripts/diag/c5_one.py
cp $out/c5/kernel_stats.csv $out/r4_c5_kernel_stats.csv
python scripts/hess_eig_bench.py 2>/dev/null | grep "^{" > $out/r4_hess_eig_bench.jsonl
scripts/prof_stats.sh r4p/iar python $root/scripts/iar_runs.py 9
cp $out/iar/kernel_stats.csv $out/r4_iar_kernel_stats.csv
NEP_IAR_EIG=host scripts/prof_stats.sh r4p/iarhost python $root/scripts/iar_runs.py 9
cp $out/iarhost/kernel_stats.csv $out/r4_iar_host_eig_kernel_stats.csv
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr4 -o tr -- python $root/scripts/iar_runs.py 6 > $root/$out/trace_run.log 2>&1)
f=$(find /tmp/tr4 -name "*kernel_trace.csv" | head -1)
python scripts/trace_k6.py $f > $out/r4_iar_trace_k6.json
python scripts/diag/trace_eig.py $f > $out/r4_iar_eig_timeline.txt
python scripts/diag/trace_steps.py $f 5 50 90 --setup > $out/r4_iar_steps_and_setup.txt
scripts/prof_stats.sh r4p/c4 python $root/scripts/diag/c4_runs.py
cp $out/c4/kernel_stats.csv $out/r4_c4_kernel_stats.csv; grep "^call" $out/c4/cmd.log > $out/r4_c4_runs.txt
scripts/prof_stats.sh r4p/c3 python $root/scripts/diag/c3_runs.py
cp $out/c3/kernel_stats.csv $out/r4_c3_kernel_stats.csv; grep "^call" $out/c3/cmd.log > $out/r4_c3_runs.txt
scripts/prof_stats.sh r4p/lufac python $root/scripts/diag/lufac_check.py
cp $out/lufac/kernel_stats.csv $out/r4_lufac_kernel_stats.csv; grep "rel diff\|factor_dev (" $out/lufac/cmd.log > $out/r4_lufac_check.txt
bash scripts/pmc_collect.sh gun $out/pmc
cp $out/pmc/gun_traffic.json $out/r4_gun_traffic.json
rm -rf $out/pmc/*_FETCH_SIZE $out/pmc/*_WRITE_SIZE $out/orth $out/k5 $out/mlincomb $out/iar $out/iarhost $out/c5step $out/c5 $out/c4 $out/c3 $out/lufac
