#!/bin/bash
# regenerates the round-6 rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun, from the repo root):
#   r6_orth / r6_k5 / r6_mlincomb _kernel_stats.csv + _bench_line.json   python bench.py --only <what>  (fixed-shape loops behind `roofline*`)
#   r6_wepscale_kernel_stats.csv + _bench_line.json    python bench.py --only wepscale (K1 / K2 super-panels, both layouts / K7 at n = 1e6)
#   r6_c5step_kernel_stats.csv + _bench_line.json      python bench.py --only c5step
#   r6_iar_kernel_stats.csv        9 full iar runs (config C2);  r6_iar_trace_k6.json, r6_iar_eig_timeline.txt, r6_iar_steps_and_setup.txt: per-dispatch
#                                  analyses of the last of 6 traced runs (19 launches per Arnoldi step: K1 is the SpMV alone)
#   r6_c4_kernel_stats.csv / r6_c4_runs.txt, r6_c3_kernel_stats.csv / r6_c3_runs.txt   contour_beyn (device tail) / nleigs (asynchronous Gram-Schmidt) runs
#   pmc2/r6_wepscale_traffic.json  separate --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --only wepscale (K2: k_tile_resid_sp / _spp)
#   pmc2/r6_gun_traffic.json       the same of scripts/kernel_bench.py gun (K6 traffic behind `roofline.traffic`)
#   r6_iar_timeline.txt            queues, busy / idle time, per-step durations and the tail of one nep_iar_run call (scripts/diag/r6_timeline.py)
#   pmc2/r6_mfma_counters.json     MFMA busy cycles of the K7 kernels on the current gemm.hip (scripts/pmc_mfma.sh)
set -u
root=$(pwd)
out=gpurun_out/r6p
mkdir -p $out
for what in orth k5 mlincomb; do
  scripts/prof_stats.sh r6p/$what python $root/bench.py --only $what --reps 50
  cp $out/$what/kernel_stats.csv $out/r6_${what}_kernel_stats.csv
  grep "^{" $out/$what/cmd.log > $out/r6_${what}_bench_line.json
done
for what in wepscale c5step; do
  scripts/prof_stats.sh r6p/$what python $root/bench.py --only $what
  cp $out/$what/kernel_stats.csv $out/r6_${what}_kernel_stats.csv
  grep "^{" $out/$what/cmd.log > $out/r6_${what}_bench_line.json
done
scripts/prof_stats.sh r6p/iar python $root/scripts/iar_runs.py 9
cp $out/iar/kernel_stats.csv $out/r6_iar_kernel_stats.csv
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr6 -o tr -- python $root/scripts/iar_runs.py 6 > $root/$out/trace_run.log 2>&1)
f=$(find /tmp/tr6 -name "*kernel_trace.csv" | head -1)
python scripts/trace_k6.py $f > $out/r6_iar_trace_k6.json
python scripts/diag/trace_eig.py $f > $out/r6_iar_eig_timeline.txt
python scripts/diag/trace_steps.py $f 5 50 90 --setup > $out/r6_iar_steps_and_setup.txt
python scripts/diag/r6_timeline.py $f > $out/r6_iar_timeline.txt
scripts/prof_stats.sh r6p/c4 python $root/scripts/diag/c4_runs.py
cp $out/c4/kernel_stats.csv $out/r6_c4_kernel_stats.csv; grep "^call\|tail" $out/c4/cmd.log > $out/r6_c4_runs.txt
scripts/prof_stats.sh r6p/c3 python $root/scripts/diag/c3_runs.py
cp $out/c3/kernel_stats.csv $out/r6_c3_kernel_stats.csv; grep "^call" $out/c3/cmd.log > $out/r6_c3_runs.txt
bash scripts/pmc_collect.sh gun $out/pmc
cp $out/pmc/gun_traffic.json $out/r6_gun_traffic.json
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --output-format csv -d "$root/$out/pmc/wepscale_$c" -o p -- python "$root/bench.py" --only wepscale > "$root/$out/pmc/wepscale_$c.log" 2>&1
  done )
python scripts/pmc_summary.py "$out/pmc" wepscale && cp $out/pmc/wepscale_traffic.json $out/r6_wepscale_traffic.json
bash scripts/pmc_mfma.sh $out/pmc_mfma && cp $out/pmc_mfma/mfma_counters.json $out/r6_mfma_counters.json
rm -rf $out/pmc/*_FETCH_SIZE $out/pmc/*_WRITE_SIZE $out/orth $out/k5 $out/mlincomb $out/iar $out/c5step $out/wepscale $out/c4 $out/c3 $out/pmc_mfma/run
