#!/bin/bash
# regenerates the round-3 rocprofv3 summaries committed under profiles/ (run on the GPU box through gpurun, from the repo root):
#   r3_orth_kernel_stats.csv      python bench.py --only orth      (the loop `roofline` times: fixed shape, AverageNs comparable)
#   r3_k5_kernel_stats.csv        python bench.py --only k5        (`roofline_k5`)
#   r3_mlincomb_kernel_stats.csv  python bench.py --only mlincomb  (`roofline_compute_Mlincomb`)
#   r3_iar_kernel_stats.csv       9 full iar runs (config C2) + r3_iar_trace_k6.json (per-dispatch analysis of the last run)
#   r3_k1k2_tiles_kernel_stats.csv  scripts/k1_tile_bench.py wep (tiled and two-launch K1, tiled and wave-per-row K2, n = 1e6)
#   r3_wepscale_kernel_stats.csv  python bench.py --only wepscale  (K1 / K2 / K7 at n = 1e6: `roofline_wep_scale`)
#   r3_c5step_kernel_stats.csv    python bench.py --only c5step    (one preconditioned operator step of config C5; the profile also holds the
#                                 1517 Sylvester solves of the preconditioner set-up)
#   r3_ub_k7.jsonl                scripts/ub_k7.hip (K7 kernel variants side by side, k = p = 60 and 64)
#   pmc2/r3_mfma_counters.json    MFMA busy cycles of the K7 kernels (scripts/pmc_mfma.sh)
#   pmc2/r3_gun_traffic.json, pmc2/r3_tiles_traffic.json   separate --pmc FETCH_SIZE / WRITE_SIZE passes
set -u
root=$(pwd)
out=gpurun_out/r3p
mkdir -p $out
for what in orth k5 mlincomb; do
  scripts/prof_stats.sh r3p/$what python $root/bench.py --only $what --reps 50
  cp $out/$what/kernel_stats.csv $out/r3_${what}_kernel_stats.csv
  grep "^{" $out/$what/cmd.log > $out/r3_${what}_bench_line.json
done
for what in wepscale c5step; do
  scripts/prof_stats.sh r3p/$what python $root/bench.py --only $what
  cp $out/$what/kernel_stats.csv $out/r3_${what}_kernel_stats.csv
  grep "^{" $out/$what/cmd.log > $out/r3_${what}_bench_line.json
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ub_k7 scripts/ub_k7.hip 2> $out/ub_k7_build.log
/tmp/ub_k7 1003995 60 60 50 > $out/r3_ub_k7.jsonl 2>&1
/tmp/ub_k7 1003984 64 64 50 >> $out/r3_ub_k7.jsonl 2>&1
scripts/prof_stats.sh r3p/iar python $root/scripts/iar_runs.py 9
cp $out/iar/kernel_stats.csv $out/r3_iar_kernel_stats.csv
scripts/prof_stats.sh r3p/tiles python $root/scripts/k1_tile_bench.py wep
cp $out/tiles/kernel_stats.csv $out/r3_k1k2_tiles_kernel_stats.csv
grep "^{" $out/tiles/cmd.log > $out/r3_k1k2_tiles_bench_lines.jsonl
# per-dispatch trace of 6 runs -> K6 run-weighted figure of the last one
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -o tr -- python $root/scripts/iar_runs.py 6 > $root/$out/trace_run.log 2>&1)
f=$(find /tmp/tr3 -name "*kernel_trace.csv" | head -1)
python scripts/trace_k6.py $f > $out/r3_iar_trace_k6.json
# HBM traffic from the counters (one pass per counter, never together with tracing)
bash scripts/pmc_collect.sh gun $out/pmc
cp $out/pmc/gun_traffic.json $out/r3_gun_traffic.json
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d "$root/$out/pmc/tiles_$c" -o p -- python "$root/scripts/k1_tile_bench.py" wep > "$root/$out/pmc/tiles_$c.log" 2>&1
done
cd "$root" && python scripts/pmc_summary.py "$out/pmc" tiles && cp $out/pmc/tiles_traffic.json $out/r3_tiles_traffic.json
bash scripts/pmc_mfma.sh $out/pmc_mfma && cp $out/pmc_mfma/mfma_counters.json $out/r3_mfma_counters.json
rm -rf $out/pmc/*_FETCH_SIZE $out/pmc/*_WRITE_SIZE $out/orth $out/k5 $out/mlincomb $out/iar $out/tiles $out/wepscale $out/c5step $out/pmc_mfma/run
