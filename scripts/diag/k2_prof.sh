root=$(pwd)
for m in "1 1" "1 0" "0 0"; do set -- $m; NEP_K2_SP=$1 NEP_K2_SP_PERSIST=$2 scripts/prof_stats.sh k2_prof/sp$1p$2 python $root/bench.py --only wepscale; echo "== SP=$1 PERSIST=$2"; grep -E "k_tile_resid|k_spmm_rm|k_sum_partials|copyBuffer" gpurun_out/k2_prof/sp$1p$2/kernel_stats.csv | cut -d, -f1-5 | cut -c1-150; done
