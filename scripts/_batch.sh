set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b3
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tiled or mlincomb or resid" > gpurun_out/b3/t_tile.log 2>&1; echo "rc=$?" >> gpurun_out/b3/t_tile.log
timeout 600 python scripts/k1_tile_bench.py all > gpurun_out/b3/k1_default.jsonl 2> gpurun_out/b3/k1.err
NEP_K1_TILE_PF=0 timeout 600 python scripts/k1_tile_bench.py wep > gpurun_out/b3/k1_wep_nopf.jsonl 2>> gpurun_out/b3/k1.err
NEP_K1_TILE_XP=8 timeout 600 python scripts/k1_tile_bench.py wep > gpurun_out/b3/k1_wep_8x64.jsonl 2>> gpurun_out/b3/k1.err
NEP_K1_TILE_XP=2 timeout 600 python scripts/k1_tile_bench.py wep > gpurun_out/b3/k1_wep_2x64.jsonl 2>> gpurun_out/b3/k1.err
NEP_K2_TILE_PS=4 timeout 600 python scripts/k1_tile_bench.py wep > gpurun_out/b3/k1_wep_ps4.jsonl 2>> gpurun_out/b3/k1.err
for shape in "4 16" "2 16" "8 8"; do set -- $shape
  NEP_K1_TILE_XP=$1 NEP_K1_TILE_ZP=$2 timeout 600 python scripts/k1_tile_bench.py gun > gpurun_out/b3/k1_gun_$1x$2.jsonl 2>> gpurun_out/b3/k1.err
done
NEP_K1_TILE_THREADS=512 timeout 600 python scripts/k1_tile_bench.py gun > gpurun_out/b3/k1_gun_t512.jsonl 2>> gpurun_out/b3/k1.err
echo done
