cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b53
NEP_TILE_BENCH_KS=8,12,16,24,32 python scripts/k1_tile_bench.py wep > gpurun_out/b53/tiles_u16.jsonl 2>&1
