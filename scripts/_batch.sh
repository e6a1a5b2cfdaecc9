cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b58
export NEP_TILE_BENCH_KS=8,16,24
python scripts/k1_tile_bench.py wep 2>&1 | grep "K2" > gpurun_out/b58/ps2.jsonl
NEP_K2_TILE_PS=4 python scripts/k1_tile_bench.py wep 2>&1 | grep "K2" > gpurun_out/b58/ps4.jsonl
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "resid" > gpurun_out/b58/pytest.log 2>&1
