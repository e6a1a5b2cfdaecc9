cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b41
python scripts/k1_tile_bench.py wep > gpurun_out/b41/tiles2.jsonl 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "tile or mlincomb" > gpurun_out/b41/pytest.log 2>&1
