cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b48
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu -k "lu or solve or iar or refine or trsv or factor" > gpurun_out/b48/pytest.log 2>&1
python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b48/bench.json 2> gpurun_out/b48/bench.err
