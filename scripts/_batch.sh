cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b32
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu > gpurun_out/b32/pytest.log 2>&1
python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b32/bench_g.json 2> gpurun_out/b32/bench.err
NEP_SPMM_GROUPED=0 python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b32/bench_g0.json 2>> gpurun_out/b32/bench.err
python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b32/bench_g_b.json 2>> gpurun_out/b32/bench.err
