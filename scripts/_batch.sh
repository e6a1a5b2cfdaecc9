cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b61
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_fullsize.py tests/test_gpu_dist2.py -x -q -m gpu -k "lu or solve or iar or refine or factor or beyn or nleigs or c3 or c4 or plan or thread or dist or rank" > gpurun_out/b61/pytest.log 2>&1
python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b61/bench_g1.json 2> gpurun_out/b61/bench.err
NEP_LU_GRAPH=0 python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b61/bench_g0.json 2>> gpurun_out/b61/bench.err
python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b61/bench_g1b.json 2>> gpurun_out/b61/bench.err
