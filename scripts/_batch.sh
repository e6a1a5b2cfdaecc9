cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b43
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu -k "cw or refine or iar or gemm or k7 or backward" > gpurun_out/b43/pytest.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -o i -- python $GRAFT_REPO_ROOT/scripts/iar_runs.py 6 > $GRAFT_REPO_ROOT/gpurun_out/b43/prof.log 2>&1)
cp $(find /tmp/pi -name "*kernel_stats.csv" | head -1) gpurun_out/b43/iar_kernel_stats.csv
python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b43/bench.json 2> gpurun_out/b43/bench.err
