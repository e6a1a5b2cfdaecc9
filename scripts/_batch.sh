cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b38
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py tests/test_gpu_fullsize.py -x -q -m gpu -k "lu or solve or iar or refine or trsv or factor or beyn or nleigs or c3 or c4 or plan" > gpurun_out/b38/pytest.log 2>&1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pi -o i -- python $GRAFT_REPO_ROOT/scripts/iar_runs.py 6 > $GRAFT_REPO_ROOT/gpurun_out/b38/prof.log 2>&1)
cp $(find /tmp/pi -name "*kernel_stats.csv" | head -1) gpurun_out/b38/iar_kernel_stats.csv
python bench.py --steps 30 --warmup 5 --no-c5 --no-cold > gpurun_out/b38/bench.json 2> gpurun_out/b38/bench.err
