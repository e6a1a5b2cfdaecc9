set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b18
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu -k "refine or lin_solve or iar or backward or linsolver or lu" > gpurun_out/b18/t.log 2>&1; echo "rc=$?" >> gpurun_out/b18/t.log
for i in 1 2 3; do
timeout 300 python scripts/iar_runs.py 12 > gpurun_out/b18/iar_$i.log 2>&1
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr5 -o tr -- python $GRAFT_REPO_ROOT/scripts/iar_runs.py 6 > $GRAFT_REPO_ROOT/gpurun_out/b18/trace_run.log 2>&1)
cp $(find /tmp/tr5 -name "*kernel_stats.csv" | head -1) gpurun_out/b18/iar_kernel_stats.csv
echo done
