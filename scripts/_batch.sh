set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b6
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu -k "gemm or resid or wep or zgemm" > gpurun_out/b6/t_new.log 2>&1; echo "rc=$?" >> gpurun_out/b6/t_new.log
timeout 300 python scripts/run_configs.py c5 > gpurun_out/b6/c5.log 2>&1
bash scripts/make_profiles_r3.sh > gpurun_out/b6/profiles.log 2>&1
echo done
