set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b4
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_goldens.py -x -q -m gpu > gpurun_out/b4/t_kernels.log 2>&1; echo "rc=$?" >> gpurun_out/b4/t_kernels.log
timeout 300 python bench.py --only k5 --reps 200 > gpurun_out/b4/k5_new.json 2> gpurun_out/b4/k5.err
NEP_ML_U0FUSE=0 NEP_ML_GEMV1=0 timeout 300 python bench.py --only k5 --reps 200 > gpurun_out/b4/k5_old.json 2>> gpurun_out/b4/k5.err
NEP_ML_U0FUSE=0 timeout 300 python bench.py --only k5 --reps 200 > gpurun_out/b4/k5_gemv_only.json 2>> gpurun_out/b4/k5.err
timeout 300 python scripts/iar_runs.py 10 > gpurun_out/b4/iar_new.log 2>&1
NEP_ML_U0FUSE=0 NEP_ML_GEMV1=0 NEP_IAR_RESID_OVERLAP=0 timeout 300 python scripts/iar_runs.py 10 > gpurun_out/b4/iar_old.log 2>&1
NEP_IAR_RESID_OVERLAP=0 timeout 300 python scripts/iar_runs.py 10 > gpurun_out/b4/iar_nooverlap.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_solvers.py -x -q -m gpu > gpurun_out/b4/t_solvers.log 2>&1; echo "rc=$?" >> gpurun_out/b4/t_solvers.log
echo done
