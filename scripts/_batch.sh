set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b10
timeout 1200 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_kernels.py -x -q -m gpu -k "wep or tiar or resid or column" > gpurun_out/b10/t.log 2>&1; echo "rc=$?" >> gpurun_out/b10/t.log
timeout 600 python scripts/run_configs.py c5 --wep-nx 1003 --wep-nz 999 --wep-solver gmres > gpurun_out/b10/c5_cm.log 2>&1
NEP_K2_CM=0 timeout 600 python scripts/run_configs.py c5 --wep-nx 1003 --wep-nz 999 --wep-solver gmres > gpurun_out/b10/c5_rm.log 2>&1
NEP_K2_CM=0 NEP_WEP_RESID_SPLIT=0 timeout 600 python scripts/run_configs.py c5 --wep-nx 1003 --wep-nz 999 --wep-solver gmres > gpurun_out/b10/c5_old.log 2>&1
echo done
