set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b17
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu -k "orth or dgks or iar or gmres" > gpurun_out/b17/t.log 2>&1; echo "rc=$?" >> gpurun_out/b17/t.log
NEP_ORTH_FUSED_DOTS=0 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "orth_dev_fused" > gpurun_out/b17/t_unfused.log 2>&1; echo "rc=$?" >> gpurun_out/b17/t_unfused.log
for i in 1 2; do
NEP_ORTH_FUSED_DOTS=0 timeout 300 python scripts/iar_runs.py 12 > gpurun_out/b17/iar_unfused_$i.log 2>&1
timeout 300 python scripts/iar_runs.py 12 > gpurun_out/b17/iar_fused_$i.log 2>&1
done
NEP_ORTH_FUSED_DOTS=0 timeout 300 python bench.py --no-c5 --no-c3 --no-beyn --no-cpu-baseline --no-wep-roofline --no-cold > gpurun_out/b17/bench_unfused.json 2> gpurun_out/b17/bench_unfused.err
timeout 300 python bench.py --no-c5 --no-c3 --no-beyn --no-cpu-baseline --no-wep-roofline --no-cold > gpurun_out/b17/bench_fused.json 2> gpurun_out/b17/bench_fused.err
timeout 300 python scripts/run_configs.py c5 --wep-nx 1003 --wep-nz 999 --wep-solver gmres > gpurun_out/b17/c5_fused.log 2>&1
NEP_ORTH_FUSED_DOTS=0 timeout 300 python scripts/run_configs.py c5 --wep-nx 1003 --wep-nz 999 --wep-solver gmres > gpurun_out/b17/c5_unfused.log 2>&1
echo done
