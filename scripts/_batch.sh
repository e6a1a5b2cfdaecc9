set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b15
for i in 1 2; do
NEP_ORTH_NT=0 timeout 300 python scripts/iar_runs.py 12 > gpurun_out/b15/iar_nt0_$i.log 2>&1
timeout 300 python scripts/iar_runs.py 12 > gpurun_out/b15/iar_auto_$i.log 2>&1
NEP_ORTH_NT=1 timeout 300 python scripts/iar_runs.py 12 > gpurun_out/b15/iar_nt1_$i.log 2>&1
NEP_ORTH_NT_MB=64 timeout 300 python scripts/iar_runs.py 12 > gpurun_out/b15/iar_mb64_$i.log 2>&1
done
NEP_ORTH_NT=0 timeout 300 python bench.py --only orth --reps 50 > gpurun_out/b15/orth_nt0.json 2>&1
NEP_ORTH_NT=1 timeout 300 python bench.py --only orth --reps 50 > gpurun_out/b15/orth_nt1.json 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solvers.py -x -q -m gpu -k "orth or dgks or iar" > gpurun_out/b15/t.log 2>&1; echo "rc=$?" >> gpurun_out/b15/t.log
echo done
