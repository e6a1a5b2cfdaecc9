out=gpurun_out/r5f; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_solvers.py tests/test_gpu_fullsize.py -x -q -m gpu -k "nleigs or c3" > $out/pytest_nleigs.log 2>&1; tail -5 $out/pytest_nleigs.log
python scripts/diag/c3_runs.py 8 > $out/c3_runs.txt 2>&1; tail -9 $out/c3_runs.txt
NEP_NLEIGS_SYNC=1 python scripts/diag/c3_runs.py 6 > $out/c3_runs_sync.txt 2>&1; tail -4 $out/c3_runs_sync.txt
