out=gpurun_out/r5e; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_solvers.py tests/test_goldens.py tests/test_gpu_dist2.py -x -q -m gpu -k "beyn" > $out/pytest_beyn.log 2>&1; tail -8 $out/pytest_beyn.log
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c4" > $out/pytest_c4.log 2>&1; tail -4 $out/pytest_c4.log
python scripts/diag/c4_runs.py > $out/c4_runs.txt 2>&1; tail -12 $out/c4_runs.txt
