out=gpurun_out/r5c; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_goldens.py -x -q -m gpu > $out/pytest_kernels.log 2>&1; tail -5 $out/pytest_kernels.log
timeout 1500 python -m pytest tests/test_gpu_solvers.py -x -q -m gpu -k "wep or tiar or iar" > $out/pytest_solvers.log 2>&1; tail -5 $out/pytest_solvers.log
