cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b49
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "long_and_empty or resid_batch" > gpurun_out/b49/pytest.log 2>&1
