cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b57
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/b57/pytest.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/b57/bench.json 2> gpurun_out/b57/bench.err
