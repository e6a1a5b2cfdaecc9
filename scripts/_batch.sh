cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b52
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/b52/pytest.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/b52/bench.json 2> gpurun_out/b52/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/b52/smoke.log 2>&1
bash scripts/prof_stats.sh b52/iar python $GRAFT_REPO_ROOT/scripts/iar_runs.py 9 > /dev/null 2>&1
bash scripts/prof_stats.sh b52/k5 python $GRAFT_REPO_ROOT/bench.py --only k5 --reps 50 > /dev/null 2>&1
grep "^{" gpurun_out/b52/k5/cmd.log > gpurun_out/b52/k5_bench_line.json
