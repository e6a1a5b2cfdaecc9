cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b37
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/b37/pytest.log 2>&1
rm -rf gpurun_out/r3p
bash scripts/make_profiles_r3.sh > gpurun_out/make_profiles.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/b37/bench.json 2> gpurun_out/b37/bench.err
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/b37/smoke.log 2>&1
