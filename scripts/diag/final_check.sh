out=gpurun_out/r5final; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 300 $out/bench.json
