cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b54
timeout 900 python -m pytest tests -x -q -m gpu -k "wep or sylv or waveguide" > gpurun_out/b54/pytest.log 2>&1
python bench.py --only c5step > gpurun_out/b54/c5step.json 2> gpurun_out/b54/c5step.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -o c5 -- python $GRAFT_REPO_ROOT/bench.py --only c5step > $GRAFT_REPO_ROOT/gpurun_out/b54/prof.log 2>&1)
cp $(find /tmp/pc5 -name "*kernel_stats.csv" | head -1) gpurun_out/b54/c5step_kernel_stats.csv
