set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/b13
for i in 1 2 3; do
timeout 300 python scripts/iar_runs.py 14 > gpurun_out/b13/iar_def_$i.log 2>&1
IAR_RUNS_HIGH_PRIO=1 timeout 300 python scripts/iar_runs.py 14 > gpurun_out/b13/iar_hp_$i.log 2>&1
done
echo done
