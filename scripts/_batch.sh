cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b46
for at in 12 1 4 8 12 1 6; do
NEP_ML_APEX_AT=$at python bench.py --steps 30 --warmup 5 --no-c5 --no-cold 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$at', j['value'], j['ms_per_step'])" >> gpurun_out/b46/apex_at.txt
done
