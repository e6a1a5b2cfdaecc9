cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b22
python scripts/diag/k7_timing.py > gpurun_out/b22/k7_timing.jsonl 2>&1
