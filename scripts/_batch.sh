cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b44
python scripts/diag/omega_first_solve.py > gpurun_out/b44/omega.json 2>&1
