cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/b35
python scripts/diag/repro_bits.py 60 8 > gpurun_out/b35/repro60.txt 2>&1
python scripts/diag/repro_bits.py 100 8 > gpurun_out/b35/repro100.txt 2>&1
python scripts/diag/repro_bits.py 60 8 > gpurun_out/b35/repro60b.txt 2>&1
