cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/k7
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ub_k7 scripts/ub_k7.hip 2> gpurun_out/k7/build.log
/tmp/ub_k7 1003995 60 60 20 > gpurun_out/k7/run60.jsonl 2>&1
/tmp/ub_k7 1003984 64 64 20 > gpurun_out/k7/run64.jsonl 2>&1
