cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/r3p
bash scripts/make_profiles_r3.sh > gpurun_out/make_profiles.log 2>&1
