export R=${R:-r06}
cd $GRAFT_REPO_ROOT
bash tools/make_profiles.sh > gpurun_out/make_profiles.log 2>&1
timeout 2400 bash tools/pmc.sh > gpurun_out/pmc_$R.log 2>&1
tail -3 gpurun_out/make_profiles.log; tail -30 gpurun_out/pmc_$R.log
