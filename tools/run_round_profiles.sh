#!/bin/bash
# The round's judged artefacts in one GPU job: kernel-trace stats of the default bench + the plain bench line (tools/make_profiles.sh),
# the per-workload counter passes (tools/pmc.sh), the rank emulation. Copy gpurun_out/profiles_new/*, gpurun_out/pmc_$R/${R}_pmc.csv and
# gpurun_out/${R}_rank_emulation.json into profiles/ afterwards.
export R=${R:-r06}
cd $GRAFT_REPO_ROOT
bash tools/make_profiles.sh > gpurun_out/make_profiles.log 2>&1
timeout 1800 bash tools/pmc.sh > gpurun_out/pmc_$R.log 2>&1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export STEPS=6
for P in 2 4 8; do python tools/rank_emulation.py $P > gpurun_out/${R}_rank_emulation_p$P.json 2> gpurun_out/rank_emulation_p$P.log; done  # one rank count per process
tail -3 gpurun_out/make_profiles.log; tail -25 gpurun_out/pmc_$R.log
