#!/bin/bash
# the whole GPU suite with per-test durations (what to sample to keep it under 600 s), then the one-process rank-emulation repro
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -x -m gpu --durations=60 > gpurun_out/r06_suite.log 2>&1
echo "suite rc $? in $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/r06_suite.log
bash tools/run_r06_emu_repro.sh
