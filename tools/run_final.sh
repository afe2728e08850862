#!/bin/bash
# end of round: the full GPU suite, then the judged profiles (tools/run_round_profiles.sh)
mkdir -p gpurun_out/final; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1
grep -E "passed|failed" gpurun_out/final/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
bash tools/vendor_vs_ours.sh > gpurun_out/final/vendor_vs_ours.txt 2>&1; cat gpurun_out/final/vendor_vs_ours.txt | grep -v amdgpu
bash tools/run_round_profiles.sh
