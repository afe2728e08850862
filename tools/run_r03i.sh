#!/bin/bash
mkdir -p gpurun_out/r03i; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_dist2.py -x -q -m gpu -k "one_launch or staged" > $O/dist2_$i.txt 2>&1; grep -E "passed|failed" $O/dist2_$i.txt | tail -1; done
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1
grep -E "passed|failed" $O/pytest_gpu.txt | tail -2
