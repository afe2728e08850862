#!/bin/bash
# round 3, GPU job f: one launch per step -- bit identity with the panel launches (1 rank RCCL / staged, 2 processes staged)
mkdir -p gpurun_out/r03f; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "one_launch" > $O/pytest_one_launch.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_dist2.py -x -q -m gpu > $O/pytest_dist2.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "config5" > $O/pytest_config5.txt 2>&1
