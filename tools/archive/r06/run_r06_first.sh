#!/bin/bash
# round 6, first GPU call: the driver's own command line, the line's size, the new default-line test
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06_first_detail.json > gpurun_out/r06_first_line.json 2> gpurun_out/r06_first.log
echo "rc $? bytes $(wc -c < gpurun_out/r06_first_line.json)"
tail -c 2200 gpurun_out/r06_first_line.json
timeout 900 python -m pytest tests/test_gpu_dist2.py -q -x -m gpu -k "default_bench_line or ends_with_targets" 2>&1 | tail -5
