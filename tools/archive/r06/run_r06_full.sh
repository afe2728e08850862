#!/bin/bash
# the whole GPU suite (timed), smoke, then the default bench line of this box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -q -x -m gpu --durations=12 > gpurun_out/r06_suite.log 2>&1
echo "suite rc $? in $(( $(date +%s) - t0 )) s"; tail -16 gpurun_out/r06_suite.log | grep -v "^RCCL\|^HIP v\|^ROCm\|^Hostn\|^Librccl"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06_bench_box_detail.json > gpurun_out/r06_bench_box.json 2> gpurun_out/r06_bench_box.log
echo "bench rc $? bytes $(wc -c < gpurun_out/r06_bench_box.json)"; tail -c 1500 gpurun_out/r06_bench_box.json
