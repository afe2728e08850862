#!/bin/bash
# end of round: the full GPU suite (timed), smoke, the driver's own bench command line
mkdir -p gpurun_out/final; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
t0=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/final/pytest_gpu.txt 2>&1
echo "suite rc $? in $(( $(date +%s) - t0 )) s"; grep -E "passed|failed" gpurun_out/final/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/final/bench_detail.json > gpurun_out/final/bench.json 2> gpurun_out/final/bench.log
echo "bench rc $? bytes $(wc -c < gpurun_out/final/bench.json)"
