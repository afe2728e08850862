#!/bin/bash
# round 3, GPU job h: full GPU suite + the default bench line
mkdir -p gpurun_out/r03h; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.txt 2>&1
for i in 1 2; do WG_BENCH_NO_CHECK=1 python bench.py --steps 2000 --warmup 100 --workload gemv_f32_1024 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gemv_f32_1024', d['value'], d['roofline'])"; WG_BENCH_NO_CHECK=1 python bench.py --steps 200 --warmup 20 --workload gemv_f32_1024_graph --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gemv_f32_1024_graph', d['value'], d['roofline'])"; done > $O/gemv1024.txt 2>&1
python bench.py > $O/bench.json 2> $O/bench.log
