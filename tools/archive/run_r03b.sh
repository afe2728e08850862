#!/bin/bash
# round 3, GPU job b: parity of the balance units + comm fixes, A/B of calibrated shares on 8192^3
mkdir -p gpurun_out/r03d; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "balance or scheduler or f16_shapes" > $O/pytest_f16.txt 2>&1

ab() { # workload
  for i in 1 2 3; do
    for bal in 0 -1; do
      v=$(WG_F16_BALANCE=$bal WG_BENCH_NO_CHECK=1 python bench.py --steps 300 --warmup 30 --workload $1 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])")
      echo "$1 balance=$bal $v"
    done
  done
}
ab gemm_f16_8192 > $O/ab_8192.txt 2>&1
ab gemmtr_f16_8192 >> $O/ab_8192.txt 2>&1
T=$PWD/wgmath_amd/libwgebra_hip_trace1.so
WG_TRACE_REPS=16 WGEBRA_HIP_LIB=$T python tools/f16_trace.py 8192 8192 8192 > $O/trace_8192_bal.txt 2>&1
WG_F16_BALANCE=0 WG_TRACE_REPS=16 WGEBRA_HIP_LIB=$T python tools/f16_trace.py 8192 8192 8192 > $O/trace_8192_nobal.txt 2>&1
python - > $O/balance_info.txt 2>&1 <<'PY'
import numpy as np, wgmath_amd as wg
from bench import device_random
gpu = wg.GpuInstance.new(0); dev, shapes = gpu.device(), wg.ViewShapeBuffers(); S = wg.BufferUsages
a = device_random(wg, gpu, (8192, 8192), np.float16, 1); b = device_random(wg, gpu, (8192, 8192), np.float16, 2)
c = wg.TensorBuilder.matrix(8192, 8192, S.STORAGE).build(dev, np.float16)
gemm = wg.Gemm.from_device(dev)
enc = dev.create_command_encoder(); p = enc.compute_pass("t", None)
for i in range(40):
    for _ in range(10): gemm.dispatch(dev, shapes, p, c, a, b)
    gpu.sync(); print(i, gpu.f16_balance_info())
PY
