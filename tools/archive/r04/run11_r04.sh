cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -60 > gpurun_out/r04_flaky_$i.log; tail -4 gpurun_out/r04_flaky_$i.log; done
grep -l "FAILED" gpurun_out/r04_flaky_*.log | head -1 | xargs -r cat | head -120
