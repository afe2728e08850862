set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_ulp.py tests/test_gpu_dist2.py tests/test_cpp_facade.py "tests/test_gpu_parity.py::test_gemv_inf_in_the_last_column_stays_inf" -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r04_tests1.log
tail -5 gpurun_out/r04_tests1.log
timeout 900 python bench.py > gpurun_out/r04_bench_first.json 2> gpurun_out/r04_bench_first.err
tail -c 2500 gpurun_out/r04_bench_first.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/vendor_trace -- python3 $GRAFT_REPO_ROOT/tools/vendor_kernel_name.py > $GRAFT_REPO_ROOT/gpurun_out/vendor_rates.txt 2>&1)
cat gpurun_out/vendor_rates.txt | tail -12
find gpurun_out/vendor_trace -name "*kernel_stats*" | head
