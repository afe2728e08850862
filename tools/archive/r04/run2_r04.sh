set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_ulp.py tests/test_gpu_dist2.py tests/test_cpp_facade.py "tests/test_gpu_parity.py::test_gemv_inf_in_the_last_column_stays_inf" "tests/test_gpu_fullsize.py::test_config5_gemm_f16_32768_sharded_entry_point" -m gpu -q 2>&1 | tail -60 > gpurun_out/r04_tests2.log
tail -30 gpurun_out/r04_tests2.log
