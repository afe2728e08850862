cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > gpurun_out/final_suite.log
cat gpurun_out/final_suite.log
R=r04 bash tools/run_round_profiles.sh
timeout 900 python tools/gemm_sweep.py 2>&1 | grep float > gpurun_out/r04_gemm_sweep_full_final.txt
grep -c behind gpurun_out/r04_gemm_sweep_full_final.txt; grep behind gpurun_out/r04_gemm_sweep_full_final.txt
