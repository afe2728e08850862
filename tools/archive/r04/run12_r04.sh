cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q -rfE 2>&1 | tail -40 > gpurun_out/r04_tests_full2.log
tail -40 gpurun_out/r04_tests_full2.log
