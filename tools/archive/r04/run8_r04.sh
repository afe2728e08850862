set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r04_tests_full.log
tail -25 gpurun_out/r04_tests_full.log
