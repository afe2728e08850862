#!/bin/bash
# round-2 GPU call: new multi-rank tests + bench plumbing (1 GPU: forced communicator path, 2 ranks sharing the GPU)
set -u
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests/test_gpu_dist2.py tests/test_cpp_facade.py "tests/test_gpu_parity.py::test_scratch_regrow_keeps_recorded_command_buffers_valid" "tests/test_gpu_parity.py::test_record_replay_gemm_chain" -x -q -m gpu > $OUT/r02b_pytest.log 2>&1
tail -5 $OUT/r02b_pytest.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k config5 > $OUT/r02b_config5.log 2>&1
tail -5 $OUT/r02b_config5.log
WG_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload gemm_f16_32768 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $OUT/r02b_dist1.json 2> $OUT/r02b_dist1.err
tail -c 1500 $OUT/r02b_dist1.json; tail -5 $OUT/r02b_dist1.err
WG_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus 2 --workload gemm_f16_8192 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $OUT/r02b_over2.json 2> $OUT/r02b_over2.err
tail -c 1500 $OUT/r02b_over2.json; tail -5 $OUT/r02b_over2.err
timeout 120 python bench.py --gpus 2 --steps 1 > $OUT/r02b_gpus2.json 2> $OUT/r02b_gpus2.err; echo "rc(--gpus 2 on 1 GPU)=$?"; tail -3 $OUT/r02b_gpus2.err
