"""The C++ host-side mirror (include/wgebra.hpp): compiles and links on CPU; on the GPU box the reference's four tests
written in C++ (tests/cpp/reference_tests.cpp) run against the HIP kernels through the C ABI."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "reference_tests")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "wgmath_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "reference_tests.cpp"), "-o", EXE, "-L", lib_dir, "-lwgebra_hip",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)


COMM_EXE = os.path.join(ROOT, "tests", "cpp", "_build", "comm_tests")


def build_comm():
    """tests/cpp/comm_tests.cpp: the multi-GPU entry points through the plain C ABI (clang++: _Float16 on the host)."""
    os.makedirs(os.path.dirname(COMM_EXE), exist_ok=True)
    lib_dir = os.path.join(ROOT, "wgmath_amd")
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "comm_tests.cpp"), "-o", COMM_EXE, "-L", lib_dir, "-lwgebra_hip",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True)


def test_cpp_comm_tests_compile_and_link():
    build_comm()
    assert os.path.exists(COMM_EXE)


@pytest.mark.gpu
def test_cpp_comm_tests_on_gpu():
    """1-rank RCCL round trip + sharded Gemm through the C ABI, the 2-rank staged gather, a peer that misses a step (time-out, report, clean
    retry) and pipelined one-launch steps of alternating shapes (no torch in the process)."""
    build_comm()
    # GPU_MAX_HW_QUEUES: the test drives two ranks of ONE device from one thread; the staged engine's wait kernel of rank 0 must not share a
    # hardware queue with rank 1's streams (HIP folds a process's streams onto 4 queues by default)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", GPU_MAX_HW_QUEUES="24")
    r = subprocess.run([COMM_EXE], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr


def test_cpp_facade_compiles_and_links():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_reference_tests_on_gpu():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr
