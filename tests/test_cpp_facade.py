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


def test_cpp_facade_compiles_and_links():
    build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_reference_tests_on_gpu():
    build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 0 and "ALL OK" in r.stdout, r.stdout + r.stderr
