"""The C++ host layer (include/dlm_host.hpp) above the C ABI: compiles on CPU, runs on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_api_check.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "host_api_check")


def _build():
    lib = os.path.join(ROOT, "bayesian_dlms_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), SRC, "-L" + lib, "-ldlm_engine",
           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    subprocess.check_call(cmd)


def test_cpp_host_layer_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_layer_matches_reference_vectors():
    _build()
    out = subprocess.run([EXE, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "HOST API OK" in out.stdout
