"""Host side of the Poiseuille operators' HODLR form (spheremanopt_amd/csrc/hodlr.hpp) on the CPU: tests/c/hodlr_host_test.cpp factorises the
inverse of a bordered banded matrix, packs it in the device layout and walks the descriptors as the kernel does, forward and transposed."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hodlr") / "hodlr_host_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "spheremanopt_amd", "csrc"),
                    os.path.join(ROOT, "tests", "c", "hodlr_host_test.cpp"), "-o", out], check=True)
    return out


@pytest.mark.parametrize("n,split", [(300, 2), (576, 1), (576, 3), (100, 0), (40, 2), (333, 6)])
def test_hodlr_layout_reproduces_dense_products(exe, n, split):
    r = subprocess.run([exe, str(n), str(split)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
