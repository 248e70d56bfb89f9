import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def in_tmp_cwd(tmp_path, monkeypatch):
    """The optimiser appends optimize_result.txt to the cwd (like the reference): isolate it."""
    monkeypatch.chdir(tmp_path)
    return tmp_path


@pytest.fixture(scope="session")
def fields384():
    """The two synthetic fields of the 256^3 cases (G = 384; 1.36 GB each, ~10 s of host FFTs apiece): made once per test session."""
    from spheremanopt_amd import kdyn
    return kdyn.synthetic_field(384, 1), kdyn.synthetic_field(384, 2)
