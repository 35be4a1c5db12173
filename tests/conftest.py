import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_oracle():
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "ndsm_oracle.c")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])


@pytest.fixture(scope="session")
def port():
    _ensure_oracle()
    from oracle import Oracle
    return Oracle("port")


@pytest.fixture(scope="session")
def ref():
    from oracle import Oracle, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return Oracle("ref")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
