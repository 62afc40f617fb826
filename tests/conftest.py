import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dr3():
    """the product package (directory name starts with a digit -> importlib)"""
    return importlib.import_module("3dr_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("3dr_amd.synth")


@pytest.fixture(scope="session")
def O():
    """the CPU oracle (checker only)"""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx(dr3):
    """one GPU context for the whole session; loud failure if the HIP library is missing"""
    c = dr3.Context(0)
    yield c
    c.close()
