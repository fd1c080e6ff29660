import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """the CPU restatement (test infrastructure); built on demand with the recipe committed under oracle/"""
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def torch():
    """for the -m gpu tests: torch with a visible GPU (they fail, not skip, without one)"""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    return torch
