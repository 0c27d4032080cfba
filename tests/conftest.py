import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from tests.oracle_c import load_oracle
    return load_oracle()


@pytest.fixture(scope="session")
def ctx():
    from nim_groth16_amd import Context
    c = Context(0)
    c.selftest()
    yield c
    c.close()
