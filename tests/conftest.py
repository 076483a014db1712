import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hk_lib():
    from hekaton_system_amd import capi
    return capi.load()


@pytest.fixture(scope="session")
def ctx_bn254():
    from hekaton_system_amd import capi
    c = capi.Context("bn254", 0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def ctx_bls():
    from hekaton_system_amd import capi
    c = capi.Context("bls12_381", 0)
    yield c
    c.close()
