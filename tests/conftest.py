import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import sqy_oracle
    sqy_oracle.lib()
    return sqy_oracle


@pytest.fixture(scope="session")
def sqy():
    import sqeazy_amd
    sqeazy_amd.lib()
    return sqeazy_amd


@pytest.fixture
def options(sqy):
    """options("block_parallel", 0): SQYAMD_Set_Option for the rest of the test; every option touched is put back afterwards
    (the library reads its environment once, at load -- include/sqeazy_amd.h)"""
    saved = {}

    def set_(name, value):
        saved.setdefault(name, sqy.get_option(name))
        sqy.set_option(name, value)
    yield set_
    for name, value in saved.items():
        sqy.set_option(name, value)
