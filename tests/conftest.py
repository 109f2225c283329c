import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import support
    return support.oracle()


@pytest.fixture(scope="session")
def corpus():
    import support
    return support.corpus()


@pytest.fixture(scope="session")
def ctx():
    import falcon_amd
    return falcon_amd.default_context(0)
