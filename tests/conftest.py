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
    # torch ships its own copy of the HIP runtime: when a test uses both, torch has to load (and
    # initialise) first, or its later initialisation finds the runtime of /opt/rocm already in place
    # and reports "No HIP GPUs are available".
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import falcon_amd
    return falcon_amd.default_context(0)
