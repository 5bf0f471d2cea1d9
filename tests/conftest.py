import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import htfx
    return htfx.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden8.htfx"))


@pytest.fixture(scope="session")
def weights():
    from hand_tracking_samples_amd import weights as W
    return W.make_cnnb()
