import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One libsfmhip context for the whole GPU session (fails loudly without a GPU)."""
    from sfm_opencv_amd.api import Context
    c = Context(0, use_torch_stream=True)   # torch allocations, copies and the library share one stream
    yield c
    c.close()
