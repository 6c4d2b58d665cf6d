import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle and the HIP library once per session if they are missing."""
    import __graft_entry__ as g
    from oracle import pyoracle
    import resampler_amd
    if not os.path.exists(resampler_amd.LIB_PATH) or not os.path.exists(pyoracle._LIB_PATH):
        g.build()
    yield


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)
