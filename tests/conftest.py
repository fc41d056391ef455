import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def reference_root():
    """The upstream reference tree; only exists in the authoring container, never on the GPU box."""
    if not os.path.isdir(REFERENCE):
        pytest.skip("reference tree not present (GPU box)")
    return REFERENCE
