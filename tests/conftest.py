import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # PyTorch (device memory + torch.distributed in these tests) is loaded BEFORE the first HIP call of
    # libpandrs_hip.so, the order bench.py uses: its import maps several GB of code objects and, once
    # on a fresh GPU box, stalled for minutes when it came in lazily after the runtime was already busy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)
