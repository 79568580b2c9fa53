import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file
    return load_file(os.path.join(ROOT, "tests", "golden", "vtp_tiny.safetensors"))


@pytest.fixture(scope="session")
def golden_sd(golden):
    return {k[3:]: v for k, v in golden.items() if k.startswith("sd.")}


TINY_HEADS = dict(vis=2, dec=2, txt=2)
