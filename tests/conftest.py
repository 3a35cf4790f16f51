import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def res_files():
    out = {}
    for name in ("test.txt", "nice.shakespeare.txt", "a_midsummer_nights_dream.txt"):
        with open(os.path.join(GOLDEN, "res", name), "rb") as f:
            out[name] = f.read()
    return out


@pytest.fixture(scope="session")
def ctx():
    """One et_ctx on cuda:0 running on torch's current stream.  Fails loudly when the
    HIP extension is missing -- there is no fallback to test instead."""
    import torch

    import entreepy_amd as E

    assert torch.cuda.is_available(), "gpu tests need a GPU"
    c = E.Context(0)
    c.use_torch_stream()
    yield c
    c.close()
