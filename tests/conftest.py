import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout holds no built artefacts (they are git-ignored): build the library, the CLI
    # and the oracle once (what __graft_entry__.build() does) when hipcc is around.  On the GPU box
    # the prebuilt files travel with the snapshot and nothing happens here.
    lib = os.path.join(ROOT, "entreepy_amd", "libentreepy_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "entreepy_amd", "csrc"), "all"], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL)


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
