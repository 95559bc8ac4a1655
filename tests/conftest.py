import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, f"{name}.pt"), weights_only=True)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _library_state_does_not_leak(request):
    """The 16-bit operand format and the loss scale are process-wide library state (clt_set_loss_scaling): a GPU test that ran
    fp16 steps leaves it set, and the next test's kernels (an Adam that unscales by 1 / 65536) would inherit it.  Back to the
    default precision and the default state after every GPU test."""
    yield
    if "gpu" in request.keywords and torch.cuda.is_available():
        import crosslayer_transcoder_amd as cta
        from crosslayer_transcoder_amd import ops
        cta.set_precision("32")
        ops.set_loss_scaling(False)
