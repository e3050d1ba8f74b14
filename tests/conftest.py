import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def maps():
    import helpers
    return helpers.load_reference_maps()


@pytest.fixture(scope="session")
def gpu_ctx():
    # torch bundles its own ROCm runtime; whichever HIP runtime initialises first serves the whole process.  Let it be torch's,
    # as in bench.py, so that tests which later bring up torch.distributed (nccl) in this process still see the GPU.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import botlab_amd
    return botlab_amd.default_context()
