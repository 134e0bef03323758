import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("LANDIFF_SKIP_INIT", "1")      # importing `landiff` would otherwise look for / download the released checkpoints
os.environ.setdefault("LD_TUNING", "1")              # the library re-reads its LD_* knobs on every call (tests alternate kernel forms in one process)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A test that hangs (a wedged GPU queue, a lost rendezvous) must fail, not stall the whole run: with pytest-timeout present
    # and no --timeout given, every test gets 30 minutes (the slowest one, the full-size VAE against the fp32 oracle on the host cores, takes ~4 on an idle host).
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 1800.0


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible (these tests must not silently skip)")
    return torch.device("cuda:0")
