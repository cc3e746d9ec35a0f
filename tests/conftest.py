import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def small_conf():
    """Reduced-depth configuration (same widths, same kernels) that the CPU oracle finishes in seconds."""
    from oracle import np_model
    conf = dict(np_model.MODEL_CONFIG)
    conf["depths"] = [1, 1, 1, 1, 1, 2, 1]
    conf["num_transformer_layers"] = 2
    return conf
