import os
import sys

import pytest

os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')   # before HIP initialises; see lsnet_amd/__init__.py
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)')


@pytest.fixture
def cpu_oracle_backend():
    """Registers the CPU oracle as the 'cpu' backend of lsnet_amd.ops for the duration of a test.
    (Test infrastructure: the product itself never registers a CPU backend.)"""
    from lsnet_amd.ops import register_backend, unregister_backend
    from tests.oracle_backend import OracleBackend
    register_backend('cpu', OracleBackend())
    yield
    unregister_backend('cpu')
