import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture(params=['clustered', 'batch'])
def forward_path(request):
    """Runs a test twice: small batches on the clustered forward kernels (the default) and with them
    switched off, so that the batch-partitioned kernels keep their small-shape coverage."""
    from paccmann_generator_amd import ops
    old = ops.USE_CLUSTER
    ops.USE_CLUSTER = request.param == 'clustered'
    yield request.param
    ops.USE_CLUSTER = old
