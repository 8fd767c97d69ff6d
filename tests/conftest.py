import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


DETERMINISTIC_ENV = {'SALT_BN_FIN': '0', 'SALT_SE_SHARDS': '0'}


@pytest.fixture
def deterministic_sums(monkeypatch):
    """Bit-for-bit comparisons of two runs need sums whose ORDER is fixed.  The default step (SALT_BN_FIN=2 / SALT_SE_SHARDS=1) adds the
    BatchNorm and SE statistics with fp64 atomics - reproducible in practice, not by construction; SALT_BN_FIN=0 /
    SALT_SE_SHARDS=0 selects the per-tile partials + fixed-order finalize protocol, which is (README: deterministic training)."""
    for k, v in DETERMINISTIC_ENV.items():
        monkeypatch.setenv(k, v)
    return dict(DETERMINISTIC_ENV)
