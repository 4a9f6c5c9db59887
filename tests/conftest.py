import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'dist-renderer_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run on the GPU box with -m gpu)')


@pytest.fixture(scope='session')
def fixture_decoder():
    from distr import fixture
    return fixture.make_decoder_weights()


@pytest.fixture(scope='session')
def cpu_oracle(fixture_decoder):
    """The CPU restatement (oracle/). Test infrastructure only."""
    from oracle import oracle as orc
    orc.build()
    Ws, bs, _ = fixture_decoder
    return orc.Oracle(Ws, bs)
