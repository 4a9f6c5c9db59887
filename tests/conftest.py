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


@pytest.fixture(scope='session')
def engine(fixture_decoder):
    """ONE packed decoder / distr_ctx on cuda:0 for the whole GPU session (fixture F1); every test module used to build its own."""
    import torch
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    from distr import binding, functions
    assert os.path.exists(binding.LIB_PATH), 'libdistr.so missing: run __graft_entry__.build()'
    Ws, bs, _ = fixture_decoder
    return functions.engine_from_weights(Ws, bs, 0)


@pytest.fixture(scope='session')
def orc():
    """The oracle module itself (make_cfg etc.), built. Test infrastructure only."""
    from oracle import oracle
    oracle.build()
    return oracle
