import os
import sys

# Idle OpenMP threads must SLEEP, not spin: the GPU suite renders its full-size oracle images in a background thread (tests/helpers.py) while
# the foreground tests run oracle renders and multi-process loops of their own -- with the default active wait policy two teams of 128
# spinning threads thrash each other (round 6, measured: 128 x 128 oracle renders 1 s -> 16 s, the suite slower than without the thread).
# Read by libgomp when it is first loaded, i.e. before anything here imports torch or the oracle.
os.environ.setdefault('OMP_WAIT_POLICY', 'passive')
os.environ.setdefault('GOMP_SPINCOUNT', '0')

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'dist-renderer_amd')
for p in (PKG, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run on the GPU box with -m gpu)')
    config.addinivalue_line('markers', 'big_oracle(key): needs a full-size CPU-oracle render (tests/helpers.py::BIG_ORACLE); such tests run last, '
                                       'their oracle renders are made by a background thread from the start of the session')


def pytest_collection_modifyitems(config, items):
    # tests that wait for a full-size oracle render go to the END of the session (stable otherwise): the render thread then has the
    # whole session's head start and the GPU is never idle behind 20-85 s of host work
    big = [it for it in items if it.get_closest_marker('big_oracle')]
    if big:
        rest = [it for it in items if not it.get_closest_marker('big_oracle')]
        items[:] = rest + big


@pytest.fixture(scope='session', autouse=True)
def _big_oracle_prefetch(request):
    keys = []
    for it in request.session.items:
        m = it.get_closest_marker('big_oracle')
        if m and m.args:
            k = m.args[0]
            if k == 'param':                    # the key is built from the test's parameters
                k = it.callspec.params.get('big_key')
            if k and k not in keys:
                keys.append(k)
    if keys and os.environ.get('DISTR_NO_ORACLE_PREFETCH') != '1':
        import helpers
        from distr import fixture
        from oracle import oracle as orc
        orc.build()
        Ws, bs, latent = fixture.make_decoder_weights()
        helpers.start_big_oracle(orc.Oracle(Ws, bs), orc, latent, keys)
    yield


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
