"""Every DISTR_* environment knob the library documents selects a kernel configuration that must give the same
results as the default one: each is run -- in a subprocess, the knobs are read at distr_create -- against a golden of the
reference and against the oracle (tests/gpu_knob_check.py). Knobs that are not worth such a test are not in the product."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KNOBS = [
    {},                                                                     # the default configuration through the same script
    {'DISTR_SAVE_MASKS': '0'},                                              # backward recomputes the decoder forward (k_bwd<BWD_FULL>)
    {'DISTR_STICKY': '0'},                                                  # cluster tiles re-compacted every step (no sticky tail)
    {'DISTR_CLUSTER': '0'},                                                 # single-workgroup 16-ray tiles only
    {'DISTR_CLUSTER': '4'},                                                 # clusters of at most 4 compute units
    {'DISTR_CLUSTER_MIN': '4'},                                             # no pair tiles
    {'DISTR_HYBRID_THRESHOLD': '4096', 'DISTR_TAIL16_THRESHOLD': '4096'},   # t16 == t32: no 32-ray role (rays were skipped)
    {'DISTR_HYBRID_THRESHOLD': '8192', 'DISTR_TAIL16_THRESHOLD': '8128'},   # t16 + t32 just below one round
    {'DISTR_HYBRID_THRESHOLD': '2048', 'DISTR_TAIL16_THRESHOLD': '512'},    # small thresholds: most remainders take a 64-ray round
    {'DISTR_HYBRID_THRESHOLD': '128', 'DISTR_TAIL16_THRESHOLD': '64'},      # the smallest legal values
]


@pytest.fixture(scope='module')
def oracle_cache(tmp_path_factory):
    """The oracle's results of gpu_knob_check.py do not depend on the knob: the first run (the default configuration) stores them
    here, the other nine load them (~15 s of CPU oracle per run otherwise)."""
    return str(tmp_path_factory.mktemp('knobs') / 'oracle.npz')


@pytest.mark.parametrize('knobs', KNOBS, ids=lambda k: ','.join('%s=%s' % kv for kv in k.items()).replace('DISTR_', '') or 'default')
def test_knob_configuration_matches_golden_and_oracle(knobs, oracle_cache):
    env = dict(os.environ)
    for k in list(env):
        if k.startswith('DISTR_'):
            env.pop(k)
    env.update(knobs)
    env['DISTR_KNOB_ORACLE_CACHE'] = oracle_cache
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'gpu_knob_check.py')], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and 'KNOB_OK' in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


@pytest.mark.parametrize('knobs', [{'DISTR_HYBRID_THRESHOLD': '8192', 'DISTR_TAIL16_THRESHOLD': '8192'},     # t16 + t32 reaches a round
                                   {'DISTR_HYBRID_THRESHOLD': '1000'}, {'DISTR_TAIL16_THRESHOLD': '0'}])
def test_illegal_thresholds_are_refused(knobs):
    """distr_create refuses threshold combinations the tile-size split cannot honour (instead of silently skipping rays)."""
    env = dict(os.environ, **knobs)
    code = ("import sys; sys.path.insert(0, %r); from distr import binding\n"
            "try:\n    binding.Context(0)\nexcept binding.DistrError as e:\n    print('REFUSED', e)\n" % os.path.join(ROOT, 'dist-renderer_amd'))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert 'REFUSED' in out.stdout and 'multiples of 64' in out.stdout, (out.stdout, out.stderr[-2000:])
