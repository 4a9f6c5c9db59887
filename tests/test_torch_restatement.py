"""Pins the PyTorch-CPU restatement (oracle/torch_restatement.py: BASELINE.md section 3 baseline (2), the 'PyTorch CPU reference
path' of config C1) against the reference's own goldens and against the C++ oracle -- two independent restatements of the same
algorithm agreeing with the reference and with each other."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import torch_restatement as tr
from distr import fixture
import helpers


def _golden(name):
    g = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: g[k] for k in g.files}


@pytest.mark.parametrize('name', ['g1_c1_pyramid_recursive_d2n.npz', 'g1_c1_recursive_agn.npz', 'g1_c1_trivial_d2n.npz',
                                  'g1_c1_pyramid_recursive_agn.npz', 'g1b_odd_pyramid.npz'])
def test_torch_restatement_matches_reference_golden(name):
    g = _golden(name)
    Ws, bs, _ = fixture.make_decoder_weights(int(g['fixture_seed']))
    H, W = int(g['H']), int(g['W'])
    out = tr.render_fwd_bwd(Ws, bs, g['latent'], H, W, g['K'], g['R'], g['T'], helpers.loss_weights(H, W, int(g['loss_seed'])),
                            marcher=str(g['marcher']), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']),
                            ratio=float(g['ratio']), use_depth2normal=bool(g['use_depth2normal']), threads=4)
    m, rm = out['mask'].astype(bool), g['mask'].astype(bool)
    assert int((m != rm).sum()) <= max(1, int(0.001 * m.size))
    both = m & rm
    assert np.abs(out['depth'] - g['depth'])[both].max() <= 1e-4
    assert np.abs(out['zdepth'].reshape(H, W) - g['zdepth'].reshape(H, W))[both].max() <= 1e-4
    assert np.abs(out['min_sdf'] - g['min_abs_query']).max() <= 1e-4
    dn = np.abs(out['normal'] - g['normal'])[both]
    fx = float(g['K'][0, 0])
    assert np.percentile(dn, 99) <= (max(1e-4, 1e-5 * fx) if bool(g['use_depth2normal']) else 1e-4)
    for k in ('g_latent', 'g_R', 'g_T'):
        rel = np.abs(out[k].reshape(-1) - g[k].reshape(-1)).max() / np.abs(g[k]).max()
        assert rel <= 2e-3, (k, rel)


def test_torch_restatement_matches_cpp_oracle(cpu_oracle, fixture_decoder):
    """The two restatements against each other on a case no golden covers (rotated camera, ragged size, ratio 2)."""
    from oracle import oracle as orc
    Ws, bs, latent = fixture_decoder
    H, W = 37, 53
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(70, -25, 1.5, 15)
    kw = dict(march_step=24, buffer_size=2, ratio=2.0, use_depth2normal=True)
    a = tr.render_fwd_bwd(Ws, bs, latent, H, W, K, R, T, helpers.loss_weights(H, W, 5), marcher='pyramid_recursive', threads=4, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, marcher='pyramid_recursive', **kw)
    assert np.array_equal(a['mask'].reshape(-1), b['mask'].reshape(-1))
    both = a['mask'].astype(bool)
    # (different summation order in the decoder: GEMM vs k-ordered fmaf chains -> the reference's own noise floor)
    assert np.abs(a["depth"] - b["depth"])[both].max() <= 1e-4 and np.abs(a["min_sdf"].reshape(-1) - b["min_sdf"].reshape(-1)).max() <= 1e-4
    assert abs(int(a['num_evals']) - int(b['num_evals'])) <= (H * W) * (kw['buffer_size'] + 1) + 3     # + the re-evaluations the tape needs
    rel = np.abs(a['g_latent'].reshape(-1) - b['g_latent'].reshape(-1)).max() / np.abs(b['g_latent']).max()
    assert rel <= 1e-2, rel       # depth2normal amplifies the decoder noise (reference's own floor for such configs: 3e-3, tests/golden/noise_floor_c2_pyramid_d2n.npz)
