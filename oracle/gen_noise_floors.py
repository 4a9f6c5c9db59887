"""TEST INFRASTRUCTURE (runs only where /root/reference exists) -- the reference's OWN sensitivity to 1e-7 relative weight noise on
the scenarios of goldens G4, G5, G9 and G11: re-runs the committed generators (oracle/gen_golden.py, gen_golden_losses.py) with a
perturbed decoder and records the residual between the reference's two answers. The GPU tests assert `residual(HIP vs golden) <=
2 x floor` and print both (VERDICT r1 'next' 7). Output: tests/golden/noise_floor_g4_g5_g9_g11.npz

    python oracle/gen_noise_floors.py
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from distr import fixture  # noqa: E402
import gen_golden as gg  # noqa: E402
import gen_golden_losses as gl  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


def main():
    clean = fixture.make_decoder_weights

    def noisy(seed=1234, latent_scale=0.003):
        Ws, bs, latent = clean(seed, latent_scale)
        rsn = np.random.RandomState(99)
        return [(W * (1 + 1e-7 * rsn.standard_normal(W.shape))).astype(np.float32) for W in Ws], bs, latent
    tmp = tempfile.mkdtemp()
    fixture.make_decoder_weights = noisy
    gg.OUT = gl.OUT = tmp
    try:
        gg.golden_g4()
        gg.golden_g5()
        gl.golden_g9()
        gl.golden_g11()
    finally:
        fixture.make_decoder_weights = clean
    L = lambda d, n: dict(np.load(os.path.join(d, n)))
    floor = {}
    a, b = L(GOLDEN, 'g4_render_warp.npz'), L(tmp, 'g4_render_warp.npz')
    floor['g4_g_latent_rel'] = rel(b['g_latent'], a['g_latent'])
    floor['g4_loss_abs'] = abs(float(b['loss_color']) - float(a['loss_color']))
    floor['g4_min_sdf'] = float(np.abs(b['min_sdf1'] - a['min_sdf1']).max())
    floor['g4_color_valid_changed_px'] = int((np.abs(b['color_valid_1'] - a['color_valid_1']).max(-1) > 1e-5).sum())
    a, b = L(GOLDEN, 'g5_adam_single_view.npz'), L(tmp, 'g5_adam_single_view.npz')
    names = ('depth', 'normal', 'mask_gt', 'mask_out', 'l2reg', 'loss', 'grad_norm')
    for i, n in enumerate(names):
        floor['g5_%s_rel' % n] = float((np.abs(b['history'][:, i] - a['history'][:, i]) / np.maximum(np.abs(a['history'][:, i]), 1e-30)).max())
    floor['g5_latent_final_abs'] = float(np.abs(b['latent_final'] - a['latent_final']).max())
    a, b = L(GOLDEN, 'g9_multi_view_round.npz'), L(tmp, 'g9_multi_view_round.npz')
    floor['g9_loss_rel'] = abs(float(b['loss_total']) - float(a['loss_total'])) / abs(float(a['loss_total']))
    for n in ('g_latent', 'g_rot', 'g_scale', 'g_trans'):
        floor['g9_%s_rel' % n] = rel(b[n], a[n])
    a, b = L(GOLDEN, 'g11_decode_sdf_grad.npz'), L(tmp, 'g11_decode_sdf_grad.npz')
    for n in ('sdf_raw', 'g_latent_raw', 'g_latent_clamped'):
        floor['g11_%s_rel' % n] = rel(b[n], a[n])
    for n in ('g_points_raw', 'g_points_clamped'):
        d = np.abs(b[n] - a[n]).max(-1)
        floor['g11_%s_changed_points' % n] = int((d > 1e-4 * np.abs(a[n]).max()).sum())      # points whose ReLU pattern flipped
        floor['g11_%s_rel_median' % n] = float(np.median(d) / np.abs(a[n]).max())
    np.savez_compressed(os.path.join(GOLDEN, 'noise_floor_g4_g5_g9_g11.npz'), **floor)
    for k, v in floor.items():
        print('%-36s %s' % (k, v))


if __name__ == '__main__':
    main()
