"""Run by tests/test_gpu_knobs.py in a subprocess with one DISTR_* environment knob set (the knobs are read at distr_create):
one golden of the reference (G1 pyramid_recursive + depth2normal, 64 x 64) and two oracle comparisons (a 20-step C1 render and
a long march at a small size that spends most steps on cluster / 16-ray tiles) through whatever kernel configuration the
knob selects. Prints KNOB_OK <residuals> on success. The oracle's side of the three comparisons does not depend on the knob: with
DISTR_KNOB_ORACLE_CACHE=<file.npz> the first run stores it and the later ones (one subprocess per knob) load it, which is most of
the wall time of those runs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import helpers  # noqa: E402
from distr import fixture, functions  # noqa: E402
from oracle import oracle as orc  # noqa: E402   (the checker)


def main():
    Ws, bs, latent = fixture.make_decoder_weights()
    eng = functions.engine_from_weights(Ws, bs, 0)
    O = orc.Oracle(Ws, bs)
    res = {}
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g1_c1_pyramid_recursive_d2n.npz')))
    H, W = int(g['H']), int(g['W'])
    a = helpers.hip_render(eng, H, W, g['K'], g['R'], g['T'], g['latent'], seed=int(g['loss_seed']), march_step=int(g['march_step']),
                           buffer_size=int(g['buffer_size']), ratio=float(g['ratio']), marcher=str(g['marcher']),
                           use_depth2normal=bool(g['use_depth2normal']))
    b = dict(mask=g['mask'], depth=g['depth'], zdepth=g['zdepth'], min_sdf=g['min_abs_query'], normal=g['normal'],
             g_latent=g['g_latent'], g_R=g['g_R'], g_T=g['g_T'])
    res['golden'] = helpers.compare(a, b, H, W, tol_depth=1e-4, tol_grad=2e-3, normal_p99=max(1e-4, 1e-5 * float(g['K'][0, 0])))
    KEYS = ('mask', 'depth', 'zdepth', 'min_sdf', 'normal', 'g_latent', 'g_R', 'g_T')
    cache = os.environ.get('DISTR_KNOB_ORACLE_CACHE')
    cached, fresh = {}, {}
    if cache and os.path.exists(cache):
        z = np.load(cache)
        for key in z.files:
            nm, k = key.split('.', 1)
            cached.setdefault(nm, {})[k] = z[key]
    for name, (H, W, kw, cam) in {
            'c1': (64, 64, dict(march_step=20, buffer_size=3, marcher='recursive', use_depth2normal=False), (30, 20, 1.6, 10)),
            'tail': (150, 130, dict(march_step=70, buffer_size=2, marcher='pyramid_recursive', use_depth2normal=True), (-40, 25, 1.6, 0)),
            'mid': (300, 300, dict(march_step=40, buffer_size=3, marcher='recursive', use_depth2normal=True), (10, 15, 1.6, 0))}.items():
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(*cam)
        a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
        if name in cached:
            b = cached[name]
        else:
            b = helpers.oracle_render(O, orc, H, W, K, R, T, latent, **kw)
            fresh[name] = {k: b[k] for k in KEYS}
        r = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=2e-4, normal_p99=1e-5, max_flip_frac=0.0)
        assert r['flips'] == 0, r
        res[name] = r
    if cache and fresh:
        tmp = cache + '.%d.tmp.npz' % os.getpid()
        allr = dict(cached)
        allr.update(fresh)
        np.savez(tmp, **{'%s.%s' % (nm, k): v for nm, d in allr.items() for k, v in d.items()})
        os.replace(tmp, cache)
    print('KNOB_OK', {k: {kk: float('%.3g' % vv) for kk, vv in v.items()} for k, v in res.items()})


if __name__ == '__main__':
    main()
