"""TEST INFRASTRUCTURE -- golden G28: pyramids other than the default three levels, rendered fwd + bwd by the REFERENCE itself on CPU (build
container only; shims in oracle/ref_harness.py; no reference source copied):

    python oracle/gen_golden_pyramid2.py        # writes tests/golden/g28_two_level_pyramid.npz

ray_marching_pyramid_recursive (core/sdfrenderer/renderer.py:713-805) builds one level per scale_list entry. Cases: scale_list=[2, 1] with
march_step_list [3, -1] (autograd normals), [6, -1] (finite-difference normals) and an explicit last entry [2, 20] (the full-resolution
step count is then independent of march_step, :724-725), on an odd-sized image (45 x 37: the coarse grids are ceil'ed, :612) and
on fixture F2's decoder; and the general form: four levels [8, 4, 2, 1], ratios other than 2 ([3, 1], [6, 2, 1], [4, 1]).
Same layout and floors as G24 (gen_golden_options.py).
"""
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, '..', 'dist-renderer_amd'))
sys.path.insert(0, _HERE)
from distr import fixture  # noqa: E402
import ref_harness as rh  # noqa: E402
import gen_golden_options as go  # noqa: E402

OUT = os.path.join(_HERE, '..', 'tests', 'golden')
H, W = 45, 37
# name -> (fixture, constructor kwargs, render kwargs)
CASES = {
    'two_level_3': ('f1', dict(scale_list=[2, 1], march_step_list=[3, -1]), dict()),
    'two_level_6_d2n': ('f1', dict(scale_list=[2, 1], march_step_list=[6, -1], use_depth2normal=True), dict()),
    'two_level_explicit_2_20': ('f1', dict(scale_list=[2, 1], march_step_list=[2, 20], march_step=50), dict()),
    'two_level_3_f2': ('f2', dict(scale_list=[2, 1], march_step_list=[3, -1]), dict()),
    # general pyramids: four levels; a ratio of 3; ratios 3 then 2; ratio 4 in one go; four levels on F2 with finite-difference normals
    'four_level_2_2_2': ('f1', dict(scale_list=[8, 4, 2, 1], march_step_list=[2, 2, 2, -1]), dict()),
    'ratio_3': ('f1', dict(scale_list=[3, 1], march_step_list=[3, -1]), dict()),
    'ratio_3_2': ('f1', dict(scale_list=[6, 2, 1], march_step_list=[2, 3, -1]), dict()),
    'ratio_4': ('f1', dict(scale_list=[4, 1], march_step_list=[4, -1], use_depth2normal=True), dict()),
    'four_level_f2_d2n': ('f2', dict(scale_list=[8, 4, 2, 1], march_step_list=[3, 1, 2, -1], use_depth2normal=True), dict()),
}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    K = np.array([[41.0, 0.0, W / 2.0], [0.0, 43.0, H / 2.0], [0.0, 0.0, 1.0]])
    R, T = fixture.make_camera(25, -14, 1.55, 5)
    out = dict(K=K, R=R, T=T, H=H, W=W, march_step=24, names=np.array(sorted(CASES)))
    rsn = np.random.RandomState(99)
    decs = {}
    for fx in ('f1', 'f2'):
        Ws, bs, latent = fixture.make_decoder_weights() if fx == 'f1' else fixture.load_fixture_f2()
        wn = dict() if fx == 'f1' else dict(weight_norm=False)
        # noise floors: the reference against itself under 1e-7 relative weight noise, MAXIMUM over three draws (a residual is dominated by
        # discrete events -- a ray stopping one step earlier -- which a single draw may or may not contain: gen_golden_f2.py)
        decs[fx] = (rh.build_reference_decoder(Ws, bs, **wn),
                    [rh.build_reference_decoder([(Wl * (1 + 1e-7 * rsn.standard_normal(Wl.shape))).astype(np.float32) for Wl in Ws], bs, **wn)
                     for _ in range(3)], latent)
        out['%s.weights_sha256' % fx] = fixture.weights_sha256(Ws, bs)
        out['%s.latent' % fx] = latent
    for name in sorted(CASES):
        fx, ckw, rkw = CASES[name]
        dec, dec_ns, latent = decs[fx]
        a = go.run(dec, latent, K, R, T, ckw, rkw, img_hw=(H, W))
        for k, v in a.items():
            out['%s.%s' % (name, k)] = v
        fl = dict(g_latent=0.0, g_R=0.0, g_T=0.0, flips=0, normal=0.0)
        for dn in dec_ns:
            b = go.run(dn, latent, K, R, T, ckw, rkw, img_hw=(H, W))
            for k in ('g_latent', 'g_R', 'g_T'):
                fl[k] = max(fl[k], float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max()))
            fl['flips'] = max(fl['flips'], int((a['mask'] != b['mask']).sum()))
            both = a['mask'].astype(bool) & b['mask'].astype(bool)
            if both.any():
                fl['normal'] = max(fl['normal'], float(np.percentile(np.abs(a['normal'] - b['normal'])[both], 99)))
        for k in ('g_latent', 'g_R', 'g_T'):
            out['%s.%s_floor_rel' % (name, k)] = fl[k]
        out['%s.flips_floor' % name] = fl['flips']
        out['%s.normal_p99_floor' % name] = fl['normal']
        out['%s.normal_scale' % name] = float(np.percentile(np.linalg.norm(a['normal'][a['mask'].astype(bool)], axis=-1), 99)) if a['mask'].any() else 1.0
        print(name, 'valid', int(a['mask'].sum()), 'loss %.4f' % a['loss'], '|g_latent| %.3g' % np.abs(a['g_latent']).max(),
              'floors', {k: '%.1e' % out['%s.%s_floor_rel' % (name, k)] for k in ('g_latent', 'g_R', 'g_T')}, flush=True)
    np.savez_compressed(os.path.join(OUT, 'g28_two_level_pyramid.npz'), **out)
    print('g28 done')


if __name__ == '__main__':
    main()
