"""TEST INFRASTRUCTURE -- golden G29: the early break of the recursive march below buffer_size steps, rendered fwd + bwd by the REFERENCE itself
on CPU (build container only; shims in oracle/ref_harness.py; no reference source copied):

    python oracle/gen_golden_early_break.py        # writes tests/golden/g29_early_break.npz

ray_marching_recursive (core/sdfrenderer/renderer.py:562-567): when no ray is unfinished after L < buffer_size steps the lists are padded to
buffer_size rows by repeating step L-1's rows; get_index_from_sdf_list (:304-318) then selects copies of a ray's last row and
get_sample_on_marching_zdepth_along_ray (:392-420) evaluates each copy again -- the same values, (1 + copies) x the row's gradient. A camera
inside the unit sphere next to the surface with exact sphere tracing (ratio 1) ends the march after 4 steps (1 behind the pyramid's coarse levels).
Cases: 'recursive' with buffer_size 7 (finite-difference normals) and 8 (autograd normals), 'pyramid_recursive' with 8, and buffer_size 3 as the
control (the march outlasts the buffer). Same layout and floors as G24 / G28 (gen_golden_options.py, gen_golden_pyramid2.py).
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
H, W = 55, 79
BASE = dict(march_step=30, ray_marching_ratio=1.0, threshold=1.5e-3, radius=1.2)
CASES = {
    'recursive_bs7_d2n': (dict(buffer_size=7, use_depth2normal=True), dict(ray_marching_type='recursive', clamp_dist=0.2)),
    'recursive_bs8': (dict(buffer_size=8), dict(ray_marching_type='recursive', clamp_dist=0.2)),
    'pyramid_bs8_d2n': (dict(buffer_size=8, use_depth2normal=True), dict(ray_marching_type='pyramid_recursive', clamp_dist=0.2)),
    'recursive_bs3_d2n': (dict(buffer_size=3, use_depth2normal=True), dict(ray_marching_type='recursive', clamp_dist=0.2)),
}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    K = np.array(fixture.make_intrinsic(H, W), dtype=np.float64)
    R, T = fixture.make_camera(-73.8, -7.6, 0.475, 26.75)
    Ws, bs, latent = fixture.make_decoder_weights()
    out = dict(K=K, R=R, T=T, H=H, W=W, latent=latent, weights_sha256=fixture.weights_sha256(Ws, bs), names=np.array(sorted(CASES)),
               **{k: np.float64(v) for k, v in BASE.items()})
    rsn = np.random.RandomState(99)
    dec = rh.build_reference_decoder(Ws, bs)
    dec_ns = [rh.build_reference_decoder([(Wl * (1 + 1e-7 * rsn.standard_normal(Wl.shape))).astype(np.float32) for Wl in Ws], bs) for _ in range(3)]
    for name in sorted(CASES):
        ckw, rkw = CASES[name]
        ckw = dict(BASE, **ckw)
        a = go.run(dec, latent, K, R, T, ckw, rkw, img_hw=(H, W))
        for k, v in a.items():
            out['%s.%s' % (name, k)] = v
        fl = dict(g_latent=0.0, g_R=0.0, g_T=0.0, flips=0, normal=0.0)
        unstable = np.zeros((H, W), bool)      # pixels whose depth / min-sdf the reference itself moves by more than 1e-5 under 1e-7 weight noise
        for dn in dec_ns:                      # (a discrete event of the march: a coarse ray stopping one step earlier moves its 2 x 2 children together)
            b = go.run(dn, latent, K, R, T, ckw, rkw, img_hw=(H, W))
            bothv = a['mask'].astype(bool) & b['mask'].astype(bool)
            unstable |= (np.abs(a['depth'] - b['depth']) > 1e-5) & bothv
            unstable |= np.abs(a['q'].reshape(H, W) - b['q'].reshape(H, W)) > 1e-5
            for k in ('g_latent', 'g_R', 'g_T'):
                fl[k] = max(fl[k], float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max()))
            fl['flips'] = max(fl['flips'], int((a['mask'] != b['mask']).sum()))
            both = a['mask'].astype(bool) & b['mask'].astype(bool)
            if both.any():
                fl['normal'] = max(fl['normal'], float(np.percentile(np.abs(a['normal'] - b['normal'])[both], 99)))
        for k in ('g_latent', 'g_R', 'g_T'):
            out['%s.%s_floor_rel' % (name, k)] = fl[k]
        out['%s.unstable' % name] = unstable
        out['%s.flips_floor' % name] = fl['flips']
        out['%s.normal_p99_floor' % name] = fl['normal']
        out['%s.normal_scale' % name] = float(np.percentile(np.linalg.norm(a['normal'][a['mask'].astype(bool)], axis=-1), 99)) if a['mask'].any() else 1.0
        print(name, 'unstable px', int(unstable.sum()), 'valid', int(a['mask'].sum()), 'loss %.4f' % a['loss'], '|g_latent| %.3g' % np.abs(a['g_latent']).max(),
              'floors', {k: '%.1e' % out['%s.%s_floor_rel' % (name, k)] for k in ('g_latent', 'g_R', 'g_T')}, flush=True)
    np.savez_compressed(os.path.join(OUT, 'g29_early_break.npz'), **out)
    print('g29 done')


if __name__ == '__main__':
    main()
