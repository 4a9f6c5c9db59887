"""Diagnostic (not a test): one iteration of the single-view loop over the MULTI-SCALE renderer list of run_single_shape.py:110-113 (full
resolution with buffer_size 1, 1/2 with 3, 1/4 with 5; march_step 100) -- the three scales one after the other (the reference's order)
against the three on a pool of HIP streams (optimize_single_view(streams=...)). Run: python tests/gpu_diag_multiscale.py [size ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
from core.graph.deep_sdf_decoder import Decoder  # noqa: E402
from core.inv_optimizer import optimize_single_view  # noqa: E402
from core.sdfrenderer import SDFRenderer  # noqa: E402
from core.utils.render_utils import downsize_camera_intrinsic  # noqa: E402
from distr import fixture  # noqa: E402


def main():
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    wd = dict(w_depth=10.0, w_normal=5.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
    for size in (int(a) for a in (sys.argv[1:] or ['224', '136'])):
        K = fixture.make_intrinsic(size, size)
        mk = lambda Kx, b_, **kw: SDFRenderer(dec, Kx, march_step=100, buffer_size=b_, threshold=5e-5, use_depth2normal=True, **kw)
        rs = [mk(K, 1, img_hw=(size, size), ray_marching_ratio=1.5), mk(downsize_camera_intrinsic(K, 2), 3), mk(downsize_camera_intrinsic(K, 4), 5)]
        R, T = fixture.make_camera(30.0, 20.0, 1.6, 0.0)
        RT = torch.from_numpy(np.concatenate([R, T[:, None]], 1)).cuda()
        lat_gt = torch.from_numpy(latent + 0.1 * np.random.RandomState(7).standard_normal(latent.shape).astype(np.float32)).cuda()
        with torch.no_grad():
            d, n, m, q = rs[0].render(lat_gt, RT[:, :3], RT[:, 3], no_grad=True)
        gt = {'depth': d.clone(), 'normal': n.clone(), 'silhouette': m.clone()}
        res = {}
        for label, streams, env in (('sequential', 0, {}), ('3 streams', 3, {}), ('3 streams, DISTR_STICKY=0 engines', 3, {'DISTR_STICKY': '0'})):
            if env:
                continue          # (the knob is read at distr_create: run the script again with DISTR_STICKY=0 in the environment)
            lat = torch.from_numpy(latent).cuda().requires_grad_(True)
            opt = torch.optim.Adam([lat], lr=1e-3)
            optimize_single_view(rs, None, opt, lat, RT, gt, wd, num_iters=3, silent=True, streams=streams)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters = 12
            optimize_single_view(rs, None, opt, lat, RT, gt, wd, num_iters=iters, silent=True, streams=streams)
            torch.cuda.synchronize()
            res[label] = (1e3 * (time.perf_counter() - t0) / iters, lat.detach().cpu().numpy().copy())
        a, b = res['sequential'], res['3 streams']
        print('multi-scale %d / %d / %d, 100 steps, sticky=%s: sequential %.2f ms/iter, 3 streams %.2f ms/iter; final shape codes identical: %s'
              % (size, size // 2, size // 4, os.environ.get('DISTR_STICKY', '1'), a[0], b[0], np.array_equal(a[1], b[1])))


if __name__ == '__main__':
    main()
