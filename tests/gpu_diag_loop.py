"""Diagnostic (not a test): where one iteration of the single-view optimisation loop (optimize_single.py:50-84 setting:
march_step 100, buffer_size 3, pyramid_recursive) spends its time on the GPU box -- render forward / losses / backward /
Adam -- at the image sizes the reference's drivers use. Run: python tests/gpu_diag_loop.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))

from core.graph.deep_sdf_decoder import Decoder  # noqa: E402
from core.inv_optimizer.loss_single import compute_all_loss  # noqa: E402
from core.sdfrenderer import SDFRenderer  # noqa: E402
from distr import fixture  # noqa: E402


FUSED = os.environ.get('DIAG_FUSED', '1') == '1'   # 0: the PyTorch restatement of the losses (core/utils/loss_utils.py)


def main():
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = Decoder(256, [512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs))
                         for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    wd = dict(w_depth=10.0, w_normal=5.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
    for size in (int(a) for a in (sys.argv[1:] or ['137', '224', '512'])):
        for d2n in (True,):
            K = fixture.make_intrinsic(size, size)
            r = SDFRenderer(dec, K, img_hw=(size, size), march_step=100, buffer_size=3, use_depth2normal=d2n)
            R, T = fixture.make_camera(30.0, 20.0, 1.6, 0.0)
            RT = torch.from_numpy(np.concatenate([R, T[:, None]], 1)).cuda()
            lat_gt = torch.from_numpy(latent + 0.1 * np.random.RandomState(7).standard_normal(latent.shape).astype(np.float32)).cuda()
            with torch.no_grad():
                d, n, m, q = r.render(lat_gt, RT[:, :3], RT[:, 3], no_grad=True)
            gt = {'depth': d.clone(), 'normal': n.clone(), 'silhouette': m.clone()}
            lat = torch.from_numpy(latent).cuda().requires_grad_(True)
            opt = torch.optim.Adam([lat], lr=1e-3)
            sync = torch.cuda.synchronize
            acc = np.zeros(5)
            iters = 12
            for it in range(iters + 3):
                sync(); t0 = time.perf_counter()
                opt.zero_grad()
                out = r.render(lat, RT[:, :3], RT[:, 3])
                sync(); t1 = time.perf_counter()
                # losses on the already rendered images (same code path as compute_all_loss after its render call)
                from core.utils import loss_utils as LU
                from distr import functions
                depth, normal, mask, min_sdf = out
                if FUSED:
                    tm = functions.single_view_losses(r._engine, depth, normal, mask, min_sdf, gt['depth'], gt['normal'], gt['silhouette'],
                                                      r.get_threshold())
                    lg, lo, ld, ln = tm[0], tm[1], tm[2], tm[3]
                else:
                    lg, lo, _ = LU.compute_loss_mask(min_sdf, mask, gt['silhouette'], threshold=r.get_threshold())
                    ld, _ = LU.compute_loss_depth(depth, mask, gt['depth'], gt['silhouette'])
                    ln, _ = LU.compute_loss_normal(normal, mask, gt['normal'], gt['silhouette'])
                loss = 10.0 * ld + 5.0 * ln + lg + lo + lat.pow(2).mean()
                sync(); t2 = time.perf_counter()
                loss.backward()
                sync(); t3 = time.perf_counter()
                opt.step()
                sync(); t4 = time.perf_counter()
                if it >= 3:
                    acc += np.array([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0])
            acc *= 1e3 / iters
            # free-running loop (no phase syncs): what the user sees
            sync(); t0 = time.perf_counter()
            for it in range(iters):
                opt.zero_grad()
                pack, _ = compute_all_loss(r, lat, RT, gt, threshold=r.get_threshold())
                loss = 10.0 * pack['depth'] + 5.0 * pack['normal'] + pack['mask_gt'] + pack['mask_out'] + pack['l2reg']
                loss.backward()
                opt.step()
            sync(); free = (time.perf_counter() - t0) * 1e3 / iters
            print('fused=%d %4dx%-4d d2n=%d valid=%6d | render %.2f  losses %.2f  backward %.2f  adam %.2f  sum %.2f ms | free-running %.2f ms/iter'
                  % (FUSED, size, size, d2n, int(m.sum()), acc[0], acc[1], acc[2], acc[3], acc[4], free))


if __name__ == '__main__':
    main()
