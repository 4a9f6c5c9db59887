"""Diagnostic (not a test): one round of the multi-view optimisation (optimize_multi.py:50-79 setting of
run_multi_pmodata.py:92: 8 view pairs, 137x137... here 128x128, march_step 100, buffer_size 1, 'recursive' marcher) --
ms per round as a function of the number of HIP streams the pairs are issued on. Run: python tests/gpu_diag_multiview.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
sys.path.insert(0, ROOT)

from core.graph.deep_sdf_decoder import Decoder  # noqa: E402
from core.inv_optimizer import multi_view_round  # noqa: E402
from core.inv_optimizer.optimize_multi import _StreamPool, pair_indices  # noqa: E402
from core.sdfrenderer import SDFRenderer_warp  # noqa: E402
from distr import fixture  # noqa: E402
from oracle.gen_synth import procedural_images  # noqa: E402  (synthetic images only)


class Cam(object):
    def __init__(self, R, T):
        self.extrinsic = np.concatenate([R, T[:, None]], 1).astype(np.float32)


def main():
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs))
                         for n, a in (('weight', W), ('bias', b))})
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 137
    K = fixture.make_intrinsic(size, size)
    r = SDFRenderer_warp(dec.cuda(), K, img_hw=(size, size), march_step=100, buffer_size=1)
    n_img = 24
    cams = [Cam(*fixture.make_camera(15.0 * i, 20.0, 1.6, 0.0)) for i in range(n_img)]
    a, b = procedural_images(size, size)
    imgs = [torch.from_numpy(np.roll(a if i % 2 else b, 3 * i, axis=1).copy()).cuda() for i in range(n_img)]
    lat = torch.from_numpy(latent).cuda().requires_grad_(True)
    opt = torch.optim.Adam([lat], lr=1e-3)
    pairs = [pair_indices(0, i, n_img / 8, 1, n_img) for i in range(8)]
    w = {'color': 5.0, 'l2reg': 1.0}
    for ns in (-1, 0, 2, 4, 8):                 # -1: the batched round (all 16 depth renders in one launch sequence)
        pool = _StreamPool(max(ns, 0), lat.device)
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            opt.zero_grad()
            total, _ = multi_view_round(r, lat, imgs, cams, pairs, w, pool=pool, batched=(ns < 0))
            total.backward()
            opt.step()
        torch.cuda.synchronize()
        print('%dx%d, 8 view pairs, %s: %.2f ms per round (fwd+bwd+Adam), loss %.5f'
              % (size, size, 'batched' if ns < 0 else 'streams=%d' % ns, (time.perf_counter() - t0) * 1e3 / 5, float(total)))


if __name__ == '__main__':
    main()
