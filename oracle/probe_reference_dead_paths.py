"""TEST INFRASTRUCTURE (runs only where /root/reference exists): which optional paths of the reference's renderer run at all?
Calls SDFRenderer.render on CPU through oracle/ref_harness.py for every sample_index_type, for num_forward_sampling > 0 and for
SDFRenderer_warp.render_warp_gt, and prints what happens. Result recorded in DESIGN.md section 5 ("not reproduced"):

    sample_index_type 'min' / 'max_neg' / 'last'   RuntimeError (shape mismatch (N) vs (k) in get_min_sdf_sample, renderer.py:304-341, 382-390)
    sample_index_type 'last_valid'                  IndexError  (renderer.py:331)
    num_forward_sampling = 1, 3                     runs (5-tuple, inside_samples (h, w, k))  -> mirrored (SDFRenderer.forward_sampling)
    render_warp_gt                                  TypeError: self.device is not callable (renderer_warp.py:228) -> dead code

    python oracle/probe_reference_dead_paths.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np   # noqa: E402
import torch         # noqa: E402
import ref_harness as rh   # noqa: E402
from distr import fixture  # noqa: E402


def main():
    rh.install_shims()
    mods = rh.reference_modules()
    SDFRenderer, SDFRenderer_warp = mods[0], mods[1]
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    H = W = 32
    K = fixture.make_intrinsic(H, W)
    R, T = (torch.from_numpy(a) for a in fixture.make_camera(30, 20, 1.6, 10))
    r = SDFRenderer(dec, K, img_hw=(H, W), march_step=20, buffer_size=3, use_gpu=False)
    for it in ('min_abs', 'min', 'max_neg', 'last_valid', 'last'):
        for marcher in ('recursive', 'pyramid_recursive'):
            lat = torch.from_numpy(latent).clone().requires_grad_(True)
            try:
                d, n, m, q = r.render(lat, R, T, sample_index_type=it, ray_marching_type=marcher)
                (d[m.bool()].sum() + q.sum()).backward()
                print('sample_index_type=%-10s %-18s runs: %d valid px' % (it, marcher, int(m.sum())))
            except Exception as e:   # noqa: BLE001
                print('sample_index_type=%-10s %-18s %s: %s' % (it, marcher, type(e).__name__, str(e)[:110]))
    for k in (1, 3):
        out = r.render(torch.from_numpy(latent), R, T, num_forward_sampling=k)
        print('num_forward_sampling=%d runs: %d outputs, inside_samples %s' % (k, len(out), tuple(out[-1].shape)))
    rw = SDFRenderer_warp(dec, K, img_hw=(H, W), march_step=20, buffer_size=1, use_gpu=False)
    rw.device = torch.device('cpu')
    try:
        z = torch.zeros(H, W)
        rw.render_warp_gt(torch.from_numpy(latent), R, T, R, T, z, z, torch.zeros(3, H * W), torch.zeros(3, H * W), torch.ones(H * W, dtype=torch.bool),
                          torch.ones(H * W, dtype=torch.bool), torch.zeros(H, W, 3), torch.zeros(H, W, 3))
        print('render_warp_gt runs')
    except Exception as e:   # noqa: BLE001
        print('render_warp_gt %s: %s' % (type(e).__name__, str(e)[:110]))


if __name__ == '__main__':
    main()
