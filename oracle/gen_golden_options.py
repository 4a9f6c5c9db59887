"""TEST INFRASTRUCTURE -- golden G24: the less-travelled constructor / call options of SDFRenderer (core/sdfrenderer/renderer.py:13-59, 943),
each rendered fwd + bwd by the REFERENCE itself on CPU (build container only; shims in oracle/ref_harness.py; no reference source copied):

    python oracle/gen_golden_options.py        # writes tests/golden/g24_renderer_options.npz

Until round 4 these options were compared HIP <-> oracle only (tests/test_gpu_parity.py::test_option_sweep_matches_oracle), i.e. against
the restatement's reading of them. Cases (48 x 40 image, rotated camera, the seeded dense loss of gen_golden.py): identity
transform_matrix (run_multi_realdata.py:96), render(use_transform=False), normalize_normal=False, clamp_dist 0.05, threshold 1e-3,
radius 0.9, march_step_list [2, 4, -1], ray_marching_ratio 1.0 and 2.0, buffer_size 1 and 8, a permuting transform_matrix, img_hw=None
(size from the intrinsic, :31-33), and render_depth's own default marcher.
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
import gen_golden as gg  # noqa: E402

OUT = os.path.join(_HERE, '..', 'tests', 'golden')
H, W = 48, 40
PERM = np.array([[0., 1., 0.], [1., 0., 0.], [0., 0., -1.]])
# name -> (constructor kwargs, render kwargs)
CASES = {
    'identity_transform': (dict(transform_matrix=np.eye(3)), dict()),
    'permuting_transform': (dict(transform_matrix=PERM), dict(ray_marching_type='recursive')),
    'no_use_transform': (dict(), dict(use_transform=False)),
    'unnormalized_normal': (dict(), dict(normalize_normal=False, ray_marching_type='recursive')),
    'clamp_005': (dict(), dict(clamp_dist=0.05)),
    'threshold_1e-3': (dict(threshold=1e-3), dict()),
    'radius_09': (dict(radius=0.9), dict()),
    'coarse_2_4': (dict(march_step_list=[2, 4, -1]), dict()),
    'ratio_10': (dict(ray_marching_ratio=1.0), dict(ray_marching_type='recursive')),
    'ratio_20': (dict(ray_marching_ratio=2.0), dict()),
    'buffer_1': (dict(buffer_size=1), dict()),
    'buffer_8': (dict(buffer_size=8), dict(ray_marching_type='trivial')),
    'd2n_threshold_ratio': (dict(use_depth2normal=True, threshold=2e-4, ray_marching_ratio=1.2), dict()),
}


def run(dec, latent, K, R, T, ckw, rkw, img_hw=(H, W)):
    SDFRenderer = rh.reference_modules()[0]
    kw = dict(march_step=24, buffer_size=3, ray_marching_ratio=1.5, use_gpu=False, use_depth2normal=False)
    kw.update(ckw)
    r = SDFRenderer(dec, K, img_hw=img_hw, **kw)
    h, w = r.get_img_hw()
    lat = torch.from_numpy(latent).clone().requires_grad_(True)
    Rt = torch.from_numpy(R).clone().requires_grad_(True)
    Tt = torch.from_numpy(T).clone().requires_grad_(True)
    depth, normal, mask, mq = r.render(lat, Rt, Tt, **rkw)
    wd, wq, wn = gg.loss_weights(h, w, 5)
    mb = mask.bool()
    L = (depth * torch.from_numpy(wd))[mb].sum() + (mq * torch.from_numpy(wq)).sum() + (normal * torch.from_numpy(wn)).sum()
    if L.requires_grad:
        L.backward()
    return dict(depth=depth.detach().numpy(), normal=normal.detach().numpy(), mask=mask.numpy(), q=mq.detach().numpy(), loss=np.float64(L.item()),
                g_latent=(lat.grad if lat.grad is not None else torch.zeros_like(lat)).numpy(), g_R=(Rt.grad if Rt.grad is not None else torch.zeros_like(Rt)).numpy(),
                g_T=(Tt.grad if Tt.grad is not None else torch.zeros_like(Tt)).numpy(), hw=np.array([h, w]))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    rsn = np.random.RandomState(99)
    dec_n = rh.build_reference_decoder([(Wl * (1 + 1e-7 * rsn.standard_normal(Wl.shape))).astype(np.float32) for Wl in Ws], bs)
    K = np.array([[44.0, 0.0, W / 2.0], [0.0, 46.0, H / 2.0], [0.0, 0.0, 1.0]])          # fx != fy, image size = (2 cy, 2 cx)
    R, T = fixture.make_camera(-35, 18, 1.6, 8)
    out = dict(weights_sha256=fixture.weights_sha256(Ws, bs), latent=latent, K=K, R=R, T=T, H=H, W=W, march_step=24, perm=PERM,
               names=np.array(sorted(CASES) + ['img_hw_none', 'render_depth_default']))
    for name in sorted(CASES):
        ckw, rkw = CASES[name]
        a, b = run(dec, latent, K, R, T, ckw, rkw), run(dec_n, latent, K, R, T, ckw, rkw)
        for k, v in a.items():
            out['%s.%s' % (name, k)] = v
        for k in ('g_latent', 'g_R', 'g_T'):
            out['%s.%s_floor_rel' % (name, k)] = float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max())
        out['%s.flips_floor' % name] = int((a['mask'] != b['mask']).sum())
        # the reference's own floor of the NORMAL image (99th percentile over pixels valid in both renders) and the size of its normal
        # vectors (1 when normalised; 3 |grad f| otherwise): the bar of the normal comparison is derived from these, not hand-tuned
        both = a['mask'].astype(bool) & b['mask'].astype(bool)
        out['%s.normal_p99_floor' % name] = float(np.percentile(np.abs(a['normal'] - b['normal'])[both], 99)) if both.any() else 0.0
        out['%s.normal_scale' % name] = float(np.percentile(np.linalg.norm(a['normal'][a['mask'].astype(bool)], axis=-1), 99)) if a['mask'].any() else 1.0
        print(name, 'valid', int(a['mask'].sum()), 'loss %.4f' % a['loss'], '|g_latent| %.3g' % np.abs(a['g_latent']).max(), flush=True)
    a = run(dec, latent, K, R, T, {}, {}, img_hw=None)                                    # size from the intrinsic (renderer.py:31-33)
    for k, v in a.items():
        out['img_hw_none.%s' % k] = v
    # render_depth with ITS default marcher ('recursive', renderer.py:836) and the gradient entering through Zdepth
    SDFRenderer = rh.reference_modules()[0]
    r = SDFRenderer(dec, K, img_hw=(H, W), march_step=24, buffer_size=2, use_gpu=False)
    lat = torch.from_numpy(latent).clone().requires_grad_(True)
    Rt, Tt = torch.from_numpy(R).clone().requires_grad_(True), torch.from_numpy(T).clone().requires_grad_(True)
    z, m, q = r.render_depth(lat, Rt, Tt)
    gz = np.random.RandomState(1).rand(H * W).astype(np.float32)
    ((z * torch.from_numpy(gz))[m].sum() + 0.5 * q.sum()).backward()
    out.update({'render_depth_default.zdepth': z.detach().numpy(), 'render_depth_default.mask': m.numpy().astype(np.uint8), 'render_depth_default.q': q.detach().numpy(),
                'render_depth_default.gz': gz, 'render_depth_default.g_latent': lat.grad.numpy(), 'render_depth_default.g_R': Rt.grad.numpy(),
                'render_depth_default.g_T': Tt.grad.numpy()})
    np.savez_compressed(os.path.join(OUT, 'g24_renderer_options.npz'), **out)
    print('g24 done')


if __name__ == '__main__':
    main()
