"""TEST INFRASTRUCTURE -- golden vectors for the fused loss kernels, produced by running the
REFERENCE's own functions on CPU on synthetic inputs (needs /root/reference; see oracle/ref_harness.py for the shims).

    python oracle/gen_golden_losses.py     # writes tests/golden/g7_single_losses.npz, g8_warp_loss.npz, g9_multi_view_round.npz

G7: core/utils/loss_utils.py compute_loss_mask / compute_loss_depth / compute_loss_normal on a synthetic render +
    ground truth: the four loss values and their autograd gradients w.r.t. depth, normal, min_sdf.
G8: SDFRenderer_warp.get_valid_points + compute_loss_color (core/sdfrenderer/renderer_warp.py:18-101) on analytic
    sphere depth maps of two views: loss_color, the kept-point images, and the autograd gradients w.r.t. the view-1
    depth and all four camera tensors.
G9: two view pairs of the multi-view round (loss_multi.py:6-49 via optimize_multi.py:50-79) with a sim(3): summed loss
    and gradients w.r.t. the shape code and the sim(3) parameters.
G10 (--g10): decode_color + SDFRenderer_color.render with a seed-defined colour decoder (row f4).
G11 (--g11): decode_sdf differentiated by autograd w.r.t. latent and points.
G6 (--g6): per-call point counts of the reference's decoder calls during render_depth (live rays per march step).
The goldens are data (inputs + reference outputs); no reference source is copied.
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
from gen_synth import sphere_view, procedural_images  # noqa: E402

OUT = os.path.join(_HERE, '..', 'tests', 'golden')


def golden_g7():
    rh.install_shims()
    rh.reference_modules()
    from core.utils import loss_utils as LU
    H, W = 40, 56
    rs = np.random.RandomState(11)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    r2 = ((xx - 27.0) / 20.0) ** 2 + ((yy - 19.0) / 14.0) ** 2
    mask = (r2 < 1.0)
    gt_mask = (((xx - 30.0) / 19.0) ** 2 + ((yy - 18.0) / 15.0) ** 2) < 1.0
    depth = np.where(mask, 1.2 + 0.3 * r2 + 0.01 * rs.standard_normal((H, W)), 1e11).astype(np.float32)
    gt_depth = np.where(gt_mask, 1.22 + 0.28 * r2, 0.0).astype(np.float32)
    gt_depth[5:9, 20:30] = 0.0          # sparse-depth holes (loss_utils.py:118)
    normal = rs.standard_normal((H, W, 3)).astype(np.float32) * mask[:, :, None]
    normal[18:20, 25:31] = 0.0          # depth2normal boundary zeros (loss_utils.py:156-157)
    gt_normal = rs.standard_normal((H, W, 3)).astype(np.float32)
    min_sdf = (np.where(mask, -2e-5, 8e-4 * np.sqrt(r2)) + 3e-5 * rs.standard_normal((H, W))).astype(np.float32)
    thr = 5e-5
    d, n, q = (torch.from_numpy(a).clone().requires_grad_(True) for a in (depth, normal, min_sdf))
    m, g = torch.from_numpy(mask), torch.from_numpy(gt_mask)
    lg, lo, _ = LU.compute_loss_mask(q, m, g, threshold=thr)
    ld, _ = LU.compute_loss_depth(d, m, torch.from_numpy(gt_depth), g)
    ln, _ = LU.compute_loss_normal(n, m, torch.from_numpy(gt_normal), g)
    w = np.array([1.0, 0.7, 10.0, 5.0], np.float32)      # upstream gradients of the four terms
    (w[0] * lg + w[1] * lo + w[2] * ld + w[3] * ln).backward()
    np.savez_compressed(os.path.join(OUT, 'g7_single_losses.npz'), H=H, W=W, threshold=thr, depth=depth, normal=normal,
                        mask=mask.astype(np.uint8), min_sdf=min_sdf, gt_depth=gt_depth, gt_normal=gt_normal,
                        gt_mask=gt_mask.astype(np.uint8), weights=w,
                        losses=np.array([lg.item(), lo.item(), ld.item(), ln.item()], np.float64),
                        g_depth=d.grad.numpy(), g_normal=n.grad.numpy(), g_min_sdf=q.grad.numpy())
    print('g7 losses', lg.item(), lo.item(), ld.item(), ln.item(), 'sets', int((g & ~m).sum()), int((m & ~g).sum()))
    # empty-set case: identical masks -> both hinge terms are 0 with zero gradient
    lg0, lo0, _ = LU.compute_loss_mask(q.detach().requires_grad_(True), m, m, threshold=thr)
    assert lg0.item() == 0.0 and lo0.item() == 0.0


def golden_g8():
    rh.install_shims()
    SDFRenderer_warp = rh.reference_modules()[1]
    Ws, bs, _ = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    H, W = 40, 56
    K = fixture.make_intrinsic(H, W)
    R1, T1 = fixture.make_camera(10, 15, 1.6, 0)
    R2, T2 = fixture.make_camera(24, 12, 1.7, 3.0)
    z1, hit1 = sphere_view(K, R1, T1, H, W, 0.55, (0.03, -0.02, 0.05))
    z2, hit2 = sphere_view(K, R2, T2, H, W, 0.55, (0.03, -0.02, 0.05))
    z2 = z2.copy()
    z2[hit2] += np.where(np.arange(int(hit2.sum())) % 7 == 0, 0.2, 0.0).astype(np.float32)   # inconsistent depth -> dropped points
    img1, img2 = procedural_images(H, W)
    r = SDFRenderer_warp(dec, K, img_hw=(H, W), march_step=10, buffer_size=1, use_gpu=False)
    r.device = torch.device('cpu')
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32)).clone().requires_grad_(True)
    Z1, tR1, tT1, tR2, tT2 = t(z1), t(R1), t(T1), t(R2), t(T2)
    m1 = torch.from_numpy(hit1)
    out1 = (Z1, m1, torch.zeros(H * W))
    out2 = (torch.from_numpy(z2), torch.from_numpy(hit2), torch.zeros(H * W))
    thres = 1e-3
    xy, vmi, vdi = r.get_valid_points(out1, out2, tR1, tT1, tR2, tT2, thres)
    loss, c1, c2 = r.compute_loss_color(torch.from_numpy(img1), torch.from_numpy(img2), xy, m1, vmi, vdi)
    (2.5 * loss).backward()
    keep = np.zeros(H * W, np.uint8)
    keep[np.nonzero(hit1)[0][vdi.numpy().astype(bool)]] = 1
    np.savez_compressed(os.path.join(OUT, 'g8_warp_loss.npz'), H=H, W=W, K=K, R1=R1, T1=T1, R2=R2, T2=T2, thres_depth=thres,
                        zdepth1=z1, mask1=hit1.astype(np.uint8), zdepth2=z2, img1=img1, img2=img2, g_loss=np.float32(2.5),
                        loss_color=np.float64(loss.item()), keep=keep, color_valid_1=c1.detach().numpy(), color_valid_2=c2.detach().numpy(),
                        g_zdepth1=Z1.grad.numpy(), g_R1=tR1.grad.numpy(), g_T1=tT1.grad.numpy(), g_R2=tR2.grad.numpy(), g_T2=tT2.grad.numpy())
    print('g8 loss', loss.item(), 'valid', int(hit1.sum()), 'kept', int(keep.sum()),
          'gz', float(Z1.grad.abs().max()), 'gR1', float(tR1.grad.abs().max()), 'gT2', float(tT2.grad.abs().max()))


class _Cam(object):
    def __init__(self, R, T):
        self.extrinsic = np.concatenate([R, T[:, None]], 1).astype(np.float32)


def golden_g9():
    """G9: two view pairs of the multi-view round (optimize_multi.py:50-79 -> compute_loss_color_warp, loss_multi.py:6-49)
    with a non-trivial sim(3): summed loss and its gradients w.r.t. the shape code and the sim(3) parameters."""
    rh.install_shims()
    rh.install_device_shims()
    SDFRenderer_warp = rh.reference_modules()[1]
    from core.inv_optimizer.loss_multi import compute_loss_color_warp
    from core.utils.train_utils import params_to_mtrx
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    H = W = 48
    K = fixture.make_intrinsic(H, W)
    cams = [fixture.make_camera(az, 15, 1.6, 0) for az in (10, 22, 34)]
    img_a, img_b = procedural_images(H, W)
    imgs = [img_a, img_b, (0.5 * (img_a + img_b[::-1])).astype(np.float32)]
    r = SDFRenderer_warp(dec, K, img_hw=(H, W), march_step=50, buffer_size=1, use_gpu=False)
    r.device = torch.device('cpu')
    lat = torch.from_numpy(latent).clone().requires_grad_(True)
    sim3 = {'rot': torch.tensor([0.02, -0.01, 0.015], requires_grad=True), 'scale': torch.tensor(0.03, requires_grad=True),
            'trans': torch.tensor([0.01, 0.02, -0.015], requires_grad=True)}
    sim3_init = torch.cat([torch.eye(3), torch.zeros(3, 1)], 1)
    sim_mtrx = params_to_mtrx(sim3).clone()
    sim_mtrx[:, 3] = torch.matmul(sim_mtrx[:3, :3].clone(), sim3_init[:, 3]) + sim_mtrx[:, 3].clone()
    sim_mtrx[:3, :3] = torch.matmul(sim_mtrx[:3, :3].clone(), sim3_init[:3, :3])
    sim3_scale = torch.norm(sim_mtrx[:3, :3].clone()) / np.sqrt(3)
    weights = {'color': 5.0, 'l2reg': 1.0}
    cameras = [_Cam(R, T) for R, T in cams]
    images = [torch.from_numpy(i) for i in imgs]
    total, packs = 0.0, []
    for (i1, i2) in ((0, 1), (1, 2)):
        loss, pack = compute_loss_color_warp(r, lat, images, cameras, i1, i2, weights, sim3=sim_mtrx, sim3_scale=sim3_scale)
        total = total + loss
        packs.append([float(pack['color']), float(pack['l2reg'])])
    total.backward()
    np.savez_compressed(os.path.join(OUT, 'g9_multi_view_round.npz'), weights_sha256=fixture.weights_sha256(Ws, bs), latent=latent, K=K,
                        H=H, W=W, march_step=50, buffer_size=1, extrinsics=np.stack([c.extrinsic for c in cameras]),
                        images=np.stack(imgs), sim3_rot=sim3['rot'].detach().numpy(), sim3_scale=sim3['scale'].detach().numpy(),
                        sim3_trans=sim3['trans'].detach().numpy(), w_color=5.0, w_l2reg=1.0, pairs=np.array([[0, 1], [1, 2]]),
                        loss_total=np.float64(total.item()), packs=np.array(packs), sim_mtrx=sim_mtrx.detach().numpy(),
                        g_latent=lat.grad.numpy(), g_rot=sim3['rot'].grad.numpy(), g_scale=sim3['scale'].grad.numpy(),
                        g_trans=sim3['trans'].grad.numpy())
    print('g9 total', total.item(), packs, 'glat', float(lat.grad.norm()), 'grot', sim3['rot'].grad.numpy(),
          'gscale', float(sim3['scale'].grad), 'gtrans', sim3['trans'].grad.numpy())


def golden_g10():
    """G10 (row f4): the reference's decode_color (decoder_utils.py:94-112) on fixed points and
    SDFRenderer_color.render (renderer_rgb.py:73-125) without and with a point light, using the seed-defined colour
    decoder fixture (distr.fixture.make_color_decoder_weights)."""
    rh.install_shims()
    mods = rh.reference_modules()
    Decoder, decoder_utils = mods[2], mods[3]
    from core.sdfrenderer.renderer_rgb import SDFRenderer_color
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    cs = 256
    Wc, bc, color_code = fixture.make_color_decoder_weights(color_size=cs)
    dims = [512] * 8
    dims[3] += cs
    dec_c = Decoder(256 + cs, dims, last_dim=3, dropout=list(range(8)), dropout_prob=0.2, norm_layers=(), latent_in=[4])
    dec_c.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a.copy()) for l, (W, b) in enumerate(zip(Wc, bc))
                           for n, a in (('weight', W), ('bias', b))})
    dec_c.eval()
    rs = np.random.RandomState(21)
    pts = (rs.rand(2048, 3) * 1.6 - 0.8).astype(np.float32)
    with torch.no_grad():
        rgb = decoder_utils.decode_color(dec_c, torch.from_numpy(color_code), torch.from_numpy(latent), torch.from_numpy(pts), no_grad=True)
    H = W = 48
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(25, 20, 1.6, 0)
    r = SDFRenderer_color(dec, dec_c, K, img_hw=(H, W), march_step=40, buffer_size=1, use_gpu=False)
    r.device = torch.device('cpu')
    tR, tT = torch.from_numpy(R), torch.from_numpy(T)
    lat, cc = torch.from_numpy(latent), torch.from_numpy(color_code)
    d, n, c, m, q = r.render(cc, lat, tR, tT, no_grad=True)
    lights = np.array([[1.5, 1.0, -1.0]], np.float32)     # one light: the reference's torch.bmm(R[None], directions) (renderer_rgb.py:58) only accepts M = 1
    energies = np.array([0.8], np.float32)
    d2, n2, c2, m2, q2 = r.render(cc, lat, tR, tT, no_grad=True, lighting_locations=torch.from_numpy(lights),
                                  lighting_energies=torch.from_numpy(energies))
    np.savez_compressed(os.path.join(OUT, 'g10_color_render.npz'), weights_sha256=fixture.weights_sha256(Ws, bs),
                        color_weights_sha256=fixture.weights_sha256(Wc, bc), color_size=cs, latent=latent, color_code=color_code,
                        points=pts, rgb=rgb.numpy(), K=K, R=R, T=T, H=H, W=W, march_step=40, buffer_size=1,
                        depth=d.detach().numpy(), normal=n.detach().numpy(), color=c.detach().numpy(), mask=m.numpy(), min_sdf=q.detach().numpy(),
                        lights=lights, energies=energies, color_shaded=c2.detach().numpy())
    print('g10 rgb range', float(rgb.min()), float(rgb.max()), 'valid', int(m.sum()), 'color mean', float(c.detach().abs().mean()),
          'shaded mean', float(c2.detach().abs().mean()))


def golden_g11():
    """G11: the reference's decode_sdf (decoder_utils.py:53-74) differentiated by autograd w.r.t. the latent code and
    the points (clamped and unclamped), with seeded per-point upstream gradients."""
    rh.install_shims()
    decoder_utils = rh.reference_modules()[3]
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    rs = np.random.RandomState(31)
    pts = (rs.rand(777, 3) * 1.4 - 0.7).astype(np.float32)
    wts = rs.standard_normal((777, 1)).astype(np.float32)
    out = {}
    for name, clamp in (('clamped', 0.1), ('raw', None)):
        lat = torch.from_numpy(latent).clone().requires_grad_(True)
        x = torch.from_numpy(pts).clone().requires_grad_(True)
        y = decoder_utils.decode_sdf(dec, lat, x, clamp_dist=clamp)
        (y * torch.from_numpy(wts)).sum().backward()
        out['sdf_' + name], out['g_latent_' + name], out['g_points_' + name] = y.detach().numpy(), lat.grad.numpy(), x.grad.numpy()
    np.savez_compressed(os.path.join(OUT, 'g11_decode_sdf_grad.npz'), weights_sha256=fixture.weights_sha256(Ws, bs), latent=latent,
                        points=pts, weights=wts, **out)
    print('g11', {k: float(np.abs(v).max()) for k, v in out.items()}, 'clamped pts', int((np.abs(out['sdf_raw']) > 0.1).sum()))


def golden_g13():
    """G13: the reference's decode_color (decoder_utils.py:94-112) differentiated by autograd w.r.t. the colour code, the shape code
    and the points, with seeded per-point upstream gradients (colour decoder fixture of G10)."""
    rh.install_shims()
    mods = rh.reference_modules()
    Decoder, decoder_utils = mods[2], mods[3]
    _, _, latent = fixture.make_decoder_weights()
    cs = 256
    Wc, bc, color_code = fixture.make_color_decoder_weights(color_size=cs)
    dims = [512] * 8
    dims[3] += cs
    dec_c = Decoder(256 + cs, dims, last_dim=3, dropout=list(range(8)), dropout_prob=0.2, norm_layers=(), latent_in=[4])
    dec_c.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a.copy()) for l, (W, b) in enumerate(zip(Wc, bc))
                           for n, a in (('weight', W), ('bias', b))})
    dec_c.eval()
    rs = np.random.RandomState(33)
    pts = (rs.rand(333, 3) * 1.4 - 0.7).astype(np.float32)
    wts = rs.standard_normal((333, 3)).astype(np.float32)
    cc = torch.from_numpy(color_code).clone().requires_grad_(True)
    sc = torch.from_numpy(latent).clone().requires_grad_(True)
    x = torch.from_numpy(pts).clone().requires_grad_(True)
    rgb = decoder_utils.decode_color(dec_c, cc, sc, x)
    (rgb * torch.from_numpy(wts)).sum().backward()
    np.savez_compressed(os.path.join(OUT, 'g13_decode_color_grad.npz'), color_weights_sha256=fixture.weights_sha256(Wc, bc), color_size=cs,
                        latent=latent, color_code=color_code, points=pts, weights=wts, rgb=rgb.detach().numpy(), g_color_code=cc.grad.numpy(),
                        g_shape_code=sc.grad.numpy(), g_points=x.grad.numpy())
    print('g13', float(rgb.abs().max()), float(cc.grad.abs().max()), float(sc.grad.abs().max()), float(x.grad.abs().max()))


def golden_g12():
    """G12: SDFRenderer.render(num_forward_sampling=3) (renderer.py:912-941, 982-985): the k samples behind the surface and the
    gradients of a seeded weighted sum of them w.r.t. the latent code and the camera."""
    rh.install_shims()
    SDFRenderer = rh.reference_modules()[0]
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    H = W = 40
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(-25, 18, 1.6, 5)
    r = SDFRenderer(dec, K, img_hw=(H, W), march_step=30, buffer_size=3, use_gpu=False)
    lat = torch.from_numpy(latent).clone().requires_grad_(True)
    Rt, Tt = torch.from_numpy(R).clone().requires_grad_(True), torch.from_numpy(T).clone().requires_grad_(True)
    depth, normal, mask, q, inside = r.render(lat, Rt, Tt, num_forward_sampling=3)
    wts = np.random.RandomState(12).rand(H, W, 3).astype(np.float32)
    (inside * torch.from_numpy(wts)).sum().backward()
    np.savez_compressed(os.path.join(OUT, 'g12_forward_sampling.npz'), weights_sha256=fixture.weights_sha256(Ws, bs), latent=latent, K=K, R=R, T=T,
                        H=H, W=W, march_step=30, buffer_size=3, num_forward_sampling=3, weights=wts, mask=mask.numpy(),
                        inside_samples=inside.detach().numpy(), g_latent=lat.grad.numpy(), g_R=Rt.grad.numpy(), g_T=Tt.grad.numpy())
    print('g12 valid', int(mask.sum()), 'inside range', float(inside.min()), float(inside.max()), 'glat', float(lat.grad.norm()))


def golden_g6():
    """G6: structure of the reference's march, recorded by wrapping its decode_sdf (core/utils/decoder_utils.py:53): the
    number of points of every decoder call of render_depth in call order (= live rays per march step, then the
    re-evaluation calls), plus init_zdepth / valid_mask of the unit-sphere test, for the three marchers at C1 size."""
    rh.install_shims()
    mods = rh.reference_modules()
    SDFRenderer = mods[0]
    import core.sdfrenderer.renderer as ref_renderer
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    H = W = 64
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(30, 20, 1.6, 10)
    out = dict(weights_sha256=fixture.weights_sha256(Ws, bs), latent=latent, K=K, R=R, T=T, H=H, W=W, march_step=20, buffer_size=3)
    orig = ref_renderer.decode_sdf
    for marcher in ('trivial', 'recursive', 'pyramid_recursive'):
        sizes = []

        def counting(decoder, latent_vector, points, *a, **k):
            sizes.append(int(points.shape[0]))
            return orig(decoder, latent_vector, points, *a, **k)
        ref_renderer.decode_sdf = counting
        try:
            r = SDFRenderer(dec, K, img_hw=(H, W), march_step=20, buffer_size=3, use_gpu=False)
            with torch.no_grad():
                z, m, q = r.render_depth(torch.from_numpy(latent), torch.from_numpy(R), torch.from_numpy(T), ray_marching_type=marcher, no_grad=True)
                cam_pos = r.get_camera_location(torch.from_numpy(R), torch.from_numpy(T))
                cam_rays = r.get_camera_rays(torch.from_numpy(R))
                init_z, valid = r.get_intersections_with_unit_spheres(cam_pos, cam_rays)
        finally:
            ref_renderer.decode_sdf = orig
        out['calls_' + marcher] = np.array(sizes, np.int64)
        out['valid_final_' + marcher] = m.numpy()
        print('g6', marcher, sizes)
    out['init_zdepth'] = init_z.numpy()
    out['in_sphere'] = valid.numpy()
    np.savez_compressed(os.path.join(OUT, 'g6_march_structure.npz'), **out)


if __name__ == '__main__':
    if sys.argv[1:2] == ['--g6']:
        golden_g6()
        sys.exit(0)
    if sys.argv[1:2] == ['--g11']:
        golden_g11()
        sys.exit(0)
    if sys.argv[1:2] == ['--g10']:
        golden_g10()
        sys.exit(0)
    if sys.argv[1:2] == ['--g12']:
        golden_g12()
        sys.exit(0)
    if sys.argv[1:2] == ['--g13']:
        golden_g13()
        sys.exit(0)
    if sys.argv[1:2] != ['--g9']:
        golden_g7()
        golden_g8()
    golden_g9()
