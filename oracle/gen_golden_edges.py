"""TEST INFRASTRUCTURE -- golden G25: edge cases of the ray / unit-sphere set-up and of empty results, rendered (or refused) by the
REFERENCE itself on CPU (build container only; shims in oracle/ref_harness.py; no reference source copied):

    python oracle/gen_golden_edges.py        # writes tests/golden/g25_edge_cases.npz

inside    camera INSIDE the unit sphere (get_intersections_with_unit_spheres, renderer.py:254-272: start depth 0 everywhere)
far       camera far away: the sphere covers part of the image; rays that miss it get Zdepth / depth 1e11 and
          min_sdf = dist + threshold - radius with its camera gradient (:863)
away      camera looking away: NO ray meets the sphere -- what the reference does is recorded (it raises)
nosurf    a shape code whose level set is empty inside the sphere: every in-sphere ray marches through, no valid pixel
Per case and marcher: outputs, loss, gradients -- or the exception type and message.
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
H = W = 32


def run(dec, latent, K, R, T, marcher, d2n):
    SDFRenderer = rh.reference_modules()[0]
    r = SDFRenderer(dec, K, img_hw=(H, W), march_step=12, buffer_size=3, use_gpu=False, use_depth2normal=d2n)
    lat = torch.from_numpy(latent).clone().requires_grad_(True)
    Rt, Tt = torch.from_numpy(R).clone().requires_grad_(True), torch.from_numpy(T).clone().requires_grad_(True)
    try:
        depth, normal, mask, mq = r.render(lat, Rt, Tt, ray_marching_type=marcher)
    except Exception as e:          # noqa: BLE001
        return dict(raised=True, exc_type=type(e).__name__, exc_msg=str(e)[:200])
    wd, wq, wn = gg.loss_weights(H, W, 5)
    mb = mask.bool()
    L = (depth * torch.from_numpy(wd))[mb].sum() + (mq * torch.from_numpy(wq)).sum() + (normal * torch.from_numpy(wn)).sum()
    L.backward()
    z = lambda t: np.zeros(t.shape, np.float32) if t.grad is None else t.grad.numpy().copy()
    with torch.no_grad():
        Zdepth = r.render_depth(lat, Rt, Tt, ray_marching_type=marcher, no_grad=True)[0].numpy()
    return dict(raised=False, depth=depth.detach().numpy(), normal=normal.detach().numpy(), mask=mask.numpy(), q=mq.detach().numpy(), zdepth=Zdepth,
                loss=np.float64(L.item()), g_latent=z(lat), g_R=z(Rt), g_T=z(Tt))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = rh.build_reference_decoder(Ws, bs)
    K = fixture.make_intrinsic(H, W)
    R, _ = fixture.make_camera(15, 5, 1.6, 0)
    du = rh.reference_modules()[3]
    # a shape code with no zero level set inside the unit sphere
    pts = (np.random.RandomState(3).rand(20000, 3) * 2 - 1).astype(np.float32)
    pts = pts[(pts ** 2).sum(1) <= 1.0]
    nosurf = None
    for c in (60.0, -60.0, 150.0, -150.0, 400.0, -400.0):
        with torch.no_grad():
            s = du.decode_sdf(dec, torch.from_numpy(c * latent), torch.from_numpy(pts), clamp_dist=None).numpy()
        print('latent x', c, 'sdf range', s.min(), s.max())
        if s.min() > 1e-3:
            nosurf = (c * latent).astype(np.float32)
            break
    assert nosurf is not None
    cases = {'inside': (latent, np.array([0, 0, 0.8], np.float32)), 'far': (latent, np.array([0.3, -0.2, 4.0], np.float32)),
             'away': (latent, np.array([0, 0, -3.0], np.float32)), 'nosurf': (nosurf, np.array([0, 0, 1.6], np.float32))}
    out = dict(weights_sha256=fixture.weights_sha256(Ws, bs), K=K, R=R, H=H, W=W, march_step=12, buffer_size=3, names=np.array(sorted(cases)))
    for name, (lat, T) in sorted(cases.items()):
        out['%s.latent' % name], out['%s.T' % name] = lat, T
        for marcher, d2n in (('recursive', False), ('pyramid_recursive', False), ('pyramid_recursive', True)):
            key = '%s.%s_%s' % (name, marcher, 'd2n' if d2n else 'agn')
            a = run(dec, lat, K, R, T, marcher, d2n)
            for k, v in a.items():
                out['%s.%s' % (key, k)] = v
            if a['raised']:
                print(key, 'RAISES', a['exc_type'], a['exc_msg'])
            else:
                print(key, 'valid', int(a['mask'].sum()), 'in-sphere', int((a['zdepth'] < 1e10).sum()), 'loss %.4f' % a['loss'], '|g_latent| %.3g |g_T| %.3g' % (
                    np.abs(a['g_latent']).max(), np.abs(a['g_T']).max()))
    np.savez_compressed(os.path.join(OUT, 'g25_edge_cases.npz'), **out)


if __name__ == '__main__':
    main()
