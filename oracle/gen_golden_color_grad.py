"""TEST INFRASTRUCTURE -- golden G26: SDFRenderer_color.render (core/sdfrenderer/renderer_rgb.py:73-125) WITH gradients, run by the
REFERENCE itself on CPU (build container only; shims in oracle/ref_harness.py; no reference source copied):

    python oracle/gen_golden_color_grad.py        # writes tests/golden/g26_color_render_grad.npz

G10 pins the forward of the colour render; without no_grad (the default of render()) the colour image stays on the tape: through
decode_color to the colour code and the shape code, and through the surface points to the camera (renderer_rgb.py:20-38, Zdepth detached).
Recorded: outputs, a seeded loss over depth / normal / colour / min-sdf, its gradients w.r.t. colour code, shape code, R, T -- plain and
with a point light -- and the noise floor (decoder weights perturbed by 1e-7 relative).
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

OUT = os.path.join(_HERE, '..', 'tests', 'golden')
H = W = 40
CS = 32


def run(dec, dec_c, latent, color_code, K, R, T, lights, energies, w):
    from core.sdfrenderer.renderer_rgb import SDFRenderer_color
    r = SDFRenderer_color(dec, dec_c, K, img_hw=(H, W), march_step=30, buffer_size=2, use_gpu=False)
    r.device = torch.device('cpu')
    lat = torch.from_numpy(latent).clone().requires_grad_(True)
    cc = torch.from_numpy(color_code).clone().requires_grad_(True)
    Rt, Tt = torch.from_numpy(R).clone().requires_grad_(True), torch.from_numpy(T).clone().requires_grad_(True)
    kw = {} if lights is None else dict(lighting_locations=torch.from_numpy(lights), lighting_energies=torch.from_numpy(energies))
    d, n, c, m, q = r.render(cc, lat, Rt, Tt, **kw)
    mb = m.bool()
    L = (d * torch.from_numpy(w['d']))[mb].sum() + (n * torch.from_numpy(w['n'])).sum() + (c * torch.from_numpy(w['c'])).sum() + (q * torch.from_numpy(w['q'])).sum()
    L.backward()
    return dict(depth=d.detach().numpy(), normal=n.detach().numpy(), color=c.detach().numpy(), mask=m.numpy(), q=q.detach().numpy(), loss=np.float64(L.item()),
                g_color_code=cc.grad.numpy(), g_latent=lat.grad.numpy(), g_R=Rt.grad.numpy(), g_T=Tt.grad.numpy())


def main():
    rh.install_shims()
    Decoder = rh.reference_modules()[2]
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Ws, bs, latent = fixture.make_decoder_weights()
    Wc, bc, color_code = fixture.make_color_decoder_weights(color_size=CS)
    dims = [512] * 8
    dims[3] += CS

    def color_dec(Wl, bl):
        dc = Decoder(256 + CS, list(dims), last_dim=3, dropout=list(range(8)), dropout_prob=0.2, norm_layers=(), latent_in=[4])
        dc.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a.copy()) for l, (W_, b) in enumerate(zip(Wl, bl)) for n, a in (('weight', W_), ('bias', b))})
        return dc.eval()
    rsn = np.random.RandomState(99)
    noisy = lambda Wl: [(Wx * (1 + 1e-7 * rsn.standard_normal(Wx.shape))).astype(np.float32) for Wx in Wl]
    decs = (rh.build_reference_decoder(Ws, bs), color_dec(Wc, bc))
    decs_n = (rh.build_reference_decoder(noisy(Ws), bs), color_dec(noisy(Wc), bc))
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(25, 20, 1.6, 0)
    rs = np.random.RandomState(26)
    w = dict(d=rs.rand(H, W).astype(np.float32), n=rs.rand(H, W, 3).astype(np.float32), c=rs.rand(H, W, 3).astype(np.float32), q=rs.rand(H, W).astype(np.float32))
    lights, energies = np.array([[1.5, 1.0, -1.0]], np.float32), np.array([0.8], np.float32)
    out = dict(weights_sha256=fixture.weights_sha256(Ws, bs), color_weights_sha256=fixture.weights_sha256(Wc, bc), color_size=CS, latent=latent, color_code=color_code,
               K=K, R=R, T=T, H=H, W=W, march_step=30, buffer_size=2, lights=lights, energies=energies, **{'w_' + k: v for k, v in w.items()})
    for tag, (lg, en) in (('plain', (None, None)), ('lit', (lights, energies))):
        a = run(decs[0], decs[1], latent, color_code, K, R, T, lg, en, w)
        b = run(decs_n[0], decs_n[1], latent, color_code, K, R, T, lg, en, w)
        for k, v in a.items():
            out['%s.%s' % (tag, k)] = v
        for k in ('g_color_code', 'g_latent', 'g_R', 'g_T'):
            out['%s.%s_floor_rel' % (tag, k)] = float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max())
        print(tag, 'valid', int(a['mask'].sum()), 'loss %.4f' % a['loss'], {k: float(np.abs(a[k]).max()) for k in ('g_color_code', 'g_latent', 'g_R', 'g_T')},
              {k: out['%s.%s_floor_rel' % (tag, k)] for k in ('g_color_code', 'g_latent', 'g_R', 'g_T')})
    np.savez_compressed(os.path.join(OUT, 'g26_color_render_grad.npz'), **out)


if __name__ == '__main__':
    main()
