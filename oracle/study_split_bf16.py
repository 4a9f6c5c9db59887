"""Numerics study (CPU, PyTorch restatement; test/analysis infrastructure, never imported by the product): what would a split-bf16
decoder do to the render? Every f32 product a*w of the decoder's layers is replaced by the products of bf16 pieces
(a = a0 + a1 + a2, w = w0 + w1 + w2, each piece bf16, accumulated in f32 -- what v_mfma_f32_32x32x16_bf16 does):
    3 products:  a0*w0 + a0*w1 + a1*w0                          (3 bf16 MFMAs per f32 MFMA: 5.3x the f32 MFMA rate at best)
    6 products:  + a1*w1 + a0*w2 + a2*w0                        (2.7x at best)
Round 3 adds the split-f16 form (products = 'h3'): a = (a0 + a1) / 64, w = (w0 + w1) / 64 with f16 pieces of the scaled operands,
    3 products:  a0*w0 + a0*w1 + a1*w0                          (11 + 11 significant bits per operand: what distr_mlp_h3.hpp computes)
and compares every form with a float64 decoder as well.
The whole render (march, sample selection, depth2normal, loss, backward through a straight-through rounding) runs with that
decoder and is compared with the exact f32 render of the same restatement: mask flips, depth / min-sdf residuals, latent-gradient
error -- next to the reference's own noise floor under 1e-7 relative weight noise (0 flips, depth 6.7e-5, normal
1.6e-3..3.2e-3). Prints a markdown table.   python oracle/study_split_bf16.py [sizes ...]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
from distr import fixture  # noqa: E402
from oracle.torch_restatement import TorchRenderer  # noqa: E402


def pieces(x, n):
    """x ~ sum of n bf16 pieces; rounding is straight-through for autograd."""
    out, r = [], x
    for _ in range(n):
        p = r.detach().to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    out[0] = out[0] + (x - x.detach())      # gradient flows through the leading piece as if unrounded
    return out


def pieces16(x, n, scale=64.0):
    """x ~ sum of n f16 pieces of scale * x (denormals kept, as v_cvt_pk_f16_f32 / the f16 MFMA do), divided by scale again."""
    out, r = [], x
    for _ in range(n):
        p = (r.detach() * scale).to(torch.float16).to(torch.float32) / scale
        out.append(p)
        r = r - p
    out[0] = out[0] + (x - x.detach())
    return out


class SplitRenderer(TorchRenderer):
    products = 3

    def _lin(self, x, l):
        if self.products == 'h3':
            xs, ws = pieces16(x, 2), self._wp[l]
            return F.linear(xs[0], ws[0]) + F.linear(xs[0], ws[1]) + F.linear(xs[1], ws[0]) + self.bs[l]
        npc = 2 if self.products == 3 else 3
        xs, ws = pieces(x, npc), self._wp[l]
        y = F.linear(xs[0], ws[0]) + F.linear(xs[0], ws[1]) + F.linear(xs[1], ws[0])
        if self.products == 6:
            y = y + F.linear(xs[1], ws[1]) + F.linear(xs[0], ws[2]) + F.linear(xs[2], ws[0])
        return y + self.bs[l]

    def decode(self, latent, pts):
        if not hasattr(self, '_wp'):
            self._wp = [pieces16(w, 2) if self.products == 'h3' else pieces(w, 3) for w in self.Ws]
        n = pts.shape[0]
        self.num_evals += n
        inp = torch.cat([latent.reshape(1, -1).expand(n, -1), pts], 1)
        x = inp
        for l in range(9):
            if l == 4:
                x = torch.cat([x, inp], 1)
            x = self._lin(x, l)
            if l < 8:
                x = torch.relu(x)
        return torch.tanh(x).reshape(-1)


def run(cls, products, Ws, bs, latent, H, S, weights):
    K = fixture.make_intrinsic(H, H)
    R, T = fixture.make_camera(30.0, 20.0, 1.6, 0.0)
    r = cls(Ws, bs, H, H, K, march_step=S, buffer_size=3, use_depth2normal=True)
    r.products = products
    lat = torch.from_numpy(latent).clone().requires_grad_(True)
    depth, normal, mask, q, z = r.render(lat, torch.from_numpy(R), torch.from_numpy(T), 'pyramid_recursive')
    wd, wq, wn = (torch.from_numpy(a) for a in weights)
    L = (depth * wd)[mask].sum() + (q * wq).sum() + (normal * wn).sum()
    L.backward()
    return dict(depth=depth.detach().numpy(), mask=mask.numpy(), q=q.detach().numpy(), normal=normal.detach().numpy(), g=lat.grad.numpy().copy())


MODES = (3, 6, 'h3')
NAMES = {3: 'split-bf16, 3 products', 6: 'split-bf16, 6 products', 'h3': 'split-f16 (x64), 3 products'}


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    args = [a for a in sys.argv[1:] if a != 'f2']
    sizes = [int(a) for a in args] or [64, 128, 256]
    Ws, bs, latent = fixture.load_fixture_f2() if 'f2' in sys.argv[1:] else fixture.make_decoder_weights()
    print('fixture %s\n' % ('F2' if 'f2' in sys.argv[1:] else 'F1'))
    # decoder value error on random points in the unit ball
    rs = np.random.RandomState(3)
    pts = rs.randn(20000, 3).astype(np.float32)
    pts *= (rs.rand(20000, 1) ** (1 / 3) / np.linalg.norm(pts, axis=1, keepdims=True)).astype(np.float32)
    ref = TorchRenderer(Ws, bs, 8, 8, fixture.make_intrinsic(8, 8))
    lat = torch.from_numpy(latent)
    with torch.no_grad():
        f = ref.decode(lat, torch.from_numpy(pts)).numpy()
        # float64 decoder
        x = inp = torch.cat([lat.double().reshape(1, -1).expand(pts.shape[0], -1), torch.from_numpy(pts).double()], 1)
        for l in range(9):
            if l == 4:
                x = torch.cat([x, inp], 1)
            x = F.linear(x, torch.from_numpy(np.asarray(Ws[l])).double(), torch.from_numpy(np.asarray(bs[l])).double())
            if l < 8:
                x = torch.relu(x)
        f64 = torch.tanh(x).reshape(-1).numpy()
        print('decoder value on 20 000 points of the unit ball (|sdf| up to %.2f; the march stops at |sdf| < 5e-5):\n' % np.abs(f).max())
        print('| decoder arithmetic | max abs error of sdf vs the f32 chain | p99 | mean | max abs error vs a float64 decoder |\n|---|---|---|---|---|')
        print('| exact f32 chain | 0 | 0 | 0 | %.2e |' % np.abs(f - f64).max())
        for k in MODES:
            sr = SplitRenderer(Ws, bs, 8, 8, fixture.make_intrinsic(8, 8))
            sr.products = k
            v = sr.decode(lat, torch.from_numpy(pts)).numpy()
            e = np.abs(v - f)
            print('| %s | %.2e | %.2e | %.2e | %.2e |' % (NAMES[k], e.max(), np.percentile(e, 99), e.mean(), np.abs(v - f64).max()))
    print('\nwhole render (pyramid_recursive, depth2normal, dense loss), split-bf16 decoder against the exact f32 decoder:\n')
    print('| image / steps | products | mask flips (of valid px) | max abs depth residual on common px | p99 | max abs min-sdf residual | normal p99 | latent gradient relative error | seconds |')
    print('|---|---|---|---|---|---|---|---|---|')
    for H in sizes:
        S = 20 if H == 64 else 50
        wrs = np.random.RandomState(5)
        weights = tuple(wrs.rand(*s).astype(np.float32) for s in ((H, H), (H, H), (H, H, 3)))
        exact = run(TorchRenderer, 0, Ws, bs, latent, H, S, weights)
        for k in MODES:
            t0 = time.time()
            o = run(SplitRenderer, k, Ws, bs, latent, H, S, weights)
            both = exact['mask'] & o['mask']
            dd = np.abs(o['depth'] - exact['depth'])[both]
            dn = np.abs(o['normal'] - exact['normal'])[both]
            print('| %dx%d / %d | %s | %d (of %d) | %.2e | %.2e | %.2e | %.2e | %.2e | %.0f |' % (
                H, H, S, NAMES[k], int((exact['mask'] != o['mask']).sum()), int(exact['mask'].sum()), dd.max() if dd.size else 0.0,
                np.percentile(dd, 99) if dd.size else 0.0, np.abs(o['q'] - exact['q']).max(), np.percentile(dn, 99) if dn.size else 0.0,
                np.abs(o['g'] - exact['g']).max() / np.abs(exact['g']).max(), time.time() - t0))
        sys.stdout.flush()


if __name__ == '__main__':
    main()
