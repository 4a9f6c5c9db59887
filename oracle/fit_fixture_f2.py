"""TEST INFRASTRUCTURE -- fits fixture "F2" (SURVEY.md 8c): a DeepSDF 8x512 decoder whose zero level set is a NON-CONVEX shape with
thin parts, concavities and several surface crossings along a ray -- a torus pierced by a thin plate -- so that parity evidence
does not rest on the smooth blob of fixture F1 alone (grazing rays, thin parts and multiple hits are where stop-step, top-k and
mask decisions differ).

    python oracle/fit_fixture_f2.py        # ~10 min on 8 CPU cores; writes tests/golden/fixture_f2.npz (3.7 MB)

A CPU Adam fit is not bit-reproducible across machines, so the fitted weights themselves are the fixture: every weight is rounded
to a bf16-representable value and stored as its upper 16 bits (the decoder IS those rounded values, exactly, in f32; the fit only
chooses them). distr.fixture.load_fixture_f2() expands them. Nothing of the reference is involved here: the network below is the
architecture core/graph/deep_sdf_decoder.py:19-111 describes (latent 256, 8 x 512, latent_in=[4], ReLU, tanh), written with
plain torch ops, trained DeepSDF-style (clamped L1, deepsdf/train_deep_sdf.py) on an analytic signed distance function.
"""
import os
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, '..', 'dist-renderer_amd'))
from distr import fixture  # noqa: E402

OUT = os.path.join(_HERE, '..', 'tests', 'golden', 'fixture_f2.npz')

TORUS_R, TORUS_r = 0.50, 0.13                 # ring in the x-z plane (axis y)
PLATE = (0.62, 0.045, 0.20)                   # half extents of the thin plate through the hole


def sdf_target(p):
    """Exact signed distance of (torus U plate); p (n,3) torch f32."""
    q = torch.stack([torch.sqrt(p[:, 0] ** 2 + p[:, 2] ** 2) - TORUS_R, p[:, 1]], 1)
    torus = torch.norm(q, dim=1) - TORUS_r
    d = torch.abs(p) - torch.tensor(PLATE)
    box = torch.norm(torch.clamp(d, min=0.0), dim=1) + torch.clamp(d.max(1).values, max=0.0)
    return torch.minimum(torus, box)


def forward(Ws, bs, latent, xyz):
    n = xyz.shape[0]
    x0 = torch.cat([latent.expand(n, -1), xyz], 1)
    x = x0
    for l in range(9):
        if l == 4:
            x = torch.cat([x, x0], 1)
        x = torch.nn.functional.linear(x, Ws[l], bs[l])
        if l < 8:
            x = torch.relu(x)
    return torch.tanh(x).squeeze(1)


def sample(rs, n):
    """Half of the batch near the surface (the closest of a larger uniform draw, jittered), half uniform in the unit ball."""
    u = rs.uniform(-1.0, 1.0, size=(16 * n, 3)).astype(np.float32)
    u = u[(u ** 2).sum(1) <= 1.02]
    t = torch.from_numpy(u)
    d = sdf_target(t).abs()
    near = t[torch.argsort(d)[:n // 2]] + 0.02 * torch.from_numpy(rs.standard_normal((n // 2, 3)).astype(np.float32))
    far = t[torch.from_numpy(rs.permutation(t.shape[0])[:n - n // 2])]
    return torch.cat([near, far], 0)


def to_bf16_bits(a):
    """f32 array -> uint16 upper halves after round-to-nearest-even to 8 significant bits."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return r


def from_bf16_bits(b):
    return (b.astype(np.uint32) << 16).view(np.float32)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
    batch = 8192
    torch.manual_seed(0)
    torch.set_num_threads(8)
    Ws0, bs0, _ = fixture.make_decoder_weights(seed=4321, latent_scale=0.01)
    latent = torch.from_numpy(fixture.make_latent(4322))
    Ws = [torch.from_numpy(w).clone().requires_grad_(True) for w in Ws0]
    bs = [torch.from_numpy(b).clone().requires_grad_(True) for b in bs0]
    opt = torch.optim.Adam(Ws + bs, lr=1e-4)
    rs = np.random.RandomState(99)
    t0 = time.time()
    for it in range(steps):
        for g in opt.param_groups:
            g['lr'] = 1e-4 * min(1.0, (it + 1) / 50.0) * (0.5 ** (it // 1000))
        x = sample(rs, batch)
        y = sdf_target(x)
        pred = forward(Ws, bs, latent, x)
        loss = (torch.clamp(pred, -0.1, 0.1) - torch.clamp(y, -0.1, 0.1)).abs().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 100 == 0 or it == steps - 1:
            print('step %4d  clamped L1 %.5f  (%.0f s)' % (it, float(loss), time.time() - t0), flush=True)
    Wq = [to_bf16_bits(w.detach().numpy()) for w in Ws]
    bq = [to_bf16_bits(b.detach().numpy()) for b in bs]
    # what the rounded decoder is: error against the analytic field on fresh points
    Wr = [torch.from_numpy(from_bf16_bits(w)) for w in Wq]
    br = [torch.from_numpy(from_bf16_bits(b)) for b in bq]
    with torch.no_grad():
        x = sample(np.random.RandomState(5), 65536)
        y = sdf_target(x)
        e_fit = (torch.clamp(forward(Ws, bs, latent, x), -0.1, 0.1) - torch.clamp(y, -0.1, 0.1)).abs()
        e_rnd = (torch.clamp(forward(Wr, br, latent, x), -0.1, 0.1) - torch.clamp(y, -0.1, 0.1)).abs()
        sign = ((forward(Wr, br, latent, x) > 0) == (y > 0)).float().mean()
    print('held-out clamped L1: fitted %.5f, rounded-to-bf16 weights %.5f (max %.4f); sign agreement %.4f'
          % (float(e_fit.mean()), float(e_rnd.mean()), float(e_rnd.max()), float(sign)))
    out = {'latent': latent.numpy()}
    for l in range(9):
        out['W%d' % l] = Wq[l]
        out['b%d' % l] = bq[l]
    np.savez_compressed(OUT, **out)
    Wf, bf, _ = fixture.load_fixture_f2(OUT)
    print('wrote', OUT, os.path.getsize(OUT), 'bytes; sha256', fixture.weights_sha256(Wf, bf))


if __name__ == '__main__':
    main()
