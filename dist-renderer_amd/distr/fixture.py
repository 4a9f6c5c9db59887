"""Seed-defined synthetic DeepSDF 8x512 decoder ("fixture F1") and synthetic cameras.

There are no pretrained DeepSDF weights in the reference tree (they ship in an
FTP tarball, deepsdf/download_scripts/download_models.sh:2-4) and a plain
``nn.Linear`` random init gives a constant-sign field with no surface, so every
test, golden vector and benchmark in this repository uses the geometric
initialisation below: it yields a closed blob-like zero level set inside the unit
sphere whose shape depends on the latent code.

Only numpy is needed here; the arrays are consumed by
  * ``distr.decoder_pack`` (product: packs them for the HIP kernels),
  * ``oracle/ref_harness.py`` (this container only: loads them into the reference's
    ``Decoder`` to generate golden vectors),
  * ``oracle/oracle.py`` (CPU restatement).

Layer shapes follow core/graph/deep_sdf_decoder.py:36-61 with
``latent_size=256, dims=[512]*8, latent_in=[4]``:
lin0 259->512, lin1-2 512->512, lin3 512->253, lin4 512->512, lin5-7 512->512, lin8 512->1.
"""
import hashlib
import math

import numpy as np

LATENT_SIZE = 256
HIDDEN = 512
NUM_LINEAR = 9


def layer_shapes(latent_size=LATENT_SIZE, hidden=HIDDEN):
    """(out, in) of lin0..lin8 for the DeepSDF '8x512, latent_in=[4]' decoder."""
    d0 = latent_size + 3
    shapes = []
    dims = [d0] + [hidden] * 8 + [1]
    for l in range(len(dims) - 1):
        out = dims[l + 1] - d0 if (l + 1) == 4 else dims[l + 1]
        shapes.append((out, dims[l]))
    return shapes


def make_decoder_weights(seed=1234, latent_scale=0.003):
    """Returns (weights, biases, latent): lists of f32 arrays W_l (out,in), b_l (out,), latent (1,256).

    Draw order is fixed (one RandomState stream) so the arrays are reproducible
    bit-for-bit on any machine: for l in 0..8 the (out,in) matrix, then for l==0 and
    l==4 an extra (512,256) draw for the latent columns; finally the latent code.
    """
    rs = np.random.RandomState(seed)
    shapes = layer_shapes()
    Ws, bs = [], []
    for l, (o, i) in enumerate(shapes):
        W = rs.standard_normal((o, i)) * (math.sqrt(2.0) / math.sqrt(o))
        b = np.zeros((o,))
        if l == 0:
            # input = [latent(256) | xyz(3)]
            W[:, :LATENT_SIZE] = latent_scale * rs.standard_normal((o, LATENT_SIZE))
        if l == 4:
            # input = [x3(253) | latent(256) | xyz(3)]
            W[:, 253:253 + LATENT_SIZE] = latent_scale * rs.standard_normal((o, LATENT_SIZE))
            W[:, 253 + LATENT_SIZE:] = 0.0
        if l == 8:
            W = math.sqrt(math.pi) / math.sqrt(i) + 1e-5 * rs.standard_normal((o, i))
            b = np.full((o,), -0.5)
        Ws.append(np.ascontiguousarray(W, dtype=np.float32))
        bs.append(np.ascontiguousarray(b, dtype=np.float32))
    latent = (0.1 * rs.standard_normal((1, LATENT_SIZE))).astype(np.float32)
    return Ws, bs, latent


def color_layer_shapes(color_size=256):
    """(out, in) of lin0..lin8 of the colour decoder load_decoder(color_size=...) builds (decoder_utils.py:16-24):
    latent = 256 + color_size, dims[3] += color_size (so lin3 still ends at 253 rows), last_dim = 3."""
    lat = LATENT_SIZE + color_size
    return [(512, lat + 3), (512, 512), (512, 512), (253, 512), (512, 512 + color_size), (512, 512), (512, 512), (512, 512), (3, 512)]


def make_color_decoder_weights(seed=4321, color_size=256):
    """Seed-defined synthetic colour decoder (no pretrained one exists in the tree): He-style hidden layers, small
    latent columns, a last layer whose three rows differ so that r, g, b vary over the surface. Returns
    (weights, biases, color_code (1, color_size))."""
    rs = np.random.RandomState(seed)
    Ws, bs = [], []
    for l, (o, i) in enumerate(color_layer_shapes(color_size)):
        W = rs.standard_normal((o, i)) * (math.sqrt(2.0) / math.sqrt(i))
        b = 0.05 * rs.standard_normal((o,))
        if l == 0:
            W[:, :-3] *= 0.05                    # latent columns
            W[:, -3:] *= 8.0                     # xyz columns: spatial variation
        if l == 4:
            W[:, 253:-3] *= 0.05
        if l == 8:
            W = rs.standard_normal((o, i)) * (1.5 / math.sqrt(i))
        Ws.append(np.ascontiguousarray(W, dtype=np.float32))
        bs.append(np.ascontiguousarray(b, dtype=np.float32))
    color_code = (0.3 * rs.standard_normal((1, color_size))).astype(np.float32)
    return Ws, bs, color_code


def weights_sha256(Ws, bs):
    h = hashlib.sha256()
    for W, b in zip(Ws, bs):
        h.update(np.ascontiguousarray(W, dtype=np.float32).tobytes())
        h.update(np.ascontiguousarray(b, dtype=np.float32).tobytes())
    return h.hexdigest()


def make_latent(seed):
    """Extra latent codes (C5 'batch of shapes'): seed -> (1,256) f32."""
    rs = np.random.RandomState(seed)
    return (0.1 * rs.standard_normal((1, LATENT_SIZE))).astype(np.float32)


def load_fixture_f2(path=None):
    """Fixture "F2": a DeepSDF 8x512 decoder fitted to a NON-CONVEX analytic shape (a torus pierced by a thin plate: thin parts,
    concavities, several surface crossings along a ray; oracle/fit_fixture_f2.py). Unlike F1 it cannot be regenerated from a seed
    (a CPU Adam fit is not bit-reproducible), so the weights are data: every weight is a bf16-representable f32 value stored as its
    upper 16 bits in tests/golden/fixture_f2.npz. Returns (weights, biases, latent) like make_decoder_weights."""
    import os
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'fixture_f2.npz')
    z = np.load(path)
    expand = lambda b: np.ascontiguousarray((b.astype(np.uint32) << 16).view(np.float32))
    Ws = [expand(z['W%d' % l]) for l in range(NUM_LINEAR)]
    bs = [expand(z['b%d' % l]) for l in range(NUM_LINEAR)]
    for W, (o, i) in zip(Ws, layer_shapes()):
        assert W.shape == (o, i), (W.shape, o, i)
    return Ws, bs, np.ascontiguousarray(z['latent'], dtype=np.float32).reshape(1, LATENT_SIZE)


def make_intrinsic(h, w):
    """K = [[w,0,w/2],[0,h,h/2],[0,0,1]]  (SURVEY 8d synthetic camera, ~53 deg FOV)."""
    return np.array([[float(w), 0.0, w / 2.0], [0.0, float(h), h / 2.0], [0.0, 0.0, 1.0]], dtype=np.float64)


def make_camera(azimuth_deg=0.0, elevation_deg=0.0, distance=1.6, roll_deg=0.0):
    """World->camera extrinsic (R (3,3), T (3,)) looking at the origin from a point on a sphere.

    azimuth=elevation=0 gives R=I, T=(0,0,distance). The camera position is -R^T T.
    Own construction (Rz(roll) Rx(elev) Ry(azim)); not the Blender-quaternion path of
    common/geometry/view.py -- only '3x3 R + 3-vector T' matter to the renderer.
    """
    a, e, r = (math.radians(v) for v in (azimuth_deg, elevation_deg, roll_deg))
    Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, math.cos(e), -math.sin(e)], [0, math.sin(e), math.cos(e)]])
    Rz = np.array([[math.cos(r), -math.sin(r), 0], [math.sin(r), math.cos(r), 0], [0, 0, 1]])
    R = (Rz @ Rx @ Ry).astype(np.float32)
    T = np.array([0.0, 0.0, distance], dtype=np.float32)
    return R, T
