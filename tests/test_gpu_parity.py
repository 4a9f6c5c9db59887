"""GPU parity tests: the HIP kernels (called through the C ABI of libdistr.so) against the CPU oracle on the same
seeded inputs, against the committed reference goldens, and -- at BASELINE.json's full sizes -- through
size-independent properties. Run on the MI355X box: python -m pytest tests -m gpu.

Tolerances: north_star asks for depth / normal / silhouette within 1e-4 of the reference. HIP vs oracle is held
to 1e-6 wherever both sides are the same IEEE op sequence (decoder, march); the reduction order of the backward
differs, so gradients are compared at 1e-4 relative.
"""
import glob
import os
import re
import sys

import numpy as np
import pytest

from conftest import GOLDEN
import helpers

pytestmark = pytest.mark.gpu




def _floors():
    """The reference's own residuals under 1e-7 relative weight noise on the G4/G5/G9/G11 scenarios (oracle/gen_noise_floors.py)."""
    return {k: float(v) for k, v in np.load(os.path.join(GOLDEN, 'noise_floor_g4_g5_g9_g11.npz')).items()}


def _points(n, seed=3):
    rs = np.random.RandomState(seed)
    p = (rs.rand(n, 3) * 1.8 - 0.9).astype(np.float32)
    p[:4] = 0.0
    return p


def test_decoder_layers_bitwise(engine, cpu_oracle, fixture_decoder):
    """Every hidden layer of the fused MFMA decoder == the oracle's k-ordered fmaf chains, bit for bit."""
    import torch
    from distr import functions
    _, _, latent = fixture_decoder
    pts = _points(64 * 3 + 17)
    for layer in range(8):
        got = functions.debug_mlp_layer(engine, torch.from_numpy(latent), torch.from_numpy(pts), layer).cpu().numpy()
        ref = cpu_oracle.layer_activations(latent, pts, layer)
        width = 256 if layer == 3 else 512
        diff = np.abs(got[:, :width] - ref[:, :width])
        assert diff.max() == 0.0, 'layer %d: max diff %g at %s' % (layer, diff.max(), np.unravel_index(diff.argmax(), diff.shape))


@pytest.mark.parametrize('n', [1, 63, 64, 65, 4096 + 37])
def test_decode_sdf_matches_oracle(engine, cpu_oracle, fixture_decoder, n):
    import torch
    from distr import functions
    _, _, latent = fixture_decoder
    pts = _points(n)
    got = functions.mlp_eval(engine, torch.from_numpy(latent), torch.from_numpy(pts)).cpu().numpy().reshape(-1)
    ref = cpu_oracle.decode_sdf(latent, pts)
    assert np.abs(got - ref).max() <= 1e-7, np.abs(got - ref).max()
    got_c = functions.mlp_eval(engine, torch.from_numpy(latent), torch.from_numpy(pts), 0.1).cpu().numpy().reshape(-1)
    assert np.abs(got_c - np.clip(ref, -0.1, 0.1)).max() <= 1e-7


def test_decode_sdf_matches_reference_golden(engine):
    import torch
    from distr import functions
    g = np.load(os.path.join(GOLDEN, 'g2_decode_sdf.npz'))
    got = functions.mlp_eval(engine, torch.from_numpy(g['latent']), torch.from_numpy(g['points'])).cpu().numpy().reshape(-1)
    assert np.abs(got - g['sdf']).max() <= 2e-6
    sdf, grad = functions.mlp_grad(engine, torch.from_numpy(g['latent']), torch.from_numpy(g['points']))
    sdf, grad = sdf.cpu().numpy(), grad.cpu().numpy()
    inclamp = np.abs(g['sdf']) <= 0.1 - 1e-5
    assert np.abs(3.0 * grad[inclamp] - g['gradient_x3_clamped'][inclamp]).max() <= 5e-5


def test_decode_sdf_gradient_matches_oracle(engine, cpu_oracle, fixture_decoder):
    import torch
    from distr import functions
    _, _, latent = fixture_decoder
    pts = _points(300)
    sdf, grad = functions.mlp_grad(engine, torch.from_numpy(latent), torch.from_numpy(pts))
    s_ref, g_ref = cpu_oracle.decode_sdf_and_gradient(latent, pts)
    assert np.abs(sdf.cpu().numpy() - s_ref).max() <= 1e-7
    assert np.abs(grad.cpu().numpy() - g_ref).max() <= 1e-5 * max(1.0, np.abs(g_ref).max())


CASES = [(m, d) for m in ('trivial', 'recursive', 'pyramid_recursive') for d in (False, True)]


@pytest.mark.parametrize('marcher,d2n', CASES)
def test_render_c1_matches_oracle(engine, cpu_oracle, orc, fixture_decoder, marcher, d2n):
    """C1 (64x64, 20 steps, bs=3, rotated camera): forward outputs and latent/camera gradients vs the oracle."""
    from distr import fixture
    _, _, latent = fixture_decoder
    H = W = 64
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(30, 20, 1.6, 10)
    kw = dict(march_step=20, buffer_size=3, marcher=marcher, use_depth2normal=d2n)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=1e-4, normal_p99=1e-5, max_flip_frac=0.0)
    assert res['flips'] == 0, res


G1_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'g1*_*.npz')) if re.match(r'g1[bc]?_', os.path.basename(p)))
# every G1 golden in the exact f32 arithmetic (the product default, the credited path); the two opt-in split arithmetics on a
# representative pair (one marcher with compaction + finite-difference normals, one dense marcher with autograd normals)
G1_RUNS = [(n, 'f32') for n in G1_FILES] + [(n, a) for n in ('g1_c1_pyramid_recursive_d2n.npz', 'g1_c1_trivial_agn.npz') for a in ('bf16x6', 'f16x3')]


@pytest.mark.parametrize('name,arith', G1_RUNS)
def test_render_matches_reference_goldens(engine, name, arith):
    """HIP path directly against outputs of the reference itself (tests/golden, made by oracle/gen_golden.py) -- in the default exact
    f32 arithmetic and in the opt-in split-bf16 arithmetic, both at the north-star bar (1e-4, <= 0.1 % mask flips)."""
    from distr import fixture, decoder_pack, functions
    g = dict(np.load(os.path.join(GOLDEN, name)))
    H, W = int(g['H']), int(g['W'])
    eng = engine
    if bool(g['weight_norm']):
        Ws, bs, _ = fixture.make_decoder_weights(int(g['fixture_seed']))
        Wse, bse = decoder_pack.effective_weights(decoder_pack.fixture_state_dict(Ws, bs, weight_norm=True))
        eng = functions.engine_from_weights(Wse, bse, 0)
    a = helpers.hip_render(eng, H, W, g['K'], g['R'], g['T'], g['latent'], seed=int(g['loss_seed']),
                           march_step=int(g['march_step']), buffer_size=int(g['buffer_size']), ratio=float(g['ratio']),
                           marcher=str(g['marcher']), use_depth2normal=bool(g['use_depth2normal']), arith=arith)
    b = dict(mask=g['mask'], depth=g['depth'], zdepth=g['zdepth'], min_sdf=g['min_abs_query'], normal=g['normal'],
             g_latent=g['g_latent'], g_R=g['g_R'], g_T=g['g_T'])
    fx = float(g['K'][0, 0])
    res = helpers.compare(a, b, H, W, tol_depth=1e-4, tol_grad=2e-3,
                          normal_p99=max(1e-4, 1e-5 * fx) if bool(g['use_depth2normal']) else 1e-4)
    print(name, arith, res)


def test_render_depth_and_warp_gradient_path(engine, cpu_oracle, orc, fixture_decoder):
    """render_depth() callers (render_warp) send their gradient through Zdepth of every in-sphere pixel."""
    import torch
    from distr import binding, fixture, functions
    _, _, latent = fixture_decoder
    H = W = 48
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(-20, 10, 1.6, 0)
    kw = dict(march_step=30, buffer_size=1, marcher='recursive', want_normal=False)
    dev = engine.device
    cfg = binding.make_cfg((H, W), K, **kw)
    lat = torch.from_numpy(latent).to(dev).requires_grad_(True)
    Rt = torch.from_numpy(R).to(dev).requires_grad_(True)
    Tt = torch.from_numpy(T).to(dev).requires_grad_(True)
    z, m, q, _, _ = functions.render_call(engine, cfg, lat, Rt, Tt)
    gz = torch.from_numpy(np.random.RandomState(1).rand(H * W).astype(np.float32)).to(dev)
    (z * gz)[m.bool()].sum().backward()
    ocfg = orc.make_cfg(H, W, K, **kw)
    out = cpu_oracle.render(ocfg, latent, R, T)
    assert np.array_equal(out['mask'], m.cpu().numpy())
    gl, gR, gT, _ = out['state'].backward(g_zdepth=gz.cpu().numpy() * out['mask'])
    assert np.abs(z.detach().cpu().numpy() - out['zdepth'])[out['mask'].astype(bool)].max() <= 1e-6
    for mine, ref in ((lat.grad, gl), (Rt.grad, gR), (Tt.grad, gT)):
        assert np.abs(mine.cpu().numpy().reshape(-1) - ref.reshape(-1)).max() <= 1e-4 * np.abs(ref).max()


def test_camera_inside_sphere_and_empty_view(engine, cpu_oracle, orc, fixture_decoder):
    """Edge cases of get_intersections_with_unit_spheres (renderer.py:266-268): camera inside the unit sphere
    (init depth 0 everywhere) and a camera looking away (the reference intersects LINES with the sphere, so rays still "meet" it behind the
    camera and march from a negative depth; none finds a surface: golden G25 pins that against the reference itself)."""
    from distr import fixture
    _, _, latent = fixture_decoder
    H = W = 32
    K = fixture.make_intrinsic(H, W)
    R, _ = fixture.make_camera(15, 5, 1.6, 0)
    for T in (np.array([0, 0, 0.8], np.float32), np.array([0, 0, -3.0], np.float32)):
        for marcher in ('recursive', 'pyramid_recursive'):
            kw = dict(march_step=12, buffer_size=3, marcher=marcher, use_depth2normal=False)
            a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
            b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
            helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=1e-4, normal_p99=1e-5, max_flip_frac=0.0)


def test_c3_full_size_properties(engine, fixture_decoder):
    """C3 = 512x512, 50 steps, depth2normal (the benchmark workload): properties that need no CPU reference."""
    import torch
    from distr import binding, fixture, functions
    _, _, latent = fixture_decoder
    H = W = 512
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(0, 0, 1.6, 0)
    kw = dict(march_step=50, buffer_size=3, use_depth2normal=True)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, marcher='pyramid_recursive', **kw)
    b = helpers.hip_render(engine, H, W, K, R, T, latent, marcher='pyramid_recursive', **kw)
    # idempotence: same inputs -> same bits, forward AND backward (live-ray compaction order never enters a value; the
    # gradient-sample list and every reduction of the backward have a fixed order)
    for k in ('zdepth', 'mask', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T'):
        assert np.array_equal(a[k], b[k]), k
    m = a['mask'].reshape(H, W).astype(bool)
    assert 0.15 * H * W < m.sum() < 0.40 * H * W
    # silhouette is inside the unit-sphere footprint, depth is within the sphere's extent, normals are unit length
    assert np.all(a['zdepth'][a['mask'].astype(bool)] < 1e5)
    d = a['depth'][m]
    assert d.min() > 0.6 and d.max() < 2.6
    nl = np.linalg.norm(a['normal'][m], axis=-1)
    assert np.abs(nl - 1).max() < 1e-5
    assert np.all(a['depth'][~m] == 0.0) and np.all(a['normal'][~m] == 0.0)      # depth2normal background convention
    # converged pixels carry |sdf| <= threshold
    assert np.all(np.abs(a['min_sdf'][a['mask'].astype(bool)]) <= 5e-5)
    # the dense ('trivial') and compacted ('recursive') marchers agree on the surface they find
    t = helpers.hip_render(engine, H, W, K, R, T, latent, marcher='trivial', **kw)
    r = helpers.hip_render(engine, H, W, K, R, T, latent, marcher='recursive', **kw)
    both = t['mask'].astype(bool) & r['mask'].astype(bool)
    assert (t['mask'] != r['mask']).mean() < 0.01
    # (the dense marcher keeps over-stepping around the surface after |sdf| < threshold: agreement is O(step), not O(eps))
    assert np.abs(t['zdepth'] - r['zdepth'])[both].max() < 5e-3
    # backward is linear in the upstream gradient
    dev = engine.device
    cfg = binding.make_cfg((H, W), K, marcher='pyramid_recursive', **kw)
    lat = torch.from_numpy(latent).to(dev).requires_grad_(True)
    Rt, Tt = torch.from_numpy(R).to(dev), torch.from_numpy(T).to(dev)
    z, mk, q, dep, nrm = functions.render_call(engine, cfg, lat, Rt, Tt)
    g1 = torch.rand(H, W, device=dev)
    g2 = torch.rand(H * W, device=dev)
    ga, = torch.autograd.grad((dep * g1)[mk.reshape(H, W).bool()].sum(), lat, retain_graph=True)
    gb, = torch.autograd.grad((q * g2).sum(), lat, retain_graph=True)
    gab, = torch.autograd.grad((dep * g1)[mk.reshape(H, W).bool()].sum() + (q * g2).sum(), lat)
    assert (ga + gb - gab).abs().max() <= 2e-4 * gab.abs().max()


def test_dropin_api_with_module(fixture_decoder):
    """`from core.sdfrenderer import SDFRenderer, SDFRenderer_warp` with the reference's constructor/call signatures
    (run_single_shape.py:110-117, run_multi_pmodata.py:92-100) on a weight-normed Decoder module."""
    import torch
    from core.sdfrenderer import SDFRenderer, SDFRenderer_warp
    from core.graph.deep_sdf_decoder import Decoder
    from core.utils.decoder_utils import decode_sdf, decode_sdf_gradient
    from distr import decoder_pack, fixture
    Ws, bs, latent = fixture_decoder
    dec = Decoder(256, [512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4],
                  weight_norm=True, xyz_in_all=False, use_tanh=False, latent_dropout=False)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in decoder_pack.fixture_state_dict(Ws, bs, True).items()})
    dec = torch.nn.DataParallel(dec.cuda()).module        # drivers unwrap DataParallel (decoder_utils.py:29-30)
    H = W = 64
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(25, 15, 1.6, 0)
    r = SDFRenderer(dec, K, march_step=30, buffer_size=3, threshold=5e-5, ray_marching_ratio=1.5, use_depth2normal=True)
    assert r.get_img_hw() == (H, W) and r.get_threshold() == 5e-5
    lat = torch.from_numpy(latent).cuda().requires_grad_(True)
    Rt, Tt = torch.from_numpy(R).cuda(), torch.from_numpy(T).cuda()
    depth, normal, mask, q = r.render(lat, Rt, Tt)
    assert depth.shape == (H, W) and normal.shape == (H, W, 3) and mask.shape == (H, W) and q.shape == (H, W)
    assert mask.dtype == torch.uint8 and depth.dtype == torch.float32
    (depth[mask.bool()].sum() + q.sum() + normal.sum()).backward()
    assert lat.grad is not None and torch.isfinite(lat.grad).all() and lat.grad.abs().max() > 0
    # PyTorch evaluation of the same module agrees with the fused kernel
    pts = torch.from_numpy(_points(500)).cuda()
    with torch.no_grad():
        ref = dec.inference(torch.cat([lat.detach().expand(500, -1), pts], 1))
        got = decode_sdf(dec, lat.detach(), pts, clamp_dist=None, no_grad=True)
    assert (ref - got).abs().max() < 5e-6
    g = decode_sdf_gradient(dec, lat.detach(), pts, clamp_dist=0.1, no_grad=True)
    assert g.shape == (500, 3)
    # multi-view warp loss (run_multi_pmodata.py:92: march_step=100, buffer_size=1, default 'recursive' marcher)
    rw = SDFRenderer_warp(dec, K, march_step=40, buffer_size=1)
    R2, T2 = fixture.make_camera(35, 15, 1.6, 0)
    img1, img2 = torch.rand(H, W, 3).cuda(), torch.rand(H, W, 3).cuda()
    lat2 = torch.from_numpy(latent).cuda().requires_grad_(True)
    out = rw.render_warp(lat2, Rt, Tt, torch.from_numpy(R2).cuda(), torch.from_numpy(T2).cuda(), img1, img2)
    assert len(out) == 9
    out[0].backward()
    assert lat2.grad is not None and torch.isfinite(lat2.grad).all()


SWEEP = [
    # (H, W, march_step, bs, ratio, marcher, d2n, camera(az,el,dist,roll), extra kwargs)
    (33, 47, 14, 1, 1.5, 'recursive', False, (12, -8, 1.7, 5), {}),
    (40, 40, 18, 5, 1.0, 'trivial', True, (-60, 30, 1.5, 0), {}),
    (50, 70, 30, 8, 1.5, 'pyramid_recursive', False, (0, 0, 1.6, 0), {}),
    (64, 48, 25, 3, 2.0, 'pyramid_recursive', True, (100, 45, 1.4, -20), dict(coarse_steps=(2, 4))),
    (45, 59, 22, 3, 1.5, 'pyramid_recursive', True, (-40, 15, 1.5, 12), dict(coarse_steps=(3, 0))),        # two levels: scale_list=[2, 1] (G28)
    (31, 33, 40, 5, 1.0, 'pyramid_recursive', False, (70, -25, 1.8, 0), dict(coarse_steps=(1, 0))),
    (53, 41, 30, 3, 1.5, 'pyramid_recursive', True, (15, 35, 1.6, -8), dict(scale_list=[8, 4, 2, 1], march_step_list=[1, 2, 3, -1])),      # general pyramids
    (38, 50, 26, 2, 1.5, 'pyramid_recursive', False, (-75, 5, 1.7, 0), dict(scale_list=[6, 3, 1], march_step_list=[2, 2, -1])),
    (64, 64, 36, 4, 2.0, 'pyramid_recursive', True, (140, -30, 1.5, 20), dict(scale_list=[5, 1], march_step_list=[4, -1])),
    (36, 36, 16, 3, 1.5, 'recursive', False, (20, 10, 1.6, 0), dict(use_transform=False)),
    (36, 36, 16, 3, 1.5, 'pyramid_recursive', False, (20, 10, 1.6, 0), dict(grad_camera=False)),
    (36, 36, 16, 3, 1.5, 'recursive', True, (20, 10, 1.6, 0), dict(grad_depth=False)),
    (36, 36, 16, 3, 1.5, 'pyramid_recursive', True, (20, 10, 1.6, 0), dict(grad_mask=False)),
    (36, 36, 16, 2, 1.5, 'recursive', False, (20, 10, 1.6, 0), dict(normalize_normal=False)),
    (36, 36, 16, 3, 1.5, 'pyramid_recursive', False, (20, 10, 1.6, 0), dict(threshold=1e-3, clamp_dist=0.05, radius=0.9)),
    (36, 36, 16, 3, 1.5, 'trivial', False, (20, 10, 1.6, 0),
     dict(transform_matrix=np.array([[0., 1., 0.], [1., 0., 0.], [0., 0., -1.]]))),
]


@pytest.mark.parametrize('case', range(len(SWEEP)))
def test_option_sweep_matches_oracle(engine, cpu_oracle, orc, fixture_decoder, case):
    """Less-travelled options (odd sizes, buffer_size 1..8, ratios, coarse step lists, no_grad_* flags, use_transform,
    custom transform_matrix, radius/threshold/clamp) -- HIP vs oracle, same bars as C1."""
    from distr import fixture
    H, W, S, bs, ratio, marcher, d2n, cam, extra = SWEEP[case]
    _, _, latent = fixture_decoder
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    kw = dict(march_step=S, buffer_size=bs, ratio=ratio, marcher=marcher, use_depth2normal=d2n)
    kw.update(extra)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=2e-4, normal_p99=1e-5, max_flip_frac=0.0)
    assert res['flips'] == 0, res


def test_c5_size_runs(engine, fixture_decoder):
    """C5-sized call (1024x1024, 100 march steps): workspace sizing (~3 GB incl. saved masks), step counters, and the
    same size-independent properties as C3 on the bigger grid."""
    from distr import fixture
    _, _, latent = fixture_decoder
    H = W = 1024
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(45, 25, 1.6, 0)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, march_step=100, buffer_size=3, marcher='pyramid_recursive',
                           use_depth2normal=True)
    m = a['mask'].reshape(H, W).astype(bool)
    assert 0.10 * H * W < m.sum() < 0.45 * H * W
    assert np.all(np.abs(a['min_sdf'][a['mask'].astype(bool)]) <= 5e-5)
    assert np.isfinite(a['g_latent']).all() and np.abs(a['g_latent']).max() > 0
    assert np.isfinite(a['g_R']).all() and np.isfinite(a['g_T']).all()
    st = engine.ctx.render_stats(a['cfg'], _last_ws(engine, a['cfg'], latent, R, T))
    # 6 coarse + 94 full-resolution steps: one launch each, or -- this configuration has been rendered before, so the previous render's hint
    # moved the steps of the sticky regime into the persistent tail launch -- 6 + tail_from + 1
    fine = 94
    assert st['num_march_launches'] == (100 if st['tail_from'] == fine else 6 + st['tail_from'] + 1) and st['num_in_sphere'] == H * W
    assert 0 < st['tail_from'] <= fine and st['tail_steals'] == 0
    assert st['num_valid'] == int(m.sum())


def _last_ws(engine, cfg, latent, R, T):
    """Runs one more forward through the raw C ABI and returns its workspace (for distr_get_render_stats)."""
    import ctypes as C
    import torch
    from distr import binding
    dev = engine.device
    P = cfg.H * cfg.W
    fwd_bytes, _ = engine.ctx.workspace_bytes(cfg)
    ws = torch.empty(fwd_bytes, dtype=torch.uint8, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(-1)).to(dev)
    lat, Rt, Tt = t(latent), t(R), t(T)
    z, q = torch.empty(P, device=dev), torch.empty(P, device=dev)
    mk = torch.empty(P, dtype=torch.uint8, device=dev)
    d, n = torch.empty(P, device=dev), torch.empty(3 * P, device=dev)
    p = binding.ptr
    engine.ctx.check(engine.ctx.L.distr_render_forward(engine.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(z), p(mk), p(q), p(d), p(n),
                                                      p(ws), ws.numel(), engine.ctx.stream()))
    return ws


def test_adam_single_view_matches_reference_golden(fixture_decoder):
    """G5: five Adam iterations of the single-view shape optimisation (optimize_single.py:50-84) through the drop-in
    API (SDFRenderer + core.inv_optimizer) against the trajectory the reference itself produced on CPU."""
    import torch
    from core.sdfrenderer import SDFRenderer
    from core.graph.deep_sdf_decoder import Decoder
    from core.inv_optimizer import optimize_single_view
    g = dict(np.load(os.path.join(GOLDEN, 'g5_adam_single_view.npz')))
    Ws, bs, _ = fixture_decoder
    dec = Decoder(256, [512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs))
                         for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer(dec, g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']),
                    use_depth2normal=True)
    RT = torch.from_numpy(np.concatenate([g['R'], g['T'][:, None]], 1)).cuda()
    gt_pack = {'depth': torch.from_numpy(g['gt_depth']).cuda(), 'normal': torch.from_numpy(g['gt_normal']).cuda(),
               'silhouette': torch.from_numpy(g['gt_mask']).cuda()}
    # the GT itself was rendered by the reference from latent_gt: our render of it must agree
    with torch.no_grad():
        d, n, m, q = r.render(torch.from_numpy(g['latent_gt']).cuda(), RT[:, :3], RT[:, 3], no_grad=True)
    both = (m.cpu().numpy() > 0) & (g['gt_mask'] > 0)
    assert (m.cpu().numpy() != g['gt_mask']).sum() <= 2
    assert np.abs(d.cpu().numpy() - g['gt_depth'])[both].max() <= 1e-4
    lat = torch.from_numpy(g['latent0']).cuda().requires_grad_(True)
    opt = torch.optim.Adam([lat], lr=1e-3)
    wd = dict(w_depth=10.0, w_normal=5.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
    hist = []

    def record(i, pack, loss):
        hist.append([float(pack[k].detach()) for k in ('depth', 'normal', 'mask_gt', 'mask_out', 'l2reg')] + [float(loss.detach())])
    optimize_single_view([r], None, opt, lat, RT, gt_pack, wd, optimizer_type='shape', num_iters=5, on_iteration=record, silent=True)
    hist, ref = np.array(hist), g['history']
    print('ours\n', hist, '\nreference\n', ref[:, :6])
    fl = _floors()
    for i, nm in enumerate(('depth', 'normal', 'mask_gt', 'mask_out', 'l2reg', 'loss')):
        r_ = (np.abs(hist[:, i] - ref[:, i]) / np.maximum(np.abs(ref[:, i]), 1e-30)).max()
        print('G5 %s: max relative residual over the 5 iterations %.3e (reference noise floor %.3e)' % (nm, r_, fl['g5_%s_rel' % nm]))
    print('G5 final latent: max abs residual %.3e (reference noise floor %.3e)' % (np.abs(lat.detach().cpu().numpy() - g['latent_final']).max(), fl['g5_latent_final_abs']))
    # Observed agreement: 3-6 significant digits. The first iteration is exact to ~1e-7; later ones depend on Adam, whose
    # first update is lr*sign(g): a coordinate whose gradient is ~0 can flip sign through float-summation order alone
    # (ours vs the reference's autograd), which moves the loss by ~1e-5.
    assert np.abs(hist[0, :] - ref[0, :6]).max() <= 2e-6                 # first iteration: nothing has diverged yet
    # later iterations: Adam's first update is lr*sign(g), so a coordinate whose gradient is ~0 flips through float-summation order
    # alone; the reference itself moves by the recorded floors under 1e-7 weight noise. Bar = 2 x that floor per term.
    # (The bar is met by the default kernel configuration. DISTR_SAVE_MASKS=0 changes the summation order of the latent gradient and
    # lands at a different, equally valid point of that noise: it can exceed 2 x floor on iterations 2-5 while every single-render
    # golden still passes, tests/test_gpu_knobs.py.)
    for i, nm in enumerate(('depth', 'normal', 'mask_gt', 'mask_out', 'l2reg', 'loss')):
        r_ = (np.abs(hist[:, i] - ref[:, i]) / np.maximum(np.abs(ref[:, i]), 1e-30)).max()
        assert r_ <= 2.0 * fl['g5_%s_rel' % nm], (nm, r_, fl['g5_%s_rel' % nm])
    assert np.abs(lat.detach().cpu().numpy() - g['latent_final']).max() <= 2.0 * fl['g5_latent_final_abs']


def test_bulk_sdf_grid(cpu_oracle, fixture_decoder):
    """Row f1: N^3 SDF grid for meshing through the fused decoder kernel vs the oracle, and the coarse-to-fine variant."""
    import torch
    from core.evaluation import create_sdf_grid, create_sdf_grid_speedup, get_samples
    from core.graph.deep_sdf_decoder import Decoder
    Ws, bs, latent = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs))
                         for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    lat = torch.from_numpy(latent).cuda()
    N = 32
    grid = create_sdf_grid(dec, lat, N)
    pts = get_samples(N).cpu().numpy()
    ref = cpu_oracle.decode_sdf(latent, pts, clamp_dist=0.1).reshape(N, N, N)
    assert np.abs(grid.cpu().numpy() - ref).max() <= 1e-7
    assert (ref < 0).any() and (ref > 0).any()
    fast = create_sdf_grid_speedup(dec, lat, 64)
    full = create_sdf_grid(dec, lat, 64)
    band = (fast.abs() < 0.1)
    assert band.any() and torch.equal(fast[band], full[band])                  # evaluated points are the same values
    assert torch.equal(torch.sign(fast), torch.sign(full))                      # same inside/outside everywhere
    big = create_sdf_grid(dec, lat, 128)                                        # 2.1 M points in one launch
    assert big.shape == (128, 128, 128) and torch.isfinite(big).all()

    # ---- against the REFERENCE's own create_mesh / create_mesh_speedup (core/evaluation/create_mesh.py:56-68, 110-142): the SDF
    # volumes it hands to marching cubes (G14, oracle/gen_golden_big.py g14: convert_sdf_samples_to_ply replaced by a recorder,
    # torch-1.1 integer division shimmed). Sample ordering (x slowest, z fastest, :16-33; transform (x, y, z) -> (x, z, -y), :10-14),
    # half-resolution pass, nearest upsampling, check_valid's 1.5-voxel band and the +-0.1 fill are the reference's, not a restatement.
    g = dict(np.load(os.path.join(GOLDEN, 'g14_create_mesh_speedup.npz')))
    assert np.array_equal(g['latent'], latent)
    assert np.abs(get_samples(8, transform=True).cpu().numpy() - g['samples_N8_t1']).max() <= 1e-6
    got = create_sdf_grid(dec, lat, 32).cpu().numpy()
    assert np.abs(got - g['plain_N32_t0']).max() <= 2e-6, np.abs(got - g['plain_N32_t0']).max()
    for transform in (False, True):
        want = g['speedup_N64_t%d' % int(transform)]
        got = create_sdf_grid_speedup(dec, lat, 64, transform=transform).cpu().numpy()
        evaluated = np.abs(want) != np.float32(0.1)
        assert 5000 < int(evaluated.sum()) < 64 ** 3 // 4                      # the band is a thin shell: most points are filled, not evaluated
        diff = np.abs(got - want)
        # a coarse point whose |sdf| sits within rounding of the band edge (1.5 coarse voxels = 0.0968) may be filled on one side and
        # evaluated on the other: both values are then within 0.004 of +-0.1. Everything else agrees to rounding.
        off = diff > 2e-6
        assert int(off.sum()) <= 8 and diff.max() <= 0.004, (transform, int(off.sum()), float(diff.max()))
        print('G14 create_mesh_speedup N=64 transform=%d: max |d| %.2e over %d evaluated points, %d band-edge points' %
              (transform, float(diff[~off].max()), int(evaluated.sum()), int(off.sum())))
    n = 96          # a size where most of the volume is filled: same rules, checked against the plain grid
    fast, full96 = create_sdf_grid_speedup(dec, lat, n), create_sdf_grid(dec, lat, n)
    ev = fast.abs() != 0.1
    assert 0 < int(ev.sum()) < n ** 3 // 2 and torch.equal(fast[ev], full96[ev])
    assert not np.array_equal(create_sdf_grid(dec, lat, 48, transform=True).cpu().numpy(), create_sdf_grid(dec, lat, 48).cpu().numpy())


def test_rccl_allreduce_packed_single_rank():
    """The collective path of bench.py --gpus N (backend 'nccl' = RCCL) on the one GPU this box has: process-group
    init, packed all-reduce, max-reduce and barrier must work (world size 1: sums are identities)."""
    import torch
    import torch.distributed as dist
    from distr import parallel
    if dist.is_initialized():
        pytest.skip('process group already initialised')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')
    dist.init_process_group(backend='nccl', rank=0, world_size=1)
    try:
        g = torch.arange(256, dtype=torch.float32, device='cuda').reshape(1, 256)
        loss = torch.tensor([3.5], device='cuda')
        t = dist.all_reduce(torch.ones(4, device='cuda'))          # raw RCCL call
        parallel.allreduce_packed([g, loss])                        # world size 1 -> untouched
        assert float(g.sum()) == 255 * 128 and float(loss) == 3.5
        x = torch.ones(8, device='cuda')
        dist.all_reduce(x)
        dist.barrier()
        torch.cuda.synchronize()
        assert float(x.sum()) == 8.0
    finally:
        dist.destroy_process_group()


def test_render_warp_matches_reference_golden(fixture_decoder):
    """G4: SDFRenderer_warp.render_warp (multi-view photometric warp loss, renderer_warp.py:103-144) against the
    reference's own outputs: loss, masks, min-sdf maps, visualisation normal / depth, latent gradient."""
    import torch
    from core.sdfrenderer import SDFRenderer_warp
    from core.graph.deep_sdf_decoder import Decoder
    g = dict(np.load(os.path.join(GOLDEN, 'g4_render_warp.npz')))
    Ws, bs, _ = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs))
                         for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer_warp(dec, g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']))
    c = lambda k: torch.from_numpy(g[k]).cuda()
    lat = c('latent').requires_grad_(True)
    out = r.render_warp(lat, c('R1'), c('T1'), c('R2'), c('T2'), c('img1'), c('img2'), no_grad_normal=True)
    loss_color, c1, c2, m1, m2, q1, q2, n1, d1 = out
    loss_color.backward()
    assert (m1.cpu().numpy() != g['mask1']).sum() <= 1 and (m2.cpu().numpy() != g['mask2']).sum() <= 1
    both = (m1.cpu().numpy() > 0) & (g['mask1'] > 0)
    assert abs(float(loss_color) - float(g['loss_color'])) <= 2e-5
    assert np.abs(q1.detach().cpu().numpy() - g['min_sdf1']).max() <= 1e-4
    assert np.abs(q2.detach().cpu().numpy() - g['min_sdf2']).max() <= 1e-4
    assert np.abs(d1.cpu().numpy() - g['depth1'])[both].max() <= 1e-4
    dn = np.abs(n1.cpu().numpy() - g['normal1'])[both]
    assert np.percentile(dn, 99) <= 1e-4
    assert np.abs(c1.cpu().numpy() - g['color_valid_1']).max() <= 1e-5 or (np.abs(c1.cpu().numpy() - g['color_valid_1']) > 1e-5).sum() <= 6
    rel = np.abs(lat.grad.cpu().numpy() - g['g_latent']).max() / np.abs(g['g_latent']).max()
    fl = _floors()
    print('G4 residual vs reference: g_latent %.3e (reference noise floor %.3e), loss %.3e (floor %.3e), min_sdf %.3e (floor %.3e)'
          % (rel, fl['g4_g_latent_rel'], abs(float(loss_color) - float(g['loss_color'])), fl['g4_loss_abs'],
             np.abs(q1.detach().cpu().numpy() - g['min_sdf1']).max(), fl['g4_min_sdf']))
    assert rel <= 2.0 * fl['g4_g_latent_rel'], (rel, fl['g4_g_latent_rel'])          # bar = 2 x the reference's own noise floor
    assert np.abs(q1.detach().cpu().numpy() - g['min_sdf1']).max() <= 2.0 * fl['g4_min_sdf']


BAND_CASES = [  # (H, W, bands, marcher, d2n)
    (128, 96, [(0, 32), (32, 100), (100, 128)], 'pyramid_recursive', True),
    (70, 64, [(0, 36), (36, 70)], 'pyramid_recursive', True),          # H not a multiple of 4: last band ends at H
    (96, 96, [(0, 48), (48, 96)], 'recursive', False),
    (64, 64, [(0, 16), (16, 32), (32, 48), (48, 64)], 'trivial', True),
    # pyramids whose coarsest scale divides the bands' 4-row alignment: two levels (compact and general form), ratio 4 in one step
    (90, 64, [(0, 40), (40, 90)], 'pyramid_recursive', True, dict(coarse_steps=(3, 0))),
    (76, 60, [(0, 24), (24, 52), (52, 76)], 'pyramid_recursive', False, dict(scale_list=[4, 1], march_step_list=[3, -1])),
]


def _check_bands(engine, fixture_decoder, H, W, bands, marcher, d2n, pyr, march_step=24, buffer_size=3, ratio=1.5, cam=(20.0, 15.0, 1.7, 5.0), min_valid=51):
    import torch
    from distr import binding, fixture, functions
    Ws, bs, latent = fixture_decoder
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    dev = engine.device
    cfg = binding.make_cfg((H, W), K, march_step=march_step, buffer_size=buffer_size, ratio=ratio, marcher=marcher, use_depth2normal=d2n, **pyr)
    wd, wq, wn = (torch.from_numpy(a).to(dev) for a in helpers.loss_weights(H, W, 3))

    def run(r0, r1):
        lat = torch.from_numpy(latent).to(dev).requires_grad_(True)
        Rt = torch.from_numpy(R).to(dev).requires_grad_(True)
        Tt = torch.from_numpy(T).to(dev).requires_grad_(True)
        if (r0, r1) == (0, H):
            z, m, q, d, n = functions.render_call(engine, cfg, lat, Rt, Tt)
        else:
            z, m, q, d, n = functions.render_band_call(engine, cfg, lat, Rt, Tt, r0, r1)
        rows = r1 - r0
        mb = m.reshape(rows, W).bool()
        L = torch.where(mb, d * wd[r0:r1], torch.zeros_like(d)).sum() + (q.reshape(rows, W) * wq[r0:r1]).sum() + (n * wn[r0:r1]).sum()
        L.backward()
        out = [t.detach().reshape(rows, -1).cpu().numpy() for t in (z, m, q, d, n)]
        return out, [g.grad.double().cpu().numpy() for g in (lat, Rt, Tt)]

    full, gfull = run(0, H)
    assert full[1].sum() >= min_valid
    parts = [run(r0, r1) for (r0, r1) in bands]
    for k, name in enumerate(('zdepth', 'mask', 'min_sdf', 'depth', 'normal')):
        cat = np.concatenate([p[0][k] for p in parts], axis=0)
        assert cat.shape == full[k].shape
        assert cat.tobytes() == full[k].tobytes(), name
    for k, name in enumerate(('g_latent', 'g_R', 'g_T')):
        tot = sum(p[1][k] for p in parts)
        scale = np.abs(gfull[k]).max()
        if scale > 0:
            rel = np.abs(tot - gfull[k]).max() / scale
            assert rel < 2e-5, (name, rel)
        else:
            assert np.abs(tot).max() == 0, name


@pytest.mark.gpu
@pytest.mark.parametrize('case', range(len(BAND_CASES)))
def test_row_bands_equal_full_render(engine, fixture_decoder, case):
    """SURVEY.md 8e (strong scaling of one view): a partition of the image into row bands, each rendered on its own
    (what separate ranks do), reproduces the full render bit for bit per pixel, and the bands' input gradients sum to the
    full render's (different summation order only)."""
    H, W, bands, marcher, d2n = BAND_CASES[case][:5]
    pyr = BAND_CASES[case][5] if len(BAND_CASES[case]) > 5 else {}
    _check_bands(engine, fixture_decoder, H, W, bands, marcher, d2n, pyr)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_BANDS', '4'))))      # (soak runs: more seeds)
def test_random_row_band_partitions_equal_full_render(engine, fixture_decoder, seed):
    """Seeded random partitions of ragged images into 2..6 row bands (cuts at multiples of 4 rows), every marcher, pyramids whose coarsest
    scale divides the 4-row alignment, both normal modes, random cameras: the bands reproduce the full render byte for byte and their
    gradients sum to the full render's."""
    rs = np.random.RandomState(15000 + seed)
    H, W = int(rs.randint(24, 160)), int(rs.randint(17, 140))
    marcher = ['recursive', 'pyramid_recursive', 'pyramid_recursive', 'trivial'][rs.randint(4)]
    S = int(rs.randint(12, 70)) if marcher != 'trivial' else int(rs.randint(6, 14))
    pyr = {}
    if marcher == 'pyramid_recursive':
        pyr = [dict(), dict(coarse_steps=(int(rs.randint(1, 4)), int(rs.randint(1, 4)))), dict(scale_list=[2, 1], march_step_list=[int(rs.randint(1, 5)), -1]),
               dict(scale_list=[4, 1], march_step_list=[int(rs.randint(1, 4)), -1])][rs.randint(4)]
    units = (H + 3) // 4
    ncut = int(rs.randint(1, min(6, units)))
    cuts = sorted(set(int(c) * 4 for c in rs.choice(np.arange(1, units), size=min(ncut, units - 1), replace=False)))
    edges = [0] + cuts + [H]
    bands = [(a, b) for a, b in zip(edges[:-1], edges[1:])]
    cam = (float(rs.uniform(-180, 180)), float(rs.uniform(-50, 50)), float(rs.uniform(1.4, 2.0)), float(rs.uniform(-15, 15)))
    _check_bands(engine, fixture_decoder, H, W, bands, marcher, bool(rs.randint(2)), pyr, march_step=S, buffer_size=int(rs.randint(1, 6)),
                 ratio=float(rs.choice([1.0, 1.5, 2.0])), cam=cam, min_valid=0)


@pytest.mark.gpu
def test_general_form_of_the_default_pyramid_is_the_compact_form(engine, fixture_decoder):
    """distr_render_cfg.num_levels = 3 / level_scale = [4, 2, 1] / level_steps (the general description, include/distr.h ABI 6) renders byte
    for byte what coarse_steps = (a, b) renders -- outputs and gradients, tail launch included -- and [2, 1] the same against (s, 0)."""
    from distr import fixture
    _, _, latent = fixture_decoder
    for (H, W, S) in ((70, 54, 40), (137, 137, 100)):
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(35, -10, 1.6, 5)
        base = dict(march_step=S, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, ratio=1.5)
        for compact, general in ((dict(coarse_steps=(2, 4)), dict(scale_list=[4, 2, 1], march_step_list=[2, 4, -1])),
                                 (dict(coarse_steps=(5, 0)), dict(scale_list=[2, 1], march_step_list=[5, -1]))):
            a = helpers.hip_render(engine, H, W, K, R, T, latent, **base, **compact)
            b = helpers.hip_render(engine, H, W, K, R, T, latent, **base, **general)
            for k in ('zdepth', 'mask', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T'):
                assert np.asarray(a[k]).tobytes() == np.asarray(b[k]).tobytes(), (H, compact, k)


PYRAMID_MENU = ([2, 1], [3, 1], [4, 1], [8, 1], [4, 2, 1], [6, 2, 1], [6, 3, 1], [8, 2, 1], [9, 3, 1], [8, 4, 2, 1], [12, 6, 2, 1], [12, 4, 2, 1], [16, 8, 4, 1], [64, 8, 1])


@pytest.mark.gpu
def test_early_break_matches_reference_golden(engine, fixture_decoder):
    """G29: the reference's own outputs and gradients where its march breaks below buffer_size steps and the padded lists repeat the last
    step's rows (renderer.py:562-567; oracle/gen_golden_early_break.py) -- HIP at G24's bars (gradients: 1e-3 or twice the reference's floor)."""
    import test_oracle_vs_golden as tg
    from distr import fixture
    g = np.load(os.path.join(GOLDEN, 'g29_early_break.npz'))
    Ws, bs, latent = fixture_decoder
    assert fixture.weights_sha256(Ws, bs) == str(g['weights_sha256'])
    H, W = int(g['H']), int(g['W'])
    for name in sorted(tg.G29_CFG):
        a = helpers.hip_render(engine, H, W, g['K'], g['R'], g['T'], g['latent'], **tg.g29_kw(g, name))
        res = tg.check_g24(tg.g29_stable(a, g, name), g, name)
        assert abs(a['loss'] - float(g[name + '.loss'])) <= 5e-5 * abs(float(g[name + '.loss'])), name
        print('G29', name, {k: '%.1e' % v for k, v in res.items()})


@pytest.mark.gpu
@pytest.mark.parametrize('marcher,bs,d2n', [('recursive', 7, True), ('recursive', 8, False), ('pyramid_recursive', 8, True), ('recursive', 3, True)])
def test_early_break_below_buffer_size_steps(engine, cpu_oracle, orc, fixture_decoder, marcher, bs, d2n):
    """renderer.py:562-567: when every ray has finished after L < buffer_size full-resolution steps the reference pads its lists by REPEATING
    the last step's rows, and the selection then holds copies of a ray's last row -- evaluated again, each with the row's coefficient: same values,
    other gradients (up to 40 % before this was restated in k_bwd_prep / k_finalize: early_dup). A camera inside the sphere next to the surface
    with exact sphere tracing (ratio 1) ends the march after 4-6 steps. HIP vs oracle (the oracle equals the reference here: 2.5e-4 on the
    gradient of such a case, measured in the build container); buffer_size 3 is the control (the march outlasts it)."""
    from distr import fixture
    H, W = 55, 79
    _, _, latent = fixture_decoder
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(-73.8, -7.6, 0.475, 26.75)
    kw = dict(march_step=30, buffer_size=bs, ratio=1.0, marcher=marcher, use_depth2normal=d2n, threshold=1.5e-3, radius=1.2, clamp_dist=0.2)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=1e-4, normal_p99=1e-5)
    assert res['flips'] == 0 and int(b['mask'].sum()) > 500
    # the premise: the oracle's march ended after 4 full-resolution steps (1 behind the coarse levels of the pyramid) -- below the large buffers
    executed = len(list(b['state'].live_counts)) - (6 if marcher == 'pyramid_recursive' else 0)
    assert executed == (1 if marcher == 'pyramid_recursive' else 4)


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_OPTIONS', '6'))))      # (soak runs: more seeds)
def test_random_options_match_oracle(engine, cpu_oracle, orc, fixture_decoder, seed):
    """Seeded random draws over the renderer's OPTIONS (SDFRenderer.__init__ / render keywords, renderer.py:13-59, 943-999): threshold, sphere
    radius, clamp_dist, a random rotation or axis permutation as transform_matrix, use_transform, normalize_normal, the three no_grad flags,
    an off-centre principal point with fx != fy, cameras from inside the sphere to far outside, buffer sizes up to the maximum -- HIP vs oracle,
    zero mask flips (the fixed option cases are pinned to the reference by G24 / G18; this sweeps their combinations)."""
    from distr import fixture
    rs = np.random.RandomState(21000 + seed)
    H, W = int(rs.randint(17, 110)), int(rs.randint(17, 110))
    marcher = ['recursive', 'pyramid_recursive', 'pyramid_recursive', 'trivial'][rs.randint(4)]
    S = int(rs.randint(12, 70)) if marcher != 'trivial' else int(rs.randint(6, 16))
    kw = dict(march_step=S, buffer_size=int(rs.randint(1, 9)), ratio=float(rs.choice([1.0, 1.5, 2.0])), marcher=marcher, use_depth2normal=bool(rs.randint(2)),
              threshold=float(10 ** rs.uniform(-5, -2.5)), radius=float(rs.uniform(0.85, 1.3)), clamp_dist=float(rs.uniform(0.03, 0.3)),
              use_transform=bool(rs.randint(4) != 0), normalize_normal=bool(rs.randint(3) != 0),
              grad_depth=bool(rs.randint(4) != 0), grad_mask=bool(rs.randint(4) != 0), grad_camera=bool(rs.randint(4) != 0))
    kw['buffer_size'] = min(kw['buffer_size'], S)            # (fewer rows than selected rows: refused, the reference's top-k raises)
    t = rs.randint(3)
    if t == 1:
        perm = rs.permutation(3)
        M = np.zeros((3, 3)); M[np.arange(3), perm] = rs.choice([-1.0, 1.0], 3)
        kw['transform_matrix'] = M
    elif t == 2:
        q, _ = np.linalg.qr(rs.standard_normal((3, 3)))
        kw['transform_matrix'] = q
    if marcher == 'pyramid_recursive' and rs.randint(2):
        kw['coarse_steps'] = (int(rs.randint(1, 4)), int(rs.randint(1, 4)))
    cam = (float(rs.uniform(-180, 180)), float(rs.uniform(-60, 60)), float(rs.choice([rs.uniform(0.3, 0.8), rs.uniform(1.2, 2.4), rs.uniform(1.2, 2.4)])), float(rs.uniform(-30, 30)))
    _, _, latent = fixture_decoder
    K = np.array(fixture.make_intrinsic(H, W), dtype=np.float64)
    K[0, 0] *= rs.uniform(0.8, 1.25); K[1, 1] *= rs.uniform(0.8, 1.25)
    K[0, 2] += rs.uniform(-0.15, 0.15) * W; K[1, 2] += rs.uniform(-0.15, 0.15) * H
    R, T = fixture.make_camera(*cam)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-5, tol_grad=1e-3, normal_p99=1e-4)
    print(seed, (H, W), marcher, S, res)
    assert res['flips'] == 0


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_PYRAMIDS', '8'))))
def test_random_pyramids_match_oracle(engine, cpu_oracle, orc, fixture_decoder, seed):
    """Seeded random draws over the pyramid (2..4 levels, ratios 2..8 from a menu), its step counts, a ragged image size, buffer_size,
    normal mode and camera: HIP vs oracle, zero mask flips (renderer.py:713-805 through include/distr.h's general description)."""
    from distr import fixture
    rs = np.random.RandomState(7000 + seed)
    sl = list(PYRAMID_MENU[rs.randint(len(PYRAMID_MENU))])
    H, W = int(rs.randint(24, 120)), int(rs.randint(24, 120))
    msl = [int(rs.randint(1, 5)) for _ in sl[:-1]] + [-1]
    S = sum(msl[:-1]) + int(rs.randint(6, 40))
    kw = dict(march_step=S, buffer_size=int(rs.randint(1, 6)), ratio=float(rs.choice([1.0, 1.5, 2.0])), marcher='pyramid_recursive',
              use_depth2normal=bool(rs.randint(2)), scale_list=sl, march_step_list=msl)
    cam = (float(rs.uniform(-180, 180)), float(rs.uniform(-60, 60)), float(rs.uniform(1.3, 2.2)), float(rs.uniform(-30, 30)))
    if int(os.environ.get('DISTR_TEST_PYRAMIDS_WIDE', '0')):     # (soak: cameras inside the sphere and buffer sizes to 8 -- the early-break regime -- on top)
        rs2 = np.random.RandomState(77000 + seed)
        kw['buffer_size'] = int(rs2.randint(1, 9))
        if rs2.randint(3) == 0:
            cam = (cam[0], cam[1], float(rs2.uniform(0.3, 0.8)), cam[3])
    _, _, latent = fixture_decoder
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-5, tol_grad=1e-3, normal_p99=1e-4)
    print(seed, (H, W), sl, msl, S, res)
    assert res['flips'] == 0


@pytest.mark.gpu
@pytest.mark.parametrize('case', [((27, 31), [64, 8, 1], [4, 3, -1], 20, 5, 2.0, False, (95.23782009672021, 7.970981355321783, 2.170534567202053, 28.679273172187123)),
                                  ((104, 27), [64, 8, 1], [2, 1, -1], 30, 2, 1.0, True, (-8.197295606877248, 55.52519337785796, 1.849839080548518, 22.018391808824433))])
def test_coarsest_level_without_a_hit(engine, cpu_oracle, orc, fixture_decoder, case):
    """A steep pyramid whose coarsest level (1 x 1 / 2 x 1 pixels) has no pixel centre on the unit sphere while full-resolution rays have: the
    reference raises there (renderer.py:271, max() of an empty tensor -- measured, oracle/distr_oracle.cpp build_level). The MI355X path cannot
    know without a host sync and defines the level's fill depth as 0 (its rays start at the camera); the oracle restates that. Found by the
    400-seed pyramid soak of round 6 (seeds 115, 143): HIP == oracle, every output finite, a non-empty image."""
    from distr import fixture
    (H, W), sl, msl, S, bsz, ratio, d2n, cam = case
    _, _, latent = fixture_decoder
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    kw = dict(march_step=S, buffer_size=bsz, ratio=ratio, marcher='pyramid_recursive', use_depth2normal=d2n, scale_list=sl, march_step_list=msl)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    assert int(b['mask'].sum()) > 0
    for k in ('zdepth', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T'):
        assert np.isfinite(np.asarray(a[k])).all() and np.isfinite(np.asarray(b[k])).all(), k
    res = helpers.compare(a, b, H, W, tol_depth=1e-5, tol_grad=1e-3, normal_p99=1e-4)
    assert res['flips'] == 0


@pytest.mark.gpu
def test_buffer_size_above_the_marched_rows_is_refused(engine, fixture_decoder):
    """march_step 6 with buffer_size 8: the reference's torch.topk raises at render time (renderer.py:314-318, k above the number of rows);
    here check_cfg refuses the configuration (DISTR_ERR_INVALID_ARG) for every marcher -- found by the F2 random sweep ('trivial' with 6 / 7 steps
    and buffer_size 8: the kernels and the oracle each did something, not the same thing). A pyramid's coarse rows count."""
    from distr import binding, fixture
    K = fixture.make_intrinsic(48, 48)
    for marcher, S, bs, ok in (('trivial', 6, 8, False), ('recursive', 7, 8, False), ('trivial', 8, 8, True), ('pyramid_recursive', 7, 8, False),
                               ('pyramid_recursive', 8, 8, True), ('recursive', 5, 5, True)):
        cfg = binding.make_cfg((48, 48), K, march_step=S, buffer_size=bs, marcher=marcher)
        if ok:
            engine.ctx.workspace_bytes(cfg)
        else:
            with pytest.raises(binding.DistrError):
                engine.ctx.workspace_bytes(cfg)


@pytest.mark.gpu
def test_row_band_argument_checks(engine, fixture_decoder):
    from distr import binding, fixture
    K = fixture.make_intrinsic(64, 64)
    for band in [(2, 16), (0, 18), (60, 8), (-4, 8)]:
        cfg = binding.make_cfg((64, 64), K, band=band)
        with pytest.raises(binding.DistrError):
            engine.ctx.workspace_bytes(cfg)
    f_full, _ = engine.ctx.workspace_bytes(binding.make_cfg((64, 64), K))
    f_band, _ = engine.ctx.workspace_bytes(binding.make_cfg((64, 64), K, band=(16, 32)))
    assert f_band < 0.6 * f_full


# ------------------------------------------------------------------------------------------ fused losses (rows f2, f3)
def _single_loss_hip(engine, g, weights):
    import torch
    from distr import functions
    c = lambda k: None if g.get(k) is None else torch.from_numpy(np.ascontiguousarray(g[k])).cuda()
    d, n, q = (c(k).requires_grad_(True) for k in ('depth', 'normal', 'min_sdf'))
    terms = functions.single_view_losses(engine, d, n, c('mask'), q, c('gt_depth'), c('gt_normal'), c('gt_mask'), float(g['threshold']))
    (terms * torch.from_numpy(np.asarray(weights, np.float32)).cuda()).sum().backward()
    return terms.detach().cpu().numpy(), d.grad.cpu().numpy(), n.grad.cpu().numpy(), q.grad.cpu().numpy()


@pytest.mark.gpu
def test_fused_single_view_losses_match_reference_golden(engine):
    """Row f3 / G7: distr_single_loss_forward/_backward against the reference's compute_loss_mask/_depth/_normal."""
    g = dict(np.load(os.path.join(GOLDEN, 'g7_single_losses.npz')))
    terms, gd, gn, gq = _single_loss_hip(engine, g, g['weights'])
    assert np.allclose(terms, g['losses'], rtol=2e-6, atol=1e-9), (terms, g['losses'])
    for name, a in (('g_depth', gd), ('g_normal', gn), ('g_min_sdf', gq)):
        assert np.abs(a - g[name]).max() <= 2e-6 * np.abs(g[name]).max(), name
    # empty sets (identical masks, no depth / normal ground truth): every term 0, every gradient 0
    g2 = dict(g, gt_mask=g['mask'], gt_depth=None, gt_normal=None)
    terms, gd, gn, gq = _single_loss_hip(engine, g2, g['weights'])
    assert (terms == 0).all() and (gd == 0).all() and (gn == 0).all() and (gq == 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize('size', [(37, 53), (512, 512)])
def test_fused_single_view_losses_match_oracle(engine, size):
    """Seeded random inputs at an odd size and at the C3 size against the CPU oracle (float summation order differs)."""
    import torch
    from oracle import loss_oracle
    H, W = size
    rs = np.random.RandomState(H * 1000 + W)
    g = dict(threshold=5e-5, depth=(1 + rs.rand(H, W)).astype(np.float32), normal=rs.standard_normal((H, W, 3)).astype(np.float32),
             mask=(rs.rand(H, W) < 0.5).astype(np.uint8), min_sdf=(2e-4 * rs.standard_normal((H, W))).astype(np.float32),
             gt_depth=np.where(rs.rand(H, W) < 0.9, 1 + rs.rand(H, W), 0).astype(np.float32),
             gt_normal=rs.standard_normal((H, W, 3)).astype(np.float32), gt_mask=(rs.rand(H, W) < 0.5).astype(np.uint8))
    g['normal'][rs.rand(H, W) < 0.1] = 0
    w = [1.0, 2.0, 10.0, 5.0]
    terms, gd, gn, gq = _single_loss_hip(engine, g, w)
    d, n, q = (torch.from_numpy(g[k]).clone().requires_grad_(True) for k in ('depth', 'normal', 'min_sdf'))
    ref = loss_oracle.single_view_losses(d, n, torch.from_numpy(g['mask']), q, torch.from_numpy(g['gt_depth']),
                                         torch.from_numpy(g['gt_normal']), torch.from_numpy(g['gt_mask']), 5e-5)
    sum(wi * t for wi, t in zip(w, ref)).backward()
    assert np.allclose(terms, [float(t) for t in ref], rtol=2e-5, atol=1e-9)
    for a, t in ((gd, d), (gn, n), (gq, q)):
        assert np.abs(a - t.grad.numpy()).max() <= 1e-5 * np.abs(t.grad.numpy()).max()
    # bit-reproducible (ordered reductions)
    again = _single_loss_hip(engine, g, w)
    assert all(a.tobytes() == b.tobytes() for a, b in zip((terms, gd, gn, gq), again))


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_LOSSES', '4'))))      # (soak runs: more seeds)
def test_random_single_view_losses_match_oracle(engine, seed):
    """Seeded random inputs of the fused single-view losses (loss_utils.py:27-172): ragged sizes, mask densities from empty to full (every
    set of the four terms can be empty), missing ground-truth depth, zero normals, random weights and threshold -- terms and the three input
    gradients against the torch restatement G7 pins to the reference."""
    import torch
    from oracle import loss_oracle
    rs = np.random.RandomState(41000 + seed)
    H, W = int(rs.randint(5, 200)), int(rs.randint(5, 200))
    pm, pg, pd = (float(rs.choice([0.0, 1.0, rs.uniform(0.02, 0.98), rs.uniform(0.02, 0.98)])) for _ in range(3))
    thr = float(10 ** rs.uniform(-5, -3))
    g = dict(threshold=thr, depth=(1 + rs.rand(H, W)).astype(np.float32), normal=rs.standard_normal((H, W, 3)).astype(np.float32),
             mask=(rs.rand(H, W) < pm).astype(np.uint8), min_sdf=(4 * thr * rs.standard_normal((H, W))).astype(np.float32),
             gt_depth=np.where(rs.rand(H, W) < pd, 1 + rs.rand(H, W), 0).astype(np.float32),
             gt_normal=rs.standard_normal((H, W, 3)).astype(np.float32), gt_mask=(rs.rand(H, W) < pg).astype(np.uint8))
    g['normal'][rs.rand(H, W) < 0.1] = 0
    g['gt_normal'][rs.rand(H, W) < 0.05] = 0
    w = [float(v) for v in rs.uniform(0.1, 10.0, 4)]
    terms, gd, gn, gq = _single_loss_hip(engine, g, w)
    d, n, q = (torch.from_numpy(g[k]).clone().requires_grad_(True) for k in ('depth', 'normal', 'min_sdf'))
    ref = loss_oracle.single_view_losses(d, n, torch.from_numpy(g['mask']), q, torch.from_numpy(g['gt_depth']),
                                         torch.from_numpy(g['gt_normal']), torch.from_numpy(g['gt_mask']), thr)
    total = sum(wi * t for wi, t in zip(w, ref))
    if total.requires_grad:
        total.backward()
    assert np.allclose(terms, [float(t) for t in ref], rtol=5e-5, atol=1e-9), (seed, terms, [float(t) for t in ref])
    for a, t in ((gd, d), (gn, n), (gq, q)):
        rg = t.grad.numpy() if t.grad is not None else np.zeros_like(a)
        assert np.abs(a - rg).max() <= 2e-5 * max(np.abs(rg).max(), 1e-20), seed


def _warp_hip(engine, g, g_loss):
    import torch
    from distr import binding, functions
    H, W = int(g['H']), int(g['W'])
    c = lambda k: torch.from_numpy(np.ascontiguousarray(np.asarray(g[k], np.float32))).cuda()
    z1, R1, T1, R2, T2 = (c(k).requires_grad_(True) for k in ('zdepth1', 'R1', 'T1', 'R2', 'T2'))
    wcfg = binding.make_warp_cfg((H, W), g['K'], float(g['thres_depth']))
    loss, keep, c1, c2 = functions.warp_loss(engine, wcfg, z1, torch.from_numpy(g['mask1']).cuda(), c('zdepth2'), c('img1'), c('img2'),
                                             R1, T1, R2, T2)
    (float(g_loss) * loss).backward()
    return (float(loss), keep.cpu().numpy(), c1.cpu().numpy(), c2.cpu().numpy(),
            [t.grad.cpu().numpy() for t in (z1, R1, T1, R2, T2)])


@pytest.mark.gpu
def test_fused_warp_loss_matches_reference_golden(engine):
    """Row f2 / G8: distr_warp_loss_forward/_backward against the reference's get_valid_points + compute_loss_color:
    loss, kept set, sampled colours, gradients to the view-1 depth and all four camera tensors."""
    g = dict(np.load(os.path.join(GOLDEN, 'g8_warp_loss.npz')))
    loss, keep, c1, c2, grads = _warp_hip(engine, g, g['g_loss'])
    assert (keep != g['keep']).sum() == 0
    assert abs(loss - float(g['loss_color'])) <= 2e-6
    assert np.abs(c1 - g['color_valid_1']).max() == 0 and np.abs(c2 - g['color_valid_2']).max() <= 2e-5
    for name, a in zip(('g_zdepth1', 'g_R1', 'g_T1', 'g_R2', 'g_T2'), grads):
        assert np.abs(a.reshape(-1) - g[name].reshape(-1)).max() <= 2e-4 * np.abs(g[name]).max(), name


@pytest.mark.gpu
def test_fused_warp_loss_matches_oracle_and_edge_cases(engine):
    import torch
    from oracle import loss_oracle
    from oracle.gen_synth import sphere_view, procedural_images
    from distr import fixture
    H, W = 96, 128
    K = fixture.make_intrinsic(H, W)
    R1, T1 = fixture.make_camera(-20, 10, 1.5, 0)
    R2, T2 = fixture.make_camera(-5, 25, 1.8, -4.0)
    z1, hit1 = sphere_view(K, R1, T1, H, W, 0.6, (0.0, 0.05, -0.04))
    z2, _ = sphere_view(K, R2, T2, H, W, 0.6, (0.0, 0.05, -0.04))
    img1, img2 = procedural_images(H, W)
    g = dict(H=H, W=W, K=K, R1=R1, T1=T1, R2=R2, T2=T2, thres_depth=1e-3, zdepth1=z1, mask1=hit1.astype(np.uint8), zdepth2=z2,
             img1=img1, img2=img2)
    loss, keep, c1, c2, grads = _warp_hip(engine, g, 1.0)
    t = lambda k: torch.from_numpy(np.asarray(g[k], np.float32)).clone().requires_grad_(True)
    tz, tR1, tT1, tR2, tT2 = t('zdepth1'), t('R1'), t('T1'), t('R2'), t('T2')
    ol, okeep, oc1, oc2 = loss_oracle.warp_loss(K, H, W, tz, torch.from_numpy(g['mask1']), torch.from_numpy(z2), torch.from_numpy(img1),
                                                torch.from_numpy(img2), tR1, tT1, tR2, tT2, 1e-3)
    ol.backward()
    flips = int((keep.astype(bool) != okeep.numpy()).sum())
    assert keep.sum() > 500 and flips <= 2, flips
    assert abs(loss - float(ol)) <= 1e-5 + 1e-3 * flips
    if flips == 0:
        for a, v in zip(grads, (tz, tR1, tT1, tR2, tT2)):
            assert np.abs(a.reshape(-1) - v.grad.numpy().reshape(-1)).max() <= 5e-4 * np.abs(v.grad.numpy()).max()
    # no valid pixel in view 1: loss 0, zero gradients (renderer_warp.py:111-113)
    g0 = dict(g, mask1=np.zeros(H * W, np.uint8))
    loss0, keep0, _, _, grads0 = _warp_hip(engine, g0, 1.0)
    assert loss0 == 0.0 and keep0.sum() == 0 and all((a == 0).all() for a in grads0)
    # every point fails the depth test: mean over an empty set is NaN, as torch.mean of an empty tensor
    g1 = dict(g, zdepth2=(z2 + 5.0).astype(np.float32))
    loss1, keep1, _, _, _ = _warp_hip(engine, g1, 1.0)
    assert np.isnan(loss1) and keep1.sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_WARP', '4'))))      # (soak runs: more seeds)
def test_random_warp_losses_match_oracle(engine, seed):
    """Seeded random view pairs of a synthetic sphere (ragged sizes, camera pairs from near-identical to 60 degrees apart, sphere size and offset,
    depth-test threshold, off-centre intrinsics): distr_warp_loss_* against the torch restatement of renderer_warp.py:18-101 that G8 pins to the
    reference -- kept set (<= 2 flips at the depth test), loss, sampled colours, gradients to the view-1 depth and the four camera tensors."""
    import torch
    from oracle import loss_oracle
    from oracle.gen_synth import sphere_view, procedural_images
    from distr import fixture
    rs = np.random.RandomState(31000 + seed)
    H, W = int(rs.randint(24, 160)), int(rs.randint(24, 160))
    K = np.array(fixture.make_intrinsic(H, W), dtype=np.float64)
    K[0, 0] *= rs.uniform(0.85, 1.2); K[1, 1] *= rs.uniform(0.85, 1.2); K[0, 2] += rs.uniform(-0.1, 0.1) * W; K[1, 2] += rs.uniform(-0.1, 0.1) * H
    az, el, dist = float(rs.uniform(-180, 180)), float(rs.uniform(-40, 40)), float(rs.uniform(1.4, 2.2))
    R1, T1 = fixture.make_camera(az, el, dist, float(rs.uniform(-10, 10)))
    R2, T2 = fixture.make_camera(az + float(rs.uniform(-60, 60)), el + float(rs.uniform(-25, 25)), dist * float(rs.uniform(0.85, 1.2)), float(rs.uniform(-10, 10)))
    rad, ctr = float(rs.uniform(0.35, 0.75)), tuple(float(v) for v in rs.uniform(-0.08, 0.08, 3))
    z1, hit1 = sphere_view(K, R1, T1, H, W, rad, ctr)
    z2, _ = sphere_view(K, R2, T2, H, W, rad, ctr)
    img1, img2 = procedural_images(H, W)
    thres = float(10 ** rs.uniform(-3.5, -1.5))
    g = dict(H=H, W=W, K=K, R1=R1, T1=T1, R2=R2, T2=T2, thres_depth=thres, zdepth1=z1, mask1=hit1.astype(np.uint8), zdepth2=z2, img1=img1, img2=img2)
    loss, keep, c1, c2, grads = _warp_hip(engine, g, 1.0)
    t = lambda k: torch.from_numpy(np.asarray(g[k], np.float32)).clone().requires_grad_(True)
    tz, tR1, tT1, tR2, tT2 = t('zdepth1'), t('R1'), t('T1'), t('R2'), t('T2')
    ol, okeep, oc1, oc2 = loss_oracle.warp_loss(K, H, W, tz, torch.from_numpy(g['mask1']), torch.from_numpy(z2), torch.from_numpy(img1),
                                                torch.from_numpy(img2), tR1, tT1, tR2, tT2, thres)
    flips = int((keep.astype(bool) != okeep.numpy()).sum())
    assert flips <= 2, (seed, flips)
    if okeep.sum() == 0:
        assert keep.sum() == 0 and (np.isnan(loss) or loss == 0.0)
        return
    assert abs(loss - float(ol)) <= 1e-5 + 2e-3 * flips, (seed, loss, float(ol))
    if flips == 0:
        ol.backward()
        # |a - b| has a kink where a colour channel of the two views agrees (near-identical cameras over a smooth image): the sign of a 1e-6
        # difference is not a statement about either side -- those pixels are left out of the per-pixel gradient, and the camera sums are
        # compared only when there is no such pixel
        amb = ((np.abs(oc1.numpy() - oc2.numpy()) < 1e-4).any(axis=-1) & okeep.numpy().reshape(H, W)).reshape(-1)
        # ... and the bilinear sample has a kink where the projected point sits on a pixel-grid line of view 2 (seed 319: x = 38.99999)
        with torch.no_grad():
            Kinv = torch.from_numpy(np.linalg.inv(np.asarray(K, np.float64)).astype(np.float32))
            yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
            rays = torch.from_numpy(R1).t() @ (Kinv @ torch.stack([xx.reshape(-1), yy.reshape(-1), torch.ones(H * W)], 0))
            rays = rays / (torch.norm(rays, p=2, dim=0, keepdim=True) + 1e-12)
            pts = rays * torch.from_numpy(np.asarray(z1, np.float32)).reshape(-1)[None, :] + (-(torch.from_numpy(R1).t() @ torch.from_numpy(T1)))[:, None]
            proj = torch.from_numpy(np.asarray(K, np.float32)) @ (torch.from_numpy(R2) @ pts + torch.from_numpy(T2)[:, None])
            xy = (proj[:2] / proj[2]).numpy()
        fr = np.abs(xy - np.round(xy))
        amb |= (fr < 2e-3).any(axis=0) & okeep.numpy().reshape(-1)
        gz, ref_gz = grads[0].reshape(-1), tz.grad.numpy().reshape(-1)
        assert np.abs(gz - ref_gz)[~amb].max() <= 5e-4 * max(np.abs(ref_gz).max(), 1e-12), seed
        if not amb.any():
            for a, v in zip(grads[1:], (tR1, tT1, tR2, tT2)):
                assert np.abs(a.reshape(-1) - v.grad.numpy().reshape(-1)).max() <= 5e-4 * max(np.abs(v.grad.numpy()).max(), 1e-12), seed


class _Cam(object):
    def __init__(self, ext):
        self.extrinsic = np.asarray(ext, np.float32)


def _multi_view_setup(g):
    import torch
    from core.sdfrenderer import SDFRenderer_warp
    from core.graph.deep_sdf_decoder import Decoder
    from distr import fixture
    Ws, bs, _ = fixture.make_decoder_weights()
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs))
                         for n, a in (('weight', W), ('bias', b))})
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer_warp(dec.cuda(), g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']))
    cams = [_Cam(e) for e in g['extrinsics']]
    imgs = [torch.from_numpy(i).cuda() for i in g['images']]
    return r, cams, imgs


@pytest.mark.gpu
def test_multi_view_round_matches_reference_golden():
    """G9: two view pairs of the multi-view round (compute_loss_color_warp with a sim(3), loss_multi.py:6-49) against
    the reference: summed loss, gradients w.r.t. the shape code and the sim(3) parameters -- through the batched round (all
    four renders in one launch sequence, the default) and through the per-pair path, whose result does not depend on how many HIP
    streams the pairs are issued on."""
    import torch
    from core.inv_optimizer import multi_view_round
    from core.inv_optimizer.optimize_multi import _StreamPool
    from core.utils.train_utils import params_to_mtrx
    g = dict(np.load(os.path.join(GOLDEN, 'g9_multi_view_round.npz')))
    r, cams, imgs = _multi_view_setup(g)
    weights = {'color': float(g['w_color']), 'l2reg': float(g['w_l2reg'])}
    outs = []
    for nstreams, batched in ((0, True), (0, False), (3, False)):
        lat = torch.from_numpy(g['latent']).cuda().requires_grad_(True)
        sim3 = {'rot': torch.from_numpy(g['sim3_rot']).cuda().requires_grad_(True),
                'scale': torch.tensor(float(g['sim3_scale']), device='cuda', requires_grad=True),
                'trans': torch.from_numpy(g['sim3_trans']).cuda().requires_grad_(True)}
        m = params_to_mtrx(sim3)
        if batched:
            assert np.abs(m.detach().cpu().numpy() - g['sim_mtrx']).max() <= 1e-6
        scale = torch.norm(m[:3, :3]) / np.sqrt(3)
        total, pack = multi_view_round(r, lat, imgs, cams, [tuple(p) for p in g['pairs']], weights, sim3=m, sim3_scale=scale,
                                       pool=_StreamPool(nstreams, lat.device), batched=batched)
        total.backward()
        torch.cuda.synchronize()
        outs.append([float(total.detach())] + [t.grad.cpu().numpy() for t in (lat, sim3['rot'], sim3['scale'], sim3['trans'])])
    fl = _floors()
    for which, (tot, glat, grot, gscale, gtrans) in (('batched', outs[0]), ('per pair', outs[1])):
        assert abs(tot - float(g['loss_total'])) <= 2e-4 * abs(float(g['loss_total']))
        assert abs(float(pack['color']) - g['packs'][-1, 0]) <= 1e-4
        for name, a in (('g_latent', glat), ('g_rot', grot), ('g_scale', gscale), ('g_trans', gtrans)):
            rel = np.abs(a - g[name]).max() / np.abs(g[name]).max()
            print('G9 (%s) %s residual vs reference %.3e (reference noise floor %.3e)' % (which, name, rel, fl['g9_%s_rel' % name]))
            # bar = 2 x the reference's own floor, but not below 5e-5: the f32 summation-order level of a gradient summed over ~1e3
            # samples (one noise realisation under-estimates it for the smaller components)
            assert rel <= max(2.0 * fl['g9_%s_rel' % name], 5e-5), (which, name, rel)
    assert outs[0][0] == outs[1][0]        # the batched round: same loss bit for bit (every view's render is byte-identical) ...
    for a, b in zip(outs[0][1:], outs[1][1:]):   # ... gradients up to the order in which the views' contributions are added
        assert np.abs(a - b).max() <= 5e-6 * np.abs(b).max()
    for a, b in zip(outs[1], outs[2]):     # per-pair path: the stream count changes nothing
        assert np.asarray(a).tobytes() == np.asarray(b).tobytes()


# ------------------------------------------------------------------------------------------ colour render (row f4)
@pytest.mark.gpu
def test_decode_color_matches_oracle_bitwise_and_reference(orc):
    """distr_color_eval against the CPU oracle (same k-ordered chains -> bit-identical) at ragged sizes and against the
    reference's decode_color (G10)."""
    import torch
    from distr import fixture, functions
    g = dict(np.load(os.path.join(GOLDEN, 'g10_color_render.npz')))
    Wc, bc, code = fixture.make_color_decoder_weights(color_size=int(g['color_size']))
    eng = functions.ColorEngine(weights=(Wc, bc))
    O = orc.ColorOracle(Wc, bc)
    for n in (1, 63, 64, 65, 2048):
        pts = g['points'][:n]
        got = functions.color_eval(eng, torch.from_numpy(code).cuda(), torch.from_numpy(g['latent']).cuda(), torch.from_numpy(pts).cuda())
        want = O.decode_color(code, g['latent'], pts)
        assert got.shape == (n, 3) and got.cpu().numpy().tobytes() == want.tobytes(), n
    assert np.abs(got.cpu().numpy() - g['rgb']).max() <= 2e-6
    with pytest.raises(ValueError):
        functions.color_eval(eng, torch.from_numpy(code[:, :10]).cuda(), torch.from_numpy(g['latent']).cuda(), torch.from_numpy(pts).cuda())


@pytest.mark.gpu
def test_color_render_matches_reference_golden(fixture_decoder):
    """G10: SDFRenderer_color.render (renderer_rgb.py:73-125) through the drop-in API, plain and with a point light."""
    import torch
    from core.sdfrenderer import SDFRenderer_color
    from core.graph.deep_sdf_decoder import Decoder
    from core.utils.decoder_utils import decode_color
    from distr import fixture
    g = dict(np.load(os.path.join(GOLDEN, 'g10_color_render.npz')))
    Ws, bs, _ = fixture_decoder
    cs = int(g['color_size'])
    Wc, bc, code = fixture.make_color_decoder_weights(color_size=cs)

    def module(W, b, latent, dims, last):
        d = Decoder(latent, dims, last_dim=last, norm_layers=(), latent_in=[4])
        d.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (Wl, bl) in enumerate(zip(W, b))
                           for n, a in (('weight', Wl), ('bias', bl))})
        return d.cuda()
    dims_c = [512] * 8
    dims_c[3] += cs
    dec, dec_c = module(Ws, bs, 256, [512] * 8, 1), module(Wc, bc, 256 + cs, dims_c, 3)
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer_color(dec, dec_c, g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']))
    c = lambda k: torch.from_numpy(g[k]).cuda()
    rgb = decode_color(dec_c, c('color_code'), c('latent'), c('points'), no_grad=True)
    assert np.abs(rgb.cpu().numpy() - g['rgb']).max() <= 2e-6
    d, n, col, m, q = r.render(c('color_code'), c('latent'), c('R'), c('T'), no_grad=True)
    mm = m.cpu().numpy().astype(bool)
    both = mm & g['mask'].astype(bool)
    assert (mm != g['mask'].astype(bool)).sum() <= 1
    assert np.abs(d.cpu().numpy() - g['depth'])[both].max() <= 1e-4
    assert np.abs(q.cpu().numpy() - g['min_sdf']).max() <= 1e-4
    assert np.percentile(np.abs(n.cpu().numpy() - g['normal'])[both], 99) <= 1e-4
    assert col.shape == (H, W, 3)
    assert np.percentile(np.abs(col.cpu().numpy() - g['color'])[both], 99) <= 1e-4          # colours at surface points 1e-4 apart
    d2, n2, col2, m2, q2 = r.render(c('color_code'), c('latent'), c('R'), c('T'), no_grad=True, lighting_locations=c('lights'),
                                    lighting_energies=c('energies'))
    assert np.percentile(np.abs(col2.cpu().numpy() - g['color_shaded'])[both], 99) <= 1e-4
    # several lights (the reference only runs with one): additive in the lights
    two = torch.cat([c('lights'), c('lights') * torch.tensor([1.0, -1.0, 1.0], device='cuda')])
    col3 = r.render(c('color_code'), c('latent'), c('R'), c('T'), no_grad=True, lighting_locations=two)[2]
    a = r.render(c('color_code'), c('latent'), c('R'), c('T'), no_grad=True, lighting_locations=two[:1])[2]
    b = r.render(c('color_code'), c('latent'), c('R'), c('T'), no_grad=True, lighting_locations=two[1:])[2]
    assert np.abs((col3 - (a + b)).cpu().numpy()).max() <= 1e-5


@pytest.mark.gpu
def test_cluster_tiles_bit_identical_to_single_workgroup_tiles(fixture_decoder):
    """The deep tail of the march runs on cluster tiles (one 16-ray tile split over 8 / 4 workgroups on as many CUs,
    slices exchanged through uncached memory, csrc/distr_mlp.hpp), and once a whole step fits one launch of 8-CU clusters the
    tiles turn sticky (they keep their 16 rays to the end of the march inside that launch, sticky_tile16). Every output row is
    still one k-ordered chain, so a render must be bit-identical with sticky tiles off, with smaller clusters and with the
    cluster tiles switched off, forward and backward, with the same number of decoder evaluations; no barrier may time out."""
    import torch
    from distr import binding, fixture, functions
    Ws, bs, latent = fixture_decoder
    outs, stats, lives = [], [], []
    for env in ({}, {'DISTR_STICKY': '0'}, {'DISTR_CLUSTER': '4'}, {'DISTR_CLUSTER': '0'}):
        os.environ.update(env)
        try:
            eng = functions.engine_from_weights(Ws, bs, 0)       # the knobs are read at distr_create
        finally:
            for k_ in env:
                del os.environ[k_]
        H = W = 96
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(35.0, 25.0, 1.6, 10.0)
        res = helpers.hip_render(eng, H, W, K, R, T, latent, march_step=60, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)
        cfg = res['cfg']
        fwd_bytes, _ = eng.ctx.workspace_bytes(cfg)
        ws = torch.empty(fwd_bytes, dtype=torch.uint8, device='cuda')
        o = [torch.empty(H * W, device='cuda'), torch.empty(H * W, dtype=torch.uint8, device='cuda'), torch.empty(H * W, device='cuda'),
             torch.empty(H, W, device='cuda'), torch.empty(H, W, 3, device='cuda')]
        import ctypes as C
        p = binding.ptr
        lat = torch.from_numpy(latent).cuda().reshape(-1)
        Rt_, Tt_ = torch.from_numpy(R).cuda().reshape(-1), torch.from_numpy(T).cuda()      # kept alive across the call
        eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat), p(Rt_),
                                                   p(Tt_), p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(o[4]), p(ws), ws.numel(),
                                                   eng.ctx.stream()))
        st = eng.ctx.render_stats(cfg, ws)
        assert st['cluster_fallbacks'] == 0 and st['num_valid'] > 300
        outs.append(res)
        stats.append(st)
        lives.append(eng.ctx.live_counts(cfg, ws))
    for i in (1, 2, 3):
        assert stats[i]['num_point_evals'] == stats[0]['num_point_evals'] and lives[i] == lives[0], i    # same live-ray profile
    assert 0 < sum(lives[0][-20:]) and max(lives[0][-20:]) <= 496                                          # the tail really ran sticky
    for k in ('zdepth', 'mask', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T'):
        for i in (1, 2, 3):
            assert outs[i][k].tobytes() == outs[0][k].tobytes(), (i, k)


@pytest.mark.gpu
def test_decode_sdf_autograd_matches_reference_golden(fixture_decoder):
    """G11: decode_sdf differentiated w.r.t. the latent code and the points (distr_mlp_backward) against the reference's
    autograd, clamped and unclamped; plus ragged sizes against PyTorch autograd through the plain module."""
    import torch
    from core.graph.deep_sdf_decoder import Decoder
    from core.utils.decoder_utils import decode_sdf
    g = dict(np.load(os.path.join(GOLDEN, 'g11_decode_sdf_grad.npz')))
    Ws, bs, _ = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs))
                         for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    w = torch.from_numpy(g['weights']).cuda()
    for name, clamp in (('clamped', 0.1), ('raw', None)):
        lat = torch.from_numpy(g['latent']).cuda().requires_grad_(True)
        x = torch.from_numpy(g['points']).cuda().requires_grad_(True)
        y = decode_sdf(dec, lat, x, clamp_dist=clamp)
        assert y.shape == (777, 1) and np.abs(y.detach().cpu().numpy() - g['sdf_' + name]).max() <= 2e-6
        (y * w).sum().backward()
        # A hidden unit whose pre-activation is ~1e-8 can sit on different sides of the ReLU in our k-ordered f32 chains and
        # in the reference's GEMMs; that point's gradient then differs by O(1e-3) (observed: one of the 777 points, lin5
        # unit 1, pre-activation -4e-8). So: per-point gradients must agree except for <= 1 % of the points, and the summed
        # latent gradient to 2e-3.
        ref = g['g_points_' + name]
        err = np.abs(x.grad.cpu().numpy() - ref).max(1)
        nbad = int((err > 2e-5 * np.abs(ref).max()).sum())
        refl = g['g_latent_' + name]
        rel_l = np.abs(lat.grad.cpu().numpy() - refl).max() / np.abs(refl).max()
        print('G11 %s: points off by > 2e-5: %d of 777 (median residual %.2e); g_latent residual %.3e (reference noise floor %.3e)'
              % (name, nbad, np.median(err) / np.abs(ref).max(), rel_l, _floors()['g11_g_latent_%s_rel' % name]))
        # a unit whose pre-activation is ~1e-8 may sit on either side of the ReLU in two f32 summation orders (observed: ONE of the
        # 777 points in the unclamped case, lin5 unit 1, pre-activation -4e-8); that point's gradient then differs by O(1e-3) and
        # so does the summed latent gradient. Without such a point the bar is 2 x the reference's own noise floor.
        assert nbad <= 1, nbad
        assert rel_l <= (2e-3 if nbad else max(2.0 * _floors()['g11_g_latent_%s_rel' % name], 5e-6)), rel_l
    # only one of the two inputs requires grad / ragged sizes, against autograd through the module itself
    for n in (1, 63, 65, 200):
        lat = torch.from_numpy(g['latent']).cuda()
        x = torch.from_numpy(g['points'][:n]).cuda().requires_grad_(True)
        y = decode_sdf(dec, lat, x, clamp_dist=None)
        y.sum().backward()
        xr = torch.from_numpy(g['points'][:n]).cuda().requires_grad_(True)
        dec.inference(torch.cat([lat.expand(n, -1), xr], 1)).sum().backward()
        bad = (np.abs((x.grad - xr.grad).cpu().numpy()).max(1) > 5e-5 * np.abs(xr.grad.cpu().numpy()).max() + 1e-7).sum()
        assert bad <= max(1, n // 100), bad
    with torch.no_grad():
        assert not decode_sdf(dec, torch.from_numpy(g['latent']).cuda().requires_grad_(True), x.detach()).requires_grad


def _bench_ranks(n, args, env_extra=None, launcher=True, timeout=600, expect_rc=0):
    """`bench.py --gpus n` with n ranks time-sharing this one GPU (gloo for the collectives: RCCL ranks cannot share a device). Small
    images, one timed step: these tests check the multi-rank PROTOCOL of the driver's scaling run, not its speed."""
    import json
    import subprocess
    from conftest import ROOT
    env = dict(os.environ, DISTR_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.update(env_extra or {})
    if launcher:
        port = 29700 + (os.getpid() * 7 + _bench_ranks.calls * 13) % 290
        _bench_ranks.calls += 1
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', str(n)] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n)] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    if expect_rc is None:
        return out
    assert out.returncode == expect_rc, (out.stdout[-1500:], out.stderr[-2500:])
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])


_bench_ranks.calls = 0


def _check_multi_rank_line(j, n, balanced_possible):
    """The keys the first real N > 1 run must be readable from: the serial check of the all-reduced loss /
    gradients, who ran where, per-rank local milliseconds and all-reduce wait, and -- for the view-parallel workload -- BOTH the
    unbalanced and the balanced timing."""
    c = j['config']
    assert j['n_gpus'] == n and c['rccl']['world_size'] == n and len(c['rccl']['ranks']) == n
    assert sorted(r['rank'] for r in c['rccl']['ranks']) == list(range(n))
    assert c['rccl']['backend'] == 'gloo' and c['rccl']['backend_is_rccl'] is False and c['scaling_measurement'] is False     # one GPU, gloo: said so
    sc = c['serial_check']
    assert sc['ok'] is True and sc['chosen']['loss_rel'] <= 1e-5 and sc['chosen']['grad_rel'] <= 1e-4, sc
    for mode, d in c['per_rank'].items():
        assert len(d['local_ms']) == n and len(d['allreduce_and_wait_ms']) == n and all(x > 0 for x in d['local_ms']), (mode, d)
    assert 'split_bf16' not in j and 'split_f16' not in j and 'cpu_baseline' not in j           # N > 1 times the exact path only
    # round 6: what every rank renders per step under the mode `value` was timed in -- [shape, view, first row, end row] per piece
    assert len(c['partition']) == n and all(len(p) >= 1 and all(len(piece) == 4 and piece[2] < piece[3] for piece in p) for p in c['partition']), c['partition']
    assert 'extra' not in j                                                                      # (small renders: N = 1 only)
    # the line names the collective backend that really ran (gloo on this rig) and which timed mode is the N = 1 protocol on N GPUs
    assert 'gloo' in c['parallelism'] and 'NOT RCCL' in c['parallelism'] and 'RCCL all-reduce' not in c['parallelism'], c['parallelism']
    assert c['n1_protocol_equivalent'].startswith(('unbalanced', 'row_bands')), c['n1_protocol_equivalent']
    if balanced_possible:
        assert c['unbalanced_ms_per_step'] > 0 and 'balanced_ms_per_step' in c and c['value_is'] in ('balanced', 'unbalanced')
        assert 'unbalanced' in sc and sc['unbalanced']['loss_rel'] <= 1e-5 and sc['unbalanced']['grad_rel'] <= 1e-4
        best = min(x for x in (c['unbalanced_ms_per_step'], c['balanced_ms_per_step']) if x)
        if not os.environ.get('DISTR_BENCH_FAKE_TIMES') and c['value_is'] == 'unbalanced':
            assert abs(j['ms_per_step'] - c['unbalanced_ms_per_step']) <= 1e-6 * best


@pytest.mark.gpu
@pytest.mark.parametrize('workload', ['c3', 'c5'])
def test_bench_two_ranks_on_one_gpu(workload):
    """bench.py's multi-rank path end to end: two ranks (time-sharing this one GPU, gloo for the packed all-reduce since two
    RCCL ranks cannot share a device) through torch.distributed.run; checks the contract fields of the JSON line and the
    self-validation keys of an N > 1 run. c5 exercises the shape / row-band partition (each rank renders two of the four shapes)."""
    size = ['--size', '128', '--march-step', '30']
    j = _bench_ranks(2, ['--steps', '1', '--warmup', '0', '--workload', workload] + size)
    assert j['n_gpus'] == 2 and j['steps'] == 1 and j['unit'] == 'rays/s' and j['value'] > 0
    assert j['scaling'] == ('strong' if workload == 'c5' else 'weak')
    assert 'cpu_baseline' not in j and j['roofline']['achieved'] > 0
    rf = j['roofline']      # the static PMC figure is quoted only for the csrc/ digest it was measured on, else null + a note that says so
    assert (rf['traffic'] is not None and rf['traffic_note'].startswith('STATIC') and rf['traffic_csrc_sha256'] == rf['csrc_sha256']) or \
        (rf['traffic'] is None and rf['traffic_note'].startswith('null') and rf['traffic_csrc_sha256'] != rf['csrc_sha256'])
    rays = (4 if workload == 'c5' else 2) * 128 ** 2
    assert abs(j['value'] * j['ms_per_step'] * 1e-3 - rays) <= 1e-6 * rays
    _check_multi_rank_line(j, 2, balanced_possible=(workload == 'c3'))


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2 ...` started WITHOUT a launcher (the way the driver starts the N = 1 line) must spawn its own two
    ranks: rc 0, n_gpus 2, and the collective layer's own report (config.rccl) shows two ranks. On this one-GPU box the ranks
    time-share the device over gloo (DISTR_DIST_BACKEND); without that override a box with fewer GPUs than ranks is refused with a
    clear message instead of an RCCL hang."""
    drop = {k: None for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')}
    args = ['--steps', '1', '--warmup', '0', '--size', '128', '--march-step', '30', '--no-balance']
    import subprocess
    from conftest import ROOT
    env = dict(os.environ, DISTR_DIST_BACKEND='gloo')
    for k in drop:
        env.pop(k, None)                                   # bench.py must set what it needs itself
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + args
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert j['n_gpus'] == 2 and j['value'] > 0
    r = j['config']['rccl']
    assert r['world_size'] == 2 and r['backend'] == 'gloo' and r['launcher'].startswith('self-spawned'), r
    _check_multi_rank_line(j, 2, balanced_possible=False)
    import torch
    if torch.cuda.device_count() < 2:
        env.pop('DISTR_DIST_BACKEND')
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode != 0 and 'RCCL ranks cannot share a device' in (out.stderr + out.stdout)


@pytest.mark.gpu
def test_bench_view_balancing_two_ranks():
    """bench.py's row-band load balancing of the view-parallel step (distr.parallel.balance_views): two ranks sharing this GPU
    (gloo), step times forced to 70 / 40 ms so that rank 0 hands the bottom rows of its view to rank 1. The run times the step
    unbalanced AND balanced; the all-reduced loss and latent gradient of BOTH must equal the serial sum over the two views that rank 0
    renders itself (config.serial_check: the work moved, nothing else)."""
    bal = _bench_ranks(2, ['--steps', '1', '--warmup', '1', '--size', '192', '--march-step', '30'], {'DISTR_BENCH_FAKE_TIMES': '70,40'})
    free = _bench_ranks(2, ['--steps', '1', '--warmup', '1', '--size', '192', '--march-step', '30', '--view-offset', '6'])
    # second run: measured times (two ranks contending for one GPU: arbitrary) through the plan + refinement schedule; whatever the
    # plan (used or only tried), it must tile both views (views 6 and 7) exactly once
    cover = np.zeros((2, 192), np.int32)
    for p in (free['config']['balance_plan'] or free['config'].get('balance_plan_tried') or [[[0, 0, 192]], [[1, 0, 192]]]):
        for (v, r0, r1) in p:
            cover[v, r0:r1] += 1
    assert (cover == 1).all() and np.isfinite(free['config']['loss_sum_all_ranks'])
    _check_multi_rank_line(free, 2, balanced_possible=True)
    plan = bal['config']['balance_plan']
    cut = plan[0][0][2]                                   # (where exactly depends on the row cost profile of the rendered view)
    assert plan[0] == [[0, 0, cut]] and 96 <= cut < 192 and cut % 4 == 0 and plan[1] == [[1, 0, 192], [0, cut, 192]], plan
    assert bal['config']['rank0_items'] == [[0, 0, 0, cut]] and 'load balancing' in bal['config']['parallelism']
    assert bal['config']['value_is'] == 'balanced' and bal['config']['balanced_ms_per_step'] > 0
    _check_multi_rank_line(bal, 2, balanced_possible=True)
    sc = bal['config']['serial_check']
    assert 'balanced' in sc and sc['balanced']['loss_rel'] <= 1e-5 and sc['balanced']['grad_rel'] <= 1e-4, sc


@pytest.mark.gpu
def test_bench_view_balancing_eight_ranks():
    """The shape of the driver's 8-GPU run, on this one GPU: eight ranks (gloo), the step times of the eight C4 views as measured on
    one MI355X (profiles/r02_view_balance.md) forced in, so that the plan has one donor and two receivers. Every row of every view
    is rendered exactly once, and the all-reduced loss / latent gradient of the unbalanced AND the balanced step equal the serial sum
    over the eight views (config.serial_check, computed inside the run)."""
    H = 256
    times = [52.71, 54.40, 51.19, 50.40, 47.98, 46.50, 49.57, 58.67]
    bal = _bench_ranks(8, ['--steps', '1', '--warmup', '0', '--size', str(H), '--march-step', '30'],
                       {'DISTR_BENCH_FAKE_TIMES': ','.join('%.2f' % t for t in times)}, timeout=900)
    assert bal['n_gpus'] == 8
    _check_multi_rank_line(bal, 8, balanced_possible=True)
    plan = [[tuple(x) for x in p] for p in bal['config']['balance_plan']]
    cover = np.zeros((8, H), np.int32)
    for r, p in enumerate(plan):
        assert p[0][0] == r and p[0][1] == 0                       # first item: what is left of the rank's own view
        for (v, r0, r1) in p:
            assert r0 % 4 == 0 and (r1 % 4 == 0 or r1 == H)
            cover[v, r0:r1] += 1
    assert (cover == 1).all()
    donors = [r for r, p in enumerate(plan) if p[0][2] < H]
    receivers = [r for r, p in enumerate(plan) if len(p) > 1]
    mean = sum(times) / 8
    assert 7 in donors and receivers and all(times[r] < mean for r in receivers) and all(times[r] > mean for r in donors), plan
    assert abs(bal['value'] - 8 * H * H / (bal['ms_per_step'] * 1e-3)) <= 1e-6 * bal['value']      # whole-job rays / max-over-ranks time
    sc = bal['config']['serial_check']
    assert sc['images'] == 8 and sc['balanced']['grad_rel'] <= 1e-4 and sc['unbalanced']['grad_rel'] <= 1e-4, sc


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_CONFIGS', '8'))))      # (soak runs: more seeds)
def test_random_configs_match_oracle(engine, cpu_oracle, orc, fixture_decoder, seed):
    """Seeded random draws over image size (ragged), steps, buffer_size, ratio, marcher, normal mode and camera: HIP vs
    oracle with zero mask flips. Long marches at small sizes spend most steps on cluster tiles / merged launches."""
    from distr import fixture
    rs = np.random.RandomState(1000 + seed)
    H, W = int(rs.randint(17, 97)), int(rs.randint(17, 97))
    marcher = ['recursive', 'pyramid_recursive', 'pyramid_recursive', 'trivial'][rs.randint(4)]
    S = int(rs.randint(12, 70)) if marcher != 'trivial' else int(rs.randint(8, 20))
    kw = dict(march_step=S, buffer_size=int(rs.randint(1, 6)), ratio=float(rs.choice([1.0, 1.5, 2.0])), marcher=marcher,
              use_depth2normal=bool(rs.randint(2)))
    if marcher == 'pyramid_recursive':
        kw['coarse_steps'] = (int(rs.randint(1, 4)), int(rs.randint(1, 4)))
    cam = (float(rs.uniform(-180, 180)), float(rs.uniform(-60, 60)), float(rs.uniform(1.3, 2.2)), float(rs.uniform(-30, 30)))
    _, _, latent = fixture_decoder
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=3e-4, normal_p99=1e-5, max_flip_frac=0.0)
    assert res['flips'] == 0, (res, H, W, kw, cam)


@pytest.mark.gpu
def test_c_abi_error_paths(engine, fixture_decoder):
    """Error behaviour of the C ABI: non-zero return + distr_last_error text, nothing crashes: null pointers, too-small
    workspaces, bad configurations, backward on an inference-only forward, calls before distr_set_decoder."""
    import ctypes as C
    import torch
    from distr import binding, fixture
    L, h = engine.ctx.L, engine.ctx.h
    K = fixture.make_intrinsic(32, 32)
    cfg = binding.make_cfg((32, 32), K, march_step=12, buffer_size=3)
    fwd, bwd = engine.ctx.workspace_bytes(cfg)
    ws = torch.empty(fwd, dtype=torch.uint8, device='cuda')
    lat = torch.zeros(256, device='cuda'); R = torch.eye(3, device='cuda').reshape(-1); T = torch.tensor([0., 0., 1.6], device='cuda')
    z = torch.empty(1024, device='cuda'); m = torch.empty(1024, dtype=torch.uint8, device='cuda'); q = torch.empty(1024, device='cuda')
    p, s = binding.ptr, engine.ctx.stream()

    def err():
        return L.distr_last_error(h).decode()
    assert L.distr_render_forward(h, C.byref(cfg), None, p(R), p(T), p(z), p(m), p(q), None, None, p(ws), ws.numel(), s) != 0 and 'null' in err()
    assert L.distr_render_forward(h, C.byref(cfg), p(lat), p(R), p(T), p(z), p(m), p(q), None, None, p(ws), 1024, s) != 0 and 'workspace' in err()
    for field, val in (('buffer_size', 0), ('buffer_size', 9), ('marcher', 7), ('march_step', 3), ('H', 0), ('radius', -1.0)):
        bad = cfg.clone()
        setattr(bad, field, val)
        f, b = C.c_size_t(), C.c_size_t()
        assert L.distr_workspace_bytes(h, C.byref(bad), C.byref(f), C.byref(b)) != 0, field
        assert len(err()) > 0
    # pyramids the kernels are not built for (include/distr.h: num_levels / level_scale / level_steps), and row bands under a pyramid
    # whose coarsest scale does not divide the bands' 4-row alignment
    for kw, word in ((dict(scale_list=[3, 2, 1], march_step_list=[2, 2, -1]), 'multiple'), (dict(scale_list=[4, 2], march_step_list=[2, -1]), 'must be 1'),
                     (dict(scale_list=[16, 1], march_step_list=[2, -1]), 'multiple'), (dict(scale_list=[4, 2, 1], march_step_list=[2, 0, -1]), 'step'),
                     (dict(scale_list=[4, 2, 1], march_step_list=[2, 16, -1]), '15'), (dict(scale_list=[8, 4, 2, 1], march_step_list=[2, 2, 2, -1], band=(8, 8)), 'band')):
        bad = binding.make_cfg((32, 32), K, march_step=12, buffer_size=3, **kw)
        f, b = C.c_size_t(), C.c_size_t()
        assert L.distr_workspace_bytes(h, C.byref(bad), C.byref(f), C.byref(b)) != 0 and word in err(), (kw, err())
    bad = cfg.clone()
    bad.num_levels = 5
    f, b = C.c_size_t(), C.c_size_t()
    assert L.distr_workspace_bytes(h, C.byref(bad), C.byref(f), C.byref(b)) != 0 and 'num_levels' in err()
    # backward on a forward that did not save for backward
    inf = cfg.clone()
    inf.save_for_backward = 0
    inf.want_normal = 0
    assert L.distr_render_forward(h, C.byref(inf), p(lat), p(R), p(T), p(z), p(m), p(q), None, None, p(ws), ws.numel(), s) == 0
    wsb = torch.empty(bwd, dtype=torch.uint8, device='cuda')
    g = torch.empty(256, device='cuda'); gR = torch.empty(9, device='cuda'); gT = torch.empty(3, device='cuda')
    rc = L.distr_render_backward(h, C.byref(inf), p(ws), ws.numel(), p(z), p(q), None, None, p(g), p(gR), p(gT), p(wsb), wsb.numel(), s)
    assert rc != 0 and 'save_for_backward' in err()
    # a context without a decoder
    ctx2 = binding.Context(0)
    out = torch.empty(4, 1, device='cuda'); x = torch.zeros(4, 3, device='cuda')
    w2 = torch.empty(L.distr_mlp_workspace_bytes(4), dtype=torch.uint8, device='cuda')
    assert L.distr_mlp_eval(ctx2.h, p(lat), p(x), 4, -1.0, p(out), p(w2), w2.numel(), s) != 0
    assert 'distr_set_decoder' in L.distr_last_error(ctx2.h).decode()
    assert L.distr_color_eval(ctx2.h, p(lat), p(x), 4, p(out), p(w2), w2.numel(), s) != 0
    # wrong weight count
    with pytest.raises(binding.DistrError):
        ctx2.set_decoder(np.zeros(10, np.float32))
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_many_streams_with_cluster_tiles(engine, fixture_decoder):
    """Sixteen small renders (tail steps on cluster tiles) issued round-robin on eight HIP streams through ONE context: every
    stream has its own exchange region, clusters that cannot assemble fall back to their lead workgroup, and each render is
    bit-identical to the same render issued alone on the default stream."""
    import ctypes as C
    import torch
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    H = W = 64
    K = fixture.make_intrinsic(H, W)
    cfg = binding.make_cfg((H, W), K, march_step=40, buffer_size=2, marcher='recursive', want_normal=False)
    cfg.save_for_backward = 0
    fwd, _ = engine.ctx.workspace_bytes(cfg)
    p = binding.ptr
    lat = torch.from_numpy(latent).cuda().reshape(-1)
    cams = [fixture.make_camera(22.5 * i, 10.0 + i, 1.6, 0.0) for i in range(16)]
    Rs = [torch.from_numpy(R).cuda().reshape(-1) for R, _ in cams]
    Ts = [torch.from_numpy(T).cuda() for _, T in cams]

    def render(i, stream):
        ws = torch.empty(fwd, dtype=torch.uint8, device='cuda')
        o = [torch.empty(H * W, device='cuda'), torch.empty(H * W, dtype=torch.uint8, device='cuda'), torch.empty(H * W, device='cuda')]
        engine.ctx.check(engine.ctx.L.distr_render_forward(engine.ctx.h, C.byref(cfg), p(lat), p(Rs[i]), p(Ts[i]), p(o[0]), p(o[1]), p(o[2]),
                                                          None, None, p(ws), ws.numel(), C.c_void_p(stream.cuda_stream)))
        return o, ws
    main = torch.cuda.current_stream()
    ref = [render(i, main) for i in range(16)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(12)]       # more streams than exchange regions (8): idle regions are recycled
    for s in streams:
        s.wait_stream(main)
    got = []
    for rep in range(3):                      # several waves of concurrent launches
        got = []
        for i in range(16):
            st_ = streams[(i + 4 * rep) % (8 if rep < 2 else 12)]
            with torch.cuda.stream(st_):
                got.append(render(i, st_))
    torch.cuda.synchronize()
    for i in range(16):
        for a, b in zip(got[i][0], ref[i][0]):
            assert a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes(), i
        st = engine.ctx.render_stats(cfg, got[i][1])
        assert st['num_in_sphere'] > 0     # (cluster_fallbacks may be > 0 here: clusters of concurrent launches compete for CUs)


@pytest.mark.gpu
def test_decode_color_autograd_matches_reference_golden():
    """G13: decode_color differentiated (distr_color_backward: fused forward recompute + dX chain through the 3-row output layer)
    against the reference's autograd: gradients w.r.t. the colour code, the shape code and the points; ragged sizes against
    autograd through the module itself."""
    import torch
    from core.graph.deep_sdf_decoder import Decoder
    from core.utils.decoder_utils import decode_color
    from distr import fixture
    g = dict(np.load(os.path.join(GOLDEN, 'g13_decode_color_grad.npz')))
    cs = int(g['color_size'])
    Wc, bc, _ = fixture.make_color_decoder_weights(color_size=cs)
    dims = [512] * 8
    dims[3] += cs
    dec = Decoder(256 + cs, dims, last_dim=3, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Wc, bc)) for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    c = lambda k: torch.from_numpy(g[k]).cuda().requires_grad_(True)
    cc, sc, x = c('color_code'), c('latent'), c('points')
    rgb = decode_color(dec, cc, sc, x)
    assert rgb.requires_grad and np.abs(rgb.detach().cpu().numpy() - g['rgb']).max() <= 2e-6
    (rgb * torch.from_numpy(g['weights']).cuda()).sum().backward()
    err = np.abs(x.grad.cpu().numpy() - g['g_points']).max(1)
    nbad = int((err > 2e-5 * np.abs(g['g_points']).max()).sum())
    rc = np.abs(cc.grad.cpu().numpy() - g['g_color_code']).max() / np.abs(g['g_color_code']).max()
    rs_ = np.abs(sc.grad.cpu().numpy() - g['g_shape_code']).max() / np.abs(g['g_shape_code']).max()
    print('G13: points off by > 2e-5: %d of 333; g_color_code residual %.3e, g_shape_code residual %.3e' % (nbad, rc, rs_))
    assert nbad <= 1 and rc <= (2e-3 if nbad else 5e-5) and rs_ <= (2e-3 if nbad else 5e-5)
    for n in (1, 63, 65, 130):
        cc2, sc2 = torch.from_numpy(g['color_code']).cuda().requires_grad_(True), torch.from_numpy(g['latent']).cuda()
        x2 = torch.from_numpy(g['points'][:n]).cuda().requires_grad_(True)
        decode_color(dec, cc2, sc2, x2)[:, 1].sum().backward()
        cr, xr = torch.from_numpy(g['color_code']).cuda().requires_grad_(True), torch.from_numpy(g['points'][:n]).cuda().requires_grad_(True)
        dec.inference(torch.cat([sc2.expand(n, -1), cr.expand(n, -1), xr], 1))[:, 1].sum().backward()
        assert (x2.grad - xr.grad).abs().max() <= 5e-5 * xr.grad.abs().max() + 1e-7
        assert (cc2.grad - cr.grad).abs().max() <= 1e-4 * cr.grad.abs().max()
    with torch.no_grad():
        assert not decode_color(dec, cc, sc, x).requires_grad
    assert not decode_color(dec, cc, sc, x, no_grad=True).requires_grad


@pytest.mark.gpu
def test_forward_sampling_matches_reference_golden(fixture_decoder):
    """G12: render(num_forward_sampling=3) -- the k samples behind the surface (renderer.py:912-941) and the gradients of a
    weighted sum of them w.r.t. latent and camera, against the reference."""
    import torch
    from core.sdfrenderer import SDFRenderer
    from core.graph.deep_sdf_decoder import Decoder
    g = dict(np.load(os.path.join(GOLDEN, 'g12_forward_sampling.npz')))
    Ws, bs, _ = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W), ('bias', b))})
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer(dec.cuda(), g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']))
    c = lambda k: torch.from_numpy(g[k]).cuda().requires_grad_(True)
    lat, R, T = c('latent'), c('R'), c('T')
    out = r.render(lat, R, T, num_forward_sampling=int(g['num_forward_sampling']))
    assert len(out) == 5 and out[4].shape == (H, W, 3)
    inside = out[4]
    m = out[2].cpu().numpy().astype(bool)
    assert int((m != g['mask'].astype(bool)).sum()) <= 1
    both = m & g['mask'].astype(bool)
    assert np.abs(inside.detach().cpu().numpy() - g['inside_samples'])[both].max() <= 1e-4
    assert np.all(inside.detach().cpu().numpy()[~m] == 0)
    (inside * torch.from_numpy(g['weights']).cuda()).sum().backward()
    for t, k in ((lat, 'g_latent'), (R, 'g_R'), (T, 'g_T')):
        rel = np.abs(t.grad.cpu().numpy() - g[k]).max() / np.abs(g[k]).max()
        assert rel <= 2e-3, (k, rel)
    with pytest.raises(NotImplementedError):
        r.render(lat, R, T, sample_index_type='min')


@pytest.mark.gpu
def test_engine_follows_live_decoder_parameters(fixture_decoder):
    """the packed weights follow the live module (load_state_dict / in-place updates after SDFRenderer.__init__), as
    the reference does by evaluating the module itself; a training-mode decoder with dropout is rejected."""
    import torch
    from core.sdfrenderer import SDFRenderer
    from core.graph.deep_sdf_decoder import Decoder
    from core.utils.decoder_utils import decode_sdf
    from distr import decoder_pack, fixture
    Ws, bs, latent = fixture_decoder
    mk = lambda: Decoder(256, [512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4], weight_norm=True)
    dec = mk().cuda()
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in decoder_pack.fixture_state_dict(Ws, bs, True).items()})
    K = fixture.make_intrinsic(32, 32)
    R, T = (torch.from_numpy(a).cuda() for a in fixture.make_camera(10, 5, 1.6, 0))
    r = SDFRenderer(dec, K, march_step=20, buffer_size=2)
    lat = torch.from_numpy(latent).cuda()
    pts = torch.from_numpy(_points(200)).cuda()
    d0 = r.render(lat, R, T, no_grad=True)[0].clone()
    s0 = decode_sdf(dec, lat, pts, no_grad=True).clone()
    Ws2, bs2, _ = fixture.make_decoder_weights(77)
    dec.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in decoder_pack.fixture_state_dict(Ws2, bs2, True).items()})
    d1 = r.render(lat, R, T, no_grad=True)[0]
    s1 = decode_sdf(dec, lat, pts, no_grad=True)
    with torch.no_grad():
        ref = dec.inference(torch.cat([lat.expand(200, -1), pts], 1)).clamp(-0.1, 0.1)
    assert (s1 - ref).abs().max() < 5e-6 and (s1 - s0).abs().max() > 1e-3 and not torch.equal(d0, d1)
    with torch.no_grad():
        dec.lin8.bias.add_(0.05)                       # in-place edit of one parameter
        ref2 = dec.inference(torch.cat([lat.expand(200, -1), pts], 1)).clamp(-0.1, 0.1)
    assert (decode_sdf(dec, lat, pts, no_grad=True) - ref2).abs().max() < 5e-6
    dec.train()
    with pytest.raises(decoder_pack.UnsupportedDecoder):
        r.render(lat, R, T, no_grad=True)
    dec.eval()
    r.render(lat, R, T, no_grad=True)


@pytest.mark.gpu
def test_cluster_fallback_is_bit_identical(engine, fixture_decoder):
    """A cluster whose workgroups do not become co-resident in time is evaluated by its lead workgroup alone. Forced here for
    EVERY cluster (DISTR_CLUSTER_TEST_ABORT=1 on a second context): outputs and gradients equal the normal run bit for bit and
    the render stats count the fallbacks."""
    from distr import fixture, functions
    Ws, bs, latent = fixture_decoder
    os.environ['DISTR_CLUSTER_TEST_ABORT'] = '1'
    try:
        eng2 = functions.engine_from_weights(Ws, bs, 0)
    finally:
        del os.environ['DISTR_CLUSTER_TEST_ABORT']
    H = W = 64
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(33, 12, 1.6, 0)
    for marcher, d2n in (('recursive', False), ('pyramid_recursive', True)):
        kw = dict(march_step=60, buffer_size=3, marcher=marcher, use_depth2normal=d2n)
        a = helpers.hip_render(engine, H, W, K, R, T, latent, **kw)
        b = helpers.hip_render(eng2, H, W, K, R, T, latent, **kw)
        for k in ('zdepth', 'mask', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T'):
            assert np.array_equal(a[k], b[k]), (marcher, k)
        sa = engine.ctx.render_stats(a['cfg'], _last_ws(engine, a['cfg'], latent, R, T))
        sb = eng2.ctx.render_stats(b['cfg'], _last_ws(eng2, b['cfg'], latent, R, T))
        assert sa['cluster_fallbacks'] == 0 and sb['cluster_fallbacks'] > 20, (sa, sb)
        assert sa['num_point_evals'] == sb['num_point_evals']


@pytest.mark.gpu
def test_cluster_tiles_under_oversubscription():
    """ADVICE r1 (medium) / VERDICT r1 item 2: eight streams, 512x512 dense renders mixed with small tail-dominated renders,
    GPU_MAX_HW_QUEUES=8, 100 iterations (soak: DISTR_TEST_STRESS_ITERS) -- clusters that cannot assemble fall back on the device; every one of the 800 renders
    is bit-identical to its stand-alone result (tests/gpu_stress_clusters.py)."""
    import json
    import subprocess
    from conftest import ROOT
    env = dict(os.environ, GPU_MAX_HW_QUEUES='8')
    iters = os.environ.get('DISTR_TEST_STRESS_ITERS', '100')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'gpu_stress_clusters.py'), '--iters', iters, '--streams', '8'],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    print('oversubscription stress:', j)
    assert j['mismatching_renders'] == 0 and j['renders'] == 8 * int(iters)


@pytest.mark.gpu
def test_forward_is_graph_capturable(engine, fixture_decoder):
    """The forward is a fixed launch sequence with device-side control flow only, so it can be captured into a HIP graph
    and replayed with new inputs written into the same buffers (cluster tiles are left out of captured renders: their
    barrier epochs are launch-time values)."""
    import ctypes as C
    import torch
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    H = W = 48
    K = fixture.make_intrinsic(H, W)
    cfg = binding.make_cfg((H, W), K, march_step=30, buffer_size=3, use_depth2normal=True)
    cfg.save_for_backward = 0
    fwd, _ = engine.ctx.workspace_bytes(cfg)
    p = binding.ptr
    lat = torch.from_numpy(latent).cuda().reshape(-1).clone()
    R0, T0 = fixture.make_camera(10.0, 5.0, 1.6, 0.0)
    R1, T1 = fixture.make_camera(-40.0, 30.0, 1.8, 12.0)
    Rt, Tt = torch.from_numpy(R0).cuda().reshape(-1).clone(), torch.from_numpy(T0).cuda().clone()
    ws = torch.empty(fwd, dtype=torch.uint8, device='cuda')
    o = [torch.empty(H * W, device='cuda'), torch.empty(H * W, dtype=torch.uint8, device='cuda'), torch.empty(H * W, device='cuda'),
         torch.empty(H, W, device='cuda'), torch.empty(H, W, 3, device='cuda')]

    def launch():
        engine.ctx.check(engine.ctx.L.distr_render_forward(engine.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(o[0]), p(o[1]), p(o[2]), p(o[3]),
                                                          p(o[4]), p(ws), ws.numel(), engine.ctx.stream()))
    launch()
    torch.cuda.synchronize()
    want0 = [t.clone() for t in o]
    Rt.copy_(torch.from_numpy(R1).reshape(-1)); Tt.copy_(torch.from_numpy(T1))
    launch()
    torch.cuda.synchronize()
    want1 = [t.clone() for t in o]
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            launch()
    torch.cuda.synchronize()
    for (R, T, want) in ((R0, T0, want0), (R1, T1, want1), (R0, T0, want0)):
        Rt.copy_(torch.from_numpy(R).reshape(-1)); Tt.copy_(torch.from_numpy(T))
        for t in o:
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(o, want):
            assert a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes()
    assert want0[1].sum() > 50 and (want0[3] != want1[3]).any()


@pytest.mark.gpu
@pytest.mark.parametrize('marcher', ['trivial', 'recursive', 'pyramid_recursive'])
def test_march_structure_matches_reference_golden(engine, marcher):
    """G6: the number of decoder evaluations the march executes (sum of the live-ray counts of all steps) and the number of
    rays that hit the unit sphere, against the reference's own decoder-call sizes."""
    import ctypes as C
    import torch
    from distr import binding
    g = dict(np.load(os.path.join(GOLDEN, 'g6_march_structure.npz')))
    H, W = int(g['H']), int(g['W'])
    cfg = binding.make_cfg((H, W), g['K'], march_step=int(g['march_step']), buffer_size=int(g['buffer_size']), marcher=marcher, want_normal=False)
    cfg.save_for_backward = 0
    fwd, _ = engine.ctx.workspace_bytes(cfg)
    ws = torch.empty(fwd, dtype=torch.uint8, device='cuda')
    p = binding.ptr
    o = [torch.empty(H * W, device='cuda'), torch.empty(H * W, dtype=torch.uint8, device='cuda'), torch.empty(H * W, device='cuda')]
    lat, Rt, Tt = (torch.from_numpy(g[k]).cuda().reshape(-1) for k in ('latent', 'R', 'T'))      # keep the inputs alive across the call
    engine.ctx.check(engine.ctx.L.distr_render_forward(engine.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt),
                                                      p(o[0]), p(o[1]), p(o[2]), None, None, p(ws), ws.numel(), engine.ctx.stream()))
    st = engine.ctx.render_stats(cfg, ws)
    ref = g['calls_' + marcher][:int(g['march_step'])]
    assert st['num_in_sphere'] == int(g['in_sphere'].sum())
    assert abs(st['num_point_evals'] - int(ref.sum())) <= 3, (st, int(ref.sum()))
    assert (o[1].cpu().numpy().astype(bool) != g['valid_final_' + marcher].astype(bool)).sum() <= 1


@pytest.mark.gpu
def test_plain_c_program_renders(engine, fixture_decoder, tmp_path):
    """A plain-C program (tests/c_abi/abi_render.c: gcc -std=c99, device memory from the HIP runtime's C API, no Python, no
    torch) renders forward + backward through include/distr.h from a blob holding the RAW BYTES of the ctypes distr_render_cfg
    mirror; its outputs equal the ctypes path's byte for byte."""
    import ctypes as C
    import shutil
    import subprocess
    import torch
    from conftest import ROOT
    from distr import binding, decoder_pack, fixture
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    exe = str(tmp_path / 'abi_render')
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'), '-I', '/opt/rocm/include',
                           '-o', exe, os.path.join(ROOT, 'tests', 'c_abi', 'abi_render.c'), '-L', '/opt/rocm/lib', '-lamdhip64', '-ldl'])
    Ws, bs, latent = fixture_decoder
    H, W = 48, 56
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(25, 15, 1.6, 5)
    kw = dict(march_step=30, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)
    cfg = binding.make_cfg((H, W), K, **kw)
    wd, wq, wn = helpers.loss_weights(H, W, 5)
    flat = np.ascontiguousarray(decoder_pack.flatten(Ws, bs), dtype=np.float32)
    blob = str(tmp_path / 'in.bin')
    with open(blob, 'wb') as f:
        f.write(np.int64(flat.size).tobytes())
        f.write(bytes(memoryview(cfg)))
        for a in (flat, latent.reshape(-1), R.reshape(-1), T.reshape(-1), wd, wq, wn):
            f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
    outp = str(tmp_path / 'out.bin')
    env = dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    run = subprocess.run([exe, binding.LIB_PATH, blob, outp], env=env, capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and 'rendered' in run.stdout, (run.stdout, run.stderr)
    P = H * W
    raw = np.fromfile(outp, dtype=np.uint8)
    out = raw[:4 * (6 * P + 268)].view(np.float32)
    mask = raw[4 * (6 * P + 268):]
    # the same call sequence through ctypes (upstream gradients handed over as they are: no mask product here or there)
    p = binding.ptr
    ctx = engine.ctx
    dev = engine.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    lat, Rt, Tt, gd, gq, gn = t(latent.reshape(-1)), t(R.reshape(-1)), t(T), t(wd.reshape(-1)), t(wq.reshape(-1)), t(wn.reshape(-1))
    fwd, bwd = ctx.workspace_bytes(cfg)
    ws, wsb = torch.empty(fwd, dtype=torch.uint8, device=dev), torch.empty(bwd, dtype=torch.uint8, device=dev)
    z, m, q, d, nm = torch.empty(P, device=dev), torch.empty(P, dtype=torch.uint8, device=dev), torch.empty(P, device=dev), torch.empty(P, device=dev), torch.empty(3 * P, device=dev)
    gl, gR, gT = torch.empty(256, device=dev), torch.empty(9, device=dev), torch.empty(3, device=dev)
    ctx.check(ctx.L.distr_render_forward(ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(z), p(m), p(q), p(d), p(nm), p(ws), ws.numel(), ctx.stream()))
    ctx.check(ctx.L.distr_render_backward(ctx.h, C.byref(cfg), p(ws), ws.numel(), None, p(gq), p(gd), p(gn), p(gl), p(gR), p(gT), p(wsb), wsb.numel(), ctx.stream()))
    torch.cuda.synchronize()
    ref = np.concatenate([x.cpu().numpy().reshape(-1) for x in (z, q, d, nm, gl, gR, gT)])
    assert out.tobytes() == ref.tobytes() and mask.tobytes() == m.cpu().numpy().tobytes()
    assert int(mask.sum()) > 100 and np.abs(out[6 * P:6 * P + 256]).max() > 0


@pytest.mark.gpu
def test_backward_refuses_re_uploaded_decoder(fixture_decoder):
    """render A, change the decoder (the packed copy is re-uploaded on the next call), render B, back-propagate A -- A's
    saved ReLU masks belong to the OLD weights; the backward must refuse instead of mixing them with the new ones."""
    import torch
    from core.graph.deep_sdf_decoder import Decoder
    from core.sdfrenderer import SDFRenderer
    from distr import fixture
    Ws, bs, latent = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W), ('bias', b))})
    H = W = 32
    r = SDFRenderer(dec.cuda(), fixture.make_intrinsic(H, W), img_hw=(H, W), march_step=16, buffer_size=2)
    R, T = (torch.from_numpy(a).cuda() for a in fixture.make_camera(10, 10, 1.6, 0))
    lat = torch.from_numpy(latent).cuda().requires_grad_(True)
    dA = r.render(lat, R, T)[0]
    with torch.no_grad():
        dec.lin8.bias.add_(1e-3)                       # in-place edit: bumps _version -> re-upload on the next render
    dB = r.render(lat, R, T)[0]
    with pytest.raises(RuntimeError, match='re-uploaded'):
        dA[dA < 1e5].sum().backward()
    dB[dB < 1e5].sum().backward()                      # the current render back-propagates fine
    assert lat.grad is not None and float(lat.grad.abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('fix', ['f1', 'f2'])
@pytest.mark.parametrize('arith', ['bf16x6', 'f16x3'])
def test_decode_sdf_split_bf16_is_f32_equivalent(fixture_decoder, fix, arith):
    """distr_mlp_eval_bf16x6 / distr_mlp_eval_f16x3 (opt-in split-bf16 / split-f16 decoder tiles) against the exact f32 kernel -- which the oracle pins
    bit for bit -- on both fixtures: ragged point counts, the clamp, and the error bar (1e-5: two orders below the 1e-4 parity bar
    of the renderer; measured ~1e-6). Also through create_sdf_grid(arith='bf16x6')."""
    import torch
    from core.evaluation import create_sdf_grid
    from core.graph.deep_sdf_decoder import Decoder
    from distr import fixture, functions
    Ws, bs, latent = fixture_decoder if fix == 'f1' else fixture.load_fixture_f2()
    eng = functions.engine_from_weights(Ws, bs, 0)
    lat = torch.from_numpy(latent).cuda()
    rs = np.random.RandomState(4)
    for n in (1, 63, 64, 65, 4097, 50000):
        pts = torch.from_numpy((rs.rand(n, 3) * 1.8 - 0.9).astype(np.float32)).cuda()
        a = functions.mlp_eval(eng, lat, pts)
        b = functions.mlp_eval(eng, lat, pts, arith=arith)
        assert b.shape == a.shape and float((a - b).abs().max()) <= 1e-5, (n, float((a - b).abs().max()))
        ac = functions.mlp_eval(eng, lat, pts, clamp_dist=0.05)
        bc = functions.mlp_eval(eng, lat, pts, clamp_dist=0.05, arith=arith)
        assert float((ac - bc).abs().max()) <= 1e-5 and float(bc.abs().max()) <= 0.05 + 1e-7
    if n >= 50000:
        print('%s: max |sdf_%s - sdf_f32| over %d points = %.3e' % (fix, arith, n, float((a - b).abs().max())))
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n_)): torch.from_numpy(a_) for l, (W, b_) in enumerate(zip(Ws, bs)) for n_, a_ in (('weight', W), ('bias', b_))})
    dec = dec.cuda()
    g0, g1 = create_sdf_grid(dec, lat, 48), create_sdf_grid(dec, lat, 48, arith=arith)
    assert float((g0 - g1).abs().max()) <= 1e-5 and bool(((g0 > 0) == (g1 > 0)).float().mean() > 0.9999)
    from core.evaluation import create_sdf_grid_speedup
    h0, h1 = create_sdf_grid_speedup(dec, lat, 48), create_sdf_grid_speedup(dec, lat, 48, arith=arith)      # coarse-to-fine variant: same band (up to a voxel at its edge), same values
    assert bool(((h0 - h1).abs() <= 1e-5).float().mean() > 0.999) and bool(((h0 > 0) == (h1 > 0)).float().mean() > 0.9999)
    with pytest.raises(ValueError):
        functions.mlp_eval(eng, lat, pts, arith='fp8')


@pytest.mark.gpu
def test_split_f16_range_is_checked(fixture_decoder):
    """The split-f16 arithmetic only covers decoders whose scaled weights and activations fit f16 (|.| < 1023): a decoder with a
    larger weight makes the mode unavailable (DISTR_ERR_UNSUPPORTED from every call that asks for it, the other arithmetics keep
    working); an activation that overflows is reported -- NaN from distr_mlp_eval_f16x3, f16_overflows > 0 in the render's counters --
    never clamped into a plausible number."""
    import ctypes as C
    import torch
    from distr import binding, fixture, functions
    Ws, bs, latent = fixture_decoder
    lat = torch.from_numpy(latent).cuda()
    pts = torch.from_numpy((np.random.RandomState(1).rand(200, 3) * 1.2 - 0.6).astype(np.float32)).cuda()
    H = W = 48
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(20.0, 10.0, 1.6, 0.0)

    # (1) a weight beyond the range (2000 * 64 > 65504)
    W1 = [w.copy() for w in Ws]
    W1[2][5, 7] = 2000.0
    eng = functions.engine_from_weights(W1, bs, 0)
    assert torch.isfinite(functions.mlp_eval(eng, lat, pts)).all()
    assert torch.isfinite(functions.mlp_eval(eng, lat, pts, arith='bf16x6')).all()
    with pytest.raises(binding.DistrError, match='f16'):
        functions.mlp_eval(eng, lat, pts, arith='f16x3')
    with pytest.raises(binding.DistrError, match='f16'):
        helpers.hip_render(eng, H, W, K, R, T, latent, arith='f16x3', march_step=10, buffer_size=2, marcher='recursive', use_depth2normal=True)

    # (2) weights in range, activations not: lin0's bias pushed to 3000 -> x0 ~ 3000 -> 64 x0 > 65504
    b2 = [b.copy() for b in bs]
    b2[0] = b2[0] + 3000.0
    eng = functions.engine_from_weights(Ws, b2, 0)
    assert torch.isfinite(functions.mlp_eval(eng, lat, pts)).all()
    assert torch.isfinite(functions.mlp_eval(eng, lat, pts, arith='bf16x6')).all()
    assert torch.isnan(functions.mlp_eval(eng, lat, pts, arith='f16x3', clamp_dist=0.1)).all()
    cfg = binding.make_cfg((H, W), K, march_step=10, buffer_size=2, marcher='recursive', use_depth2normal=True, arith='f16x3')
    n = H * W
    dev = eng.device
    ws = torch.empty(eng.ctx.workspace_bytes(cfg)[0], dtype=torch.uint8, device=dev)
    outs = [torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev),
            torch.empty(n, 3, device=dev)]
    p = binding.ptr
    Rt, Tt = torch.from_numpy(R).to(dev).reshape(-1).contiguous(), torch.from_numpy(T).to(dev)
    eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]),
                                               p(ws), ws.numel(), eng.ctx.stream()))
    st = eng.ctx.render_stats(cfg, ws)
    assert st['f16_overflows'] > 0, st
    cfg.arith = binding.ARITH['bf16x6']
    eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]),
                                               p(ws), ws.numel(), eng.ctx.stream()))
    assert eng.ctx.render_stats(cfg, ws)['f16_overflows'] == 0


def _g18_cases():
    import test_oracle_vs_golden as tg
    return tg._g18_cases()


@pytest.mark.gpu
@pytest.mark.parametrize('marcher,d2n', sorted({(c[0], c[1]) for c in _g18_cases()}))
def test_no_grad_flags_match_reference_golden(fixture_decoder, marcher, d2n):
    """G18 through the drop-in class: SDFRenderer.render(..., no_grad_depth / _normal / _mask / _camera = True) (renderer.py:943-999, as
    loss_single.py:17 and renderer_warp.py:109 call it) against the reference's own gradients for every flag set of the golden --
    loss value, latent / R / T gradients (bar: 2 x the reference's noise floor of that very case, not below 1e-3 of the flag-free
    gradient's size), and the forward depth, which no_grad_depth changes in its last bits."""
    import torch
    from core.sdfrenderer import SDFRenderer
    from core.graph.deep_sdf_decoder import Decoder
    Ws, bs, latent = fixture_decoder
    g = np.load(os.path.join(GOLDEN, 'g18_no_grad_flags.npz'))
    assert np.array_equal(g['latent'], latent)
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W_, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W_), ('bias', b))})
    dec = dec.cuda()
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer(dec, g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']),
                    ray_marching_ratio=float(g['ratio']), use_depth2normal=d2n)
    wd, wq, wn = (torch.from_numpy(a).cuda() for a in helpers.loss_weights(H, W, int(g['loss_seed'])))
    base = '%s_%s_flag_free.' % (marcher, 'd2n' if d2n else 'agn')
    worst = {}
    for flagset in [str(f) for f in g['flag_sets']]:
        key = '%s_%s_%s' % (marcher, 'd2n' if d2n else 'agn', flagset)
        lat = torch.from_numpy(latent).cuda().requires_grad_(True)
        Rt = torch.from_numpy(g['R']).cuda().requires_grad_(True)
        Tt = torch.from_numpy(g['T']).cuda().requires_grad_(True)
        depth, normal, mask, mq = r.render(lat, Rt, Tt, ray_marching_type=marcher, **{'no_grad_' + f: True for f in flagset.split('+')})
        mb = mask.bool()
        L = (depth * wd)[mb].sum() + (mq * wq).sum() + (normal * wn).sum()
        assert bool(L.requires_grad) == bool(g[key + '.has_graph'])
        if L.requires_grad:
            L.backward()
        assert abs(float(L.detach()) - float(g[key + '.loss'])) <= 2e-5 * abs(float(g[key + '.loss']))
        assert np.array_equal(mask.cpu().numpy(), g[base + 'mask'])
        ref_depth = g[key + '.depth'] if not bool(g[key + '.depth_same_as_flag_free']) else g[base + 'depth']
        assert np.abs(depth.detach().cpu().numpy() - ref_depth)[mb.cpu().numpy()].max() <= 1e-4
        for name, t in (('g_latent', lat), ('g_R', Rt), ('g_T', Tt)):
            mine = np.zeros(t.shape, np.float32) if t.grad is None else t.grad.cpu().numpy()
            ref, scale, fl = g['%s.%s' % (key, name)], float(g['%s.%s_scale' % (key, name)]), float(g['%s.%s_floor_rel' % (key, name)])
            rel = float(np.abs(mine.reshape(-1) - ref.reshape(-1)).max() / scale)
            worst[(flagset, name)] = (rel, fl)
            assert rel <= max(2.0 * fl, 1e-3), (key, name, rel, fl)
    print('G18 %s %s: worst residual / floor per flag set:' % (marcher, 'd2n' if d2n else 'agn'),
          {k[0]: '%.1e / %.1e' % max(v for kk, v in worst.items() if kk[0] == k[0]) for k in worst})


class _StepRecorder(object):
    """Wraps optimizer.step like oracle/gen_golden_loops.py does around the reference's loop: the optimised tensor and its gradient as
    the loop hands them to Adam, iteration by iteration."""

    def __init__(self, opt, tensor):
        self.values, self.grads, self.tensor = [], [], tensor
        self._step = opt.step
        opt.step = self.step

    def step(self, *a, **k):
        self.values.append(self.tensor.detach().cpu().numpy().copy())
        self.grads.append(self.tensor.grad.detach().cpu().numpy().copy())
        return self._step(*a, **k)


def _plain_decoder(fixture_decoder):
    import torch
    from core.graph.deep_sdf_decoder import Decoder
    Ws, bs, _ = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W_, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W_), ('bias', b))})
    return dec.cuda()


def _check_loop_against_golden(rec, final, g, label, lr, grad_at):
    """rec: the drop-in loop's own five iterations; grad_at(value) -> the gradient ONE iteration of the drop-in loop computes when started
    at `value`. Two checks: (1) parity proper -- at the reference's own tensor of every iteration the loop's gradient equals the
    reference's (1e-3 relative, or 2 x the reference's noise floor); (2) the free-running trajectory stays on the reference's for
    the first iterations and ends within a few Adam steps of it. (Later free-running gradients are NOT compared point-wise: measured on
    G20, a 4.5e-7 difference in the shape code before iteration 4 -- Adam amplifying a 1e-6 gradient residual -- moves one pixel of the
    16 x 16 renderer across the mask threshold and the gradient by 6e-3, while at the reference's own code of that iteration the two
    gradients agree to 3e-4.)"""
    vals, grads = np.array(rec.values), np.array(rec.grads)
    ref_v, ref_g = g['values'], g['grads']
    assert vals.shape == ref_v.shape and grads.shape == ref_g.shape
    fg, ff = float(g['floor_grad_rel']), float(g['floor_final_abs'])
    at_ref = []
    for i in range(ref_v.shape[0]):
        mine = grad_at(ref_v[i])
        at_ref.append(float(np.abs(mine - ref_g[i]).max() / np.abs(ref_g[i]).max()))
    free = [float(np.abs(grads[i] - ref_g[i]).max() / np.abs(ref_g[i]).max()) for i in range(grads.shape[0])]
    dvals = [float(np.abs(vals[i] - ref_v[i]).max()) for i in range(vals.shape[0])]
    dv = np.abs(final.reshape(-1) - g['final'].reshape(-1))
    print('%s: gradient residual AT THE REFERENCE\'S tensor of each iteration %s (reference noise floor %.1e); free-running: gradient %s, '
          '|tensor - reference| before each step %s, final max %.2e p99 %.2e (floor %.1e)'
          % (label, ['%.1e' % r for r in at_ref], fg, ['%.1e' % r for r in free], ['%.1e' % v for v in dvals], dv.max(), np.percentile(dv, 99), ff))
    assert max(at_ref) <= max(2.0 * fg, 1e-3), at_ref
    assert dvals[0] == 0.0 and free[0] <= max(2.0 * fg, 2e-4), (dvals, free)
    assert all(v <= 0.02 * lr for v in dvals[:3]), dvals                 # three iterations on the reference's trajectory
    assert np.percentile(dv, 99) <= max(2.0 * ff, 0.1 * lr) and dv.max() <= 2.5 * lr * grads.shape[0], (dv.max(), np.percentile(dv, 99), ff)


@pytest.mark.gpu
def test_camera_optimisation_loop_matches_reference_golden(fixture_decoder):
    """G19: the reference's optimize_single_view(optimizer_type='camera') (optimize_single.py:36-103 as run_single_camera.py:76-111 drives
    it: quaternion + translation tensor, get_camera_from_tensor inside the loop, Adam lr 5e-3, weights 10 / 10 / 1 / 1 / 1) -- five
    iterations through the drop-in loop and renderer against the camera tensors and gradients the reference itself produced."""
    import torch
    from core.sdfrenderer import SDFRenderer
    from core.inv_optimizer import optimize_single_view
    g = dict(np.load(os.path.join(GOLDEN, 'g19_camera_loop.npz')))
    dec = _plain_decoder(fixture_decoder)
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer(dec, g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']), threshold=5e-5,
                    ray_marching_ratio=1.5, use_depth2normal=False)
    lat = torch.from_numpy(g['latent']).cuda()
    RT = torch.from_numpy(g['RT_true']).cuda()
    d, n, m, q = r.render(lat, RT[:, :3], RT[:, 3], no_grad=True)        # the GT the reference rendered from the true camera: ours agrees
    assert (m.cpu().numpy() != g['gt_mask']).sum() <= 2
    both = (m.cpu().numpy() > 0) & (g['gt_mask'] > 0)
    assert np.abs(d.detach().cpu().numpy() - g['gt_depth'])[both].max() <= 1e-4
    gt_pack = {'depth': torch.from_numpy(g['gt_depth']).cuda(), 'normal': torch.from_numpy(g['gt_normal']).cuda(),
               'silhouette': torch.from_numpy(g['gt_mask']).cuda()}
    cam = torch.from_numpy(g['camera0']).cuda().requires_grad_(True)
    opt = torch.optim.Adam([cam], lr=float(g['lr']))
    rec = _StepRecorder(opt, cam)
    wd = dict(w_depth=10.0, w_normal=10.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
    out, _ = optimize_single_view([r], None, opt, lat, cam, gt_pack, wd, optimizer_type='camera', num_iters=int(g['iters']), renderer_weights=[1.0], silent=True)
    assert out is cam

    def grad_at(value):
        c = torch.from_numpy(value).cuda().requires_grad_(True)
        o = torch.optim.SGD([c], lr=0.0)
        rr = _StepRecorder(o, c)
        optimize_single_view([r], None, o, lat, c, gt_pack, wd, optimizer_type='camera', num_iters=1, renderer_weights=[1.0], silent=True)
        return rr.grads[0]
    _check_loop_against_golden(rec, cam.detach().cpu().numpy(), g, 'G19 camera loop', float(g['lr']), grad_at)


@pytest.mark.gpu
def test_multiscale_shape_loop_matches_reference_golden(fixture_decoder):
    """G20: the reference's optimize_single_view over the multi-scale renderer list of run_single_shape.py:110-113 (full resolution with
    buffer_size 1, 1/2 with 3, 1/4 with 5 through downsize_camera_intrinsic; one loss summed over the three; GT downsized inside
    compute_all_loss) -- five Adam iterations through the drop-in loop against the shape codes and gradients the reference produced."""
    import torch
    from core.sdfrenderer import SDFRenderer
    from core.inv_optimizer import optimize_single_view
    from core.utils.render_utils import downsize_camera_intrinsic
    g = dict(np.load(os.path.join(GOLDEN, 'g20_multiscale_shape_loop.npz')))
    dec = _plain_decoder(fixture_decoder)
    H, W, S = int(g['H']), int(g['W']), int(g['march_step'])
    mk = lambda Kx, bs, **kw: SDFRenderer(dec, Kx, march_step=S, buffer_size=bs, threshold=5e-5, use_depth2normal=True, **kw)
    rs = [mk(g['K'], 1, img_hw=(H, W), ray_marching_ratio=1.5), mk(downsize_camera_intrinsic(g['K'], 2), 3), mk(downsize_camera_intrinsic(g['K'], 4), 5)]
    assert [list(r.get_img_hw()) for r in rs] == g['img_hw'].tolist()
    RT = torch.from_numpy(g['RT']).cuda()
    gt_pack = {'depth': torch.from_numpy(g['gt_depth']).cuda(), 'normal': torch.from_numpy(g['gt_normal']).cuda(),
               'silhouette': torch.from_numpy(g['gt_mask']).cuda()}
    lat = torch.from_numpy(g['latent0']).cuda().requires_grad_(True)
    opt = torch.optim.Adam([lat], lr=float(g['lr']))
    rec = _StepRecorder(opt, lat)
    wd = dict(w_depth=10.0, w_normal=5.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
    optimize_single_view(rs, None, opt, lat, RT, gt_pack, wd, optimizer_type='shape', num_iters=int(g['iters']), renderer_weights=[1.0, 1.0, 1.0], silent=True)

    def grad_at(value):
        l_ = torch.from_numpy(value).cuda().requires_grad_(True)
        o = torch.optim.SGD([l_], lr=0.0)
        rr = _StepRecorder(o, l_)
        optimize_single_view(rs, None, o, l_, RT, gt_pack, wd, optimizer_type='shape', num_iters=1, renderer_weights=[1.0, 1.0, 1.0], silent=True)
        return rr.grads[0]
    _check_loop_against_golden(rec, lat.detach().cpu().numpy(), g, 'G20 multi-scale shape loop', float(g['lr']), grad_at)
    # the loop above ran the three scales on a pool of HIP streams (streams=None: one per renderer, distr_render_cfg.concurrent set);
    # the reference's sequential order must give the same shape code bit for bit
    lat2 = torch.from_numpy(g['latent0']).cuda().requires_grad_(True)
    opt2 = torch.optim.Adam([lat2], lr=float(g['lr']))
    optimize_single_view(rs, None, opt2, lat2, RT, gt_pack, wd, optimizer_type='shape', num_iters=int(g['iters']), renderer_weights=[1.0, 1.0, 1.0], silent=True, streams=0)
    assert torch.equal(lat2.detach(), lat.detach())


@pytest.mark.gpu
def test_two_level_pyramid_matches_reference_golden(fixture_decoder):
    """G28 through the drop-in class with the reference's own keywords: SDFRenderer(scale_list=..., march_step_list=...) on an odd-sized image,
    fixtures F1 and F2: two levels (also with an explicit last march_step_list entry, renderer.py:724-725), four levels, ratios of 3 and 4
    (renderer.py:713-805), against the reference's outputs and gradients at G24's bars; what is not built (five levels, fractional
    ratios, a list that does not end in 1, ratios above 8) is rejected at construction."""
    import torch
    import test_oracle_vs_golden as tg
    from core.sdfrenderer import SDFRenderer
    from distr import fixture
    g = np.load(os.path.join(GOLDEN, 'g28_two_level_pyramid.npz'))
    H, W = int(g['H']), int(g['W'])
    cases = {
        'two_level_3': ('f1', dict(scale_list=[2, 1], march_step_list=[3, -1])),
        'two_level_6_d2n': ('f1', dict(scale_list=[2, 1], march_step_list=[6, -1], use_depth2normal=True)),
        'two_level_explicit_2_20': ('f1', dict(scale_list=[2, 1], march_step_list=[2, 20], march_step=50)),
        'two_level_3_f2': ('f2', dict(scale_list=[2, 1], march_step_list=[3, -1])),
        'four_level_2_2_2': ('f1', dict(scale_list=[8, 4, 2, 1], march_step_list=[2, 2, 2, -1])),
        'ratio_3': ('f1', dict(scale_list=[3, 1], march_step_list=[3, -1])),
        'ratio_3_2': ('f1', dict(scale_list=[6, 2, 1], march_step_list=[2, 3, -1])),
        'ratio_4': ('f1', dict(scale_list=[4, 1], march_step_list=[4, -1], use_depth2normal=True)),
        'four_level_f2_d2n': ('f2', dict(scale_list=[8, 4, 2, 1], march_step_list=[3, 1, 2, -1], use_depth2normal=True)),
    }
    assert sorted(cases) == sorted(str(n) for n in g['names'])
    wd, wq, wn = (torch.from_numpy(a).cuda() for a in helpers.loss_weights(H, W, 5))
    decs = {'f1': _plain_decoder(fixture_decoder), 'f2': _plain_decoder(fixture.load_fixture_f2())}
    for name, (fx, ckw) in sorted(cases.items()):
        kw = dict(march_step=int(g['march_step']), buffer_size=3, ray_marching_ratio=1.5, use_depth2normal=False)
        kw.update(ckw)
        r = SDFRenderer(decs[fx], g['K'], img_hw=(H, W), **kw)
        lat = torch.from_numpy(g[fx + '.latent']).cuda().requires_grad_(True)
        Rt, Tt = torch.from_numpy(g['R']).cuda().requires_grad_(True), torch.from_numpy(g['T']).cuda().requires_grad_(True)
        depth, normal, mask, mq = r.render(lat, Rt, Tt)
        L = (depth * wd)[mask.bool()].sum() + (mq * wq).sum() + (normal * wn).sum()
        L.backward()
        a = dict(mask=mask.cpu().numpy(), depth=depth.detach().cpu().numpy(), normal=normal.detach().cpu().numpy(), min_sdf=mq.detach().cpu().numpy(),
                 g_latent=lat.grad.cpu().numpy(), g_R=Rt.grad.cpu().numpy(), g_T=Tt.grad.cpu().numpy())
        res = tg.check_g24(a, g, name)
        assert abs(float(L.detach()) - float(g[name + '.loss'])) <= 5e-5 * abs(float(g[name + '.loss'])), (name, float(L.detach()), float(g[name + '.loss']))
        print('G28', name, {k: '%.1e' % v for k, v in res.items()})
    for bad in (dict(scale_list=[16, 8, 4, 2, 1], march_step_list=[2, 2, 2, 2, -1]), dict(scale_list=[3, 2, 1], march_step_list=[3, 3, -1]),
                dict(scale_list=[2, 1], march_step_list=[3, 3, -1]), dict(scale_list=[4, 2], march_step_list=[3, -1]),
                dict(scale_list=[16, 1], march_step_list=[3, -1])):
        with pytest.raises(NotImplementedError):
            SDFRenderer(decs['f1'], g['K'], img_hw=(H, W), **bad)
    with pytest.raises(ValueError):
        SDFRenderer(decs['f1'], g['K'], img_hw=(H, W), scale_list=[2, 1], march_step_list=[0, -1])


@pytest.mark.gpu
def test_renderer_options_match_reference_golden(fixture_decoder):
    """G24 through the drop-in class: every constructor / call option of the golden (identity and permuting transform_matrix,
    use_transform=False, normalize_normal=False, clamp_dist, threshold, radius, march_step_list, ray_marching_ratio, buffer_size 1 / 8,
    img_hw=None) with the reference's own keyword names, against the reference's outputs and gradients; plus render_depth with its own
    default marcher ('recursive', renderer.py:836) and the gradient entering through Zdepth."""
    import torch
    import test_oracle_vs_golden as tg
    from core.sdfrenderer import SDFRenderer
    g = np.load(os.path.join(GOLDEN, 'g24_renderer_options.npz'))
    dec = _plain_decoder(fixture_decoder)
    H, W = int(g['H']), int(g['W'])
    PERM = g['perm']
    cases = {
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
        'img_hw_none': (dict(), dict()),
    }
    assert sorted(list(cases) + ['render_depth_default']) == sorted(str(n) for n in g['names'])
    wd, wq, wn = (torch.from_numpy(a).cuda() for a in helpers.loss_weights(H, W, 5))

    def run(ckw, rkw, img_hw):
        kw = dict(march_step=int(g['march_step']), buffer_size=3, ray_marching_ratio=1.5, use_depth2normal=False)
        kw.update(ckw)
        r = SDFRenderer(dec, g['K'], img_hw=img_hw, **kw)
        assert tuple(r.get_img_hw()) == (H, W)
        lat = torch.from_numpy(g['latent']).cuda().requires_grad_(True)
        Rt, Tt = torch.from_numpy(g['R']).cuda().requires_grad_(True), torch.from_numpy(g['T']).cuda().requires_grad_(True)
        depth, normal, mask, mq = r.render(lat, Rt, Tt, **rkw)
        L = (depth * wd)[mask.bool()].sum() + (mq * wq).sum() + (normal * wn).sum()
        L.backward()
        return dict(mask=mask.cpu().numpy(), depth=depth.detach().cpu().numpy(), normal=normal.detach().cpu().numpy(), min_sdf=mq.detach().cpu().numpy(),
                    g_latent=lat.grad.cpu().numpy(), g_R=Rt.grad.cpu().numpy(), g_T=Tt.grad.cpu().numpy(), loss=float(L.detach()))
    for name, (ckw, rkw) in sorted(cases.items()):
        a = run(ckw, rkw, None if name == 'img_hw_none' else (H, W))
        if name == 'img_hw_none':                                  # size from the intrinsic (renderer.py:31-33): the plain default render
            assert abs(a['loss'] - float(g['img_hw_none.loss'])) <= 2e-5 * abs(float(g['img_hw_none.loss']))
            assert np.abs(a['g_latent'] - g['img_hw_none.g_latent']).max() <= 1e-3 * np.abs(g['img_hw_none.g_latent']).max()
            continue
        res = tg.check_g24(a, g, name)
        assert abs(a['loss'] - float(g[name + '.loss'])) <= 5e-5 * abs(float(g[name + '.loss'])), (name, a['loss'], float(g[name + '.loss']))
        print('G24', name, {k: '%.1e' % v for k, v in res.items()})
    # render_depth(): default marcher 'recursive', gradient through Zdepth and min_sdf
    r = SDFRenderer(dec, g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=2)
    lat = torch.from_numpy(g['latent']).cuda().requires_grad_(True)
    Rt, Tt = torch.from_numpy(g['R']).cuda().requires_grad_(True), torch.from_numpy(g['T']).cuda().requires_grad_(True)
    z, m, q = r.render_depth(lat, Rt, Tt)
    assert m.dtype == torch.bool and np.array_equal(m.cpu().numpy().astype(np.uint8), g['render_depth_default.mask'])
    ((z * torch.from_numpy(g['render_depth_default.gz']).cuda())[m].sum() + 0.5 * q.sum()).backward()
    mb = m.cpu().numpy()
    assert np.abs(z.detach().cpu().numpy() - g['render_depth_default.zdepth'])[mb].max() <= 1e-4
    # off the final mask Zdepth still holds where the march ended (every ray of this view crosses the unit sphere; 1e11 only for rays that
    # miss it, renderer.py:866-869): same values up to stop-step events on rays that never converged
    dz_off = np.abs(z.detach().cpu().numpy() - g['render_depth_default.zdepth'])[~mb]
    assert np.percentile(dz_off, 99) <= 1e-4 and dz_off.max() <= 0.2, (np.percentile(dz_off, 99), dz_off.max())
    assert np.abs(q.detach().cpu().numpy() - g['render_depth_default.q']).max() <= 1e-4
    for k, t in (('g_latent', lat), ('g_R', Rt), ('g_T', Tt)):
        ref = g['render_depth_default.' + k]
        assert np.abs(t.grad.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), k


@pytest.mark.gpu
def test_edge_cases_match_reference_golden(engine):
    """G25 on the HIP path (same checks as the oracle's, tests/test_oracle_vs_golden.py::check_g25): camera inside the unit sphere, far
    away, looking away, and a shape code without a surface."""
    import test_oracle_vs_golden as tg
    g = np.load(os.path.join(GOLDEN, 'g25_edge_cases.npz'))
    H, W = int(g['H']), int(g['W'])
    for name, marcher, d2n in tg.G25_RUNS:
        key = '%s.%s_%s' % (name, marcher, 'd2n' if d2n else 'agn')
        a = helpers.hip_render(engine, H, W, g['K'], g['R'], g[name + '.T'], g[name + '.latent'], march_step=int(g['march_step']),
                               buffer_size=int(g['buffer_size']), marcher=marcher, use_depth2normal=d2n)
        res = tg.check_g25(a, g, key, H, W)
        assert abs(a['loss'] - float(g[key + '.loss'])) <= 5e-5 * abs(float(g[key + '.loss'])), key
        print('G25', key, 'valid', int(a['mask'].sum()), {k: '%.1e' % v for k, v in res.items()})


@pytest.mark.gpu
def test_color_render_gradients_match_reference_golden(fixture_decoder):
    """G26: SDFRenderer_color.render WITHOUT no_grad (its default, renderer_rgb.py:73-125): the colour image stays on the tape -- through
    decode_color to the colour code and the shape code, through the surface points to the camera. Loss over depth, normal, colour and
    min-sdf; gradients w.r.t. colour code, shape code, R, T against the reference's, plain and with a point light (bar: 1e-3 relative or
    2 x the reference's own noise floor)."""
    import torch
    from core.sdfrenderer import SDFRenderer_color
    from core.graph.deep_sdf_decoder import Decoder
    from distr import fixture
    g = dict(np.load(os.path.join(GOLDEN, 'g26_color_render_grad.npz')))
    Ws, bs, _ = fixture_decoder
    cs = int(g['color_size'])
    Wc, bc, code = fixture.make_color_decoder_weights(color_size=cs)
    assert np.array_equal(code, g['color_code'])

    def module(W_, b_, latent, dims, last):
        d = Decoder(latent, dims, last_dim=last, norm_layers=(), latent_in=[4])
        d.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (Wl, bl) in enumerate(zip(W_, b_)) for n, a in (('weight', Wl), ('bias', bl))})
        return d.cuda()
    dims_c = [512] * 8
    dims_c[3] += cs
    dec, dec_c = module(Ws, bs, 256, [512] * 8, 1), module(Wc, bc, 256 + cs, dims_c, 3)
    H, W = int(g['H']), int(g['W'])
    r = SDFRenderer_color(dec, dec_c, g['K'], img_hw=(H, W), march_step=int(g['march_step']), buffer_size=int(g['buffer_size']))
    c = lambda k: torch.from_numpy(g[k]).cuda()
    for tag in ('plain', 'lit'):
        lat, cc = c('latent').requires_grad_(True), c('color_code').requires_grad_(True)
        Rt, Tt = c('R').requires_grad_(True), c('T').requires_grad_(True)
        kw = {} if tag == 'plain' else dict(lighting_locations=c('lights'), lighting_energies=c('energies'))
        d, n, col, m, q = r.render(cc, lat, Rt, Tt, **kw)
        mb = m.bool()
        assert int((m.cpu().numpy() != g[tag + '.mask']).sum()) <= 1
        both = mb.cpu().numpy() & g[tag + '.mask'].astype(bool)
        assert np.percentile(np.abs(col.detach().cpu().numpy() - g[tag + '.color'])[both], 99) <= 1e-4
        L = (d * c('w_d'))[mb].sum() + (n * c('w_n')).sum() + (col * c('w_c')).sum() + (q * c('w_q')).sum()
        L.backward()
        assert abs(float(L.detach()) - float(g[tag + '.loss'])) <= 5e-5 * abs(float(g[tag + '.loss']))
        res = {}
        for k, t in (('g_color_code', cc), ('g_latent', lat), ('g_R', Rt), ('g_T', Tt)):
            ref = g['%s.%s' % (tag, k)]
            res[k] = float(np.abs(t.grad.cpu().numpy().reshape(-1) - ref.reshape(-1)).max() / np.abs(ref).max())
            assert res[k] <= max(1e-3, 2.0 * float(g['%s.%s_floor_rel' % (tag, k)])), (tag, k, res[k])
        print('G26', tag, {k: '%.1e' % v for k, v in res.items()})
    # no_grad=True: nothing but the explicit `R @ normal` term is left (renderer_rgb.py:93-94: the normals are detached before it)
    lat, cc, Rt = c('latent').requires_grad_(True), c('color_code').requires_grad_(True), c('R').requires_grad_(True)
    d, n, col, m, q = r.render(cc, lat, Rt, c('T'), no_grad=True)
    assert not col.requires_grad and not d.requires_grad and not q.requires_grad and n.requires_grad
    n.sum().backward()
    assert (lat.grad is None or float(lat.grad.abs().max()) == 0.0) and cc.grad is None and float(Rt.grad.abs().max()) > 0
