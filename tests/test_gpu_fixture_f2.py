"""Second fixture + the reference drivers' real regimes under the oracle.

F2 (tests/golden/fixture_f2.npz, oracle/fit_fixture_f2.py): a decoder fitted to a NON-CONVEX shape -- a torus pierced by a thin
plate: thin parts, concavities, rays that cross the surface several times -- because a smooth blob (fixture F1) has few grazing
rays and no multiple hits, exactly the cases where stop-step, top-k and mask decisions could differ. Here: HIP vs the
reference's own goldens on F2 (G1-F2, G3-F2; oracle/gen_golden_f2.py), HIP vs oracle on every pixel at 256 x 256, and the
constructor calls the reference's drivers really make (run_single_shape.py:110-117, run_multi_pmodata.py:92,
run_multi_realdata.py:96), on both fixtures, plus the extra C5 shape codes at 256 x 256."""
import glob
import os

import numpy as np
import pytest

import helpers
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def f2():
    from distr import fixture
    return fixture.load_fixture_f2()


@pytest.fixture(scope='module')
def engine_f2(f2):
    from distr import functions
    return functions.engine_from_weights(f2[0], f2[1], 0)


@pytest.fixture(scope='module')
def engine_f1(engine):
    return engine                      # the session's F1 engine (conftest.py)


@pytest.fixture(scope='module')
def oracle_f2(f2, orc):
    return orc.Oracle(f2[0], f2[1])


def _floor():
    return {k: float(v) for k, v in np.load(os.path.join(GOLDEN, 'noise_floor_f2.npz')).items()}


G1F2_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, 'g1f2_*.npz')))
G1F2_RUNS = [(n, 'f32') for n in G1F2_FILES] + [('g1f2_c1_pyramid_recursive_d2n.npz', a) for a in ('bf16x6', 'f16x3')]     # split modes: one F2 golden


@pytest.mark.parametrize('name,arith', G1F2_RUNS)
def test_f2_hip_matches_reference_goldens(engine_f2, f2, name, arith):
    """HIP path on F2 directly against outputs of the reference itself (64 x 64, 20 steps: three marchers + autograd normals), in the
    exact f32 arithmetic and in the two opt-in split arithmetics (outputs at the same bars, gradients at 5e-3)."""
    from distr import fixture
    g = dict(np.load(os.path.join(GOLDEN, name)))
    assert str(g['fixture']) == 'f2' and fixture.weights_sha256(f2[0], f2[1]) == str(g['weights_sha256'])
    H, W = int(g['H']), int(g['W'])
    a = helpers.hip_render(engine_f2, H, W, g['K'], g['R'], g['T'], g['latent'], seed=int(g['loss_seed']),
                           march_step=int(g['march_step']), buffer_size=int(g['buffer_size']), ratio=float(g['ratio']),
                           marcher=str(g['marcher']), use_depth2normal=bool(g['use_depth2normal']), arith=arith)
    b = dict(mask=g['mask'], depth=g['depth'], zdepth=g['zdepth'], min_sdf=g['min_abs_query'], normal=g['normal'],
             g_latent=g['g_latent'], g_R=g['g_R'], g_T=g['g_T'])
    fl = _floor()
    key = 'c1_%s_%s' % (str(g['marcher']), 'd2n' if bool(g['use_depth2normal']) else 'agn')
    # gradient bar: 2 x what the reference moves by itself under 1e-7 weight noise on this fixture (not below 2e-3)
    tol_grad = max(2e-3, 2.0 * max(fl.get(key + '_g_latent_rel', 0.0), fl.get(key + '_g_R_rel', 0.0), fl.get(key + '_g_T_rel', 0.0)))
    if arith != 'f32':
        # the split arithmetics move sdf values by up to ~9e-7 -- several times the 1e-7 relative weight noise the floors were
        # recorded with -- so one more stop-step / selected-row event than the reference's own noise produces is expected on a 64 x 64
        # image; through depth2normal such an event is worth a few 1e-3 of the gradient (measured: bf16x6 2.7e-3 on the pyramid case)
        tol_grad = max(tol_grad, 5e-3)
    fx = float(g['K'][0, 0])
    res = helpers.compare(a, b, H, W, tol_depth=1e-4, tol_grad=tol_grad,
                          normal_p99=max(1e-4, 1e-5 * fx) if bool(g['use_depth2normal']) else 1e-4)
    print(name, arith, res, 'grad bar', tol_grad)
    assert int(a['mask'].sum()) > 150


def test_f2_c2_hip_matches_reference_golden(engine_f2):
    """C2 on F2 (256 x 256, 50 steps, pyramid_recursive, depth2normal) against the reference: the WHOLE mask (bit-packed in the
    golden), a 32 x 32 crop pixel by pixel, summaries, gradients at 2 x the reference's own noise floor on this fixture."""
    g = dict(np.load(os.path.join(GOLDEN, 'g3f2_c2_pyramid_recursive_d2n.npz')))
    fl = _floor()
    H, W = int(g['H']), int(g['W'])
    a = helpers.hip_render(engine_f2, H, W, g['K'], g['R'], g['T'], g['latent'], seed=int(g['loss_seed']),
                           march_step=int(g['march_step']), buffer_size=int(g['buffer_size']), ratio=float(g['ratio']),
                           marcher=str(g['marcher']), use_depth2normal=True)
    m = a['mask'].reshape(H, W).astype(bool)
    ref_full = np.unpackbits(g['mask_full'])[:H * W].reshape(H, W).astype(bool)
    flips = int((m != ref_full).sum())
    print('F2 C2: valid px %d, mask flips vs the reference %d (the reference against itself under weight noise: %d)'
          % (int(m.sum()), flips, int(fl['c2_flips'])))
    assert flips <= max(2, 2 * int(fl['c2_flips']), int(0.001 * int(g['valid_count'])))
    y0, x0 = int(g['crop_y0']), int(g['crop_x0'])
    sl = (slice(y0, y0 + 32), slice(x0, x0 + 32))
    mc, rc = m[sl], g['mask'].astype(bool)
    both = mc & rc
    assert both.sum() > 200
    assert np.abs(a['depth'][sl] - g['depth'])[both].max() <= 1e-4
    assert np.abs(a['zdepth'].reshape(H, W)[sl] - g['zdepth'])[both].max() <= 1e-4
    assert np.abs(a['min_sdf'].reshape(H, W)[sl] - g['min_abs_query']).max() <= 1e-4
    fx = float(g['K'][0, 0])
    assert np.percentile(np.abs(a['normal'][sl] - g['normal'])[both], 99) <= max(1e-4, 1e-5 * fx)
    assert abs(a['depth'][m].sum(dtype=np.float64) / m.sum() - float(g['sum_depth']) / int(g['valid_count'])) <= 1e-4
    assert abs(a['min_sdf'].sum(dtype=np.float64) - float(g['sum_q'])) / (H * W) <= 1e-5
    for k in ('g_latent', 'g_R', 'g_T'):
        rel = np.abs(a[k].reshape(-1) - g[k].reshape(-1)).max() / np.abs(g[k]).max()
        print('F2 C2 %s: residual %.3e, reference noise floor %.3e' % (k, rel, fl['c2_%s_rel' % k]))
        assert rel <= max(2.0 * fl['c2_%s_rel' % k], 1e-3), (k, rel)


@pytest.mark.parametrize('marcher,d2n', [('pyramid_recursive', True), ('recursive', False), ('trivial', True)])
def test_f2_hip_matches_oracle_256(engine_f2, oracle_f2, orc, f2, marcher, d2n):
    """F2 at 256 x 256 / 50 steps, all 65 536 pixels: HIP vs oracle with ZERO mask flips, depth <= 1e-6, the same number of decoder
    evaluations; prints evaluations per ray (F1 needs 7.0 at C3)."""
    from distr import fixture
    H = W = 128 if marcher == 'trivial' else 256          # (the dense marcher costs the oracle 50 full-image decoder passes: 30 s at 256^2)
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(-40, 35, 1.6, 0)
    kw = dict(march_step=50, buffer_size=3, marcher=marcher, use_depth2normal=d2n)
    a = helpers.hip_render(engine_f2, H, W, K, R, T, f2[2], **kw)
    b = helpers.oracle_render(oracle_f2, orc, H, W, K, R, T, f2[2], **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=2e-4, normal_p99=1e-5, max_flip_frac=0.0)
    assert res['flips'] == 0, res
    assert 3000 * (H * W) // 65536 < int(a['mask'].sum()) < 30000
    print('F2 %dx%d %s: valid px %d, oracle decoder evaluations %d = %.2f per image ray; residuals %s'
          % (H, W, marcher, int(a['mask'].sum()), b['num_evals'], b['num_evals'] / float(H * W), res))


F2_PYRAMIDS = (None, None, [2, 1], [8, 4, 2, 1], [6, 2, 1], [3, 1])


@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_F2', '4'))))      # (soak runs: more seeds)
def test_f2_random_configs_match_oracle(engine_f2, oracle_f2, orc, f2, seed):
    """Seeded random draws on the NON-CONVEX fixture (its lin4 has non-zero xyz columns; rays graze the plate and the torus' hole): image size
    up to 200 px (every tile class from cluster tiles to 64-ray rounds), steps, buffer_size (to 8), ratio, marcher, pyramid, normal mode, camera (one in four INSIDE the sphere) and a
    perturbed shape code -- HIP vs oracle, zero mask flips, depth <= 1e-5."""
    from distr import fixture
    rs = np.random.RandomState(4000 + seed)
    H, W = int(rs.randint(17, 200)), int(rs.randint(17, 200))
    marcher = ['recursive', 'pyramid_recursive', 'pyramid_recursive', 'trivial'][rs.randint(4)]
    S = int(rs.randint(12, 80)) if marcher != 'trivial' else int(rs.randint(6, 14))
    kw = dict(march_step=S, buffer_size=int(rs.randint(1, 9)), ratio=float(rs.choice([1.0, 1.5, 2.0])), marcher=marcher, use_depth2normal=bool(rs.randint(2)))
    kw['buffer_size'] = min(kw['buffer_size'], S)            # (fewer rows than selected rows: refused, the reference's top-k raises)
    sl = F2_PYRAMIDS[rs.randint(len(F2_PYRAMIDS))]
    if marcher == 'pyramid_recursive' and sl is not None:
        kw['scale_list'] = list(sl)
        kw['march_step_list'] = [int(rs.randint(1, 4)) for _ in sl[:-1]] + [-1]
    cam = (float(rs.uniform(-180, 180)), float(rs.uniform(-60, 60)), float(rs.uniform(0.3, 0.8) if rs.randint(4) == 0 else rs.uniform(1.3, 2.2)), float(rs.uniform(-20, 20)))
    latent = (f2[2] + 0.02 * rs.standard_normal(f2[2].shape)).astype(np.float32)
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    a = helpers.hip_render(engine_f2, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(oracle_f2, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-5, tol_grad=1e-3, normal_p99=1e-4)
    print(seed, (H, W), marcher, kw.get('scale_list'), S, res)
    assert res['flips'] == 0, res


# (H, W, march_step, buffer_size, marcher, depth2normal, extra cfg)   -- constructor calls of the reference's drivers
REGIMES = [
    (137, 137, 100, 1, 'pyramid_recursive', False, {}),                 # run_single_shape.py:111 (multiscale, full resolution)
    (68, 68, 100, 3, 'pyramid_recursive', True, {}),                    # run_single_shape.py:112 (1/2 scale of 137)
    (34, 34, 100, 5, 'pyramid_recursive', False, {}),                   # run_single_shape.py:113 (1/4 scale)
    (137, 137, 100, 3, 'pyramid_recursive', True, {}),                  # run_single_shape.py:116 (CLI default buffer_size 3)
    (224, 224, 100, 3, 'pyramid_recursive', True, {}),                  # run_single_shape.py:116 at the 224 x 224 image size
    (224, 224, 100, 5, 'pyramid_recursive', False, {}),
    (160, 120, 200, 1, 'recursive', False, dict(transform_matrix=np.eye(3))),   # run_multi_realdata.py:96 (march_step 200, identity)
]


# every regime on F1; on F2 the three that differ in kind (bs 1 + autograd normals, bs 3 + depth2normal, the 200-step 'recursive' march):
# the CPU oracle's side of the seven regimes costs 50-60 s per fixture (test-time budget of the driver's pytest -m gpu step)
REGIME_RUNS = [(c, 'f1') for c in range(len(REGIMES))] + [(c, 'f2') for c in (0, 3, 6)]


@pytest.mark.parametrize('case,fix', REGIME_RUNS)
def test_driver_regimes_match_oracle(engine_f1, engine_f2, cpu_oracle, oracle_f2, orc, fixture_decoder, f2, case, fix):
    """render() with the reference drivers' own constructor arguments (long marches at small sizes: most steps run on cluster /
    16-ray tiles), both fixtures: HIP vs oracle, zero flips, depth <= 1e-6."""
    from distr import fixture
    H, W, S, bsz, marcher, d2n, extra = REGIMES[case]
    eng, O, lat = (engine_f1, cpu_oracle, fixture_decoder[2]) if fix == 'f1' else (engine_f2, oracle_f2, f2[2])
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(25, 30, 1.6, 0)
    kw = dict(march_step=S, buffer_size=bsz, marcher=marcher, use_depth2normal=d2n)
    kw.update(extra)
    a = helpers.hip_render(eng, H, W, K, R, T, lat, **kw)
    b = helpers.oracle_render(O, orc, H, W, K, R, T, lat, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=2e-4, normal_p99=1e-5, max_flip_frac=0.0)
    assert res['flips'] == 0, res
    assert int(a['mask'].sum()) > 30


@pytest.mark.parametrize('fix', ['f1', 'f2'])
def test_pmo_render_depth_regime_matches_oracle(engine_f1, engine_f2, cpu_oracle, oracle_f2, orc, fixture_decoder, f2, fix):
    """run_multi_pmodata.py:92: SDFRenderer_warp(march_step=100, buffer_size=1) -> render_depth with its default 'recursive'
    marcher at 137 x 137; the gradient enters through Zdepth of every valid pixel (what render_warp sends back)."""
    import torch
    from distr import binding, fixture, functions
    eng, O, latent = (engine_f1, cpu_oracle, fixture_decoder[2]) if fix == 'f1' else (engine_f2, oracle_f2, f2[2])
    H = W = 137
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(-20, 30, 1.6, 0)
    kw = dict(march_step=100, buffer_size=1, marcher='recursive', want_normal=False)
    dev = eng.device
    cfg = binding.make_cfg((H, W), K, **kw)
    lat = torch.from_numpy(latent).to(dev).requires_grad_(True)
    Rt = torch.from_numpy(R).to(dev).requires_grad_(True)
    Tt = torch.from_numpy(T).to(dev).requires_grad_(True)
    z, m, q, _, _ = functions.render_call(eng, cfg, lat, Rt, Tt)
    gz = torch.from_numpy(np.random.RandomState(1).rand(H * W).astype(np.float32)).to(dev)
    (z * gz)[m.bool()].sum().backward()
    out = O.render(orc.make_cfg(H, W, K, **kw), latent, R, T)
    assert np.array_equal(out['mask'], m.cpu().numpy()) and int(out['mask'].sum()) > 500
    gl, gR, gT, _ = out['state'].backward(g_zdepth=gz.cpu().numpy() * out['mask'])
    assert np.abs(z.detach().cpu().numpy() - out['zdepth'])[out['mask'].astype(bool)].max() <= 1e-6
    assert np.abs(q.detach().cpu().numpy() - out['min_sdf']).max() <= 1e-6
    for mine, ref in ((lat.grad, gl), (Rt.grad, gR), (Tt.grad, gT)):
        assert np.abs(mine.cpu().numpy().reshape(-1) - ref.reshape(-1)).max() <= 2e-4 * np.abs(ref).max()


@pytest.mark.parametrize('seed', [1235, 1236, 1237])
def test_c5_shape_codes_match_oracle_256(engine_f1, cpu_oracle, orc, seed):
    """The three extra shape codes of C5's batch of four (bench.py --workload c5: seeds 1235..1237) against the oracle at
    256 x 256 / 100 steps -- they were only ever compared band-vs-full before."""
    from distr import fixture
    latent = fixture.make_latent(seed)
    H = W = 256 if seed == 1235 else 128                   # (one of the three at 256^2: 10 s of CPU oracle each)
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(0, 0, 1.6, 0)
    kw = dict(march_step=100, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)
    a = helpers.hip_render(engine_f1, H, W, K, R, T, latent, **kw)
    b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=2e-4, normal_p99=1e-5, max_flip_frac=0.0)
    assert res['flips'] == 0, res
    assert int(a['mask'].sum()) > 5000 * (H * W) // 65536


def test_normal_only_gradient_matches_reference_golden(engine_f1, engine_f2, fixture_decoder, f2):
    """G27 (oracle/gen_golden_normal_grad.py, the reference itself): the gradient of a loss on the AUTOGRAD normals alone. The reference
    differentiates decode_sdf_gradient(create_graph=True) a second time (decoder_utils.py:76-92, renderer.py:903-909); the MI355X path keeps
    the gradient of `R @ normal` w.r.t. R and omits that decoder-path term by design. Isolated here on both fixtures (F2: non-convex,
    non-zero xyz columns in lin4), normalised and raw normals: every gradient within max(2 x the reference's own noise floor, 1e-4 x the
    size of the FULL loss's gradient) -- the omitted term is 4e-6 of the full gradient at worst (|g_latent| 1.1e-3 against 256, F2 raw
    normals; with normalisation 1e-11: a ReLU decoder is piecewise linear in x, only tanh'' at a surface point where pre ~ 0 is left)."""
    import torch
    from distr import binding, fixture, functions
    g = dict(np.load(os.path.join(GOLDEN, 'g27_normal_only_grad.npz')))
    H, W = int(g['H']), int(g['W'])
    _, _, wn = helpers.loss_weights(H, W, int(g['loss_seed']))
    worst = {}
    for key in [str(c) for c in g['cases']]:
        fx, rest = key.split('_', 1)
        marcher, mode = rest.rsplit('_', 1)
        eng, (Ws, bs, _) = (engine_f1, fixture_decoder) if fx == 'f1' else (engine_f2, f2)
        assert fixture.weights_sha256(Ws, bs) == str(g[fx + '.weights_sha256'])
        dev = eng.device
        cfg = binding.make_cfg((H, W), g['K'], march_step=int(g['march_step']), buffer_size=int(g['buffer_size']), ratio=float(g['ratio']), marcher=marcher,
                               use_depth2normal=False, normalize_normal=(mode == 'unit'))
        lat = torch.from_numpy(g[fx + '.latent'].astype(np.float32)).to(dev).requires_grad_(True)
        Rt = torch.from_numpy(g['R'].astype(np.float32)).to(dev).requires_grad_(True)
        Tt = torch.from_numpy(g['T'].astype(np.float32)).to(dev).requires_grad_(True)
        z, mask, q, depth, normal = functions.render_call(eng, cfg, lat, Rt, Tt)
        assert int((mask.cpu().numpy().reshape(H, W) != g[key + '.mask'].reshape(H, W)).sum()) == 0
        dn = np.abs(normal.detach().cpu().numpy() - g[key + '.normal'])
        assert np.percentile(dn, 99) <= (1e-4 if mode == 'unit' else 3e-3), (key, float(dn.max()))      # (raw normals are 3 x grad f, up to ~30 long)
        L = (normal * torch.from_numpy(wn).to(dev)).sum()
        L.backward()
        for k, t in (('g_latent', lat), ('g_R', Rt), ('g_T', Tt)):
            got = np.zeros_like(g['%s.%s' % (key, k)]) if t.grad is None else t.grad.cpu().numpy().reshape(g['%s.%s' % (key, k)].shape)
            err = float(np.abs(got - g['%s.%s' % (key, k)]).max())
            bar = max(2.0 * float(g['%s.%s_floor' % (key, k)]), 1e-4 * float(g['%s.%s_full_scale' % (key, k)]))
            worst[(key, k)] = (err, bar, err / float(g['%s.%s_full_scale' % (key, k)]))
            assert err <= bar, (key, k, err, bar)
    print('G27 worst error / full-loss gradient size:', max(v[2] for v in worst.values()), {k: '%.2e <= %.2e' % (v[0], v[1]) for k, v in worst.items() if k[1] != 'g_R'})
