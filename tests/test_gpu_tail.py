"""The persistent tail launch (k_tail, distr_kernels.hpp): every full-resolution march step from `tail_from` on inside ONE launch --
no launch behind the last live ray (core/sdfrenderer/renderer.py:528-567 breaks out of its loop there). Checked here: renders are
bit-identical to the launch-per-step path for every start step and with clusters / sticky tiles on or off, the hint from the previous
render moves `tail_from` to the sticky regime, batches of views, workgroups that never become resident (their tiles are taken over),
and parity with the CPU oracle through the tail path."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

KEYS = ('zdepth', 'mask', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T')


def _engine(fixture_decoder, **env):
    """A context of its own created under `env` (the DISTR_* knobs are read by distr_create)."""
    from distr import functions
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        Ws, bs, _ = fixture_decoder
        return functions.engine_from_weights(Ws, bs, 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same(a, b):
    return [k for k in KEYS if not np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(b[k]).view(np.uint8))]


def _stats(eng, cfg, latent, R, T):
    """Counters of one more forward through the raw C ABI on a workspace of its own."""
    import ctypes as C
    import torch
    from distr import binding
    dev = eng.device
    P = cfg.band_rows * cfg.W
    ws = torch.empty(eng.ctx.workspace_bytes(cfg)[0], dtype=torch.uint8, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(-1)).to(dev)
    lat, Rt, Tt = t(latent), t(R), t(T)
    o = [torch.empty(P, device=dev), torch.empty(P, dtype=torch.uint8, device=dev), torch.empty(P, device=dev), torch.empty(P, device=dev),
         torch.empty(3 * P, device=dev)]
    p = binding.ptr
    eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(o[4]), p(ws),
                                               ws.numel(), eng.ctx.stream()))
    return eng.ctx.render_stats(cfg, ws), eng.ctx.live_counts(cfg, ws)


CASES = [(64, 64, 20, 'pyramid_recursive', 3), (137, 137, 100, 'pyramid_recursive', 3), (50, 70, 40, 'pyramid_recursive', 5), (64, 64, 60, 'recursive', 1),
         (96, 96, 50, 'recursive', 8)]


@pytest.fixture(scope='module')
def per_step_engine(fixture_decoder):
    return _engine(fixture_decoder, DISTR_TAIL=0)


@pytest.mark.parametrize('variant', [{'DISTR_TAIL_FROM': 0}, {'DISTR_TAIL_FROM': 5}, {'DISTR_TAIL_FROM': 11}, {'DISTR_TAIL_FROM': 0, 'DISTR_CLUSTER': 0},
                                     {'DISTR_TAIL_FROM': 2, 'DISTR_STICKY': 0}, {'DISTR_TAIL_FROM': 1, 'DISTR_CLUSTER': 4}, {'DISTR_TAIL_FROM': 0, 'DISTR_SAVE_MASKS': 0}],
                         ids=lambda v: '-'.join('%s%s' % (k.replace('DISTR_', '').lower(), x) for k, x in v.items()))
def test_tail_launch_is_bit_identical(fixture_decoder, per_step_engine, variant):
    """fwd + bwd through the tail launch from several start steps = the launch-per-step render, byte for byte (outputs AND gradients: the
    saved ReLU masks went through the tail launch's cluster members too)."""
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder, **variant)
    ref_eng = per_step_engine if 'DISTR_SAVE_MASKS' not in variant else _engine(fixture_decoder, DISTR_TAIL=0, DISTR_SAVE_MASKS=0)
    for (H, W, steps, marcher, bs) in CASES:
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(30, 20, 1.6, 10)
        kw = dict(march_step=steps, buffer_size=bs, marcher=marcher, use_depth2normal=True, ratio=1.5)
        a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
        b = helpers.hip_render(ref_eng, H, W, K, R, T, latent, **kw)
        assert _same(a, b) == [], (H, W, steps, marcher, bs, variant, _same(a, b))
        st, _ = _stats(eng, binding.make_cfg((H, W), K, **kw), latent, R, T)
        fine = steps - (6 if marcher == 'pyramid_recursive' else 0)
        want = min(int(variant['DISTR_TAIL_FROM']), fine)
        assert st['tail_from'] == want and st['tail_steals'] == 0, st
        assert st['num_march_launches'] == (steps - fine) + (want + 1 if want < fine else fine), st


def test_tail_hint_moves_the_launch_to_the_sticky_regime(fixture_decoder, per_step_engine):
    """Default policy: a configuration's FIRST render runs launch per step (nothing is known); k_finalize leaves the first step with at
    most 640 live rays in a host-mapped word, and the next render of that configuration starts its tail launch there -- the host never
    synchronises for it. Values never depend on it."""
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder)
    H = W = 137
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(30, 20, 1.6, 10)
    kw = dict(march_step=100, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, ratio=1.5)
    cfg = binding.make_cfg((H, W), K, **kw)
    fine = 94
    st0, counts = _stats(eng, cfg, latent, R, T)
    assert st0['tail_from'] == fine and st0['num_march_launches'] == 100          # first render: no hint yet
    import torch
    torch.cuda.synchronize()
    st1, _ = _stats(eng, cfg, latent, R, T)
    live = counts[6:]
    first = next(i for i, c in enumerate(live) if c <= 640)
    assert st1['tail_from'] == first, (st1, first)
    assert live[first] <= 640 and (first == 0 or live[first - 1] > 640)
    assert st1['num_march_launches'] == 6 + first + 1 and st1['num_point_evals'] == st0['num_point_evals']
    a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
    b = helpers.hip_render(per_step_engine, H, W, K, R, T, latent, **kw)
    assert _same(a, b) == []
    # another camera through the same configuration: the hint is the previous render's, the render still exact
    R2, T2 = fixture.make_camera(140, -35, 1.9, 0)
    a = helpers.hip_render(eng, H, W, K, R2, T2, latent, **kw)
    b = helpers.hip_render(per_step_engine, H, W, K, R2, T2, latent, **kw)
    assert _same(a, b) == []


def test_two_level_pyramid_through_the_tail_launch(fixture_decoder, per_step_engine, cpu_oracle, orc):
    """scale_list=[2, 1] (coarse_steps = (s, 0), golden G28): the half-resolution level feeds the full-resolution march directly; the tail
    launch from step 0 and from step 7 = the launch-per-step render byte for byte, and both match the oracle."""
    from distr import fixture
    _, _, latent = fixture_decoder
    H, W = 61, 75
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(-20, 25, 1.6, 5)
    kw = dict(march_step=60, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, ratio=1.5, coarse_steps=(4, 0))
    b = helpers.hip_render(per_step_engine, H, W, K, R, T, latent, **kw)
    for frm in (0, 7):
        a = helpers.hip_render(_engine(fixture_decoder, DISTR_TAIL_FROM=frm), H, W, K, R, T, latent, **kw)
        assert _same(a, b) == [], (frm, _same(a, b))
    o = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    print(helpers.compare(b, o, H, W, tol_depth=1e-5, tol_grad=1e-3, normal_p99=1e-4))
    # a general pyramid (four levels, ratios 2 x 3 x 2) the same way
    kw = dict(march_step=50, buffer_size=5, marcher='pyramid_recursive', use_depth2normal=False, ratio=1.5, scale_list=[12, 6, 2, 1], march_step_list=[2, 1, 3, -1])
    b = helpers.hip_render(per_step_engine, H, W, K, R, T, latent, **kw)
    a = helpers.hip_render(_engine(fixture_decoder, DISTR_TAIL_FROM=3), H, W, K, R, T, latent, **kw)
    assert _same(a, b) == [], _same(a, b)
    o = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
    print(helpers.compare(b, o, H, W, tol_depth=1e-5, tol_grad=1e-3, normal_p99=1e-4))


@pytest.mark.parametrize('absent', [1, 8, 37, 200])
def test_tail_tiles_of_absent_workgroups_are_taken_over(fixture_decoder, per_step_engine, absent):
    """Co-residency of the tail launch's 256 workgroups is not promised by the hardware. DISTR_TAIL_TEST_ABSENT=n makes the first n leave
    at once: their tiles (single-workgroup tiles, cluster tiles whose lead or helper is missing, sticky tiles) are taken over by the
    others after the wait bound -- same bytes, `tail_steals` > 0, no hang."""
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder, DISTR_TAIL_FROM=4, DISTR_TAIL_TEST_ABSENT=absent)
    H, W, steps = 72, 72, 30
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(30, 20, 1.6, 10)
    kw = dict(march_step=steps, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, ratio=1.5)
    a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
    b = helpers.hip_render(per_step_engine, H, W, K, R, T, latent, **kw)
    assert _same(a, b) == [], (absent, _same(a, b))
    st, _ = _stats(eng, binding.make_cfg((H, W), K, **kw), latent, R, T)
    assert st['tail_from'] == 4 and st['tail_steals'] > 0, st


def test_tail_launch_with_a_batch_of_views(fixture_decoder, per_step_engine):
    """Four views in one launch sequence (distr_render_forward_batch): the tail launch walks the virtual concatenation of the views' live
    lists exactly like the per-step launches; every view equals its stand-alone per-step render."""
    import torch
    from distr import binding, fixture, functions
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder, DISTR_TAIL_FROM=2)
    H = W = 48
    K = fixture.make_intrinsic(H, W)
    cams = [fixture.make_camera(40.0 * i, 15.0 * (i % 3), 1.6 + 0.1 * i, 0) for i in range(4)]
    cfg = binding.make_cfg((H, W), K, march_step=40, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, ratio=1.5)
    dev = eng.device
    lat = torch.from_numpy(np.asarray(latent, np.float32)).to(dev).requires_grad_(True)
    Rs = torch.stack([torch.from_numpy(np.asarray(c[0], np.float32)) for c in cams]).to(dev).requires_grad_(True)
    Ts = torch.stack([torch.from_numpy(np.asarray(c[1], np.float32)) for c in cams]).to(dev).requires_grad_(True)
    outs = functions.render_batch_call(eng, cfg, lat, Rs, Ts)
    wd, wq, wn = (torch.from_numpy(x).to(dev) for x in helpers.loss_weights(H, W, 5))
    L = 0
    for v in range(4):
        mb = outs[1][v].reshape(H, W).bool()
        L = L + (outs[3][v] * wd)[mb].sum() + (outs[2][v].reshape(H, W) * wq).sum() + (outs[4][v] * wn).sum()
    L.backward()
    torch.cuda.synchronize()
    g_sum = np.zeros_like(lat.grad.cpu().numpy(), dtype=np.float64)
    for v in range(4):
        b = helpers.hip_render(per_step_engine, H, W, K, cams[v][0], cams[v][1], latent, march_step=40, buffer_size=3, marcher='pyramid_recursive',
                               use_depth2normal=True, ratio=1.5)
        for k, o in (('zdepth', outs[0][v]), ('mask', outs[1][v]), ('min_sdf', outs[2][v]), ('depth', outs[3][v]), ('normal', outs[4][v])):
            assert np.array_equal(np.asarray(b[k]).reshape(-1).view(np.uint8), o.detach().cpu().numpy().reshape(-1).view(np.uint8)), (v, k)
        assert np.array_equal(b['g_R'].reshape(-1).view(np.uint8), Rs.grad[v].cpu().numpy().reshape(-1).view(np.uint8)), v
        assert np.array_equal(b['g_T'].reshape(-1).view(np.uint8), Ts.grad[v].cpu().numpy().reshape(-1).view(np.uint8)), v
        g_sum += b['g_latent'].astype(np.float64)
    assert np.abs(lat.grad.cpu().numpy() - g_sum).max() <= 1e-5 * np.abs(g_sum).max()


def test_tail_launch_matches_the_oracle(fixture_decoder, cpu_oracle, orc):
    """The oracle parity bars of test_gpu_parity.py through the tail launch (started at step 0: every full-resolution step inside it)."""
    from distr import fixture
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder, DISTR_TAIL_FROM=0)
    for (H, W, steps, marcher) in [(64, 64, 20, 'pyramid_recursive'), (57, 43, 48, 'recursive')]:
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(30, 20, 1.6, 10)
        kw = dict(march_step=steps, buffer_size=3, marcher=marcher, use_depth2normal=True, ratio=1.5)
        a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
        b = helpers.oracle_render(cpu_oracle, orc, H, W, K, R, T, latent, **kw)
        res = helpers.compare(a, b, H, W, tol_depth=1e-6, tol_grad=2e-4, normal_p99=1e-5, max_flip_frac=0.0)
        assert res['flips'] == 0, res


def test_two_tail_launches_on_two_streams(fixture_decoder, per_step_engine):
    """Renders in flight on two streams of one context, each with a tail launch (neither marked `concurrent`): every tail launch wants all
    256 compute units and waits inside the launch for its steps; whatever the dispatcher does with the two, both finish and both are exact."""
    import torch
    from distr import binding, fixture, functions
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder, DISTR_TAIL_FROM=3)
    H = W = 80
    K = fixture.make_intrinsic(H, W)
    kw = dict(march_step=60, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, ratio=1.5)
    cams = [fixture.make_camera(30, 20, 1.6, 10), fixture.make_camera(200, -10, 1.8, 0)]
    refs = [helpers.hip_render(per_step_engine, H, W, K, c[0], c[1], latent, **kw) for c in cams]
    cfg = binding.make_cfg((H, W), K, **kw)
    dev = eng.device
    lat = torch.from_numpy(np.asarray(latent, np.float32)).to(dev)
    RT = [(torch.from_numpy(np.asarray(c[0], np.float32)).to(dev), torch.from_numpy(np.asarray(c[1], np.float32)).to(dev)) for c in cams]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rep in range(8):
        outs = [None, None]
        for i, sm in enumerate(streams):
            with torch.cuda.stream(sm), torch.no_grad():
                for _ in range(3):
                    outs[i] = functions.render_call(eng, cfg, lat, RT[i][0], RT[i][1])
        torch.cuda.synchronize()
        for i in range(2):
            z, mask, q, depth, normal = outs[i]
            for k, o in (('zdepth', z), ('mask', mask), ('min_sdf', q), ('depth', depth), ('normal', normal)):
                assert np.array_equal(np.asarray(refs[i][k]).reshape(-1).view(np.uint8), o.cpu().numpy().reshape(-1).view(np.uint8)), (rep, i, k)


@pytest.mark.parametrize('knobs', [{}, {'DISTR_TAIL_FROM': 0}], ids=['per-step', 'tail'])
def test_cluster_member_dropping_out_behind_its_last_slice(fixture_decoder, per_step_engine, knobs):
    """with saved masks every cluster member stores its OWN words of the rays' mask blocks. A helper that gives up after `go`,
    behind its last slice (a timeout while staging h7), is not missed by the others -- its words would keep what an earlier render left
    there and the backward would run on wrong ReLU masks without any signal. Forced for member 0 of every cluster
    (DISTR_CLUSTER_TEST_ABORT=2): it evaluates the tile on its own and stores whole blocks; gradients stay bit-identical, the event is counted."""
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder, DISTR_CLUSTER_TEST_ABORT=2, **knobs)
    H = W = 64
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(33, 12, 1.6, 0)
    for marcher in ('recursive', 'pyramid_recursive'):
        kw = dict(march_step=60, buffer_size=3, marcher=marcher, use_depth2normal=True, ratio=1.5)
        for rep in range(2):          # (twice: the second render's workspace holds the first one's mask blocks)
            a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
        b = helpers.hip_render(per_step_engine, H, W, K, R, T, latent, **kw)
        assert _same(a, b) == [], (marcher, _same(a, b))
        st, _ = _stats(eng, binding.make_cfg((H, W), K, **kw), latent, R, T)
        assert st['cluster_fallbacks'] > 10, st


@pytest.mark.parametrize('knobs', [{}, {'DISTR_TAIL_FROM': 0}], ids=['per-step', 'tail'])
def test_clusters_spread_over_xcds(fixture_decoder, per_step_engine, knobs):
    """the members of a cluster normally share an XCD (equal workgroup index mod 8); the mixed-XCD path (write-through slice
    stores, chosen by the assembly from the members' XCC ids) was only ever forced with all members still on one XCD. DISTR_CLUSTER_SPREAD=1
    puts the members of every cluster on CONSECUTIVE workgroups = eight different XCDs: the granules really cross XCDs. Same bytes, no fallback."""
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    eng = _engine(fixture_decoder, DISTR_CLUSTER_SPREAD=1, **knobs)
    for (H, W, steps, marcher) in [(64, 64, 60, 'recursive'), (80, 80, 50, 'pyramid_recursive')]:
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(33, 12, 1.6, 0)
        kw = dict(march_step=steps, buffer_size=3, marcher=marcher, use_depth2normal=True, ratio=1.5)
        a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
        b = helpers.hip_render(per_step_engine, H, W, K, R, T, latent, **kw)
        assert _same(a, b) == [], (marcher, _same(a, b))
        st, _ = _stats(eng, binding.make_cfg((H, W), K, **kw), latent, R, T)
        assert st['cluster_fallbacks'] == 0, st


TAIL_PYRAMIDS = (None, None, None, [2, 1], [8, 4, 2, 1], [6, 2, 1], [12, 6, 2, 1])


@pytest.fixture(scope='module')
def tail_engines(fixture_decoder):
    return dict(hint=_engine(fixture_decoder), from0=_engine(fixture_decoder, DISTR_TAIL_FROM=0), from4=_engine(fixture_decoder, DISTR_TAIL_FROM=4),
                from13=_engine(fixture_decoder, DISTR_TAIL_FROM=13))


@pytest.mark.parametrize('seed', range(int(os.environ.get('DISTR_TEST_RANDOM_TAIL', '6'))))      # (soak runs: more seeds)
def test_random_configs_through_the_tail_launch(fixture_decoder, per_step_engine, tail_engines, seed):
    """Seeded random draws (ragged sizes up to 150 px, 30..130 steps, both recursive marchers, pyramids of 2..4 levels, buffer sizes, ratios,
    normal modes, cameras, perturbed shape codes): the render through the tail launch -- started where the previous render's hint puts it
    (the random sweeps against the oracle render every configuration once and therefore never get there) and forced from step 0 / 4 / 13 --
    equals the launch-per-step render byte for byte, outputs and gradients."""
    from distr import fixture
    rs = np.random.RandomState(9000 + seed)
    _, _, latent0 = fixture_decoder
    H, W = int(rs.randint(17, 150)), int(rs.randint(17, 150))
    marcher = ['recursive', 'pyramid_recursive', 'pyramid_recursive'][rs.randint(3)]
    S = int(rs.randint(30, 130))
    kw = dict(march_step=S, buffer_size=int(rs.randint(1, 9)), ratio=float(rs.choice([1.0, 1.5, 2.0])), marcher=marcher, use_depth2normal=bool(rs.randint(2)))
    sl = TAIL_PYRAMIDS[rs.randint(len(TAIL_PYRAMIDS))]
    if marcher == 'pyramid_recursive' and sl is not None:
        kw['scale_list'] = list(sl)
        kw['march_step_list'] = [int(rs.randint(1, 4)) for _ in sl[:-1]] + [-1]
    cam = (float(rs.uniform(-180, 180)), float(rs.uniform(-60, 60)), float(rs.uniform(1.3, 2.2)), float(rs.uniform(-20, 20)))
    latent = (latent0 + 0.05 * rs.standard_normal(latent0.shape)).astype(np.float32)
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    ref = helpers.hip_render(per_step_engine, H, W, K, R, T, latent, **kw)
    first = helpers.hip_render(tail_engines['hint'], H, W, K, R, T, latent, **kw)       # (no hint yet: launch per step; leaves the hint)
    assert _same(first, ref) == [], (seed, 'first', _same(first, ref))
    import torch
    torch.cuda.synchronize()
    for name in ('hint', ('from0', 'from4', 'from13')[rs.randint(3)]):
        a = helpers.hip_render(tail_engines[name], H, W, K, R, T, latent, **kw)
        assert _same(a, ref) == [], (seed, name, (H, W), kw, _same(a, ref))
