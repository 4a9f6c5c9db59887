"""Batched render (distr_render_forward_batch / _backward_batch / distr_render_normal_batch): several views -- same or different
shape codes, per-view no_grad_* options -- in ONE launch sequence. The contract under test: every view of a batch is
BYTE-IDENTICAL, outputs and gradients, to a stand-alone render of that view through the single-view entry points (which the
oracle / golden tests pin). Reference behaviour being batched: the 16 render_depth calls of a multi-view round,
core/inv_optimizer/optimize_multi.py:62-81 + core/sdfrenderer/renderer_warp.py:108-109."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cams(n, elev=20.0):
    from distr import fixture
    return [fixture.make_camera(360.0 / n * i + 7.0, elev if i % 3 else -15.0, 1.6 if i % 4 else 1.45, 5.0 * i) for i in range(n)]


def _raw_single(engine, cfg, lat, R, T, gz, gq, gd, gn):
    """One view through distr_render_forward / distr_render_backward. Returns dict of numpy arrays."""
    import torch
    from distr import binding
    p = binding.ptr
    ctx = engine.ctx
    P = cfg.band_rows * cfg.W
    fwd, bwd = ctx.workspace_bytes(cfg)
    dev = engine.device
    ws = torch.empty(fwd, dtype=torch.uint8, device=dev)
    z, m, q = torch.empty(P, device=dev), torch.empty(P, dtype=torch.uint8, device=dev), torch.empty(P, device=dev)
    d, nm = (torch.empty(P, device=dev), torch.empty(P, 3, device=dev)) if cfg.want_normal else (None, None)
    ctx.check(ctx.L.distr_render_forward(ctx.h, C.byref(cfg), p(lat), p(R), p(T), p(z), p(m), p(q), p(d), p(nm), p(ws), ws.numel(), ctx.stream()))
    gl, gR, gT = torch.empty(256, device=dev), torch.empty(9, device=dev), torch.empty(3, device=dev)
    wsb = torch.empty(bwd, dtype=torch.uint8, device=dev)
    ctx.check(ctx.L.distr_render_backward(ctx.h, C.byref(cfg), p(ws), ws.numel(), p(gz), p(gq), p(gd), p(gn), p(gl), p(gR), p(gT),
                                          p(wsb), wsb.numel(), ctx.stream()))
    st = ctx.render_stats(cfg, ws)
    torch.cuda.synchronize()
    out = dict(z=z, m=m, q=q, gl=gl, gR=gR, gT=gT)
    if cfg.want_normal:
        out.update(d=d, n=nm)
    return {k: v.cpu().numpy() for k, v in out.items()}, st


def _raw_batch(engine, cfg, lats, Rs, Ts, flags, gz, gq, gd, gn):
    import torch
    from distr import binding
    p = binding.ptr
    ctx = engine.ctx
    B = Rs.shape[0]
    P = cfg.band_rows * cfg.W
    fwd, bwd = ctx.workspace_bytes(cfg)
    dev = engine.device
    ws = torch.empty(B * fwd, dtype=torch.uint8, device=dev)
    z, m, q = torch.empty(B, P, device=dev), torch.empty(B, P, dtype=torch.uint8, device=dev), torch.empty(B, P, device=dev)
    d, nm = (torch.empty(B, P, device=dev), torch.empty(B, P, 3, device=dev)) if cfg.want_normal else (None, None)
    fl = None if flags is None else (C.c_int32 * B)(*flags)
    ctx.check(ctx.L.distr_render_forward_batch(ctx.h, C.byref(cfg), B, fl, p(lats), 0 if lats.shape[0] == 1 else 256, p(Rs), p(Ts),
                                               p(z), p(m), p(q), p(d), p(nm), p(ws), ws.numel(), ctx.stream()))
    gl, gR, gT = torch.empty(B, 256, device=dev), torch.empty(B, 9, device=dev), torch.empty(B, 3, device=dev)
    wsb = torch.empty(B * bwd, dtype=torch.uint8, device=dev)
    ctx.check(ctx.L.distr_render_backward_batch(ctx.h, C.byref(cfg), B, p(ws), ws.numel(), p(gz), p(gq), p(gd), p(gn), p(gl), p(gR), p(gT),
                                                p(wsb), wsb.numel(), ctx.stream()))
    stats = [ctx.render_stats(cfg, ws[b * fwd:(b + 1) * fwd]) for b in range(B)]
    torch.cuda.synchronize()
    out = dict(z=z, m=m, q=q, gl=gl, gR=gR, gT=gT)
    if cfg.want_normal:
        out.update(d=d, n=nm)
    return {k: v.cpu().numpy() for k, v in out.items()}, stats


def _check_batch(engine, fixture_decoder, H, W, B, shared, flags, band=None, require_valid=True, **kw):
    import torch
    from distr import binding, fixture
    _, _, latent = fixture_decoder
    dev = engine.device
    K = fixture.make_intrinsic(H, W)
    cfg = binding.make_cfg((H, W), K, band=band, **kw)
    P = cfg.band_rows * W
    lat_np = [latent] + [fixture.make_latent(1300 + i) for i in range(1, B)]
    lats = torch.from_numpy(np.concatenate(lat_np[:1 if shared else B], 0)).to(dev).contiguous()
    cams = _cams(B)
    Rs = torch.from_numpy(np.stack([c[0] for c in cams]).astype(np.float32)).to(dev).reshape(B, 9).contiguous()
    Ts = torch.from_numpy(np.stack([c[1] for c in cams]).astype(np.float32)).to(dev).contiguous()
    rs = np.random.RandomState(11)
    gz, gq, gd = (torch.from_numpy(rs.randn(B, P).astype(np.float32)).to(dev) for _ in range(3))
    gn = torch.from_numpy(rs.randn(B, P, 3).astype(np.float32)).to(dev)
    if not cfg.want_normal:
        gd = gn = None
    got, stats = _raw_batch(engine, cfg, lats, Rs, Ts, flags, gz, gq, gd, gn)
    evals = 0
    for b in range(B):
        c1 = cfg.clone()
        if flags is not None:
            c1.grad_depth, c1.grad_mask, c1.grad_camera = int(bool(flags[b] & 1)), int(bool(flags[b] & 2)), int(bool(flags[b] & 4))
            # the workspace layout of a stand-alone render follows ITS cfg; the batch's follows the union -> same values either way
        ref, st = _raw_single(engine, c1, lats[0 if shared else b], Rs[b], Ts[b], gz[b], gq[b], None if gd is None else gd[b],
                              None if gn is None else gn[b])
        for k in ref:
            assert got[k][b].tobytes() == ref[k].tobytes(), (b, k, np.abs(got[k][b].astype(np.float64) - ref[k]).max())
        for k in ('num_in_sphere', 'num_point_evals', 'num_valid', 'num_grad_samples'):
            assert stats[b][k] == st[k], (b, k, stats[b][k], st[k])
        assert stats[b]['num_valid'] > 0 or b > 0 or not require_valid
        evals += st['num_point_evals']
    return evals


def test_batch_16_views_of_137_equals_per_view(engine, fixture_decoder):
    """The multi-view round's shape: 16 views of 137 x 137 with ONE shape code, march_step 100, buffer_size 1, 'recursive'
    (run_multi_pmodata.py:92), every second view rendered with no_grad_depth (renderer_warp.py:109)."""
    flags = [7, 6] * 8
    _check_batch(engine, fixture_decoder, 137, 137, 16, True, flags, march_step=100, buffer_size=1, marcher='recursive', want_normal=False)


@pytest.mark.parametrize('case', [
    dict(H=96, W=80, B=4, shared=False, flags=None, kw=dict(march_step=40, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)),
    dict(H=64, W=72, B=3, shared=False, flags=[7, 3, 5], kw=dict(march_step=30, buffer_size=2, marcher='pyramid_recursive', use_depth2normal=False)),
    dict(H=40, W=40, B=3, shared=True, flags=None, kw=dict(march_step=12, buffer_size=3, marcher='trivial', use_depth2normal=True)),
    dict(H=256, W=256, B=5, shared=False, flags=None, kw=dict(march_step=50, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)),
    dict(H=33, W=47, B=64, shared=True, flags=None, kw=dict(march_step=25, buffer_size=1, marcher='recursive', want_normal=False)),
    dict(H=128, W=96, B=2, shared=False, flags=None, band=(32, 64), kw=dict(march_step=30, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)),
    dict(H=75, W=61, B=5, shared=False, flags=[7, 6, 7, 3, 5], kw=dict(march_step=60, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True,
                                                                         scale_list=[8, 4, 2, 1], march_step_list=[2, 1, 3, -1])),
    dict(H=90, W=64, B=3, shared=True, flags=None, band=(40, 50), kw=dict(march_step=30, buffer_size=2, marcher='pyramid_recursive', use_depth2normal=False,
                                                                             scale_list=[2, 1], march_step_list=[4, -1])),
], ids=['shapes4-pyramid-d2n', 'autograd-normals-flags', 'trivial', 'c2-size-5-shapes', 'max-batch-64', 'row-band', 'four-level-pyramid', 'two-level-pyramid-band'])
def test_batch_equals_per_view(engine, fixture_decoder, case):
    """Different shape codes per view (a batch of shapes, BASELINE.json configs[4]), all marchers, both normal modes, per-view
    flags, ragged sizes, the largest batch, a row band: byte equality with the stand-alone renders, forward and backward."""
    _check_batch(engine, fixture_decoder, case['H'], case['W'], case['B'], case['shared'], case['flags'], band=case.get('band'), **case['kw'])


@pytest.mark.parametrize('seed', range(int(__import__('os').environ.get('DISTR_TEST_RANDOM_BATCH', '4'))))      # (soak runs: more seeds)
def test_random_batches_equal_per_view(engine, fixture_decoder, seed):
    """Seeded random batches: 2..12 views, shared or per-view shape codes, random per-view no_grad flags, ragged sizes, every marcher, pyramids of
    2..4 levels, both normal modes, now and then a row band -- every view byte-identical (outputs, gradients, counters) to its stand-alone render."""
    rs = np.random.RandomState(12000 + seed)
    H, W = int(rs.randint(17, 120)), int(rs.randint(17, 120))
    B = int(rs.randint(2, 13))
    marcher = ['recursive', 'pyramid_recursive', 'pyramid_recursive', 'trivial'][rs.randint(4)]
    S = int(rs.randint(12, 90)) if marcher != 'trivial' else int(rs.randint(6, 14))
    kw = dict(march_step=S, buffer_size=int(rs.randint(1, 6)), ratio=float(rs.choice([1.0, 1.5, 2.0])), marcher=marcher)
    mode = rs.randint(3)
    if mode == 0:
        kw['want_normal'] = False
    else:
        kw['use_depth2normal'] = bool(mode == 1)
    sl = (None, None, [2, 1], [8, 4, 2, 1], [6, 2, 1])[rs.randint(5)]
    if marcher == 'pyramid_recursive' and sl is not None:
        kw['scale_list'] = list(sl)
        kw['march_step_list'] = [int(rs.randint(1, 4)) for _ in sl[:-1]] + [-1]
    band = None
    if rs.randint(4) == 0 and (marcher != 'pyramid_recursive' or sl is None or sl == [2, 1]) and H >= 24:
        r0 = 4 * int(rs.randint(0, H // 4 - 1))
        rows = min(H - r0, 4 * int(rs.randint(1, 8)))
        if r0 + rows < H:
            rows -= rows % 4
        band = (r0, max(rows, 4)) if r0 + max(rows, 4) <= H else None
    flags = None if rs.randint(2) else [int(rs.choice([7, 6, 5, 3, 7, 7])) for _ in range(B)]
    _check_batch(engine, fixture_decoder, H, W, B, bool(rs.randint(2)), flags, band=band, require_valid=False, **kw)     # (a thin band may miss the object, a short march may converge nowhere)


def test_batch_argument_checks(engine, fixture_decoder):
    import torch
    from distr import binding, fixture
    p = binding.ptr
    ctx = engine.ctx
    cfg = binding.make_cfg((32, 32), fixture.make_intrinsic(32, 32), march_step=10, buffer_size=1, marcher='recursive', want_normal=False,
                           grad_depth=False)
    fwd, _ = ctx.workspace_bytes(cfg)
    t = torch.zeros(4 * 32 * 32 * 4, device='cuda')
    ws = torch.empty(2 * fwd, dtype=torch.uint8, device='cuda')

    def call(n, flags, wsb, stride=0):
        fl = None if flags is None else (C.c_int32 * len(flags))(*flags)
        return ctx.L.distr_render_forward_batch(ctx.h, C.byref(cfg), n, fl, p(t), stride, p(t), p(t), p(t), p(t), p(t), None, None, p(ws), wsb, ctx.stream())
    assert call(0, None, ws.numel()) == -1 and call(65, None, ws.numel()) == -1
    assert call(3, None, ws.numel()) == -4                       # workspace holds two views
    assert call(2, [6, 7], ws.numel()) == -1                     # view 1 asks for the depth gradient cfg has switched off
    assert b'view_flags[1]' in ctx.L.distr_last_error(ctx.h)
    assert call(2, None, ws.numel(), stride=100) == -1           # stride shorter than a shape code
    assert call(2, [6, 2], ws.numel()) == 0
    torch.cuda.synchronize()


def test_batch_autograd_function_and_warp_round(fixture_decoder):
    """The product path: SDFRenderer_warp.render_warp_batch vs render_warp pair by pair (identical outputs), and one round of
    multi_view_round batched vs on the stream pool (same loss; gradients equal up to the order in which the views' shape-code
    gradients are added)."""
    import torch
    from core.graph.deep_sdf_decoder import Decoder
    from core.inv_optimizer import multi_view_round
    from core.inv_optimizer.optimize_multi import _StreamPool, pair_indices
    from core.sdfrenderer import SDFRenderer_warp
    from distr import fixture
    from oracle.gen_synth import procedural_images
    Ws, bs, latent = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W), ('bias', b))})
    size = 72
    K = fixture.make_intrinsic(size, size)
    r = SDFRenderer_warp(dec.cuda(), K, img_hw=(size, size), march_step=60, buffer_size=1)

    class Cam(object):
        def __init__(self, R, T):
            self.extrinsic = np.concatenate([R, T[:, None]], 1).astype(np.float32)
    n_img = 12
    cams = [Cam(*fixture.make_camera(30.0 * i, 20.0, 1.6, 0.0)) for i in range(n_img)]
    a, b = procedural_images(size, size)
    imgs = [torch.from_numpy(np.roll(a if i % 2 else b, 3 * i, axis=1).copy()).cuda() for i in range(n_img)]
    pairs = [pair_indices(0, i, n_img / 4, 1, n_img) for i in range(4)]
    w = {'color': 5.0, 'l2reg': 1.0}
    res = {}
    for mode in ('batched', 'pool'):
        lat = torch.from_numpy(latent).cuda().requires_grad_(True)
        total, pack = multi_view_round(r, lat, imgs, cams, pairs, w, pool=_StreamPool(2, lat.device), batched=(mode == 'batched'))
        total.backward()
        torch.cuda.synchronize()
        res[mode] = (float(total), lat.grad.cpu().numpy(), float(pack['color']))
    assert res['batched'][0] == res['pool'][0] and res['batched'][2] == res['pool'][2]
    ga, gb = res['batched'][1], res['pool'][1]
    assert np.abs(ga - gb).max() <= 2e-6 * np.abs(gb).max()
    # pair by pair: every element of the 9-tuple
    from core.inv_optimizer.loss_multi import pair_cameras
    lat = torch.from_numpy(latent).cuda().requires_grad_(True)
    args = []
    for (i1, i2) in pairs:
        (R1, T1), (R2, T2) = pair_cameras(cams, i1, i2, lat.device)
        args.append((R1, T1, R2, T2, imgs[i1], imgs[i2]))
    outs = r.render_warp_batch(lat, args)
    for arg, out in zip(args, outs):
        ref = r.render_warp(lat, *arg, no_grad_normal=True)
        for x, y in zip(out, ref):
            assert torch.equal(x.detach(), y.detach())


# ------------------------------------------------------------------------------------------------------ split-bf16 march (opt-in)
def _render_arith(engine, latent, H, W, cam, arith, **kw):
    import helpers
    from distr import fixture
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(*cam)
    return helpers.hip_render(engine, H, W, K, R, T, latent, arith=arith, **kw)


@pytest.mark.parametrize('case', [
    dict(H=256, W=256, cam=(-40, 25, 1.6, 0), kw=dict(march_step=50, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)),
    dict(H=160, W=120, cam=(20, 10, 1.6, 5), kw=dict(march_step=40, buffer_size=2, marcher='recursive', use_depth2normal=False)),
    dict(H=48, W=48, cam=(0, 0, 1.6, 0), kw=dict(march_step=12, buffer_size=3, marcher='trivial', use_depth2normal=True)),
], ids=['c2-pyramid-d2n', 'recursive-agn', 'trivial'])
@pytest.mark.parametrize('arith', ['bf16x6', 'f16x3'])
def test_split_bf16_march_is_f32_equivalent(engine, fixture_decoder, case, arith):
    """distr_render_cfg.arith = DISTR_ARITH_BF16X6 / DISTR_ARITH_F16X3 (opt-in): the whole render -- march, selection, outputs, backward on the ReLU masks
    this forward saved -- against the exact f32 render (which the oracle pins bit for bit): at most a handful of stop-step /
    threshold flips, depth and min-sdf within 1e-4 on commonly valid pixels (north_star's bar; measured ~1e-6 on almost all),
    gradients within 2e-3; and the mode is self-consistent: two renders are bit-identical (64- and 32-ray tiles compute a ray
    identically, so the nondeterministic order of the live lists does not matter)."""
    _, _, latent = fixture_decoder
    H, W = case['H'], case['W']
    a = _render_arith(engine, latent, H, W, case['cam'], arith, **case['kw'])
    b = _render_arith(engine, latent, H, W, case['cam'], 'f32', **case['kw'])
    a2 = _render_arith(engine, latent, H, W, case['cam'], arith, **case['kw'])
    for k in ('zdepth', 'mask', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T'):
        assert a[k].tobytes() == a2[k].tobytes(), k                          # reproducible bit for bit
    ma, mb = a['mask'].reshape(H, W).astype(bool), b['mask'].reshape(H, W).astype(bool)
    flips = int((ma != mb).sum())
    both = ma & mb
    dz = np.abs(a['zdepth'].reshape(H, W) - b['zdepth'].reshape(H, W))[both]
    dq = np.abs(a['min_sdf'] - b['min_sdf'])
    res = dict(flips=flips, valid=int(mb.sum()), zdepth_max=float(dz.max()), zdepth_p99=float(np.percentile(dz, 99)), zdepth_p999=float(np.percentile(dz, 99.9)),
               zdepth_median=float(np.median(dz)), min_sdf_max=float(dq.max()))
    for k in ('g_latent', 'g_R', 'g_T'):
        res[k] = float(np.abs(a[k] - b[k]).max() / np.abs(b[k]).max())
    print(arith, 'vs f32', case['kw']['marcher'], res)
    assert flips <= max(2, int(0.001 * mb.sum())), res
    # (a ray whose |sdf| lands within ~1e-6 of the stop threshold stops a step earlier or later: ~0.1 % of the pixels move by ~1e-5,
    # the mechanism behind the reference's own noise floor, tests/golden/noise_floor_c1.npz: depth 6.7e-5 under 1e-7 weight noise)
    assert res['zdepth_p99'] <= 5e-6 and res['zdepth_p999'] <= 5e-5 and res['zdepth_max'] <= 1e-4 and res['min_sdf_max'] <= 1e-4, res
    assert max(res['g_latent'], res['g_R'], res['g_T']) <= (2e-2 if case['kw']['use_depth2normal'] else 2e-3), res


@pytest.mark.parametrize('arith', ['bf16x6', 'f16x3'])
def test_split_bf16_batch_and_band_consistency(engine, fixture_decoder, arith):
    """Within the split arithmetics the structural identities of the exact path hold too: a batch of views equals the
    stand-alone renders byte for byte, forward and backward."""
    _check_batch(engine, fixture_decoder, 96, 80, 3, False, None, march_step=30, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True,
                 arith=arith)


@pytest.mark.parametrize('arith', ['bf16x6', 'f16x3'])
def test_split_arithmetics_over_several_shapes(engine, fixture_decoder, arith):
    """The opt-in arithmetics on the other shape codes of BASELINE config C5 (latent seeds 1235..1237) and from two more cameras: as
    close to the exact render as on the fixture's own code (<= 0.1 % flips, depth p99 <= 5e-6, max <= 1e-4), and -- for the split-f16
    form, whose validity depends on the activations' range -- not one overflow."""
    import ctypes as C
    import torch
    from distr import binding, fixture, functions
    _, _, latent0 = fixture_decoder
    H = W = 160
    K = fixture.make_intrinsic(H, W)
    dev = engine.device
    p = binding.ptr
    for i, seed in enumerate((1235, 1236, 1237)):
        latent = fixture.make_latent(seed)
        cam = ((35.0 * i, 15.0, 1.6, 0.0), (-60.0, 30.0 - 10.0 * i, 1.7, 5.0))[i % 2]
        a = _render_arith(engine, latent, H, W, cam, arith, march_step=50, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)
        b = _render_arith(engine, latent, H, W, cam, 'f32', march_step=50, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)
        ma, mb = a['mask'].reshape(H, W).astype(bool), b['mask'].reshape(H, W).astype(bool)
        both = ma & mb
        dz = np.abs(a['zdepth'].reshape(H, W) - b['zdepth'].reshape(H, W))[both]
        res = dict(seed=seed, valid=int(mb.sum()), flips=int((ma != mb).sum()), p99=float(np.percentile(dz, 99)), zmax=float(dz.max()),
                   g=float(np.abs(a['g_latent'] - b['g_latent']).max() / np.abs(b['g_latent']).max()))
        print(arith, res)
        assert mb.sum() > 500 and res['flips'] <= max(2, int(0.001 * mb.sum())) and res['p99'] <= 5e-6 and res['zmax'] <= 1e-4, res
        assert np.isfinite(a['g_latent']).all() and res['g'] <= 2e-2, res
        if arith == 'f16x3':      # the range check of the march: counters of one forward
            R, T = fixture.make_camera(*cam)
            cfg = binding.make_cfg((H, W), K, march_step=50, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, arith=arith)
            n = H * W
            ws = torch.empty(engine.ctx.workspace_bytes(cfg)[0], dtype=torch.uint8, device=dev)
            outs = [torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev),
                    torch.empty(n, 3, device=dev)]
            engine.ctx.check(engine.ctx.L.distr_render_forward(engine.ctx.h, C.byref(cfg), p(torch.from_numpy(latent).to(dev)),
                                                               p(torch.from_numpy(R).to(dev).reshape(-1).contiguous()), p(torch.from_numpy(T).to(dev)),
                                                               p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(ws), ws.numel(), engine.ctx.stream()))
            assert engine.ctx.render_stats(cfg, ws)['f16_overflows'] == 0


def test_renderer_takes_its_default_arithmetic_from_the_environment(fixture_decoder, monkeypatch):
    """What `python -m distr.launch --arith f16x3 <driver>` arranges for renderers the driver constructs without `arith=`: the
    renderer really renders in that mode (depth within the mode's distance of the exact render, not equal to it), an explicit
    `arith=` wins, an unknown name raises."""
    import torch
    from core.graph.deep_sdf_decoder import Decoder
    from core.sdfrenderer import SDFRenderer
    from distr import fixture
    Ws, bs, latent = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    size = 96
    K = fixture.make_intrinsic(size, size)
    R, T = fixture.make_camera(30.0, 20.0, 1.6, 0.0)
    lat, Rt, Tt = (torch.from_numpy(a).cuda() for a in (latent, R, T))
    monkeypatch.delenv('DISTR_ARITH', raising=False)
    exact = SDFRenderer(dec, K, img_hw=(size, size), march_step=30, buffer_size=3)
    assert exact.arith == 'f32'
    monkeypatch.setenv('DISTR_ARITH', 'f16x3')
    fast = SDFRenderer(dec, K, img_hw=(size, size), march_step=30, buffer_size=3)
    assert fast.arith == 'f16x3' and SDFRenderer(dec, K, img_hw=(size, size), arith='f32').arith == 'f32'
    with torch.no_grad():
        da, ma = exact.render_depth(lat, Rt, Tt)[:2]
        db, mb = fast.render_depth(lat, Rt, Tt)[:2]
    both = (ma.reshape(-1) > 0) & (mb.reshape(-1) > 0)
    d = (da.reshape(-1) - db.reshape(-1)).abs()[both]
    assert int(both.sum()) > 500 and float(d.max()) <= 1e-4 and float(d.max()) > 0.0
    monkeypatch.setenv('DISTR_ARITH', 'fp8')
    with pytest.raises(ValueError, match='DISTR_ARITH'):
        SDFRenderer(dec, K, img_hw=(size, size))


def test_render_depth_batch_mixed_no_grad_flags_equal_per_view_calls(fixture_decoder):
    """SDFRenderer.render_depth_batch with DIFFERENT no_grad_* options per view -- including one view with no_grad_mask AND
    no_grad_camera, whose min_sdf row render_depth detaches (renderer.py:388-389 + 863) -- must give, for a loss over all outputs of all
    views, exactly the gradients of the per-view render_depth calls with the same options (the batch used to leak that view's min-sdf
    gradient of the rays that miss the sphere into g_R / g_T)."""
    import torch
    from core.sdfrenderer import SDFRenderer
    from core.graph.deep_sdf_decoder import Decoder
    from distr import fixture
    Ws, bs, latent = fixture_decoder
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W_, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W_), ('bias', b))})
    dec = dec.cuda()
    H = W = 40
    K = fixture.make_intrinsic(H, W)
    r = SDFRenderer(dec, K, img_hw=(H, W), march_step=24, buffer_size=2)
    cams = [fixture.make_camera(20.0 * i, 10.0, 2.6 if i == 1 else 1.6, 0.0) for i in range(4)]      # view 1 is far: most of its rays miss the sphere
    ngd, ngm, ngc = [False, False, True, False], [False, True, False, True], [False, True, False, False]
    rs = np.random.RandomState(3)
    wz, wq = (torch.from_numpy(rs.rand(4, H * W).astype(np.float32)).cuda() for _ in range(2))

    def leaves():
        lat = torch.from_numpy(latent).cuda().requires_grad_(True)
        Rs = [torch.from_numpy(c[0]).cuda().requires_grad_(True) for c in cams]
        Ts = [torch.from_numpy(c[1]).cuda().requires_grad_(True) for c in cams]
        return lat, Rs, Ts
    lat, Rs, Ts = leaves()
    Z, M, Q = r.render_depth_batch(lat, Rs, Ts, no_grad_depth=ngd, no_grad_mask=ngm, no_grad_camera=ngc)
    (torch.where(M, Z * wz, torch.zeros_like(Z)).sum() + (Q * wq).sum()).backward()
    lat2, Rs2, Ts2 = leaves()
    total = 0
    for v in range(4):
        z, m, q = r.render_depth(lat2, Rs2[v], Ts2[v], no_grad_depth=ngd[v], no_grad_mask=ngm[v], no_grad_camera=ngc[v])
        assert torch.equal(z.detach(), Z[v].detach()) and torch.equal(m, M[v]) and torch.equal(q.detach(), Q[v].detach())
        total = total + torch.where(m, z * wz[v], torch.zeros_like(z)).sum() + (q * wq[v]).sum()
    total.backward()
    assert int((~(Z[1] < 1e10)).sum()) > 100                                    # view 1 has rays that miss the sphere
    for v in range(4):
        for a, b, nm in ((Rs[v].grad, Rs2[v].grad, 'R'), (Ts[v].grad, Ts2[v].grad, 'T')):
            if b is None:
                assert a is None or float(a.abs().max()) == 0.0, (v, nm)
            else:
                assert torch.equal(a, b), (v, nm, (a - b).abs().max())
    assert Rs2[1].grad is None or float(Rs2[1].grad.abs().max()) == 0.0           # both flags set: nothing reaches view 1's camera
    assert (lat.grad - lat2.grad).abs().max() <= 2e-6 * lat2.grad.abs().max()      # (a shared code: the sum over views, order differs)
