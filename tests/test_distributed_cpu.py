"""world_size-2 gloo test of the view-parallel path (runs on CPU): each rank 'renders' its own view (with the CPU
oracle standing in for the GPU renderer -- test infrastructure), the packed latent gradient is summed with ONE
all-reduce, and every rank ends with the sum over all views."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG


def _worker(rank, world, port, q):
    for p in (PKG, ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from distr import fixture, parallel
    from oracle import oracle as orc
    import helpers
    r, w, _ = parallel.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    orc.lib().orc_set_num_threads(2)
    Ws, bs, latent = fixture.make_decoder_weights()
    O = orc.Oracle(Ws, bs)
    H = W = 20
    K = fixture.make_intrinsic(H, W)
    views = parallel.shard_views(4, rank, world)
    g = torch.zeros(1, 256)
    loss = torch.zeros(1)
    for v in views:
        R, T = fixture.make_camera(45.0 * v, 20.0, 1.6, 0)
        out = helpers.oracle_render(O, orc, H, W, K, R, T, latent, march_step=12, buffer_size=2, marcher='recursive')
        g += torch.from_numpy(out['g_latent'])
        loss += float(out['mask'].sum())
    # strong-scaling partition (C5): the ranks' row bands tile every image exactly once
    cover = torch.zeros(3, 72)
    for (img, r0, r1) in parallel.shard_rows(3, 70, rank, world):
        cover[img, r0:r1] += 1
    parallel.allreduce_packed([g, loss, cover])
    assert bool((cover[:, :70] == 1).all()) and bool((cover[:, 70:] == 0).all())
    mx = parallel.allreduce_max_scalar(float(rank))
    # the view balancer: every rank gathers all step times and derives the same plan; its pieces tile every view once
    times = parallel.allgather_scalar(70.0 if rank == 0 else 40.0)
    assert times == [70.0, 40.0]
    profs = parallel.allgather_vector([1.0 + rank] * 48)
    assert profs == [[1.0] * 48, [2.0] * 48]
    plan = parallel.balance_views(times, 192, profs)
    rows = torch.zeros(2, 192)
    for (v, r0, r1) in plan[rank]:
        rows[v, r0:r1] += 1
    parallel.allreduce_packed([rows])
    assert bool((rows == 1).all()) and len(plan[1]) == 2
    parallel.barrier()
    q.put((rank, g.numpy().copy(), float(loss), mx))
    dist.destroy_process_group()


def test_view_parallel_allreduce_gloo(fixture_decoder):
    from distr import fixture
    from oracle import oracle as orc
    import helpers
    orc.build()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # serial reference: sum over all 4 views
    Ws, bs, latent = fixture_decoder
    O = orc.Oracle(Ws, bs)
    H = W = 20
    K = fixture.make_intrinsic(H, W)
    g_ref, n_ref = np.zeros((1, 256), np.float32), 0.0
    for v in range(4):
        R, T = fixture.make_camera(45.0 * v, 20.0, 1.6, 0)
        out = helpers.oracle_render(O, orc, H, W, K, R, T, latent, march_step=12, buffer_size=2, marcher='recursive')
        g_ref += out['g_latent']
        n_ref += float(out['mask'].sum())
    for rank, g, n, mx in res:
        assert np.abs(g - g_ref).max() <= 1e-5 * np.abs(g_ref).max()
        assert n == n_ref and mx == 1.0


# ------------------------------------------------------------------------------------------------------------------------------
# The PRODUCT's distributed optimisation loops (core.inv_optimizer.optimize_multi_view / optimize_single_view, distributed=True) on
# two gloo ranks. The renderer is a CPU stand-in built from the test oracles (C++ oracle for the two depth renders of a view pair,
# oracle/loss_oracle.py for the warp loss) behind SDFRenderer_warp.render_warp's interface -- test infrastructure; the loops, the
# sharding and the packed all-reduce are the shipped code.
class _OracleDepthFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, latent, R, T, O, cfg):
        out = O.render(cfg, latent.detach().numpy(), R.detach().numpy(), T.detach().numpy())
        ctx.state, ctx.shapes = out['state'], (latent.shape, R.shape, T.shape)
        mask = torch.from_numpy(out['mask'].astype(np.uint8))
        ctx.mark_non_differentiable(mask)
        return torch.from_numpy(out['zdepth'].copy()), mask, torch.from_numpy(out['min_sdf'].copy())

    @staticmethod
    def backward(ctx, gz, gm, gq):
        gl, gR, gT, _ = ctx.state.backward(g_zdepth=gz.numpy(), g_min_sdf=gq.numpy())
        ls, rs, ts = ctx.shapes
        return (torch.from_numpy(gl).reshape(ls), torch.from_numpy(gR).reshape(rs), torch.from_numpy(gT).reshape(ts), None, None)


class _OracleWarpRenderer(object):
    def __init__(self, O, orc, K, H, W, march_step, buffer_size):
        self.O, self.orc, self.K, self.H, self.W = O, orc, np.asarray(K), H, W
        self.cfg = orc.make_cfg(H, W, K, march_step=march_step, buffer_size=buffer_size, marcher='recursive', want_normal=False)

    def render_warp(self, latent, R1, T1, R2, T2, img1, img2, clamp_dist=0.1, profile=False, no_grad_normal=False, thres_depth=0.001):
        from oracle import loss_oracle
        z1, m1, q1 = _OracleDepthFn.apply(latent, R1, T1, self.O, self.cfg)
        with torch.no_grad():
            z2, m2, q2 = _OracleDepthFn.apply(latent, R2, T2, self.O, self.cfg)
        loss, keep, c1, c2 = loss_oracle.warp_loss(self.K, self.H, self.W, z1, m1, z2, img1, img2, R1, T1, R2, T2, thres_depth)
        h, w = self.H, self.W
        return (loss, c1, c2, m1.reshape(h, w), m2.reshape(h, w), q1.reshape(h, w), q2.reshape(h, w), None, None)


class _GCam(object):
    def __init__(self, ext):
        self.extrinsic = np.asarray(ext, np.float32)


def _g9_first_round(distributed):
    """Runs optimize_multi_view on the G9 scene (3 views, 2 pairs per round, sim(3)) and returns loss + gradients of its FIRST round."""
    from conftest import GOLDEN
    from core.inv_optimizer import optimize_multi_view
    from distr import fixture
    from oracle import oracle as orc
    g = dict(np.load(os.path.join(GOLDEN, 'g9_multi_view_round.npz')))
    Ws, bs, _ = fixture.make_decoder_weights()
    orc.lib().orc_set_num_threads(2)
    r = _OracleWarpRenderer(orc.Oracle(Ws, bs), orc, g['K'], int(g['H']), int(g['W']), int(g['march_step']), int(g['buffer_size']))
    lat = torch.from_numpy(g['latent']).clone().requires_grad_(True)
    sim3 = {'rot': torch.from_numpy(g['sim3_rot']).clone().requires_grad_(True), 'scale': torch.tensor(float(g['sim3_scale']), requires_grad=True),
            'trans': torch.from_numpy(g['sim3_trans']).clone().requires_grad_(True)}
    opt = torch.optim.SGD([lat] + list(sim3.values()), lr=0.0)           # gradients are what is compared; parameters stay put
    got = []

    def on_round(epoch, idx, loss, pack):
        if not got:
            got.append([float(loss)] + [t.grad.detach().clone().numpy() for t in (lat, sim3['rot'], sim3['scale'], sim3['trans'])] +
                       [np.array([float(pack['color']), float(pack['l2reg'])])])
    optimize_multi_view(r, None, lat, opt, [torch.from_numpy(i) for i in g['images']], [_GCam(e) for e in g['extrinsics']],
                        {'color': float(g['w_color']), 'l2reg': float(g['w_l2reg'])}, num_views_per_round=2, num_iters=1, sep_dist=1, sim3=sim3,
                        sim3_init=torch.cat([torch.eye(3), torch.zeros(3, 1)], 1), streams=0, on_round=on_round, distributed=distributed)
    return got[0], g


def _multi_worker(rank, world, port, q):
    for p in (PKG, ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from distr import parallel
    parallel.init_from_env(backend='gloo')
    res, _ = _g9_first_round(True)
    q.put((rank, res))
    # multi-scale single-view loop: three renderers sharded over the two ranks, three Adam steps, final code must equal the serial one
    q.put((rank, _single_view_final(True)))
    dist.destroy_process_group()


def _single_view_final(distributed):
    from core.inv_optimizer import optimize_single_view
    from test_host_logic import _FakeRenderer
    h = w = 8
    gt = {'depth': torch.full((h, w), 1.4), 'normal': torch.zeros(h, w, 3) + torch.tensor([0.0, 0.0, -1.0]), 'silhouette': torch.zeros(h, w, dtype=torch.uint8)}
    gt['silhouette'][3:7, 3:7] = 1
    lat = (0.01 * torch.arange(256, dtype=torch.float32).reshape(1, 256) / 256).requires_grad_(True)
    opt = torch.optim.Adam([lat], lr=1e-2)
    wd = dict(w_depth=10.0, w_normal=5.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
    optimize_single_view([_FakeRenderer((h, w)) for _ in range(3)], None, opt, lat, torch.eye(3, 4), gt, wd, num_iters=3, renderer_weights=[1.0, 0.5, 0.25],
                         silent=True, distributed=distributed)
    return lat.detach().numpy().copy()


def test_product_optimisation_loops_distributed_gloo():
    """VERDICT r1 'next' 4: optimize_multi_view(distributed=True) shards the round's view pairs over two ranks, one packed all-reduce of
    [g_shape | g_sim3 (7) | loss]; both ranks end with the gradients of the serial round -- which equal the reference's own numbers for
    this round (golden G9: loss, shape-code and sim(3) gradients). Same for the multi-scale optimize_single_view."""
    torch.set_num_threads(2)
    serial, g = _g9_first_round(False)
    final_serial = _single_view_final(False)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29650 + (os.getpid() % 150)
    procs = [ctx.Process(target=_multi_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # serial run of the shipped loop vs the reference's golden round
    assert abs(serial[0] - float(g['loss_total'])) <= 2e-4 * abs(float(g['loss_total']))
    for a, name in zip(serial[1:5], ('g_latent', 'g_rot', 'g_scale', 'g_trans')):
        rel = np.abs(a - g[name]).max() / np.abs(g[name]).max()
        assert rel <= 1e-2, (name, rel)
    # the loss pack handed to on_round / printed per epoch is the one of the round's LAST pair on every rank (it was rank
    # 0's last LOCAL pair); serial[5] = [color, l2reg] of the serial loop = the reference's packs[-1]
    assert abs(serial[5][0] - g['packs'][-1, 0]) <= 1e-4
    for rank, r in res:
        if isinstance(r, list):            # round gradients of a rank (+ the loss pack of the round's last pair)
            assert abs(r[0] - serial[0]) <= 1e-5 * abs(serial[0]), rank
            for a, b in zip(r[1:], serial[1:]):
                assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max(), rank
        else:                              # final shape code of the multi-scale single-view loop
            assert np.abs(r - final_serial).max() <= 1e-6, rank


def _error_worker(rank, world, port, q):
    for p in (PKG, ROOT, os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    from distr import parallel
    parallel.init_from_env(backend='gloo')
    # (1) mixed float types: an f64 camera tensor next to an f32 shape code; a parameter without a gradient on this rank
    a = torch.zeros(4, dtype=torch.float32, requires_grad=True)
    b = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    a.grad = torch.full((4,), 1.0 + rank)
    if rank == 0:
        b.grad = torch.tensor([1.0 + 2.0 ** -40, 2.0, 3.0], dtype=torch.float64)        # needs more than 24 bits
    loss, = parallel.allreduce_grads([a, b], [torch.tensor(0.5 + rank)])
    ok = bool((a.grad == 3.0).all()) and a.grad.dtype == torch.float32 and b.grad.dtype == torch.float64 and \
        float(b.grad[0]) == 1.0 + 2.0 ** -40 and float(loss) == 2.0
    # (2) a failure on ONE rank inside the step: nobody hangs, the failing rank re-raises its own exception, the other one a RemoteRankError
    kind = 'none'
    try:
        err = ValueError('rank 1 broke') if rank == 1 else None
        parallel.allreduce_grads([a], [torch.tensor(1.0)], error=err)
    except ValueError:
        kind = 'own'
    except parallel.RemoteRankError:
        kind = 'remote'
    parallel.barrier()                       # the group is still usable afterwards
    # (3) the deferred flag (what device buffers use: no device -> host sync inside the step). Step A: rank 1 fails;
    # it raises its own error at once, rank 0 leaves step A normally and must raise RemoteRankError at the START of step B, i.e. before
    # it enters a collective rank 1 will never join.
    lazy_kind, entered_b = 'none', False
    try:
        err = ValueError('rank 1 broke again') if rank == 1 else None
        parallel.allreduce_grads([a], [torch.tensor(1.0)], error=err, lazy=True)
        lazy_kind = 'returned'
        real_all_reduce = dist.all_reduce
        def spy(*args, **kw):
            nonlocal entered_b
            entered_b = True
            return real_all_reduce(*args, **kw)
        dist.all_reduce = spy
        try:
            parallel.allreduce_grads([a], [torch.tensor(1.0)], lazy=True)
        finally:
            dist.all_reduce = real_all_reduce
    except ValueError:
        lazy_kind = 'own'
    except parallel.RemoteRankError:
        lazy_kind = 'remote-next-step' if lazy_kind == 'returned' else 'remote'
    # a clean step leaves nothing pending at the end of a loop
    if rank == 0:
        parallel.check_pending_errors()
    parallel.barrier()
    q.put((rank, ok, kind, lazy_kind, entered_b))
    dist.destroy_process_group()


def test_allreduce_grads_mixed_types_and_error_flag_gloo():
    """allreduce_grads with parameters of different float types (f64 not rounded through f32), a parameter without a
    local gradient, and the error flag that makes all ranks leave a step together when one of them failed."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 90)
    procs = [ctx.Process(target=_error_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in [q.get(timeout=240) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][:2] == (True, 'remote') and res[1][:2] == (True, 'own'), res
    # deferred flag: the healthy rank learns of the failure one step later and BEFORE its next collective
    assert res[0][2:] == ('remote-next-step', False) and res[1][2:] == ('own', False), res
