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
