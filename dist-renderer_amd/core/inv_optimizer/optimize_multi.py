"""optimize_multi_view: shape-code (+ optional sim(3)) optimisation on posed multi-view images (reference:
core/inv_optimizer/optimize_multi.py:35-109). Same signature; the evaluator / visualiser hooks (mesh extraction,
chamfer distance, plots: out of scope) are called only when such objects are passed in.

MI355X-first difference: the `num_views_per_round` view pairs of one round are independent given (shape code, sim3), and
each is two small latency-bound renders (137x137 in the PMO setting) that fill a few percent of the chip. They are
therefore issued round-robin on a small pool of HIP streams (`streams=`): libdistr enqueues on the caller's current
stream and never synchronises, PyTorch replays each pair's backward on the stream its forward ran on, so forward and
backward of different pairs overlap on the idle CUs. The loss sum and every gradient are accumulated in the same order
as in the sequential loop (the per-pair losses are added on the main stream in pair order), so results do not depend on
the number of streams.

Multi-GPU (`distributed=`): with torch.distributed initialised (one process per GPU, backend nccl = RCCL), the view
pairs of a round are sharded over the ranks (rank r takes pairs r, r + world, ...: the 8 pairs of the reference's round on 8 GPUs =
one pair each), every rank back-propagates its partial loss, and ONE packed all-reduce of [g_shape_code | g_sim3 (rot 3, scale 1,
trans 3) | loss] (distr.parallel.allreduce_grads) gives every rank the gradient of the whole round; the optimiser step is then
identical everywhere -- no parameter broadcast, no other collective.
"""
import os

import numpy as np
import torch

from core.utils.train_utils import params_to_mtrx

from .loss_multi import compute_loss_color_warp, compute_loss_color_warp_batch


def _dist_state(distributed):
    """(rank, world) the loop shards over; (0, 1) when not distributed. distributed=None: follow torch.distributed."""
    from distr import parallel
    if distributed is False:
        return 0, 1
    rank, world = parallel.rank_world()
    if distributed and world == 1:
        raise RuntimeError('distributed=True but torch.distributed is not initialised with more than one rank '
                           '(distr.parallel.init_from_env() under torch.distributed.run)')
    return rank, world


class _StreamPool(object):
    def __init__(self, n, device):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(int(n), 0))]

    def run(self, i, fn):
        """Runs fn() on stream i of the pool (fork from / join into the current stream). Returns fn's result."""
        if not self.streams:
            return fn()
        main = torch.cuda.current_stream()
        side = self.streams[i % len(self.streams)]
        side.wait_stream(main)
        from distr import binding
        with torch.cuda.stream(side):
            if len(self.streams) > 1:
                with binding.concurrent_section():      # several renders share the device: no sticky tail launches (distr_render_cfg.concurrent)
                    out = fn()
            else:
                out = fn()
        return out

    def join(self, tensors=()):
        if not self.streams:
            return
        main = torch.cuda.current_stream()
        for s in self.streams:
            main.wait_stream(s)
        for t in tensors:
            if torch.is_tensor(t):
                t.record_stream(main)


def pair_indices(idx, i, rot_freq, sep_dist, num_images):
    """The (idx1, idx2) view pair of slot i in round idx (optimize_multi.py:62-65)."""
    idx1 = idx + int(np.floor(i * rot_freq))
    idx2 = idx1 + sep_dist
    if idx2 >= num_images:
        idx1 = num_images - 1
        idx2 = idx1 - sep_dist
    return idx1, idx2


def multi_view_round(renderer, shape_code, images, cameras, pairs, weight_list, sim3=None, sim3_scale=None, visualizer=None,
                     pool=None, batched=True):
    """Sum of compute_loss_color_warp over `pairs` (the body of optimize_multi.py:59-76). batched (default): the 2n depth renders of
    the round run as ONE batched launch sequence (every march step covers the live rays of all views; values and gradients are
    those of the per-pair calls); otherwise -- or with a visualizer, which wants the pairs one by one -- the pairs are issued on the
    stream pool."""
    if batched and visualizer is None and hasattr(renderer, 'render_warp_batch') and 0 < 2 * len(pairs) <= 64:
        results = compute_loss_color_warp_batch(renderer, shape_code, images, cameras, pairs, weight_list, sim3=sim3, sim3_scale=sim3_scale)
        loss_total = 0.0
        for loss, _ in results:
            loss_total = loss_total + loss
        return loss_total, results[-1][1]
    results = []
    for i, (idx1, idx2) in enumerate(pairs):
        fn = (lambda a=idx1, b=idx2: compute_loss_color_warp(renderer, shape_code, images, cameras, a, b, weight_list,
                                                             sim3=sim3, sim3_scale=sim3_scale, visualizer=visualizer))
        results.append(pool.run(i, fn) if pool is not None else fn())
    if pool is not None:
        pool.join([r[0] for r in results])
    loss_total = 0.0
    for loss, _ in results:
        loss_total = loss_total + loss
    return loss_total, results[-1][1]


def optimize_multi_view(renderer, evaluator, shape_code, shape_optimizer, images, cameras, weight_list,
                        num_views_per_round=8, num_iters=20, num_sample_points=30000, sep_dist=1, test_step=5, points_gt=None,
                        sim3=None, sim3_init=None, visualizer=None, vis_dir=None, vis_flag=None, full_flag=True, streams=4,
                        on_round=None, distributed=None, batched=True):
    from distr import parallel
    rank, world = _dist_state(distributed)
    if world > 1:
        parallel.reset_pending_errors()      # (a flag left behind by a loop that ended through another exception must not surface here)
    lead = rank == 0                         # printing / mesh extraction / evaluation happen once, on rank 0
    num_images = len(images)
    rot_freq = num_images / num_views_per_round
    pool = _StreamPool(streams if visualizer is None else 0, shape_code.device)
    if evaluator is not None and vis_dir is not None and lead:
        evaluator.latent_vec_to_points(shape_code, num_points=num_sample_points, fname=os.path.join(vis_dir, 'mesh_initial.ply'), silent=True)
    best_chamfer, best_epoch = 100, 0
    loss_pack = None
    for epoch in range(num_iters):
        for idx in range(0, int(np.ceil(rot_freq)), sep_dist):
            shape_optimizer.zero_grad()
            sim_mtrx, sim3_scale = None, None
            if sim3 is not None:
                m = params_to_mtrx(sim3)
                rot = torch.matmul(m[:3, :3], sim3_init[:3, :3])
                trans = torch.matmul(m[:3, :3], sim3_init[:, 3]) + m[:, 3]
                sim_mtrx = torch.cat([rot, trans[:, None]], dim=1)
                sim3_scale = torch.norm(rot) / np.sqrt(3)
            pairs = [pair_indices(idx, i, rot_freq, sep_dist, num_images) for i in range(num_views_per_round)]
            mine = pairs[rank::world]                  # view-parallel: this rank's share of the round (all of it when world == 1)
            err = None
            loss_total = torch.zeros((), device=shape_code.device)
            try:
                if mine:
                    loss_total, loss_pack = multi_view_round(renderer, shape_code, images, cameras, mine, weight_list, sim3=sim_mtrx,
                                                             sim3_scale=sim3_scale, visualizer=visualizer, pool=pool, batched=batched)
                    loss_total.backward()
            except Exception as e:                     # noqa: BLE001 -- reported to every rank through the step's collective
                if world == 1:
                    raise
                err = e
            if world > 1:                              # ONE collective per step: [g_shape_code | g_sim3 | loss | pack of the last pair | error flag]
                sim_params = [sim3[k] for k in ('rot', 'scale', 'trans')] if sim3 is not None else []
                # the loss pack the reference prints / plots is the one of the round's LAST pair (optimize_multi.py:76-79): its owner
                # contributes the two scalars, everybody else zeros, so every rank ends up with the serial loop's values
                owner = (len(pairs) - 1) % world
                mine_last = err is None and rank == owner and loss_pack is not None
                pk = [loss_pack['color'].t if mine_last else 0.0, loss_pack['l2reg'].t if mine_last else 0.0]
                loss_total, pc, pl = parallel.allreduce_grads([shape_code] + sim_params, [loss_total] + pk, error=err)
                from .loss_multi import _LazyScalar
                loss_pack = {'color': _LazyScalar(pc), 'l2reg': _LazyScalar(pl)}
            shape_optimizer.step()
            if on_round is not None:
                on_round(epoch, idx, loss_total.detach(), loss_pack)
        if vis_flag and visualizer is not None and lead and loss_pack is not None:
            print('[{0}] loss_color: {1:.4f}, loss_l2reg: {2:.4f}\n'.format(epoch, loss_pack['color'], loss_pack['l2reg']))
            visualizer.show_all_data_color_warp(os.path.join(vis_dir, 'vis_{}.png'.format(epoch)))
        if evaluator is not None and vis_dir is not None and epoch % test_step == 0 and lead:
            points_pred = evaluator.latent_vec_to_points(shape_code, num_points=num_sample_points,
                                                         fname=os.path.join(vis_dir, 'mesh_{}.ply'.format(epoch)), silent=True)
            if points_pred is None:
                print('The current latent code does not correspond to a valid shape.')
            elif points_gt is not None and full_flag:
                dist1, dist2 = evaluator.compute_chamfer_distance(points_gt, points_pred, separate=True)
                if (dist1 + dist2) * 1000 < best_chamfer:
                    best_chamfer, best_epoch = (dist1 + dist2) * 1000, epoch
                    evaluator.latent_vec_to_points(shape_code, num_points=num_sample_points, fname=os.path.join(vis_dir, 'mesh_best.ply'), silent=True)
                print('CHAMFER DISTANCE: {0:.3f} & {1:.3f} at epoch {2}'.format(dist1 * 1000, dist2 * 1000, epoch))
                print('BEST SUM CHAMFER DISTANCE: {0:.3f} at epoch {1}'.format(best_chamfer, best_epoch))
    if world > 1:
        parallel.check_pending_errors()      # the last step's (deferred) error flag
    return shape_code, shape_optimizer
