"""optimize_single_view: the Adam loop over the shape code or the camera (reference:
core/inv_optimizer/optimize_single.py:35-110). Same signature and the same hooks: per-iteration printing unless `silent`,
the Visualizer's loss curves / image dumps when one is passed, and every `test_step` iterations the evaluator's mesh
extraction + chamfer distance against `points_gt` (those objects are the reference's own CPU tooling; this loop only calls
them). `on_iteration(i, loss_pack, loss)` is an extra callback for callers that want the numbers without printing.

Multi-GPU (`distributed=`): with torch.distributed initialised the renderers of a multi-scale list
(run_single_shape.py:110-117: full, 1/2 and 1/4 resolution) are sharded over the ranks (renderer i on rank i mod world); every rank
back-propagates its weighted partial loss and ONE packed all-reduce of [g_shape_code or g_camera_tensor (7) | loss] makes the
optimiser step identical on all ranks. (One large view is split further by row bands: distr.functions.render_band_call.)
"""
import os

import torch

from core.utils.render_utils import get_camera_from_tensor

from .loss_single import compute_all_loss


def print_loss_pack(loss_pack, name):
    """One line with the five loss terms (same content as the reference's core/visualize/visualizer.py:16-20)."""
    def val(x):
        return float(x.detach().mean()) if hasattr(x, 'detach') else float(x)
    print('NAME = [{0}] -- loss_depth: {1:.4f}, loss_mask_gt: {2:.4f}, loss_mask_out: {3:.4f}, loss_normal: {4:.4f}, '
          'loss_l2reg: {5:.4f}'.format(name, val(loss_pack['depth']), val(loss_pack['mask_gt']), val(loss_pack['mask_out']),
                                       val(loss_pack['normal']), val(loss_pack['l2reg'])))


def _progress(n, silent):
    if not silent:
        try:
            from tqdm import tqdm
            return tqdm(range(n))
        except ImportError:
            pass
    return range(n)


def optimize_single_view(sdfrenderer_list, evaluator, optimizer, shape_code, camera_tensor, gt_pack, weight_dict,
                         optimizer_type='shape', num_iters=200, renderer_weights=None, grad_settings=None, points_gt=None,
                         test_step=50, profile=False, visualizer=None, silent=False, vis_folder=None,
                         ray_marching_type='pyramid_recursive', on_iteration=None, distributed=None, streams=None):
    """streams (not in the reference): the renderers of a multi-scale list (run_single_shape.py:110-113) are independent until their losses
    are summed; each is issued on its own HIP stream of a small pool so that their latency-bound march tails overlap (same values: the
    sum is taken in list order). None = one stream per renderer when the list has several and no visualizer wants them one by one;
    0 = the reference's sequential order."""
    if optimizer_type not in ('shape', 'camera'):
        raise NotImplementedError
    from distr import parallel
    from .optimize_multi import _dist_state, _StreamPool
    rank, world = _dist_state(distributed)
    if world > 1:
        parallel.reset_pending_errors()      # (a flag left behind by a loop that ended through another exception must not surface here)
    silent = silent or rank != 0             # printing / plots / evaluation once, on rank 0
    weights = list(renderer_weights) if renderer_weights else [1.0] * len(sdfrenderer_list)
    if grad_settings is None:
        grad_settings = {'depth': True, 'normal': True, 'silhouette': True}
    visualize = (visualizer is not None) and (not silent)
    if (visualize or (points_gt is not None and not silent)) and vis_folder is not None and not os.path.exists(vis_folder):
        os.mkdir(vis_folder)
    nstreams = (len(sdfrenderer_list) if len(sdfrenderer_list) > 1 else 0) if streams is None else int(streams)
    if visualizer is not None or not shape_code.is_cuda or profile:
        nstreams = 0             # (profile: launch counters and timing events are per context, not per stream -- one scale at a time)
    pool = _StreamPool(nstreams, shape_code.device) if nstreams > 0 else None
    for i in _progress(num_iters, silent):
        optimizer.zero_grad()
        extrinsics = camera_tensor if optimizer_type == 'shape' else get_camera_from_tensor(camera_tensor)
        loss = 0
        err = None
        try:
            if pool is not None:
                # the scales of a multi-scale list on a pool of streams: one latency-bound march per stream, overlapping
                mine = [(idx, r_, w_) for idx, (r_, w_) in enumerate(zip(sdfrenderer_list, weights)) if idx % world == rank]

                def one(renderer, rw):
                    pack, _ = compute_all_loss(renderer, shape_code, extrinsics, gt_pack, threshold=renderer.get_threshold(), profile=profile,
                                               visualizer=None, ray_marching_type=ray_marching_type, grad_settings=dict(grad_settings))
                    return pack, rw * (weight_dict['w_depth'] * pack['depth'] + weight_dict['w_normal'] * pack['normal'] +
                                       weight_dict['w_mask_gt'] * pack['mask_gt'] + weight_dict['w_mask_out'] * pack['mask_out'] +
                                       weight_dict['w_l2reg'] * pack['l2reg'])
                outs = [pool.run(k, lambda r_=r_, w_=w_: one(r_, w_)) for k, (idx, r_, w_) in enumerate(mine)]
                pool.join([o[1] for o in outs])
                for (idx, r_, w_), (pack, part) in zip(mine, outs):
                    loss = loss + part
                    if idx == 0:
                        if not silent:
                            print_loss_pack(pack, '{0}/s224'.format(i))
                        if on_iteration is not None:
                            on_iteration(i, pack, loss)
            for idx, (renderer, rw) in enumerate(zip(sdfrenderer_list, weights) if pool is None else ()):
                if idx % world != rank:          # renderer-parallel: another rank renders this scale
                    continue
                # only the first (full-resolution) renderer of a multi-scale list feeds the visualiser (optimize_single.py:63-74)
                pack, vis_out = compute_all_loss(renderer, shape_code, extrinsics, gt_pack, threshold=renderer.get_threshold(),
                                                 profile=profile, visualizer=visualizer if idx == 0 else None,
                                                 ray_marching_type=ray_marching_type, grad_settings=dict(grad_settings))
                if idx == 0:
                    visualizer = vis_out
                    if not silent:
                        print_loss_pack(pack, '{0}/s224'.format(i))
                    if visualize:
                        visualizer.show_loss_curve(os.path.join(vis_folder, 'vis_loss_curve_{}.png'.format(i)))
                        visualizer.show_all_data(os.path.join(vis_folder, 'vis_all_data_{}.png'.format(i)))
                loss = loss + rw * (weight_dict['w_depth'] * pack['depth'] + weight_dict['w_normal'] * pack['normal'] +
                                    weight_dict['w_mask_gt'] * pack['mask_gt'] + weight_dict['w_mask_out'] * pack['mask_out'] +
                                    weight_dict['w_l2reg'] * pack['l2reg'])
                if on_iteration is not None and idx == 0:
                    on_iteration(i, pack, loss)
            if torch.is_tensor(loss):
                loss.backward()
        except Exception as e:               # noqa: BLE001 -- reported to every rank through the step's collective
            if world == 1:
                raise
            err = e
        if world > 1:                        # ONE collective per step: [gradient of the optimised tensor | loss | error flag]
            target = shape_code if optimizer_type == 'shape' else camera_tensor
            loss, = parallel.allreduce_grads([target], [loss if torch.is_tensor(loss) else torch.zeros((), device=target.device)], error=err)
        if visualize:
            visualizer.add_loss(loss)
        optimizer.step()
        # evaluation every test_step iterations (optimize_single.py:87-98): mesh of the current code + chamfer distance
        if points_gt is not None and (i + 1) % test_step == 0 and not silent and evaluator is not None:
            fname = os.path.join(vis_folder, 'output_{}.ply'.format(i)) if vis_folder is not None else None
            points_tmp = evaluator.latent_vec_to_points(shape_code, fname=fname, silent=True)
            if points_tmp is None:
                print('The current latent code does not correspond to a valid shape.')
                dist = 1e11
            else:
                dist = evaluator.compute_chamfer_distance(points_gt, points_tmp)
                print('CHAMFER DISTANCE: {0:.3f}'.format(dist * 1000))
            if visualize:
                visualizer.add_chamfer(dist)
        if visualize:
            visualizer.dump_all_data(os.path.join(vis_folder, 'vis_all_data_{}.pkl'.format(i)))
    if world > 1:
        parallel.check_pending_errors()      # the last step's (deferred) error flag: a failure on another rank must not end silently
    return (shape_code if optimizer_type == 'shape' else camera_tensor), optimizer
