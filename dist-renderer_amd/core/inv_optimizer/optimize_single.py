"""optimize_single_view: the Adam loop over the shape code or the camera (reference:
core/inv_optimizer/optimize_single.py:35-110), without the visualiser / mesh-evaluation hooks (out of scope).
`on_iteration(i, loss_pack, loss)` replaces the reference's printing / plotting callbacks."""
from core.utils.render_utils import get_camera_from_tensor

from .loss_single import compute_all_loss


def optimize_single_view(sdfrenderer_list, evaluator, optimizer, shape_code, camera_tensor, gt_pack, weight_dict,
                         optimizer_type='shape', num_iters=200, renderer_weights=None, grad_settings=None, points_gt=None,
                         test_step=50, profile=False, visualizer=None, silent=True, vis_folder=None,
                         ray_marching_type='pyramid_recursive', on_iteration=None):
    if optimizer_type not in ('shape', 'camera'):
        raise NotImplementedError
    weights = list(renderer_weights) if renderer_weights else [1.0] * len(sdfrenderer_list)
    if grad_settings is None:
        grad_settings = {'depth': True, 'normal': True, 'silhouette': True}
    for i in range(num_iters):
        optimizer.zero_grad()
        extrinsics = camera_tensor if optimizer_type == 'shape' else get_camera_from_tensor(camera_tensor)
        loss = 0
        for renderer, rw in zip(sdfrenderer_list, weights):
            pack, _ = compute_all_loss(renderer, shape_code, extrinsics, gt_pack, threshold=renderer.get_threshold(),
                                       profile=profile, ray_marching_type=ray_marching_type, grad_settings=dict(grad_settings))
            loss = loss + rw * (weight_dict['w_depth'] * pack['depth'] + weight_dict['w_normal'] * pack['normal'] +
                                weight_dict['w_mask_gt'] * pack['mask_gt'] + weight_dict['w_mask_out'] * pack['mask_out'] +
                                weight_dict['w_l2reg'] * pack['l2reg'])
            if on_iteration is not None and renderer is sdfrenderer_list[0]:
                on_iteration(i, pack, loss)
        loss.backward()
        optimizer.step()
    return (shape_code if optimizer_type == 'shape' else camera_tensor), optimizer
