"""compute_all_loss: one render + the single-view loss pack (reference: core/inv_optimizer/loss_single.py:7-65)."""
import torch

from core.utils.loss_utils import compute_loss_depth, compute_loss_mask, compute_loss_normal, downsize_img_tensor


def compute_all_loss(sdf_renderer, latent_tensor, extrinsic, gt_pack, threshold=5e-5, profile=False, visualizer=None,
                     ray_marching_type='pyramid_recursive', grad_settings=None):
    want = {'depth': True, 'normal': True, 'silhouette': True} if grad_settings is None else dict(grad_settings)
    for key in want:
        want[key] = bool(want[key]) and (gt_pack.get(key) is not None)
    depth, normal, mask, min_sdf = sdf_renderer.render(latent_tensor, extrinsic[:, :3], extrinsic[:, 3], profile=profile,
                                                       sample_index_type='min_abs', ray_marching_type=ray_marching_type,
                                                       no_grad_depth=not want['depth'], no_grad_normal=not want['normal'])
    ratio = next(iter(gt_pack.values())).shape[0] / depth.shape[0]
    gt = {k: downsize_img_tensor(v, ratio) for k, v in gt_pack.items() if v is not None}
    pack = {'mask_gt': 0.0, 'mask_out': 0.0, 'depth': 0.0, 'normal': 0.0}
    if 'silhouette' in gt:
        pack['mask_gt'], pack['mask_out'], visualizer = compute_loss_mask(min_sdf, mask, gt['silhouette'], threshold=threshold,
                                                                          visualizer=visualizer)
        if not want['silhouette']:
            pack['mask_gt'], pack['mask_out'] = pack['mask_gt'].detach(), pack['mask_out'].detach()
    if 'depth' in gt:
        pack['depth'], visualizer = compute_loss_depth(depth, mask, gt['depth'], gt['silhouette'], visualizer=visualizer)
        if not want['depth']:
            pack['depth'] = pack['depth'].detach()
    if 'normal' in gt:
        pack['normal'], visualizer = compute_loss_normal(normal, mask, gt['normal'], gt['silhouette'], visualizer=visualizer)
        if not want['normal']:
            pack['normal'] = pack['normal'].detach()
    pack['l2reg'] = torch.mean(latent_tensor.pow(2))
    return pack, visualizer
