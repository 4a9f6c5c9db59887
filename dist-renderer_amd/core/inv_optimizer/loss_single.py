"""Single-view loss pack: one `SDFRenderer.render` call followed by the silhouette / depth / normal / L2 terms.

Drop-in for `compute_all_loss` of the reference (core/inv_optimizer/loss_single.py:7-65): same arguments, same
returned dict keys ('mask_gt', 'mask_out', 'depth', 'normal', 'l2reg') and the same (pack, visualizer) tuple."""
import torch

from core.utils import loss_utils as LU

_TERMS = ('depth', 'normal', 'silhouette')


def _enabled(grad_settings, gt_pack):
    """A term back-propagates only if it is switched on AND its ground truth is present."""
    cfg = dict.fromkeys(_TERMS, True) if grad_settings is None else dict(grad_settings)
    return {t: bool(cfg.get(t, False)) and gt_pack.get(t) is not None for t in _TERMS}


def _maybe_detach(value, keep_grad):
    return value if keep_grad else value.detach()


def compute_all_loss(sdf_renderer, latent_tensor, extrinsic, gt_pack, threshold=5e-5, profile=False, visualizer=None,
                     ray_marching_type='pyramid_recursive', grad_settings=None):
    on = _enabled(grad_settings, gt_pack)
    R, T = extrinsic[:, :3], extrinsic[:, 3]
    rendered = sdf_renderer.render(latent_tensor, R, T, profile=profile, ray_marching_type=ray_marching_type,
                                   no_grad_depth=not on['depth'], no_grad_normal=not on['normal'])
    depth, normal, mask, min_sdf = rendered
    # ground truth may come at a multiple of the render resolution (multi-scale renderer lists)
    scale = next(iter(gt_pack.values())).shape[0] / depth.shape[0]
    gt = {name: LU.downsize_img_tensor(img, scale) for name, img in gt_pack.items() if img is not None}
    sil = gt.get('silhouette')

    pack = dict(mask_gt=0.0, mask_out=0.0, depth=0.0, normal=0.0)
    if visualizer is not None:
        visualizer.reset_data()                     # loss_single.py:27-28: one set of images per call
    engine = getattr(sdf_renderer, '_engine', None)
    if sil is not None and engine is not None and visualizer is None and depth.is_cuda:
        # fused path (row f3): all four terms in two element-wise kernels, no host synchronisation
        from distr import functions
        terms = functions.single_view_losses(engine, depth, normal, mask, min_sdf, gt.get('depth'), gt.get('normal'), sil, threshold)
        pack['mask_gt'], pack['mask_out'] = _maybe_detach(terms[0], on['silhouette']), _maybe_detach(terms[1], on['silhouette'])
        if 'depth' in gt:
            pack['depth'] = _maybe_detach(terms[2], on['depth'])
        if 'normal' in gt:
            pack['normal'] = _maybe_detach(terms[3], on['normal'])
    else:
        if sil is not None:
            miss, extra, visualizer = LU.compute_loss_mask(min_sdf, mask, sil, threshold=threshold, visualizer=visualizer)
            pack['mask_gt'], pack['mask_out'] = _maybe_detach(miss, on['silhouette']), _maybe_detach(extra, on['silhouette'])
        if 'depth' in gt:
            value, visualizer = LU.compute_loss_depth(depth, mask, gt['depth'], sil, visualizer=visualizer)
            pack['depth'] = _maybe_detach(value, on['depth'])
        if 'normal' in gt:
            value, visualizer = LU.compute_loss_normal(normal, mask, gt['normal'], sil, visualizer=visualizer)
            pack['normal'] = _maybe_detach(value, on['normal'])
    pack['l2reg'] = latent_tensor.pow(2).mean()
    if visualizer is not None:
        visualizer.add_loss_from_pack(pack)         # loss_single.py:62-63: the loss curves
    return pack, visualizer
