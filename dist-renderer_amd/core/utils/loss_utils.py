"""Image-space losses consumed around SDFRenderer.render (reference: core/utils/loss_utils.py:9-172).

Tiny element-wise PyTorch-ROCm code: it defines the upstream gradients (g_depth, g_normal, g_min_sdf) that the fused
backward kernel receives; fusing it with the renderer epilogue is SURVEY.md's "next" row f3. Written from the
reference's documented behaviour; masks may be bool or uint8.
"""
import torch
import torch.nn.functional as F


def grid_sample_on_img(img, xy):
    """Bilinear sampling of img (B,C,H,W) at pixel coordinates xy (B,2,Hout,Wout) -> (B,C,Hout,Wout).
    align_corners=True is the torch-1.1 behaviour the (W-1) normalisation assumes (loss_utils.py:9-25)."""
    _, _, H, W = img.shape
    gx = 2.0 * xy[:, 0] / max(W - 1, 1) - 1.0
    gy = 2.0 * xy[:, 1] / max(H - 1, 1) - 1.0
    return F.grid_sample(img, torch.stack([gx, gy], -1), mode='bilinear', padding_mode='zeros', align_corners=True)


def downsize_img_tensor(img, factor):
    """Average-pools an (H,W) / (H,W,3) tensor by an integer factor (loss_utils.py:27-57); uint8/bool stay masks."""
    h, w = img.shape[0], img.shape[1]
    if abs(h / factor - round(h / factor)) > 1e-12 or abs(w / factor - round(w / factor)) > 1e-12:
        raise ValueError('The image size {0} should be divisible by the factor {1}.'.format((h, w), factor))
    out_hw = (int(round(h / factor)), int(round(w / factor)))
    is_mask = img.dtype in (torch.uint8, torch.bool)
    x = img.float()
    x = x.permute(2, 0, 1) if x.dim() == 3 else x[None]
    with torch.no_grad():
        y = F.adaptive_avg_pool2d(x, out_hw)
    if is_mask:
        y = y.to(torch.uint8)
    return y[0] if y.shape[0] == 1 else y.permute(1, 2, 0)


def _b(mask):
    return mask if mask.dtype == torch.bool else mask != 0


def _mean_or_zero(values, like):
    return values.mean() if values.numel() else torch.zeros_like(like, dtype=torch.float32).mean()


def _show(visualizer, key, tensor, mask=None):
    """Hands an image (or the values of the pixels selected by `mask`) to the reference's Visualizer.add_data
    (core/visualize/visualizer.py:117-129), which wants numpy arrays."""
    if visualizer is not None:
        visualizer.add_data(key, tensor.detach().cpu().numpy(), None if mask is None else mask.detach().cpu().numpy())


def compute_loss_mask(min_sdf_sample, valid_mask, valid_mask_gt, threshold=5e-5, visualizer=None,
                      name=('mask_output', 'mask_gt', 'loss_mask_gt', 'loss_mask_out'), handle_first_query_corner_case=False):
    """Silhouette hinge losses on the min-|sdf| sample (loss_utils.py:59-103): pixels of the GT mask the render misses
    are pulled below the threshold, pixels the render adds are pushed above it. Returns (loss_gt, loss_out, visualizer)."""
    m, g = _b(valid_mask), _b(valid_mask_gt)
    _show(visualizer, name[0], valid_mask)
    _show(visualizer, name[1], valid_mask_gt)
    if handle_first_query_corner_case:
        neg_first = (min_sdf_sample < threshold) & (~m)
        min_sdf_sample = torch.where(neg_first, -min_sdf_sample + 2.0 * threshold + 0.1, min_sdf_sample)
    miss, extra = g & ~m, m & ~g
    hinge_gt = torch.clamp(min_sdf_sample[miss] - threshold, min=0.0)
    hinge_out = torch.clamp(threshold - min_sdf_sample[extra], min=0.0)
    if visualizer is not None:
        if hinge_gt.numel():
            _show(visualizer, name[2], hinge_gt, miss)
        else:
            _show(visualizer, name[2], torch.zeros_like(miss, dtype=torch.float32))
        if hinge_out.numel():      # (the reference plots the MEAN on the out-only pixels, loss_utils.py:94-96)
            _show(visualizer, name[3], hinge_out.mean(), extra)
        else:
            _show(visualizer, name[3], torch.zeros_like(extra, dtype=torch.float32))
    return _mean_or_zero(hinge_gt, miss), _mean_or_zero(hinge_out, extra), visualizer


def compute_loss_depth(depth_output, valid_mask, depth_gt, valid_mask_gt, visualizer=None):
    """Mean |depth - depth_gt| over pixels valid in both masks with a usable GT depth (loss_utils.py:105-133)."""
    _show(visualizer, 'depth_output', depth_output)
    _show(visualizer, 'depth_gt', depth_gt)
    both = _b(valid_mask) & _b(valid_mask_gt) & (depth_gt > 0) & (depth_gt < 1e5)
    diff = depth_output[both] - depth_gt[both]
    if visualizer is not None:
        if diff.numel():
            _show(visualizer, 'loss_depth', diff, both)
        else:
            _show(visualizer, 'loss_depth', torch.zeros_like(both, dtype=torch.float32))
    return _mean_or_zero(diff.abs(), both), visualizer


def normalize_vectors(x, dim=0):
    """Rows of an (n,3) tensor scaled to unit length, eps 1e-12 added to the norm (loss_utils.py:134-138; `dim` is the axis
    the norm runs over, 1 for (n,3) rows as in every caller)."""
    return x / (torch.norm(x, p=2, dim=dim)[:, None] + 1e-12)


def compute_loss_normal(normal_output, valid_mask, normal_gt, valid_mask_gt, visualizer=None):
    """Negative cosine between rendered and GT normals over the common mask (loss_utils.py:140-172)."""
    _show(visualizer, 'normal_output', normal_output)
    _show(visualizer, 'normal_gt', normal_gt)
    both = _b(valid_mask) & _b(valid_mask_gt) & (torch.norm(normal_output, p=2, dim=2) != 0)
    a, b = normal_output[both], normal_gt[both]
    if a.numel() == 0:
        _show(visualizer, 'loss_normal', torch.zeros_like(both, dtype=torch.float32))
        return torch.zeros_like(both, dtype=torch.float32).mean(), visualizer
    cos = (normalize_vectors(a, dim=1) * normalize_vectors(b, dim=1)).sum(1)
    _show(visualizer, 'loss_normal', -cos, both)
    return (-cos).mean(), visualizer


def compute_loss_color(color_output, valid_mask, color_gt, valid_mask_gt, visualizer=None,
                       name=('color_output', 'color_gt', 'loss_color'), use_ssim=False):
    """Mean L1 colour difference over the pixels valid in both masks (loss_utils.py:174-205; the SSIM variant needs the
    reference's pytorch_ssim package and is taken from there when a reference checkout is importable)."""
    _show(visualizer, name[0], color_output)
    _show(visualizer, name[1], color_gt)
    both = _b(valid_mask) & _b(valid_mask_gt)
    diff = color_output[both] - color_gt[both]
    if visualizer is not None:
        if diff.numel():
            _show(visualizer, name[2], diff.abs().mean(1), both)
        else:
            _show(visualizer, name[2], torch.zeros_like(both, dtype=torch.float32))
    if not diff.numel():
        raise ValueError('compute_loss_color: the rendered and ground-truth masks do not overlap')   # (the reference fails here too)
    loss = diff.abs().mean()
    if use_ssim:
        # (loss_utils.py:202-208: SSIM of the two images with everything outside the common mask zeroed; returned as a third value)
        from core.utils import pytorch_ssim        # the reference's own package (extended __path__); absent -> ImportError
        a = torch.where(both[..., None], color_gt, torch.zeros_like(color_gt)).permute(2, 0, 1)[None]
        b = torch.where(both[..., None], color_output, torch.zeros_like(color_output)).permute(2, 0, 1)[None]
        return loss, pytorch_ssim.loss_ssim(a, b)[0], visualizer
    return loss, visualizer
