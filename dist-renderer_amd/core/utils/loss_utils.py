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


def compute_loss_mask(min_sdf_sample, valid_mask, valid_mask_gt, threshold=5e-5, visualizer=None, name=None,
                      handle_first_query_corner_case=False):
    """Silhouette hinge losses on the min-|sdf| sample (loss_utils.py:59-103): pixels of the GT mask the render misses
    are pulled below the threshold, pixels the render adds are pushed above it."""
    m, g = _b(valid_mask), _b(valid_mask_gt)
    if handle_first_query_corner_case:
        neg_first = (min_sdf_sample < threshold) & (~m)
        min_sdf_sample = torch.where(neg_first, -min_sdf_sample + 2.0 * threshold + 0.1, min_sdf_sample)
    miss, extra = g & ~m, m & ~g
    q_miss, q_extra = min_sdf_sample[miss], min_sdf_sample[extra]
    loss_gt = _mean_or_zero(torch.clamp(q_miss - threshold, min=0.0), miss)
    loss_out = _mean_or_zero(torch.clamp(threshold - q_extra, min=0.0), extra)
    return loss_gt, loss_out, visualizer


def compute_loss_depth(depth_output, valid_mask, depth_gt, valid_mask_gt, visualizer=None):
    """Mean |depth - depth_gt| over pixels valid in both masks with a usable GT depth (loss_utils.py:105-133)."""
    both = _b(valid_mask) & _b(valid_mask_gt) & (depth_gt > 0) & (depth_gt < 1e5)
    return _mean_or_zero((depth_output[both] - depth_gt[both]).abs(), both), visualizer


def compute_loss_normal(normal_output, valid_mask, normal_gt, valid_mask_gt, visualizer=None):
    """Negative cosine between rendered and GT normals over the common mask (loss_utils.py:140-172)."""
    both = _b(valid_mask) & _b(valid_mask_gt) & (torch.norm(normal_output, p=2, dim=2) != 0)
    a, b = normal_output[both], normal_gt[both]
    if a.numel() == 0:
        return torch.zeros_like(both, dtype=torch.float32).mean(), visualizer
    a = a / (torch.norm(a, p=2, dim=1, keepdim=True) + 1e-12)
    b = b / (torch.norm(b, p=2, dim=1, keepdim=True) + 1e-12)
    return (-(a * b).sum(1)).mean(), visualizer
