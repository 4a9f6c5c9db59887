"""TEST INFRASTRUCTURE -- CPU restatement (PyTorch float32 on CPU, gradients by autograd) of the two image-space
consumers the fused loss kernels replace. Only tests/ may import this; the product path
(dist-renderer_amd/csrc/distr_losses.hpp) never does. Pinned against goldens produced by the reference's own
functions (tests/golden/g7_single_losses.npz, g8_warp_loss.npz; generator oracle/gen_golden_losses.py).

  single_view_losses   core/utils/loss_utils.py:59-172   (compute_loss_mask / _depth / _normal)
  warp_loss            core/sdfrenderer/renderer_warp.py:18-101 (get_valid_points, valid_points_depth,
                       compute_loss_color) + grid_sample_on_img core/utils/loss_utils.py:9-25 (torch-1.1 semantics:
                       align_corners=True, zero padding)
"""
import numpy as np
import torch
import torch.nn.functional as F


def _mean_or_zero(v):
    return v.mean() if v.numel() else torch.zeros(())


def single_view_losses(depth, normal, mask, min_sdf, gt_depth, gt_normal, gt_mask, threshold):
    """-> [mask_gt, mask_out, depth, normal] (torch scalars with autograd history)."""
    m, g = mask.bool(), gt_mask.bool()
    miss, extra = g & ~m, m & ~g                                         # loss_utils.py:75, 89
    l_gt = _mean_or_zero(torch.clamp(min_sdf[miss] - threshold, min=0.0))
    l_out = _mean_or_zero(torch.clamp(threshold - min_sdf[extra], min=0.0))
    l_d = torch.zeros(())
    if gt_depth is not None:
        sel = m & g & (gt_depth > 0) & (gt_depth < 1e5)                  # loss_utils.py:118-121
        l_d = _mean_or_zero((depth[sel] - gt_depth[sel]).abs())
    l_n = torch.zeros(())
    if gt_normal is not None:
        sel = m & g & (torch.norm(normal, p=2, dim=2) != 0)              # loss_utils.py:155-160
        a, b = normal[sel], gt_normal[sel]
        if a.numel():
            a = a / (torch.norm(a, p=2, dim=1, keepdim=True) + 1e-12)
            b = b / (torch.norm(b, p=2, dim=1, keepdim=True) + 1e-12)
            l_n = (-(a * b).sum(1)).mean()
    return [l_gt, l_out, l_d, l_n]


def _sample(img, xy):
    """img (1,C,H,W), xy (2,n) pixel coordinates -> (C,n)."""
    _, _, H, W = img.shape
    gx = 2.0 * xy[0] / max(W - 1, 1) - 1.0
    gy = 2.0 * xy[1] / max(H - 1, 1) - 1.0
    grid = torch.stack([gx, gy], -1)[None, :, None, :]
    return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=True)[0, :, :, 0]


def warp_loss(K, H, W, z1, m1, z2, img1, img2, R1, T1, R2, T2, thres_depth):
    """-> (loss_color, keep (P) bool, color1 (H,W,3), color2 (H,W,3)). z1, R*, T* may require grad."""
    Kt = torch.from_numpy(np.asarray(K, np.float32))
    Kinv = torch.from_numpy(np.linalg.inv(np.asarray(K, np.float64)).astype(np.float32))
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing='ij')
    homo = Kinv @ torch.stack([xx.reshape(-1), yy.reshape(-1), torch.ones(H * W)], 0)
    calib = (homo / (torch.norm(homo, p=2, dim=0, keepdim=True) + 1e-12))[2]
    m1 = m1.bool().reshape(-1)
    c1 = torch.zeros(H * W, 3)
    c2 = torch.zeros(H * W, 3)
    keep_full = torch.zeros(H * W, dtype=torch.bool)
    if int(m1.sum()) == 0:
        return torch.zeros(()), keep_full, c1.reshape(H, W, 3), c2.reshape(H, W, 3)
    rays = R1.t() @ homo
    rays = rays / (torch.norm(rays, p=2, dim=0, keepdim=True) + 1e-12)
    cam = -(R1.t() @ T1)
    pts = rays[:, m1] * z1.reshape(-1)[m1][None, :] + cam[:, None]           # renderer_warp.py:23-28
    proj = Kt @ (R2 @ pts + T2[:, None])                                      # :31
    xy = proj[:2] / proj[2]
    d2 = (z2.reshape(-1) * calib).reshape(1, 1, H, W)                          # :65-68
    keep = (proj[2] - _sample(d2, xy)[0]) ** 2 < thres_depth                   # :70-71
    xyk = xy[:, keep]
    a = img1.reshape(H * W, 3)[m1][keep]
    b = _sample(img2.permute(2, 0, 1)[None], xyk).t()
    loss = torch.mean(torch.abs(a - b))                                        # :85
    idx = m1.nonzero().reshape(-1)[keep]
    keep_full[idx] = True
    c1[idx] = a.detach()
    c2[idx] = b.detach()
    return loss, keep_full, c1.reshape(H, W, 3), c2.reshape(H, W, 3)
