"""SDFRenderer_warp: multi-view photometric warp loss (reference: core/sdfrenderer/renderer_warp.py:13-144).

The two render_depth calls and render_normal are libdistr kernels; the warp itself (back-project view-1 depth,
project into view 2, bilinear sampling, depth-consistency test, L1 colour) is the handful of small PyTorch-ROCm ops
SURVEY.md lists as "next" row f2 and stays in PyTorch here. Gradients reach the latent through Zdepth of view 1
(g_zdepth input of distr_render_backward).
"""
import torch
import torch.nn.functional as F

from .renderer import SDFRenderer


def _sample_img(img, xy):
    """Bilinear sample img (1,C,H,W) at pixel coords xy (2,n) -> (C,n); align_corners=True is what torch 1.1 did
    (core/utils/loss_utils.py:9-25)."""
    _, _, H, W = img.shape
    gx = 2.0 * xy[0] / max(W - 1, 1) - 1.0
    gy = 2.0 * xy[1] / max(H - 1, 1) - 1.0
    grid = torch.stack([gx, gy], -1)[None, :, None, :]
    return F.grid_sample(img, grid, mode='bilinear', padding_mode='zeros', align_corners=True)[0, :, :, 0]


class SDFRenderer_warp(SDFRenderer):
    # reference: renderer_warp.py:14
    def __init__(self, decoder, intrinsic, img_hw=None, march_step=50, buffer_size=5, ray_marching_ratio=1.5,
                 max_sample_dist=0.2, threshold=5e-5, use_gpu=True, is_eval=True, transform_matrix=None):
        super(SDFRenderer_warp, self).__init__(decoder, intrinsic, img_hw=img_hw, transform_matrix=transform_matrix,
                                               march_step=march_step, buffer_size=buffer_size,
                                               ray_marching_ratio=ray_marching_ratio, max_sample_dist=max_sample_dist,
                                               threshold=threshold, use_gpu=use_gpu, is_eval=is_eval)
        self.counter = 0

    # reference: renderer_warp.py:103
    def render_warp(self, latent, R1, T1, R2, T2, img1, img2, clamp_dist=0.1, profile=False, no_grad_normal=False,
                    thres_depth=0.001):
        h, w = self.img_hw
        dev = self.calib_map.device
        Z1, m1, q1 = self.render_depth(latent, R1, T1, clamp_dist=clamp_dist, profile=profile)
        Z2, m2, q2 = self.render_depth(latent, R2, T2, clamp_dist=clamp_dist, profile=profile, no_grad_depth=True)
        img1, img2 = img1.to(dev), img2.to(dev)
        color1, color2 = torch.zeros_like(img1), torch.zeros_like(img1)
        if int(m1.sum()) == 0:
            loss_color = torch.zeros((), device=dev, requires_grad=True)
        else:
            # view-1 surface points (depth keeps its gradient) -> view-2 pixels            renderer_warp.py:18-36
            pts = self.generate_point_samples(self.get_camera_location(R1, T1), self.get_camera_rays(R1)[:, m1], Z1[m1],
                                              inv_transform=False, has_zdepth_grad=True)
            proj = torch.matmul(self.K, torch.matmul(R2, pts) + T2[:, None])
            xy = proj[:2] / proj[2]
            # depth consistency against view 2 (renderer_warp.py:63-72)
            d2 = (Z2 * self.calib_map).reshape(1, 1, h, w)
            keep = (proj[2] - _sample_img(d2, xy)[0]) ** 2 < thres_depth
            xy = xy[:, keep]
            c1 = img1.reshape(h * w, 3)[m1][keep]
            c2 = _sample_img(img2.permute(2, 0, 1)[None], xy).t()
            loss_color = torch.mean(torch.abs(c1 - c2))                                     # renderer_warp.py:85
            final = torch.zeros(h * w, dtype=torch.bool, device=dev)
            final[m1.nonzero().reshape(-1)[keep]] = True
            color1.reshape(h * w, 3)[final] = c1.detach()
            color2.reshape(h * w, 3)[final] = c2.detach()
        # visualisation outputs (renderer_warp.py:131-144)
        n1 = self.render_normal(latent, R1, T1, Z1, m1, no_grad=no_grad_normal, clamp_dist=clamp_dist)
        n1 = torch.matmul(R1.detach(), n1)
        n1 = torch.cat([-n1[:1], n1[1:]], 0).reshape(3, h, w).permute(1, 2, 0)
        depth1 = torch.where(m1, Z1.detach() * self.calib_map, torch.zeros_like(Z1)).reshape(h, w)
        return (loss_color, color1, color2, m1.reshape(h, w).to(torch.uint8), m2.reshape(h, w).to(torch.uint8),
                q1.reshape(h, w), q2.reshape(h, w), n1, depth1)
