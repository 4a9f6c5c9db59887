"""SDFRenderer_warp: multi-view photometric warp loss (reference: core/sdfrenderer/renderer_warp.py:13-144).

The two render_depth calls, render_normal and the warp itself (back-project view-1 depth, project into view 2,
bilinear depth + colour sampling, depth-consistency test, L1 colour) are libdistr kernels. Gradients
reach the latent through Zdepth of view 1 (g_zdepth input of distr_render_backward) and the four camera tensors.
"""
import torch

from distr import binding, functions

from .renderer import SDFRenderer


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
        # (`profile` is accepted for signature compatibility and ignored: the reference prints host-side timers around its two
        # render_depth calls, renderer_warp.py:105-110; the batched launch sequence has no per-view boundary to time. self._last is
        # not updated on this path either -- use render_depth(profile=True) to time one view.)
        h, w = self.img_hw
        dev = self.calib_map.device
        # the two depth renders of the pair (renderer_warp.py:108-109: view 2 with no_grad_depth) as ONE batched launch sequence: every
        # march step covers the live rays of both views, their latency-bound tails overlap; each view is byte-identical to its own
        # render_depth call (tests/test_gpu_batch.py)
        Z, M, Q = self.render_depth_batch(latent, [R1, R2], [T1, T2], clamp_dist=clamp_dist, no_grad_depth=[False, True])
        Z1, m1, q1, Z2, m2, q2 = Z[0], M[0], Q[0], Z[1], M[1], Q[1]
        # warp + consistency test + L1 colour, forward and backward, fused (row f2): distr_warp_loss_forward/_backward.
        # Gradients reach the latent through Zdepth of view 1 and the cameras directly.
        wcfg = binding.make_warp_cfg((h, w), self.intrinsic, thres_depth)
        loss_color, keep, color1, color2 = functions.warp_loss(self._engine, wcfg, Z1, m1, Z2, img1.to(dev), img2.to(dev),
                                                               R1, T1, R2, T2)
        # visualisation outputs (renderer_warp.py:131-144)
        n1 = self.render_normal(latent, R1, T1, Z1, m1, no_grad=no_grad_normal, clamp_dist=clamp_dist)
        n1 = torch.matmul(R1, n1)
        n1 = torch.cat([-n1[:1], n1[1:]], 0).reshape(3, h, w).permute(1, 2, 0)
        depth1 = torch.where(m1, Z1.detach() * self.calib_map, torch.zeros_like(Z1)).reshape(h, w)
        return (loss_color, color1, color2, m1.reshape(h, w).to(torch.uint8), m2.reshape(h, w).to(torch.uint8),
                q1.reshape(h, w), q2.reshape(h, w), n1, depth1)

    def render_warp_batch(self, latent, pairs, clamp_dist=0.1, thres_depth=0.001, want_vis=True):
        """render_warp for several view pairs that share the shape code -- the body of the multi-view round
        (core/inv_optimizer/optimize_multi.py:62-81 calls render_warp once per pair; renderer_warp.py:108-109 renders both views of a
        pair with the same latent) -- with ALL 2n depth renders in one batched launch sequence and all n visualisation normals in
        another. `pairs` = [(R1, T1, R2, T2, img1, img2), ...]. Returns one 9-tuple per pair, each identical (bit for bit, values
        and gradients) to render_warp(latent, *pair). want_vis=False (the optimisation loop, which only consumes the
        loss): the visualisation normal / depth images (entries 7, 8) are None and their normal pass (2.3 ms of a 93 ms round, plus its
        workspace) is skipped."""
        h, w = self.img_hw
        dev = self.calib_map.device
        n = len(pairs)
        Rs = [t for p in pairs for t in (p[0], p[2])]
        Ts = [t for p in pairs for t in (p[1], p[3])]
        ngd = [False, True] * n                      # view 2 of every pair: no_grad_depth (renderer_warp.py:109)
        Z, M, Q = self.render_depth_batch(latent, Rs, Ts, clamp_dist=clamp_dist, no_grad_depth=ngd)
        wcfg = binding.make_warp_cfg((h, w), self.intrinsic, thres_depth)
        N1 = self.render_normal_batch(latent, Rs[0::2], Ts[0::2], Z[0::2].detach(), M[0::2], clamp_dist=clamp_dist) if want_vis else None
        outs = []
        for i, (R1, T1, R2, T2, img1, img2) in enumerate(pairs):
            Z1, m1, q1, Z2, m2, q2 = Z[2 * i], M[2 * i], Q[2 * i], Z[2 * i + 1], M[2 * i + 1], Q[2 * i + 1]
            loss_color, keep, color1, color2 = functions.warp_loss(self._engine, wcfg, Z1, m1, Z2, img1.to(dev), img2.to(dev),
                                                                   R1, T1, R2, T2)
            n1 = depth1 = None
            if want_vis:
                n1 = torch.matmul(R1, N1[i])
                n1 = torch.cat([-n1[:1], n1[1:]], 0).reshape(3, h, w).permute(1, 2, 0)
                depth1 = torch.where(m1, Z1.detach() * self.calib_map, torch.zeros_like(Z1)).reshape(h, w)
            outs.append((loss_color, color1, color2, m1.reshape(h, w).to(torch.uint8), m2.reshape(h, w).to(torch.uint8),
                         q1.reshape(h, w), q2.reshape(h, w), n1, depth1))
        return outs
