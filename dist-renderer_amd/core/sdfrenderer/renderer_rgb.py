"""SDFRenderer_color: textured rendering with a second (colour) decoder + optional point-light shading (reference:
core/sdfrenderer/renderer_rgb.py:12-125). Same constructor / `render` signature and return tuples.

Depth, mask, min-sdf sample and normals come from one fused `distr_render_forward` (the 'recursive' marcher that
`render_depth` defaults to, autograd-style normals); the colour decoder runs on the fused decoder tile too
(`distr_color_eval`: latent = [shape code | colour code] folded into per-call constants, lin8 with three rows). The
colour image is differentiable like the reference's (golden G26): `render_color` / `render` without `no_grad` keep the colour
code, the shape code and -- through the surface points -- the camera on the tape (ColorDecodeFunction -> distr_color_backward)."""
import torch

from distr import functions

from .renderer import SDFRenderer


class SDFRenderer_color(SDFRenderer):
    # reference: renderer_rgb.py:13
    def __init__(self, decoder, decoder_color, intrinsic, img_hw=None, march_step=50, buffer_size=5, ray_marching_ratio=1.5,
                 max_sample_dist=0.2, threshold=5e-5, use_gpu=True, is_eval=True):
        super(SDFRenderer_color, self).__init__(decoder, intrinsic, img_hw=img_hw, march_step=march_step, buffer_size=buffer_size,
                                                ray_marching_ratio=ray_marching_ratio, max_sample_dist=max_sample_dist,
                                                threshold=threshold, use_gpu=use_gpu, is_eval=is_eval)
        self.decoder_color = decoder_color.eval() if is_eval else decoder_color
        functions.get_color_engine(self.decoder_color, self.device)

    @property
    def _color_engine(self):
        return functions.get_color_engine(self.decoder_color, self.device)

    # reference: renderer_rgb.py:20
    def render_color(self, latent_color, latent, cam_pos, cam_rays, Zdepth, valid_mask, no_grad=False):
        h, w = self.img_hw
        color = torch.zeros(h * w, 3, device=Zdepth.device)
        idx = torch.nonzero(valid_mask.reshape(-1)).reshape(-1)
        if idx.numel() == 0:
            return color.reshape(h, w, 3)
        points = self.generate_point_samples(cam_pos, cam_rays[:, idx], Zdepth.reshape(-1)[idx], has_zdepth_grad=False)
        if no_grad or not torch.is_grad_enabled():
            with torch.no_grad():
                rgb = functions.color_eval(self._color_engine, latent_color, latent, points.t())
        else:
            # renderer_rgb.py:32-36: without no_grad the colours stay on the tape -- gradients to the colour code, the shape code and, through
            # the surface points (Zdepth detached: has_zdepth_grad=False), to the camera (distr_color_backward; golden G26)
            rgb = functions.color_eval_autograd(self._color_engine, latent_color, latent, points.t().contiguous())
        return color.index_copy(0, idx, rgb).reshape(h, w, 3)

    # reference: renderer_rgb.py:39
    def compute_shading_maps(self, R, T, lighting_locations, Zdepth, Znormal, valid_mask):
        """Lambertian n.l terms for M point lights -> (M, H*W); zero off the mask. (The reference's torch.bmm(R[None],
        directions), renderer_rgb.py:58, only runs for M = 1; any M works here.)"""
        idx = torch.nonzero(valid_mask.reshape(-1)).reshape(-1)
        cam_pos, cam_rays = self.get_camera_location(R, T), self.get_camera_rays(R)
        pts = self.generate_point_samples(cam_pos, cam_rays[:, idx], Zdepth.reshape(-1)[idx], inv_transform=False).t()   # (N,3)
        to_light = lighting_locations[:, None, :] - pts[None, :, :]                                                     # (M,N,3)
        to_light = to_light / torch.norm(to_light, p=2, dim=2, keepdim=True)
        z_dirs = torch.matmul(to_light, R.t())                                   # rows = R @ direction
        lambert = (z_dirs * Znormal.reshape(-1, 3)[idx][None]).sum(2)            # (M,N)
        out = torch.zeros(lighting_locations.shape[0], Zdepth.numel(), device=Zdepth.device, dtype=lambert.dtype)
        return out.index_copy(1, idx, lambert)

    # reference: renderer_rgb.py:73
    def render(self, latent_color, latent, R, T, clamp_dist=0.1, profile=False, no_grad=False, lighting_locations=None,
               lighting_energies=None):
        h, w = self.img_hw
        cfg = self._cfg(clamp_dist, 'recursive', True, want_normal=True, no_grad_depth=no_grad, no_grad_mask=no_grad,
                        no_grad_camera=no_grad)
        cfg.use_depth2normal = 0
        zdepth, mask, min_sdf, depth, normal = functions.render_call(self._engine, cfg, latent, R, T)
        if no_grad:
            # (the normals are detached inside render_normal, before `R @ normal` (renderer_rgb.py:93-94): their gradient w.r.t. R stays,
            # exactly as in SDFRenderer.render -- golden G18)
            depth, min_sdf = depth.detach(), min_sdf.detach()
        valid = mask.bool()
        color = self.render_color(latent_color, latent, self.get_camera_location(R, T), self.get_camera_rays(R), zdepth.detach(),
                                  valid, no_grad=no_grad)
        mask_img, min_sdf = mask.reshape(h, w), min_sdf.reshape(h, w)
        if lighting_locations is None:
            return depth, normal, color, mask_img, min_sdf
        if lighting_energies is None:
            lighting_energies = torch.ones_like(lighting_locations[:, 0])
        shading = self.compute_shading_maps(R, T, lighting_locations, zdepth.detach(), normal.reshape(-1, 3), valid)
        shading = (shading * lighting_energies[:, None]).sum(0).reshape(h, w)
        return depth, normal, color * shading[:, :, None], mask_img, min_sdf
