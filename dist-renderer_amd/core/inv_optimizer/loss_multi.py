"""Multi-view photometric loss of one view pair (reference: core/inv_optimizer/loss_multi.py:6-49): the optional sim(3)
alignment is folded into both extrinsics, `SDFRenderer_warp.render_warp` produces the colour loss (two fused renders +
the fused warp kernel), and the latent L2 regulariser is added. Same arguments and return values as the reference;
`loss_pack` values stay device tensors until they are read (the reference copies them to the host per pair)."""
import torch


class _LazyScalar(object):
    """Holds a detached device scalar; converts to numpy / float only when somebody looks at it (no host sync in the loop)."""
    __slots__ = ('t',)

    def __init__(self, t):
        self.t = t.detach()

    def __float__(self):
        return float(self.t)

    def __array__(self, dtype=None, copy=None):
        a = self.t.cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def __format__(self, spec):
        return format(float(self.t), spec)

    def __repr__(self):
        return repr(float(self.t))


def _as_device(x, device):
    t = torch.from_numpy(x) if not torch.is_tensor(x) else x
    return t.to(device=device, dtype=torch.float32)


def apply_sim3(R, T, sim3, sim3_scale):
    """World -> camera extrinsic composed with the object's similarity transform, rescaled (loss_multi.py:14-22)."""
    T = torch.matmul(R, sim3[:, 3]) + T
    R = torch.matmul(R, sim3[:3, :3])
    return R / sim3_scale, T / sim3_scale


def pair_cameras(cameras, idx1, idx2, device, sim3=None, sim3_scale=None):
    """Extrinsics of a view pair with the optional sim(3) folded in (loss_multi.py:10-22)."""
    cams = []
    for idx in (idx1, idx2):
        ext = cameras[idx].extrinsic
        R, T = _as_device(ext[:, :3], device), _as_device(ext[:, 3], device)
        if sim3 is not None:
            R, T = apply_sim3(R, T, sim3, sim3_scale)
        cams.append((R, T))
    return cams


def compute_loss_color_warp_batch(sdf_renderer, shape_code, images, cameras, pairs, weight_list, sim3=None, sim3_scale=None):
    """compute_loss_color_warp for all view pairs of a round at once: one batched render of the 2n views
    (SDFRenderer_warp.render_warp_batch). Returns [(loss, loss_pack), ...] in pair order, each equal to the per-pair call."""
    dev = shape_code.device
    args = []
    for (idx1, idx2) in pairs:
        (R1, T1), (R2, T2) = pair_cameras(cameras, idx1, idx2, dev, sim3, sim3_scale)
        args.append((R1, T1, R2, T2, images[idx1], images[idx2]))
    outs = sdf_renderer.render_warp_batch(shape_code, args, want_vis=False)     # only out[0] (the loss) is consumed here
    loss_l2reg = torch.mean(shape_code.pow(2))
    res = []
    for out in outs:
        loss = weight_list['color'] * out[0] + weight_list['l2reg'] * loss_l2reg
        res.append((loss, {'color': _LazyScalar(out[0]), 'l2reg': _LazyScalar(loss_l2reg)}))
    return res


def compute_loss_color_warp(sdf_renderer, shape_code, images, cameras, idx1, idx2, weight_list, sim3=None, sim3_scale=None,
                            visualizer=None):
    (R1, T1), (R2, T2) = pair_cameras(cameras, idx1, idx2, shape_code.device, sim3, sim3_scale)
    view1, view2 = images[idx1], images[idx2]
    out = sdf_renderer.render_warp(shape_code, R1, T1, R2, T2, view1, view2, no_grad_normal=True)
    loss_color, color_valid_1, color_valid_2 = out[0], out[1], out[2]
    if visualizer is not None:
        visualizer.reset_data()
        for name, img in (('color_gt-1', view1), ('color_gt-2', view2), ('color_valid-1', color_valid_1),
                          ('color_valid-2', color_valid_2), ('color_valid_loss', torch.abs(color_valid_1 - color_valid_2))):
            visualizer.add_data(name, img.detach().cpu().numpy())
    loss_l2reg = torch.mean(shape_code.pow(2))
    loss = weight_list['color'] * loss_color + weight_list['l2reg'] * loss_l2reg
    return loss, {'color': _LazyScalar(loss_color), 'l2reg': _LazyScalar(loss_l2reg)}
