"""Camera / normal helpers used around the hot path (reference: core/utils/render_utils.py)."""
import numpy as np
import torch


def depth2normal(depth, f_pix_x, f_pix_y=None):
    """Finite-difference normal map of a depth image (h,w) -> (h,w,3); reference: render_utils.py:9-43.
    Standalone PyTorch version for callers that hold their own depth map; SDFRenderer.render(use_depth2normal=True)
    uses the fused HIP kernel. Like the reference it overwrites background depth (>1e5 or ==0) with 0 IN PLACE."""
    if f_pix_y is None:
        f_pix_y = f_pix_x
    bg = (depth > 1e5) | (depth == 0)
    depth[bg] = 0.0
    d = depth
    dzdx = torch.zeros_like(d)
    dzdy = torch.zeros_like(d)
    dzdx[:, 1:-1] = (d[:, 2:] - d[:, :-2]) * f_pix_x / 2.0
    dzdy[1:-1, :] = (d[2:, :] - d[:-2, :]) * f_pix_y / 2.0
    n = torch.stack([dzdx, dzdy, -torch.ones_like(d)], -1)
    n = n / (torch.norm(n, p=2, dim=2, keepdim=True) + 1e-12)
    return torch.where(bg[..., None], torch.zeros_like(n), n)


def quad2rotation(quad):
    """(bs,4) unit quaternions (r,i,j,k) -> (bs,3,3); reference: render_utils.py:45-62."""
    r, i, j, k = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    rows = [1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r),
            2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r),
            2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)]
    return torch.stack(rows, -1).reshape(-1, 3, 3)


def get_camera_from_tensor(inputs):
    """(7,) or (bs,7) [quaternion | T] -> (3,4) / (bs,3,4) [R|T]; reference: render_utils.py:64-74."""
    single = inputs.dim() == 1
    x = inputs[None] if single else inputs
    RT = torch.cat([quad2rotation(x[:, :4]), x[:, 4:, None]], 2)
    return RT[0] if single else RT


def downsize_camera_intrinsic(intrinsic, factor):
    """Intrinsic of the image downsized by `factor` (pixel-centre convention); reference: render_utils.py:92-115."""
    img_h, img_w = int(2 * intrinsic[1, 2]), int(2 * intrinsic[0, 2])
    if img_h % factor or img_w % factor:
        raise ValueError('The image size {0} should be divisible by the factor {1}.'.format((img_h, img_w), factor))
    K = np.array(intrinsic, dtype=np.float64).copy()
    K[0, 0], K[1, 1] = K[0, 0] / factor, K[1, 1] / factor
    K[0, 2], K[1, 2] = K[0, 2] / factor, K[1, 2] / factor
    return K
