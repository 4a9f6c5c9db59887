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


def get_tensor_from_camera(RT):
    """(3,4) [R|T] (numpy or tensor) -> (7,) [unit quaternion (r,i,j,k) | T], the inverse of get_camera_from_tensor
    (reference: render_utils.py:75-90, which goes through Blender's mathutils; here the standard branch-on-largest-diagonal
    conversion, positive scalar part). A tensor input returns a tensor on the same device."""
    dev = RT.device if torch.is_tensor(RT) else None
    M = RT.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(RT) else np.asarray(RT, dtype=np.float64)
    R, T = M[:, :3], M[:, 3]
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = 2.0 * np.sqrt(tr + 1.0)
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = 2.0 * np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 0.0))
        q = [0.0, 0.0, 0.0, 0.0]
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.asarray(q)
    if q[0] < 0:
        q = -q
    out = torch.from_numpy(np.concatenate([q / np.linalg.norm(q), T])).float()
    return out.to(dev) if dev is not None else out


def downsize_camera_intrinsic(intrinsic, factor):
    """Intrinsic of the image downsized by `factor`: the first two ROWS (focal lengths, skew, principal point) are divided
    by it (reference: render_utils.py:92-107)."""
    img_h, img_w = int(2 * intrinsic[1, 2]), int(2 * intrinsic[0, 2])
    if (img_h / factor - round(img_h / factor)) > 1e-12 or (img_w / factor - round(img_w / factor)) > 1e-12:
        raise ValueError('The image size {0} should be divisible by the factor {1}.'.format((img_h, img_w), factor))
    K = np.array(intrinsic, dtype=np.float64).copy()
    K[0, :] = K[0, :] / factor
    K[1, :] = K[1, :] / factor
    return K


def sample_points_from_mesh(mesh, N=30000):
    """(N,3) surface samples of a trimesh mesh (render_utils.py:109-115)."""
    import trimesh
    return trimesh.sample.sample_surface(mesh, N)[0]


def transform_point_cloud(points):
    """Mesh-file axes -> point-cloud axes: (x, y, z) -> (x, -z, y) (render_utils.py:117-124)."""
    out = np.array(points, copy=True)
    out[:, 1] = -points[:, 2]
    out[:, 2] = points[:, 1]
    return out


def read_pickle(fname):
    import pickle
    with open(fname, 'rb') as f:
        return pickle.load(f, encoding='latin1')


def save_pkl(data, fname):
    import pickle
    with open(fname, 'wb') as f:
        pickle.dump(data, f)


def save_render_output(render_output, fname):
    """Pickles (depth, normal, valid_mask) of a `SDFRenderer.render` result as numpy arrays (render_utils.py:131-138)."""
    depth, normal, mask, _ = render_output
    save_pkl({'depth': depth.detach().cpu().numpy(), 'normal': normal.detach().cpu().numpy(),
              'valid_mask': mask.detach().cpu().numpy()}, fname)
