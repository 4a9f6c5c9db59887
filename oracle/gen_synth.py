"""TEST INFRASTRUCTURE -- synthetic inputs shared by the golden generator and the tests (analytic sphere depth maps,
procedural images). No reference code involved."""
import numpy as np


def sphere_view(K, R, T, H, W, radius, centre):
    """render_depth-style outputs of an analytic sphere: Zdepth along the normalised camera ray (1e11 off the sphere)
    and the hit mask, flattened row-major."""
    Kinv = np.linalg.inv(np.asarray(K, np.float64))
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    homo = Kinv @ np.stack([xx.reshape(-1), yy.reshape(-1), np.ones(H * W)], 0)
    rays = np.asarray(R, np.float64).T @ homo
    rays /= np.linalg.norm(rays, axis=0, keepdims=True)
    c = -np.asarray(R, np.float64).T @ np.asarray(T, np.float64) - np.asarray(centre, np.float64)
    b = (rays * c[:, None]).sum(0)
    disc = b * b - (c @ c - radius * radius)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0))
    return np.where(hit, t, 1e11).astype(np.float32), hit


def procedural_images(H, W):
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    img1 = np.stack([0.5 + 0.5 * np.sin(xx / 3.0), 0.5 + 0.5 * np.cos(yy / 4.0), ((xx // 6 + yy // 6) % 2).astype(np.float64)], -1)
    img2 = np.stack([0.5 + 0.5 * np.sin(xx / 3.0 + 0.7), 0.5 + 0.5 * np.cos(yy / 4.0 - 0.3), ((xx // 5 + yy // 7) % 2).astype(np.float64)], -1)
    return img1.astype(np.float32), img2.astype(np.float32)
