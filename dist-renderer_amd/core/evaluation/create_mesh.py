"""Bulk SDF grid evaluation for meshing (SURVEY.md "next" row f1; reference: core/evaluation/create_mesh.py:16-142).

The reference evaluates the N^3 grid in 32^3-point batches with a host<->device round trip per batch
(`.cuda()` / `.cpu()` inside the loop, create_mesh.py:46-52). Here the grid lives on the GPU and one
`distr_mlp_eval` launch (fused MFMA decoder, 64 points per workgroup) evaluates all of it; the coarse-to-fine
variant only evaluates the full-resolution points within 1.5 coarse voxels of the surface, like
`create_mesh_speedup`. Mesh extraction (skimage marching cubes + plyfile) is CPU tooling outside the hot path:
`create_mesh*` call it when those packages are importable and raise otherwise.
"""
import torch

from core.utils.decoder_utils import decode_sdf


def get_samples(N, voxel_origin=(-1.0, -1.0, -1.0), voxel_size=None, transform=False, device='cuda'):
    """(N^3, 3) grid coordinates, x slowest / z fastest (create_mesh.py:16-33); transform = (x, z, -y) swap (:10-14)."""
    if voxel_size is None:
        voxel_size = 2.0 / (N - 1)
    idx = torch.arange(N ** 3, device=device)
    ijk = torch.stack([(idx // (N * N)) % N, (idx // N) % N, idx % N], 1).float()
    pts = ijk * voxel_size + torch.tensor(voxel_origin, dtype=torch.float32, device=device)
    if transform:
        pts = torch.stack([pts[:, 0], pts[:, 2], -pts[:, 1]], 1)
    return pts


def infer_samples(decoder, latent_vec, samples, max_batch=None, arith='f32'):
    """SDF (clamped to +-0.1 like decode_sdf's default) of (M,3|4) samples -> (M,). One kernel launch; `max_batch` is
    accepted for signature compatibility and ignored. arith='bf16x6': the split-bf16 decoder tile (values within ~1e-6 of the
    exact ones: marching cubes cannot tell, the evaluation is faster)."""
    with torch.no_grad():
        return decode_sdf(decoder, latent_vec, samples[:, :3].contiguous(), no_grad=True, arith=arith).reshape(-1)


def create_sdf_grid(decoder, latent_vec, N=256, transform=False, arith='f32'):
    """(N,N,N) SDF grid on [-1,1]^3 (the tensor create_mesh hands to marching cubes, create_mesh.py:56-68)."""
    dev = next(decoder.parameters()).device
    return infer_samples(decoder, latent_vec, get_samples(N, transform=transform, device=dev), arith=arith).reshape(N, N, N)


def create_sdf_grid_speedup(decoder, latent_vec, N=256, transform=False, relaxation=1.5, arith='f32'):
    """Coarse-to-fine grid (create_mesh.py:100-131): evaluate N/2 per axis, nearest-upsample, and evaluate the full
    grid only where |sdf_half| <= relaxation * coarse voxel size; elsewhere +-0.1. `arith`: as in create_sdf_grid."""
    assert N % 2 == 0
    dev = next(decoder.parameters()).device
    Nh = N // 2
    vs_half = 2.0 / (Nh - 1)
    half = infer_samples(decoder, latent_vec, get_samples(Nh, voxel_size=vs_half, transform=transform, device=dev), arith=arith).reshape(Nh, Nh, Nh)
    up = half.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(-1)
    band = up.abs() <= vs_half * relaxation
    grid = torch.where(up > 0, torch.full_like(up, 0.1), torch.full_like(up, -0.1))
    pts = get_samples(N, transform=transform, device=dev)
    if bool(band.any()):
        grid[band] = infer_samples(decoder, latent_vec, pts[band], arith=arith)
    return grid.reshape(N, N, N)


def _to_ply(sdf_grid, filename, N):
    try:
        import plyfile
        import numpy as np
        from skimage import measure
    except Exception as e:      # noqa: BLE001
        raise RuntimeError('mesh extraction needs scikit-image and plyfile (CPU tooling outside the rendering hot path); '
                           'use create_sdf_grid / create_sdf_grid_speedup for the SDF volume') from e
    vol = sdf_grid.detach().cpu().numpy()
    try:
        verts, faces, _, _ = measure.marching_cubes(vol, level=0.0, spacing=[2.0 / (N - 1)] * 3)
    except Exception:           # noqa: BLE001  (no zero crossing: invalid shape, create_mesh.py returns False)
        return False
    verts = verts - 1.0
    v = np.array([tuple(p) for p in verts], dtype=[('x', 'f4'), ('y', 'f4'), ('z', 'f4')])
    f = np.array([(list(t),) for t in faces], dtype=[('vertex_indices', 'i4', (3,))])
    plyfile.PlyData([plyfile.PlyElement.describe(v, 'vertex'), plyfile.PlyElement.describe(f, 'face')]).write(filename)
    return True


def create_mesh(decoder, latent_vec, filename, N=256, max_batch=32 ** 3, silent=False, transform=False):
    return _to_ply(create_sdf_grid(decoder, latent_vec, N, transform), filename + '.ply', N)


def create_mesh_speedup(decoder, latent_vec, filename, N=256, max_batch=32 ** 3, silent=False, transform=False):
    return _to_ply(create_sdf_grid_speedup(decoder, latent_vec, N, transform), filename + '.ply', N)
