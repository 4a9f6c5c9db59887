"""TEST INFRASTRUCTURE -- golden G21: the Python helper functions around the hot path, evaluated by the REFERENCE itself on CPU (build
container only; shims in oracle/ref_harness.py; no reference source is copied):

    python oracle/gen_golden_helpers.py        # writes tests/golden/g21_python_helpers.npz

core/utils/render_utils.py: depth2normal (:9-43, incl. the in-place zeroing of its input and its autograd gradient), quad2rotation
(:45-62), get_camera_from_tensor (:64-73), downsize_camera_intrinsic (:92-107); core/utils/loss_utils.py: grid_sample_on_img (:9-25),
downsize_img_tensor (:27-57: float images, 3-channel images, uint8 masks), compute_loss_color (:174-205); core/utils/train_utils.py:
get_lie_rotation_matrix / params_to_mtrx (:155-177) with gradients. The drop-in package restates each of them
(dist-renderer_amd/core/utils/): tests/test_host_logic.py::test_python_helpers_match_reference_golden compares on the CPU.
"""
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, '..', 'dist-renderer_amd'))
sys.path.insert(0, _HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(_HERE, '..', 'tests', 'golden')


def main():
    rh.install_shims()
    rh.install_device_shims()
    rh.reference_modules()
    import core.utils.render_utils as ru
    import core.utils.loss_utils as lu
    import core.utils.train_utils as tu
    for m in (ru, lu, tu):
        assert os.path.abspath(m.__file__).startswith(rh.REFERENCE_ROOT)
    rs = np.random.RandomState(21)
    out = {}
    # ---- depth2normal: random depth with both background conventions (1e11 and 0), fx != fy, gradient of a seeded loss
    h, w = 20, 24
    depth = (1.0 + 0.5 * rs.rand(h, w)).astype(np.float32)
    depth[rs.rand(h, w) < 0.2] = 1e11
    depth[rs.rand(h, w) < 0.1] = 0.0
    wn = rs.rand(h, w, 3).astype(np.float32)
    d = torch.from_numpy(depth.copy()).requires_grad_(True)
    dd = d * 1.0                                       # (a non-leaf, like the renderer's depth: the function writes into it)
    n = ru.depth2normal(dd, 31.0, 27.0)
    (n * torch.from_numpy(wn)).sum().backward()
    out.update(d2n_depth_in=depth, d2n_wn=wn, d2n_normal=n.detach().numpy(), d2n_depth_after=dd.detach().numpy(), d2n_grad=d.grad.numpy(), d2n_fx=31.0, d2n_fy=27.0)
    n1 = ru.depth2normal(torch.from_numpy(depth.copy()), 40.0)
    out['d2n_normal_single_f'] = n1.numpy()
    # ---- quaternions / camera tensors
    q = rs.standard_normal((6, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    out.update(quat=q, quat_rot=ru.quad2rotation(torch.from_numpy(q)).numpy())
    cam = np.concatenate([q, rs.standard_normal((6, 3)).astype(np.float32)], 1)
    out.update(cam=cam, cam_RT_batch=ru.get_camera_from_tensor(torch.from_numpy(cam)).numpy(), cam_RT_single=ru.get_camera_from_tensor(torch.from_numpy(cam[2])).numpy())
    c = torch.from_numpy(cam[1]).clone().requires_grad_(True)
    wRT = rs.rand(3, 4).astype(np.float32)
    (ru.get_camera_from_tensor(c) * torch.from_numpy(wRT)).sum().backward()
    out.update(cam_wRT=wRT, cam_grad=c.grad.numpy())
    # ---- intrinsics
    K = np.array([[300.0, 1.5, 112.0], [0.0, 310.0, 96.0], [0.0, 0.0, 1.0]])
    out.update(K=K, K_half=ru.downsize_camera_intrinsic(K, 2), K_quarter=ru.downsize_camera_intrinsic(K, 4))
    try:
        ru.downsize_camera_intrinsic(K, 5)
        out['K_fifth_raises'] = False
    except ValueError:
        out['K_fifth_raises'] = True
    # ---- image downsizing
    img = rs.rand(16, 24).astype(np.float32)
    img3 = rs.rand(16, 24, 3).astype(np.float32)
    msk = (rs.rand(16, 24) < 0.7).astype(np.uint8)
    for f in (2, 4):
        out['ds_img_%d' % f] = lu.downsize_img_tensor(torch.from_numpy(img), f).numpy()
        out['ds_img3_%d' % f] = lu.downsize_img_tensor(torch.from_numpy(img3), f).numpy()
        out['ds_mask_%d' % f] = lu.downsize_img_tensor(torch.from_numpy(msk), f).numpy()
    out.update(ds_img=img, ds_img3=img3, ds_mask=msk)
    # ---- bilinear sampling at pixel coordinates (torch-1.1 semantics = align_corners: shim 7)
    src = rs.rand(2, 3, 9, 11).astype(np.float32)
    xy = np.stack([rs.uniform(-1.5, 12.0, (2, 5, 6)), rs.uniform(-1.5, 10.0, (2, 5, 6))], 1).astype(np.float32)
    out.update(gs_img=src, gs_xy=xy, gs_out=lu.grid_sample_on_img(torch.from_numpy(src), torch.from_numpy(xy)).numpy())
    # ---- colour loss
    co = torch.from_numpy(rs.rand(12, 10, 3).astype(np.float32)).requires_grad_(True)
    cg = torch.from_numpy(rs.rand(12, 10, 3).astype(np.float32))
    m1 = torch.from_numpy(rs.rand(12, 10) < 0.6)
    m2 = torch.from_numpy(rs.rand(12, 10) < 0.7)
    lc, _ = lu.compute_loss_color(co, m1, cg, m2)
    lc.backward()
    out.update(lc_out=co.detach().numpy(), lc_gt=cg.numpy(), lc_m1=m1.numpy(), lc_m2=m2.numpy(), lc_loss=np.float64(lc.item()), lc_grad=co.grad.numpy())
    # ---- sim(3)
    sim3 = {'rot': torch.tensor([0.11, -0.07, 0.23], requires_grad=True), 'scale': torch.tensor(0.13, requires_grad=True),
            'trans': torch.tensor([0.05, -0.02, 0.3], requires_grad=True)}
    M = tu.params_to_mtrx(sim3)
    wM = rs.rand(3, 4).astype(np.float32)
    (M * torch.from_numpy(wM)).sum().backward()
    out.update(sim3_rot=sim3['rot'].detach().numpy(), sim3_scale=np.float32(0.13), sim3_trans=sim3['trans'].detach().numpy(), sim3_mtrx=M.detach().numpy(),
               sim3_w=wM, sim3_g_rot=sim3['rot'].grad.numpy(), sim3_g_scale=np.float32(sim3['scale'].grad.item()), sim3_g_trans=sim3['trans'].grad.numpy(),
               lie_big=tu.get_lie_rotation_matrix(torch.tensor([1.3, -0.8, 2.1])).numpy())
    np.savez_compressed(os.path.join(OUT, 'g21_python_helpers.npz'), **out)
    print('g21 done:', len(out), 'arrays')


if __name__ == '__main__':
    main()
