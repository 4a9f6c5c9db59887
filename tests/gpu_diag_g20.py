"""Diagnostic (not a pytest file): G20's per-iteration gradients evaluated AT THE REFERENCE'S OWN latent of each iteration (no trajectory),
per scale: which renderer of the multi-scale list carries a residual, and whether a mask pixel differs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p_)
from distr import fixture
from core.sdfrenderer import SDFRenderer
from core.graph.deep_sdf_decoder import Decoder
from core.inv_optimizer.loss_single import compute_all_loss
from core.utils.render_utils import downsize_camera_intrinsic
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g20_multiscale_shape_loop.npz')))
Ws, bs, _ = fixture.make_decoder_weights()
dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W_, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W_), ('bias', b))})
dec = dec.cuda()
H, W, S = int(g['H']), int(g['W']), int(g['march_step'])
mk = lambda Kx, b_, **kw: SDFRenderer(dec, Kx, march_step=S, buffer_size=b_, threshold=5e-5, use_depth2normal=True, **kw)
rs = [mk(g['K'], 1, img_hw=(H, W), ray_marching_ratio=1.5), mk(downsize_camera_intrinsic(g['K'], 2), 3), mk(downsize_camera_intrinsic(g['K'], 4), 5)]
RT = torch.from_numpy(g['RT']).cuda()
gt_pack = {'depth': torch.from_numpy(g['gt_depth']).cuda(), 'normal': torch.from_numpy(g['gt_normal']).cuda(), 'silhouette': torch.from_numpy(g['gt_mask']).cuda()}
wd = dict(w_depth=10.0, w_normal=5.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
for i in range(int(g['iters'])):
    lat = torch.from_numpy(g['values'][i]).cuda().requires_grad_(True)
    tot = 0
    per = []
    for r in rs:
        pack, _ = compute_all_loss(r, lat, RT, gt_pack, threshold=r.get_threshold())
        L = sum(wd['w_' + k] * pack[k] for k in ('depth', 'normal', 'mask_gt', 'mask_out', 'l2reg'))
        gi, = torch.autograd.grad(L, lat, retain_graph=False)
        per.append(gi.cpu().numpy())
        with torch.no_grad():
            m = r.render(lat.detach(), RT[:, :3], RT[:, 3], no_grad=True)[2]
        per[-1] = (per[-1], int(m.sum()))
    gsum = sum(p[0] for p in per)
    ref = g['grads'][i]
    print('iter %d: |ours(ref latent) - ref| / max|ref| = %.2e ; per-scale grad max %s valid px %s' % (
        i, np.abs(gsum - ref).max() / np.abs(ref).max(), ['%.2f' % np.abs(p[0]).max() for p in per], [p[1] for p in per]))
