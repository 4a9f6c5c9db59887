"""Diagnostic (not a test): where the time of one batched multi-view round goes (run_multi_pmodata.py:92 setting: 8 view pairs of
137 x 137, march_step 100, buffer_size 1, 'recursive' marcher, one shape code) -- phases timed with events, decoder evaluations
and march-kernel roofline of the batched depth render, live-ray profile. Run: python tests/gpu_diag_batch.py [size] [pairs]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
sys.path.insert(0, ROOT)

from core.graph.deep_sdf_decoder import Decoder  # noqa: E402
from core.inv_optimizer.loss_multi import pair_cameras  # noqa: E402
from core.inv_optimizer.optimize_multi import pair_indices  # noqa: E402
from core.sdfrenderer import SDFRenderer_warp  # noqa: E402
from distr import binding, fixture, functions  # noqa: E402
from oracle.gen_synth import procedural_images  # noqa: E402  (synthetic images only)

FLOP = 3146752


class Cam(object):
    def __init__(self, R, T):
        self.extrinsic = np.concatenate([R, T[:, None]], 1).astype(np.float32)


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) / n


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 137
    npairs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    marcher = sys.argv[3] if len(sys.argv) > 3 else 'recursive'
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W), ('bias', b))})
    K = fixture.make_intrinsic(size, size)
    r = SDFRenderer_warp(dec.cuda(), K, img_hw=(size, size), march_step=100, buffer_size=1)
    n_img = 24
    cams = [Cam(*fixture.make_camera(15.0 * i, 20.0, 1.6, 0.0)) for i in range(n_img)]
    a, b = procedural_images(size, size)
    imgs = [torch.from_numpy(np.roll(a if i % 2 else b, 3 * i, axis=1).copy()).cuda() for i in range(n_img)]
    lat = torch.from_numpy(latent).cuda().requires_grad_(True)
    pairs = [pair_indices(0, i, n_img / 8, 1, n_img) for i in range(npairs)]
    args = []
    for (i1, i2) in pairs:
        (R1, T1), (R2, T2) = pair_cameras(cams, i1, i2, lat.device)
        args.append((R1, T1, R2, T2, imgs[i1], imgs[i2]))
    Rs = torch.stack([t for p in args for t in (p[0], p[2])])
    Ts = torch.stack([t for p in args for t in (p[1], p[3])])
    B = Rs.shape[0]
    ngd = [False, True] * npairs
    eng = r._engine

    (Z, M, Q), t_depth = timed(lambda: r.render_depth_batch(lat, Rs, Ts, no_grad_depth=ngd, ray_marching_type=marcher))
    _, t_norm = timed(lambda: r.render_normal_batch(lat, Rs[0::2], Ts[0::2], Z[0::2].detach(), M[0::2]))
    wcfg = binding.make_warp_cfg((size, size), r.intrinsic, 0.001)

    def warps():
        tot = 0.0
        for i, (R1, T1, R2, T2, i1, i2) in enumerate(args):
            tot = tot + functions.warp_loss(eng, wcfg, Z[2 * i], M[2 * i], Z[2 * i + 1], i1, i2, R1, T1, R2, T2)[0]
        return tot
    loss, t_warp = timed(warps)

    def fwd_bwd():
        lat.grad = None
        outs = r.render_warp_batch(lat, args)
        tot = 0.0
        for o in outs:
            tot = tot + 5.0 * o[0]
        tot.backward()
        return tot
    _, t_all = timed(fwd_bwd)
    print('%dx%d, %d views (%s): depth batch %.2f ms, normals batch %.2f ms, %d warp losses %.2f ms, whole fwd+bwd %.2f ms -> backward + glue %.2f ms'
          % (size, size, B, marcher, t_depth, t_norm, npairs, t_warp, t_all, t_all - t_depth - t_norm - t_warp))

    # decoder evaluations + march-kernel roofline of the batched depth render (raw C ABI: the workspace is needed for the counters)
    import ctypes as C
    cfg = r._cfg(0.1, marcher, True, want_normal=False)
    cfg.save_for_backward = 1
    fwd, _ = eng.ctx.workspace_bytes(cfg)
    ws = torch.empty(B * fwd, dtype=torch.uint8, device='cuda')
    P = size * size
    z, m, q = torch.empty(B, P, device='cuda'), torch.empty(B, P, dtype=torch.uint8, device='cuda'), torch.empty(B, P, device='cuda')
    p = binding.ptr
    fl = (C.c_int32 * B)(*[6 if x else 7 for x in ngd])
    latc, Rc, Tc = lat.detach().reshape(-1).contiguous(), Rs.detach().reshape(B, 9).contiguous(), Ts.detach().contiguous()

    def raw():
        eng.ctx.check(eng.ctx.L.distr_render_forward_batch(eng.ctx.h, C.byref(cfg), B, fl, p(latc), 0, p(Rc), p(Tc), p(z), p(m), p(q), None, None,
                                                           p(ws), ws.numel(), eng.ctx.stream()))
    raw()
    eng.ctx.profile_enable(True)
    raw()
    ms = eng.ctx.profile_read_list()
    eng.ctx.profile_enable(False)
    stats = [eng.ctx.render_stats(cfg, ws[b * fwd:(b + 1) * fwd]) for b in range(B)]
    live = np.array([eng.ctx.live_counts(cfg, ws[b * fwd:(b + 1) * fwd]) for b in range(B)])
    evals = sum(s['num_point_evals'] for s in stats)
    tot_ms = float(np.sum(ms))
    print('batched depth render: %d rays in sphere, %d decoder evaluations (%.1f per image ray), %d march launches, %.2f ms of march kernels '
          '-> %.1f TFLOP/s = %.3f of the f32-MFMA peak; cluster fallbacks %d'
          % (sum(s['num_in_sphere'] for s in stats), evals, evals / (B * P), len(ms), tot_ms, FLOP * evals / tot_ms / 1e9,
             FLOP * evals / tot_ms / 1e9 / 157.3, sum(s['cluster_fallbacks'] for s in stats)))
    tot_live = live.sum(0)
    print('step: live rays of all views | kernel us | TFLOP/s')
    for i, (n, t) in enumerate(zip(tot_live, ms)):
        if i < 30 or i % 10 == 0:
            print('  %3d %7d %8.1f %6.1f' % (i, n, 1e3 * t, FLOP * n / t / 1e9 if t > 0 else 0.0))


if __name__ == '__main__':
    main()
