"""Shared helpers of the GPU parity tests (HIP kernels through the C ABI vs the CPU oracle)."""
import numpy as np


def loss_weights(H, W, seed):
    rs = np.random.RandomState(seed)
    return rs.rand(H, W).astype(np.float32), rs.rand(H, W).astype(np.float32), rs.rand(H, W, 3).astype(np.float32)


def oracle_render(O, orc, H, W, K, R, T, latent, seed=5, **kw):
    cfg = orc.make_cfg(H, W, K, **kw)
    out = O.render(cfg, latent, R, T)
    wd, wq, wn = loss_weights(H, W, seed)
    m = out['mask'].reshape(H, W).astype(np.float32)
    gl, gR, gT, ns = out['state'].backward(g_min_sdf=wq.reshape(-1), g_depth=(wd * m).reshape(-1), g_normal=wn.reshape(-1))
    out.update(g_latent=gl, g_R=gR, g_T=gT, num_samples=ns, num_evals=out['state'].num_evals)
    return out


def hip_render(engine, H, W, K, R, T, latent, seed=5, **kw):
    """Same computation through libdistr (torch only owns the buffers / runs the tiny loss)."""
    import torch
    from distr import binding, functions
    dev = engine.device
    cfg = binding.make_cfg((H, W), K, **kw)
    lat = torch.from_numpy(np.asarray(latent, np.float32)).to(dev).requires_grad_(True)
    Rt = torch.from_numpy(np.asarray(R, np.float32)).to(dev).requires_grad_(True)
    Tt = torch.from_numpy(np.asarray(T, np.float32)).to(dev).requires_grad_(True)
    zdepth, mask, min_sdf, depth, normal = functions.render_call(engine, cfg, lat, Rt, Tt)
    wd, wq, wn = (torch.from_numpy(a).to(dev) for a in loss_weights(H, W, seed))
    mb = mask.reshape(H, W).bool()
    L = (depth * wd)[mb].sum() + (min_sdf.reshape(H, W) * wq).sum() + (normal * wn).sum()
    L.backward()
    torch.cuda.synchronize()
    return dict(zdepth=zdepth.detach().cpu().numpy(), mask=mask.cpu().numpy(), min_sdf=min_sdf.detach().cpu().numpy(),
                depth=depth.detach().cpu().numpy(), normal=normal.detach().cpu().numpy(),
                g_latent=lat.grad.cpu().numpy(), g_R=Rt.grad.cpu().numpy(), g_T=Tt.grad.cpu().numpy(), cfg=cfg, loss=float(L.detach()))


def compare(a, b, H, W, tol_depth=1e-4, tol_grad=1e-3, normal_p99=1e-4, max_flip_frac=0.001):
    """a: HIP, b: oracle/golden dicts. Returns a dict of residuals after asserting the bars."""
    ma, mb = a['mask'].reshape(H, W).astype(bool), b['mask'].reshape(H, W).astype(bool)
    flips = int((ma != mb).sum())
    both = ma & mb
    res = dict(flips=flips,
               depth=float(np.abs(a['depth'] - b['depth'])[both].max()) if both.any() else 0.0,
               zdepth=float(np.abs(a['zdepth'].reshape(H, W) - b['zdepth'].reshape(H, W))[both].max()) if both.any() else 0.0,
               min_sdf=float(np.abs(a['min_sdf'].reshape(-1) - b['min_sdf'].reshape(-1)).max()),
               normal_p99=float(np.percentile(np.abs(a['normal'] - b['normal'])[both], 99)) if both.any() else 0.0,
               normal_max=float(np.abs(a['normal'] - b['normal'])[both].max()) if both.any() else 0.0)
    for k in ('g_latent', 'g_R', 'g_T'):
        res[k] = float(np.abs(a[k].reshape(-1) - b[k].reshape(-1)).max() / max(np.abs(b[k]).max(), 1e-30))
    assert flips <= max(1, int(max_flip_frac * H * W)), res
    assert res['depth'] <= tol_depth and res['zdepth'] <= tol_depth and res['min_sdf'] <= tol_depth, res
    assert res['normal_p99'] <= normal_p99, res
    assert max(res['g_latent'], res['g_R'], res['g_T']) <= tol_grad, res
    return res
