"""TEST / BASELINE INFRASTRUCTURE -- a PyTorch (CPU) restatement of the hot path in this build's own Python.

BASELINE.md section 3, baseline (2): "the build's own PyTorch-CPU restatement (what 'PyTorch CPU reference path' of config C1
means on the GPU box; the reference's Python never ships there)". Same algorithm as the reference's SDFRenderer.render
(core/sdfrenderer/renderer.py:472-583, 713-805, 836-999), written with batched ATen ops the way the
reference executes it -- per march step one decoder evaluation over the live rays (nine GEMMs on the (n, 259) [latent|xyz]
input, decoder_utils.py:53-74), step histories kept as (steps, N) tensors, top-k selection, re-evaluation of the selected
samples WITH grad so that loss.backward() replays the autograd tape. It is an independent second restatement (the C++ oracle is
the first) and is pinned against the same reference goldens (tests/test_torch_restatement.py).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

DEFAULT_M = np.array([[1., 0., 0.], [0., 0., -1.], [0., 1., 0.]])


def _normalize(v, dim=0):
    return v / (torch.norm(v, p=2, dim=dim, keepdim=True) + 1e-12)


class TorchRenderer(object):
    def __init__(self, Ws, bs, H, W, K, march_step=50, buffer_size=5, ratio=1.5, threshold=5e-5, radius=1.0, clamp_dist=0.1,
                 use_depth2normal=False, transform_matrix=None, coarse_steps=(3, 3), dtype=torch.float32):
        self.Ws = [torch.from_numpy(np.asarray(w, np.float32)).to(dtype) for w in Ws]
        self.bs = [torch.from_numpy(np.asarray(b, np.float32)).to(dtype) for b in bs]
        self.H, self.W = int(H), int(W)
        K = np.asarray(K, np.float64)
        self.K_inv = torch.from_numpy(np.linalg.inv(K).astype(np.float32)).to(dtype)
        self.fx, self.fy = float(np.float32(K[0, 0])), float(np.float32(K[1, 1]))
        self.S, self.bsz, self.ratio, self.thr, self.radius, self.cd = int(march_step), int(buffer_size), float(ratio), float(threshold), float(radius), float(clamp_dist)
        self.d2n = bool(use_depth2normal)
        self.M = torch.from_numpy(np.asarray(DEFAULT_M if transform_matrix is None else transform_matrix, np.float32)).to(dtype)
        self.coarse_steps = tuple(coarse_steps)
        self.dtype = dtype
        self.num_evals = 0

    # ------------------------------------------------------------------ decoder (deep_sdf_decoder.py:80-111, decoder_utils.py:53-74)
    def decode(self, latent, pts):
        """pts (n,3) -> (n,) unclamped sdf; the latent is expanded and concatenated per point like decode_sdf does."""
        n = pts.shape[0]
        self.num_evals += n
        inp = torch.cat([latent.reshape(1, -1).expand(n, -1), pts], 1)
        x = inp
        for l in range(9):
            if l == 4:
                x = torch.cat([x, inp], 1)
            x = F.linear(x, self.Ws[l], self.bs[l])
            if l < 8:
                x = torch.relu(x)
        return torch.tanh(x).reshape(-1)

    # ------------------------------------------------------------------ geometry (renderer.py:171-282)
    def _rays(self, R, px, py):
        h = self.K_inv @ torch.stack([px, py, torch.ones_like(px)], 0)          # (3,n)
        calib = h[2] / (torch.norm(h, p=2, dim=0) + 1e-12)
        return _normalize(R.t() @ h, 0), calib

    def _sphere(self, cam_pos, rays):
        ptq = (cam_pos[:, None] * rays).sum(0)
        dist = torch.norm(cam_pos[:, None] - ptq[None] * rays, p=2, dim=0)
        inside = dist <= self.radius
        chord = torch.where(inside, 2.0 * torch.sqrt(torch.clamp(self.radius ** 2 - dist ** 2, min=0.0)), torch.zeros_like(dist))
        cdist = torch.norm(cam_pos)
        if float(cdist.detach()) < self.radius:
            init = torch.zeros_like(dist)
        else:
            init = torch.sqrt(torch.clamp(cdist ** 2 - dist ** 2, min=0.0)) - chord / 2.0
            fill = init[inside].max() if bool(inside.any()) else init.new_zeros(())
            init = torch.where(inside, init, fill.expand_as(init))
        return dist, inside, chord, init

    def _points(self, cam_pos, rays, z):
        return (self.M.t() @ (cam_pos[:, None] + rays * z[None])).t()           # (n,3)

    # ------------------------------------------------------------------ marchers (no grad; renderer.py:472-583)
    def _march(self, latent, cam_pos, rays, init, maxbound, steps, recursive):
        """-> rows: sdf (S,n), zb (S,n) depth along the ray before the step (= init + m_before), ma (S,n) m after the step;
        padded rows of finished rays: sdf 1, zb NaN (their sample point is the origin), frozen m."""
        n = rays.shape[1]
        m = torch.zeros(n, dtype=self.dtype)
        live = (init < maxbound) if recursive else torch.ones(n, dtype=torch.bool)
        sdf_rows, zb_rows, ma_rows = [], [], []
        with torch.no_grad():
            for t in range(steps):
                s = torch.ones(n, dtype=self.dtype)
                zb = torch.full((n,), float('nan'), dtype=self.dtype)
                idx = torch.nonzero(live).reshape(-1)
                if idx.numel():
                    z = init[idx] + m[idx]
                    sv = self.decode(latent, self._points(cam_pos, rays[:, idx], z))
                    s[idx] = sv
                    zb[idx] = z
                    m[idx] = m[idx] + torch.clamp(sv, -self.cd, self.cd) * self.ratio
                    if recursive:
                        keep = (m[idx] + init[idx] < maxbound[idx]) & (sv.abs() >= self.thr)
                        live = live.clone()
                        live[idx] = keep
                sdf_rows.append(s); zb_rows.append(zb); ma_rows.append(m.clone())
                if recursive and not bool(live.any()):
                    while len(sdf_rows) < self.bsz:                     # renderer.py:563-566: pad to buffer_size rows
                        sdf_rows.append(sdf_rows[-1].clone()); zb_rows.append(zb_rows[-1].clone()); ma_rows.append(ma_rows[-1].clone())
                    break
        return torch.stack(sdf_rows), torch.stack(zb_rows), torch.stack(ma_rows)

    # ------------------------------------------------------------------ render (renderer.py:836-999)
    def render(self, latent, R, T, marcher='pyramid_recursive'):
        H, W = self.H, self.W
        P = H * W
        ys, xs = torch.meshgrid(torch.arange(H, dtype=self.dtype), torch.arange(W, dtype=self.dtype), indexing='ij')
        px, py = xs.reshape(-1), ys.reshape(-1)
        cam_pos = -(R.t() @ T)
        rays, calib = self._rays(R, px, py)
        dist, inside, chord, init_all = self._sphere(cam_pos, rays)
        ins = torch.nonzero(inside).reshape(-1)
        N = ins.numel()
        zdepth = torch.full((P,), 1e11, dtype=self.dtype)
        min_sdf = dist + self.thr - self.radius                                   # out-of-sphere px (renderer.py:863)
        valid = torch.zeros(P, dtype=torch.bool)
        if N:
            with torch.no_grad():
                cam_d, rays_d = cam_pos.detach(), rays.detach()
                init0 = init_all.detach()[ins]
                maxb = init0 + chord.detach()[ins]
                # every history row: sdf, depth-before along ITS OWN ray, which ray that is (level grid + index), z after (for z_sel)
                if marcher == 'trivial':
                    s, zb, ma = self._march(latent.detach(), cam_d, rays_d[:, ins], init0, maxb, self.S, False)
                    rows = dict(s=s, zb=zb, za=ma, lvl=torch.zeros_like(s, dtype=torch.long), ray=ins[None].expand_as(s))
                    z_last = ma[-1] + init0
                    vr = (z_last < maxb) & (s.abs().min(0)[0] <= self.thr) & (s[0] > self.thr)
                    m_of = lambda za: za                                           # rows store m after the step
                elif marcher == 'recursive':
                    s, zb, ma = self._march(latent.detach(), cam_d, rays_d[:, ins], init0, maxb, self.S, True)
                    rows = dict(s=s, zb=zb, za=ma, lvl=torch.zeros_like(s, dtype=torch.long), ray=ins[None].expand_as(s))
                    vr = (ma[-1] + init0 < maxb) & (s.abs().min(0)[0] <= self.thr) & (s[0] > self.thr)
                    m_of = lambda za: za
                else:
                    rows, vr, grids = self._pyramid(latent.detach(), R.detach(), cam_d, ins, init0, maxb)
                    m_of = lambda za: za - init0[None]                             # rows store zdepth after the step (renderer.py:804)
                # ---- selection (renderer.py:304-362, 382-420): bs smallest |sdf|, earliest step wins ties
                order = torch.sort(rows['s'].abs(), dim=0, stable=True)[1][:self.bsz]            # (bs, N)
                gather = lambda a: torch.gather(a, 0, order)
                s_sel, zb_sel, za_sel, lvl_sel, ray_sel = (gather(rows[k]) for k in ('s', 'zb', 'za', 'lvl', 'ray'))
                z_sel = m_of(za_sel)[0] + (1.0 - self.ratio) * torch.clamp(s_sel[0], -self.cd, self.cd)
            # ---- re-evaluate the selected samples WITH grad (latent + camera); padded rows sample the origin
            def sample_points(k):
                pts = torch.zeros(N, 3, dtype=self.dtype)
                real = ~torch.isnan(zb_sel[k])
                for lvl in torch.unique(lvl_sel[k][real]).tolist():
                    sel = real & (lvl_sel[k] == lvl)
                    if lvl == 0:
                        r_l = rays[:, ray_sel[k][sel]]
                    else:
                        gx, gy = grids[lvl]
                        r_l, _ = self._rays(R, gx[ray_sel[k][sel]], gy[ray_sel[k][sel]])
                    pts = pts.index_put((torch.nonzero(sel).reshape(-1),), self._points(cam_pos, r_l, zb_sel[k][sel]))
                return pts
            if marcher != 'pyramid_recursive':
                grids = None
            q = self.decode(latent, sample_points(0))
            z_fin = z_sel
            for k in range(self.bsz):
                sk = torch.clamp(self.decode(latent, sample_points(k)), -self.cd, self.cd)
                z_fin = z_fin + self.ratio * (sk - sk.detach())
            zdepth = zdepth.index_put((ins,), init0 + z_fin)
            min_sdf = min_sdf.index_put((ins,), q)
            valid = valid.index_put((ins,), vr)
        depth = torch.where(valid, zdepth * calib.detach(), torch.full_like(zdepth, 1e11)).reshape(H, W)
        if self.d2n:
            normal, depth = self._depth2normal(depth)
        else:
            normal = self._autograd_normal(latent, R, cam_pos, rays, zdepth, valid).reshape(H, W, 3)
        return depth, normal, valid.reshape(H, W), min_sdf.reshape(H, W), zdepth

    def _pyramid(self, latent, R, cam_pos, ins, init0, maxb):
        """renderer.py:713-805. Returns rows over the in-sphere level-0 rays, valid_render, and the coarse grids' pixel centres."""
        H, W = self.H, self.W
        hs, ws = [H], [W]
        for _ in range(2):
            hs.append((hs[-1] + 1) // 2); ws.append((ws[-1] + 1) // 2)
        valid0 = torch.zeros(H * W, dtype=torch.bool); valid0[ins] = True
        valids = [valid0.reshape(H, W)]
        for l in (1, 2):                                                          # OR-pool of the 2x2 children
            v = F.max_pool2d(F.pad(valids[-1][None, None].float(), (0, ws[l] * 2 - ws[l - 1], 0, hs[l] * 2 - hs[l - 1])), 2)[0, 0] > 0
            valids.append(v)
        grids = {}
        rows_s, rows_zb, rows_za, rows_lvl, rows_ray = [], [], [], [], []
        up = None                                                                 # last zdepth row, upsampled to the level below
        yy0, xx0 = torch.div(ins, W, rounding_mode='floor'), ins % W
        for l in (2, 1):
            scale = float(2 ** l)
            ys, xs = torch.meshgrid(torch.arange(hs[l], dtype=self.dtype), torch.arange(ws[l], dtype=self.dtype), indexing='ij')
            gx, gy = xs.reshape(-1) * scale + (scale - 1) / 2.0, ys.reshape(-1) * scale + (scale - 1) / 2.0
            grids[l] = (gx, gy)
            r_l, _ = self._rays(R, gx, gy)
            _, inside_l, chord_l, init_l = self._sphere(cam_pos, r_l)
            idx = torch.nonzero(valids[l].reshape(-1)).reshape(-1)
            start = init_l if up is None else up
            steps = self.coarse_steps[2 - l]
            s, zb, ma = self._march(latent, cam_pos, r_l[:, idx], start[idx], start[idx] + 1e30, steps, False)
            n_l = hs[l] * ws[l]
            full_s = torch.ones(steps, n_l, dtype=self.dtype); full_s[:, idx] = s
            full_zb = torch.full((steps, n_l), float('nan'), dtype=self.dtype); full_zb[:, idx] = zb
            full_za = start[None].repeat(steps, 1); full_za[:, idx] = ma + start[idx][None]                  # unmapped rows: m = 0 (renderer.py:776-779)
            # parent of every in-sphere level-0 pixel at this level
            par = torch.div(yy0, 2 ** l, rounding_mode='floor') * ws[l] + torch.div(xx0, 2 ** l, rounding_mode='floor')
            rows_s.append(full_s[:, par]); rows_zb.append(full_zb[:, par]); rows_za.append(full_za[:, par])
            rows_lvl.append(torch.full((steps, ins.numel()), l, dtype=torch.long)); rows_ray.append(par[None].expand(steps, -1))
            # upsample the last zdepth row to level l-1 (nearest parent)
            hh, wwid = hs[l - 1], ws[l - 1]
            yy, xx = torch.meshgrid(torch.arange(hh), torch.arange(wwid), indexing='ij')
            up = full_za[-1][(torch.div(yy, 2, rounding_mode='floor') * ws[l] + torch.div(xx, 2, rounding_mode='floor')).reshape(-1)]
        # level 0: recursive from the upsampled depth, no first-query check
        rays0, _ = self._rays(R, (ins % W).to(self.dtype), torch.div(ins, W, rounding_mode='floor').to(self.dtype))
        start0 = up[ins]
        fine = self.S - sum(self.coarse_steps)
        s, zb, ma = self._march(latent, cam_pos, rays0, start0, maxb, fine, True)
        rows_s.append(s); rows_zb.append(zb); rows_za.append(ma + start0[None])
        rows_lvl.append(torch.zeros_like(s, dtype=torch.long)); rows_ray.append(ins[None].expand_as(s))
        rows = dict(s=torch.cat(rows_s), zb=torch.cat(rows_zb), za=torch.cat(rows_za), lvl=torch.cat(rows_lvl), ray=torch.cat(rows_ray))
        vr = (ma[-1] + start0 < maxb) & (s.abs().min(0)[0] <= self.thr)          # the level-0 marcher's own rows (renderer.py:573-577)
        return rows, vr, grids

    def _depth2normal(self, depth):
        """core/utils/render_utils.py:9-43 incl. the in-place zeroing of the background depth."""
        bg = (depth > 1e5) | (depth == 0)
        d = torch.where(bg, torch.zeros_like(depth), depth)
        zx, zy = torch.zeros_like(d), torch.zeros_like(d)
        zx = torch.cat([zx[:, :1], (d[:, 2:] - d[:, :-2]) * self.fx / 2.0, zx[:, -1:]], 1)
        zy = torch.cat([zy[:1], (d[2:] - d[:-2]) * self.fy / 2.0, zy[-1:]], 0)
        n = torch.stack([zx, zy, -torch.ones_like(d)], -1)
        n = n / (torch.norm(n, p=2, dim=2, keepdim=True) + 1e-12)
        return torch.where(bg[..., None], torch.zeros_like(n), n), d

    def _autograd_normal(self, latent, R, cam_pos, rays, zdepth, valid):
        """renderer.py:880-910, 977-980: n = normalize(3 * grad f) at the surface points, M n, R (M n), x flipped. The normalised
        gradient is piecewise constant in (latent, point) for a ReLU decoder, so only the explicit R product carries gradient."""
        P = zdepth.shape[0]
        out = torch.zeros(P, 3, dtype=self.dtype)
        idx = torch.nonzero(valid).reshape(-1)
        if idx.numel() == 0:
            return out
        pts = self._points(cam_pos.detach(), rays.detach()[:, idx], zdepth.detach()[idx]).requires_grad_(True)
        with torch.enable_grad():
            f = self.decode(latent.detach(), pts)
            g, = torch.autograd.grad(torch.clamp(f, -self.cd, self.cd).sum(), pts)
        n = _normalize(3.0 * g.t(), 0)                                           # (3,n)
        t = self.M @ n
        o = R @ t
        o = torch.stack([-o[0], o[1], o[2]], 1)
        return out.index_put((idx,), o)


def render_fwd_bwd(Ws, bs, latent, H, W, K, R, T, weights, marcher='pyramid_recursive', threads=None, **kw):
    """One forward + backward of the goldens' loss (sum depth*wd on the mask + sum min_sdf*wq + sum normal*wn). Returns a dict of
    numpy outputs and gradients."""
    if threads:
        torch.set_num_threads(int(threads))
    r = TorchRenderer(Ws, bs, H, W, K, **kw)
    lat = torch.from_numpy(np.asarray(latent, np.float32)).clone().requires_grad_(True)
    Rt = torch.from_numpy(np.asarray(R, np.float32)).clone().requires_grad_(True)
    Tt = torch.from_numpy(np.asarray(T, np.float32)).clone().requires_grad_(True)
    depth, normal, mask, q, zdepth = r.render(lat, Rt, Tt, marcher)
    wd, wq, wn = (torch.from_numpy(a) for a in weights)
    L = (depth * wd)[mask].sum() + (q * wq).sum() + (normal * wn).sum()
    L.backward()
    z = lambda t: None if t is None else t.detach().numpy()
    return dict(depth=z(depth), normal=z(normal), mask=mask.numpy().astype(np.uint8), min_sdf=z(q), zdepth=z(zdepth),
                g_latent=z(lat.grad), g_R=z(Rt.grad if Rt.grad is not None else torch.zeros(3, 3)),
                g_T=z(Tt.grad if Tt.grad is not None else torch.zeros(3)), num_evals=r.num_evals, loss=float(L))
