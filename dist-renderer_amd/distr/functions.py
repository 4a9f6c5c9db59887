"""torch.autograd glue around the C ABI: forward -> distr_render_forward, backward -> distr_render_backward.

Replaces the autograd tape the reference records through ~50 decode_sdf calls per render
(core/sdfrenderer/renderer.py:382-420, 836-878): the forward keeps O(buffer_size) selected rows per ray
in the workspace tensor together with the ReLU masks of those rows, the backward kernel runs the dX chain at exactly those points.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import binding, decoder_pack


def _param_key(module):
    """Identity + in-place version of every parameter: changes after load_state_dict / an optimiser step / .to().
    NOT seen: edits through `.data` (`p.data.copy_()`, `p.data.mul_()`, EMA-style updates) -- they do not bump `_version`. Code
    that edits decoder weights that way must call `get_engine(decoder, dev).refresh(decoder)` itself (a content checksum per
    render would cost a device reduction and a host sync in the hot path)."""
    return tuple((p.data_ptr(), p._version) for p in module.parameters())


def _check_eval(module):
    if module.training and getattr(module, 'dropout', None) and float(getattr(module, 'dropout_prob', 0.0) or 0.0) > 0.0:
        raise decoder_pack.UnsupportedDecoder('decoder is in training mode with dropout: the fused kernels evaluate the deterministic '
                                              '(eval) network only -- call decoder.eval() (SDFRenderer(is_eval=True) does)')


class DecoderEngine(object):
    """Packed decoder on one device (distr_ctx). One per (decoder module, device)."""

    def __init__(self, decoder, device_index):
        self.ctx = binding.Context(device_index)
        self.device = torch.device('cuda', device_index)
        self._key = None
        self.refresh(decoder)

    generation = 0     # number of weight uploads so far: a forward remembers it, its backward refuses to run on newer weights

    def refresh(self, decoder):
        """(Re)uploads the weights."""
        self.ctx.set_decoder(decoder_pack.pack_module(decoder))
        self._key = _param_key(decoder)
        self.generation += 1

    def sync(self, decoder):
        """The reference reads the live module on every call (decoder_utils.py:53-74); the packed copy follows it: re-packed
        when any parameter was modified in place, replaced or moved since the last upload (a few microseconds to check)."""
        _check_eval(decoder)
        if _param_key(decoder) != self._key:
            self.refresh(decoder)
        return self


_engines = weakref.WeakKeyDictionary()


def get_engine(decoder, device_index):
    per_dec = _engines.setdefault(decoder, {})
    if device_index not in per_dec:
        _check_eval(decoder)
        per_dec[device_index] = DecoderEngine(decoder, device_index)
    return per_dec[device_index].sync(decoder)


def engine_from_weights(Ws, bs, device_index=0):
    """Engine straight from numpy weights (tests / bench; no nn.Module needed)."""
    eng = DecoderEngine.__new__(DecoderEngine)
    eng.ctx = binding.Context(device_index)
    eng.device = torch.device('cuda', device_index)
    eng.ctx.set_decoder(decoder_pack.flatten(Ws, bs))
    return eng


class ColorEngine(object):
    """Packed colour decoder on one device (decode_color / SDFRenderer_color)."""

    def __init__(self, decoder_color=None, device_index=0, weights=None):
        self.ctx = binding.Context(device_index)
        self.device = torch.device('cuda', device_index)
        self._key = None
        if weights is None:
            self.refresh(decoder_color)
        else:
            flat, nlat = decoder_pack.flatten_color(*weights)
            self.latent_size = nlat
            self.ctx.set_color_decoder(flat, nlat)

    def refresh(self, decoder_color):
        flat, nlat = decoder_pack.pack_color_module(decoder_color)
        self.latent_size = nlat
        self.ctx.set_color_decoder(flat, nlat)
        self._key = _param_key(decoder_color)

    def sync(self, decoder_color):
        _check_eval(decoder_color)
        if _param_key(decoder_color) != self._key:
            self.refresh(decoder_color)
        return self


_color_engines = weakref.WeakKeyDictionary()


def get_color_engine(decoder_color, device_index):
    per_dec = _color_engines.setdefault(decoder_color, {})
    if device_index not in per_dec:
        _check_eval(decoder_color)
        per_dec[device_index] = ColorEngine(decoder_color, device_index)
    return per_dec[device_index].sync(decoder_color)


def color_eval(engine, color_code, shape_code, points):
    """decode_color forward (core/utils/decoder_utils.py:94-112): points (n,3) -> rgb (n,3); the decoder input is
    [shape_code | color_code | xyz] (decoder_utils.py:101-103)."""
    dev = engine.device
    lat = torch.cat([_f32c(shape_code, dev).reshape(-1), _f32c(color_code, dev).reshape(-1)])
    if lat.numel() != engine.latent_size:
        raise ValueError('shape code + colour code have %d entries, the colour decoder expects %d' % (lat.numel(), engine.latent_size))
    x = _f32c(points, dev).reshape(-1, 3)
    n = x.shape[0]
    out = torch.empty(n, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(engine.ctx.L.distr_mlp_workspace_bytes(n), dtype=torch.uint8, device=dev)
    p = binding.ptr
    engine.ctx.check(engine.ctx.L.distr_color_eval(engine.ctx.h, p(lat), p(x), n, p(out), p(ws), ws.numel(), engine.ctx.stream()))
    return out


class ColorDecodeFunction(torch.autograd.Function):
    """decode_color with autograd (decoder_utils.py:94-112 called with no_grad=False): (color_code (1,cs), shape_code (1,256),
    points (n,3)) -> rgb (n,3); backward = distr_color_backward (fused forward recompute + dX chain with the 3-row output layer;
    the code gradients come from the per-tile delta sums)."""

    @staticmethod
    def forward(ctx, color_code, shape_code, points, engine):
        out = color_eval(engine, color_code, shape_code, points)
        ctx.engine = engine
        ctx.save_for_backward(color_code.detach(), shape_code.detach(), points.detach())
        ctx.need = (color_code.requires_grad, shape_code.requires_grad, points.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        engine = ctx.engine
        dev = engine.device
        color_code, shape_code, points = ctx.saved_tensors
        lat = torch.cat([_f32c(shape_code, dev).reshape(-1), _f32c(color_code, dev).reshape(-1)])
        x = _f32c(points, dev).reshape(-1, 3)
        n = x.shape[0]
        gs = _f32c(g, dev).reshape(-1, 3)
        g_x = torch.empty(n, 3, dtype=torch.float32, device=dev) if ctx.need[2] else None
        g_l = torch.empty(engine.latent_size, dtype=torch.float32, device=dev) if (ctx.need[0] or ctx.need[1]) else None
        ws = torch.empty(engine.ctx.L.distr_mlp_backward_workspace_bytes(n), dtype=torch.uint8, device=dev)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_color_backward(engine.ctx.h, p(lat), p(x), n, p(gs), p(g_x), p(g_l), p(ws), ws.numel(), engine.ctx.stream()))
        ns = shape_code.numel()
        return ((g_l[ns:].reshape(color_code.shape) if ctx.need[0] else None), (g_l[:ns].reshape(shape_code.shape) if ctx.need[1] else None),
                (g_x.reshape(points.shape) if ctx.need[2] else None), None)


def color_eval_autograd(engine, color_code, shape_code, points):
    return ColorDecodeFunction.apply(color_code, shape_code, points, engine)


def _f32c(t, device):
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _check_generation(ctx):
    """A backward must run on the decoder weights its forward used: the saved ReLU masks (and every selected row) belong to THAT
    evaluation. Rendering A, updating the decoder, rendering B, then back-propagating A would silently mix new weights with old
    masks -- refuse instead."""
    if getattr(ctx.engine, 'generation', 0) != ctx.generation:
        raise RuntimeError('the decoder weights were re-uploaded (load_state_dict / optimiser step / refresh) between this render\'s '
                           'forward and its backward; back-propagate before changing the decoder, or render again')


F16_CHECK_EVERY = 256          # renders between two range checks of the split-f16 arithmetic (DISTR_F16_CHECK_EVERY; the first render of
                               # every decoder upload is always checked)


def _check_f16_range(engine, cfg, ws, nviews=1, view_bytes=0, needs_grad=False):
    """arith = 'f16x3' only. A decoder activation that leaves the f16 range makes the march see a NaN sdf, which fails both `stay`
    comparisons and silently ends the ray -- a driver would optimise on corrupted renders. The march kernel counts such evaluations
    (distr_render_stats.f16_overflows); reading the counter costs a device sync, so for gradient-free renders it is read on the
    first render after every decoder upload and then every F16_CHECK_EVERY renders. A render that will be back-propagated
    (`needs_grad`: an optimisation loop moving the shape code, where overflow depends on the current latent) is checked EVERY time:
    a backward taken on NaN-terminated rays is the failure that must not pass silently. Anything but zero RAISES (use
    arith='bf16x6' or 'f32' for this decoder). Residual window: up to F16_CHECK_EVERY - 1 consecutive gradient-free renders."""
    if cfg.arith != binding.ARITH['f16x3']:
        return
    import os
    every = int(os.environ.get('DISTR_F16_CHECK_EVERY', F16_CHECK_EVERY))
    gen = getattr(engine, 'generation', 0)
    st = engine.__dict__.setdefault('_f16_state', {'gen': None, 'n': 0})
    first = st['gen'] != gen
    if first:
        st['gen'], st['n'] = gen, 0
    st['n'] += 1
    if not (first or needs_grad or (every > 0 and st['n'] % every == 0)):
        return
    bad = 0
    for v in range(nviews):
        bad += engine.ctx.render_stats(cfg, ws[v * view_bytes:] if nviews > 1 else ws)['f16_overflows']
    if bad:
        raise binding.DistrError("arith='f16x3': %d decoder evaluation(s) of this render left the f16 range (|activation| x 64 >= 65504): the "
                                 "render is not to be trusted. Use arith='bf16x6' or 'f32' for this decoder / shape code." % bad)


class RenderFunction(torch.autograd.Function):
    """(latent, R, T) -> (zdepth[P], mask[P] uint8, min_sdf[P], depth[H,W], normal[H,W,3])"""

    @staticmethod
    def forward(ctx, latent, R, T, engine, cfg):
        dev = engine.device
        H, W = cfg.band_rows, cfg.W
        P = H * W
        lat, Rc, Tc = _f32c(latent, dev).reshape(-1), _f32c(R, dev).reshape(-1), _f32c(T, dev).reshape(-1)
        if lat.numel() != 256 or Rc.numel() != 9 or Tc.numel() != 3:
            raise ValueError('expected latent (1,256), R (3,3), T (3)')
        fwd_bytes, bwd_bytes = engine.ctx.workspace_bytes(cfg)
        ws = torch.empty(fwd_bytes, dtype=torch.uint8, device=dev)
        zdepth = torch.empty(P, dtype=torch.float32, device=dev)
        mask = torch.empty(P, dtype=torch.uint8, device=dev)
        min_sdf = torch.empty(P, dtype=torch.float32, device=dev)
        if cfg.want_normal:
            depth = torch.empty(H, W, dtype=torch.float32, device=dev)
            normal = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        else:
            depth = torch.empty(0, dtype=torch.float32, device=dev)
            normal = torch.empty(0, dtype=torch.float32, device=dev)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_render_forward(
            engine.ctx.h, C.byref(cfg), p(lat), p(Rc), p(Tc), p(zdepth), p(mask), p(min_sdf),
            p(depth) if cfg.want_normal else None, p(normal) if cfg.want_normal else None,
            p(ws), ws.numel(), engine.ctx.stream()))
        _check_f16_range(engine, cfg, ws, needs_grad=any(ctx.needs_input_grad[:3]))
        ctx.engine, ctx.cfg, ctx.ws, ctx.bwd_bytes = engine, cfg, ws, bwd_bytes
        ctx.generation = getattr(engine, 'generation', 0)
        ctx.shapes = (latent.shape, R.shape, T.shape)
        ctx.in_meta = tuple((t.device, t.dtype) for t in (latent, R, T))
        ctx.mark_non_differentiable(mask)
        return zdepth, mask, min_sdf, depth, normal

    @staticmethod
    def backward(ctx, g_zdepth, g_mask, g_min_sdf, g_depth, g_normal):
        engine, cfg, ws = ctx.engine, ctx.cfg, ctx.ws
        _check_generation(ctx)
        dev = engine.device

        def prep(g, n):
            if g is None or g.numel() != n:
                return None
            return g.to(dtype=torch.float32).contiguous()
        P = cfg.band_rows * cfg.W
        gz, gq = prep(g_zdepth, P), prep(g_min_sdf, P)
        gd, gn = (prep(g_depth, P), prep(g_normal, 3 * P)) if cfg.want_normal else (None, None)
        g_lat = torch.empty(256, dtype=torch.float32, device=dev)
        g_R = torch.empty(9, dtype=torch.float32, device=dev)
        g_T = torch.empty(3, dtype=torch.float32, device=dev)
        ws_b = torch.empty(ctx.bwd_bytes, dtype=torch.uint8, device=dev)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_render_backward(
            engine.ctx.h, C.byref(cfg), p(ws), ws.numel(), p(gz), p(gq), p(gd), p(gn), p(g_lat), p(g_R), p(g_T),
            p(ws_b), ws_b.numel(), engine.ctx.stream()))
        ctx.last_ws = ws
        ls, rs, ts = ctx.shapes
        # gradients go back in the inputs' own device / dtype (a host-resident or f64 camera tensor keeps working)
        return tuple(g.reshape(sh).to(device=d, dtype=dt) for g, sh, (d, dt) in zip((g_lat, g_R, g_T), (ls, rs, ts), ctx.in_meta)) + (None, None)


def render_call(engine, cfg, latent, R, T):
    # inference (torch.no_grad() or no input requires grad): skip saving the ReLU masks for the backward pass
    need_bwd = torch.is_grad_enabled() and any(getattr(t, 'requires_grad', False) for t in (latent, R, T))
    cfg = cfg.clone()          # the autograd node keeps ITS cfg: a caller reusing one cfg object for a later no-grad render must
    cfg.save_for_backward = 1 if need_bwd else 0      # not flip save_for_backward under a pending backward
    return RenderFunction.apply(latent, R, T, engine, cfg)


class RenderBatchFunction(torch.autograd.Function):
    """Several views in one launch sequence (distr_render_forward_batch): (latent (1,256) shared by all views or (B,256), R (B,3,3),
    T (B,3)) -> (zdepth (B,P), mask (B,P) uint8, min_sdf (B,P), depth (B,H,W), normal (B,H,W,3)). Every view's values and
    gradients are bit-identical to its own RenderFunction call; what changes is the schedule (one march launch per step for all
    views: the views' latency-bound tails overlap). `view_flags`: per-view DISTR_VIEW_GRAD_* (the no_grad_* options of each view)."""

    @staticmethod
    def forward(ctx, latent, R, T, engine, cfg, view_flags):
        dev = engine.device
        H, W = cfg.band_rows, cfg.W
        P = H * W
        Rc, Tc = _f32c(R, dev).reshape(-1, 9), _f32c(T, dev).reshape(-1, 3)
        B = Rc.shape[0]
        lat = _f32c(latent, dev).reshape(-1, 256)
        if Tc.shape[0] != B or lat.shape[0] not in (1, B) or not (1 <= B <= binding.MAX_VIEWS):
            raise ValueError('expected latent (1,256) or (B,256), R (B,3,3), T (B,3) with 1 <= B <= %d' % binding.MAX_VIEWS)
        shared = lat.shape[0] == 1
        fwd_bytes, bwd_bytes = engine.ctx.workspace_bytes(cfg)
        ws = torch.empty(B * fwd_bytes, dtype=torch.uint8, device=dev)
        zdepth = torch.empty(B, P, dtype=torch.float32, device=dev)
        mask = torch.empty(B, P, dtype=torch.uint8, device=dev)
        min_sdf = torch.empty(B, P, dtype=torch.float32, device=dev)
        if cfg.want_normal:
            depth = torch.empty(B, H, W, dtype=torch.float32, device=dev)
            normal = torch.empty(B, H, W, 3, dtype=torch.float32, device=dev)
        else:
            depth = torch.empty(0, dtype=torch.float32, device=dev)
            normal = torch.empty(0, dtype=torch.float32, device=dev)
        flags = None if view_flags is None else (C.c_int32 * B)(*[int(f) for f in view_flags])
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_render_forward_batch(
            engine.ctx.h, C.byref(cfg), B, flags, p(lat), 0 if shared else 256, p(Rc), p(Tc), p(zdepth), p(mask), p(min_sdf),
            p(depth) if cfg.want_normal else None, p(normal) if cfg.want_normal else None, p(ws), ws.numel(), engine.ctx.stream()))
        _check_f16_range(engine, cfg, ws, B, fwd_bytes, needs_grad=any(ctx.needs_input_grad[:3]))
        ctx.engine, ctx.cfg, ctx.ws, ctx.bwd_bytes, ctx.B, ctx.shared = engine, cfg, ws, bwd_bytes, B, shared
        ctx.generation = getattr(engine, 'generation', 0)
        ctx.shapes = (latent.shape, R.shape, T.shape)
        ctx.in_meta = tuple((t.device, t.dtype) for t in (latent, R, T))
        ctx.view_bytes = fwd_bytes
        ctx.mark_non_differentiable(mask)
        return zdepth, mask, min_sdf, depth, normal

    @staticmethod
    def backward(ctx, g_zdepth, g_mask, g_min_sdf, g_depth, g_normal):
        engine, cfg, ws, B = ctx.engine, ctx.cfg, ctx.ws, ctx.B
        _check_generation(ctx)
        dev = engine.device

        def prep(g, n):
            if g is None or g.numel() != n:
                return None
            return g.to(dtype=torch.float32).contiguous()
        P = cfg.band_rows * cfg.W
        gz, gq = prep(g_zdepth, B * P), prep(g_min_sdf, B * P)
        gd, gn = (prep(g_depth, B * P), prep(g_normal, 3 * B * P)) if cfg.want_normal else (None, None)
        g_lat = torch.empty(B, 256, dtype=torch.float32, device=dev)
        g_R = torch.empty(B, 9, dtype=torch.float32, device=dev)
        g_T = torch.empty(B, 3, dtype=torch.float32, device=dev)
        ws_b = torch.empty(B * ctx.bwd_bytes, dtype=torch.uint8, device=dev)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_render_backward_batch(
            engine.ctx.h, C.byref(cfg), B, p(ws), ws.numel(), p(gz), p(gq), p(gd), p(gn), p(g_lat), p(g_R), p(g_T),
            p(ws_b), ws_b.numel(), engine.ctx.stream()))
        ls, rs, ts = ctx.shapes
        if ctx.shared:
            g_lat = g_lat.sum(0)                   # one shape code rendered from B cameras: the views' gradients add up (fixed order)
        return tuple(g.reshape(sh).to(device=d, dtype=dt) for g, sh, (d, dt) in zip((g_lat, g_R, g_T), (ls, rs, ts), ctx.in_meta)) + (None, None, None)


def render_batch_call(engine, cfg, latent, R, T, view_flags=None):
    """Batched render_call: R (B,3,3), T (B,3), latent (1,256) (shared) or (B,256)."""
    need_bwd = torch.is_grad_enabled() and any(getattr(t, 'requires_grad', False) for t in (latent, R, T))
    cfg = cfg.clone()
    cfg.save_for_backward = 1 if need_bwd else 0
    return RenderBatchFunction.apply(latent, R, T, engine, cfg, view_flags)


def render_normal_batch_call(engine, cfg, latent, R, T, zdepth, mask):
    """Batched render_normal_call: R (B,3,3), T (B,3), zdepth / mask (B,P) -> (B,3,P)."""
    dev = engine.device
    P = cfg.band_rows * cfg.W
    Rc, Tc = _f32c(R, dev).reshape(-1, 9), _f32c(T, dev).reshape(-1, 3)
    B = Rc.shape[0]
    lat = _f32c(latent, dev).reshape(-1, 256)
    if lat.shape[0] not in (1, B) or not (1 <= B <= binding.MAX_VIEWS):
        raise ValueError('expected latent (1,256) or (B,256), R (B,3,3), T (B,3) with 1 <= B <= %d' % binding.MAX_VIEWS)
    z = _f32c(zdepth, dev).reshape(B, P)
    m = mask.detach().to(device=dev).reshape(B, P).to(torch.uint8).contiguous()
    fwd_bytes, _ = engine.ctx.workspace_bytes(cfg)
    ws = torch.empty(B * fwd_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty(B, 3, P, dtype=torch.float32, device=dev)
    p = binding.ptr
    engine.ctx.check(engine.ctx.L.distr_render_normal_batch(engine.ctx.h, C.byref(cfg), B, p(lat), 0 if lat.shape[0] == 1 else 256, p(Rc), p(Tc),
                                                            p(z), p(m), p(out), p(ws), ws.numel(), engine.ctx.stream()))
    return out


HALO = 4   # rows; depth2normal needs 1, the 4x4 pyramid parents need the band aligned to 4


def band_cfg(cfg, r0, r1):
    """cfg of the launch that produces rows [r0, r1): the band plus its depth2normal halo. Returns (cfg, b0, b1)."""
    H = cfg.H
    if not (0 <= r0 < r1 <= H) or (r0 & 3) or ((r1 & 3) and r1 != H):
        raise ValueError('band [%d, %d) must lie in [0, %d) on multiples of 4 (or end at H)' % (r0, r1, H))
    halo = HALO if (cfg.want_normal and cfg.use_depth2normal) else 0
    b0, b1 = max(0, r0 - halo), min(H, r1 + halo)
    bcfg = cfg.clone()
    bcfg.row0, bcfg.rows = b0, b1 - b0
    return bcfg, b0, b1


def render_band_call(engine, cfg, latent, R, T, r0, r1):
    """Rows [r0, r1) of the render `cfg` describes (r0, r1 multiples of 4, or r1 == H): what one rank computes when one
    large view is split over several GPUs (row-band tiles whose height is a multiple of 4 px, halo
    recomputed locally for depth2normal). Renders the band plus a HALO-row halo and crops it; autograd through the crop
    zero-pads the halo's upstream gradients, so the sum over a partition of [0, H) of the bands' input gradients equals
    the full render's. Returns (zdepth (n*W), mask (n*W) uint8, min_sdf (n*W), depth (n, W), normal (n, W, 3)), n = r1 - r0;
    every value is bit-identical to the same pixel of the full render."""
    W = cfg.W
    bcfg, b0, b1 = band_cfg(cfg, r0, r1)
    z, mask, q, depth, normal = render_call(engine, bcfg, latent, R, T)
    lo, n = r0 - b0, r1 - r0
    z, mask, q = (t.reshape(b1 - b0, W)[lo:lo + n].reshape(-1) for t in (z, mask, q))
    if cfg.want_normal:
        depth, normal = depth[lo:lo + n], normal[lo:lo + n]
    return z, mask, q, depth, normal


def render_normal_call(engine, cfg, latent, R, T, zdepth, mask):
    """SDFRenderer.render_normal forward (renderer.py:880-910) -> (3, P). Gradient-free: for ReLU decoders the
    normalised SDF gradient is piecewise constant in (latent, point), its autograd contribution is identically ~0
    (SURVEY.md A.6-1)."""
    dev = engine.device
    P = cfg.band_rows * cfg.W
    lat, Rc, Tc = _f32c(latent, dev).reshape(-1), _f32c(R, dev).reshape(-1), _f32c(T, dev).reshape(-1)
    z = _f32c(zdepth, dev).reshape(-1)
    m = mask.detach().to(device=dev).reshape(-1).to(torch.uint8).contiguous()
    fwd_bytes, _ = engine.ctx.workspace_bytes(cfg)
    ws = torch.empty(fwd_bytes, dtype=torch.uint8, device=dev)
    out = torch.empty(3, P, dtype=torch.float32, device=dev)
    p = binding.ptr
    engine.ctx.check(engine.ctx.L.distr_render_normal(engine.ctx.h, C.byref(cfg), p(lat), p(Rc), p(Tc), p(z), p(m), p(out),
                                                      p(ws), ws.numel(), engine.ctx.stream()))
    return out


def mlp_eval(engine, latent, points, clamp_dist=None, arith='f32'):
    """decode_sdf forward (core/utils/decoder_utils.py:53-74): points (n,3) -> (n,1). arith='f32' (default): exact f32 MFMA, bit-identical
    to the oracle; 'bf16x6': six-product split-bf16 arithmetic (distr_mlp_eval_bf16x6: f32-equivalent accuracy, faster, not bit-identical);
    'f16x3': three-product split-f16 arithmetic (distr_mlp_eval_f16x3: the same for decoders inside the f16 range, faster again; NaN
    for a point whose activations leave that range)."""
    if arith not in binding.ARITH:
        raise ValueError("arith must be one of %s" % sorted(binding.ARITH))
    dev = engine.device
    lat = _f32c(latent, dev).reshape(-1)
    x = _f32c(points, dev).reshape(-1, 3)
    n = x.shape[0]
    out = torch.empty(n, 1, dtype=torch.float32, device=dev)
    ws = torch.empty(engine.ctx.L.distr_mlp_workspace_bytes(n), dtype=torch.uint8, device=dev)
    p = binding.ptr
    fn = {'f32': engine.ctx.L.distr_mlp_eval, 'bf16x6': engine.ctx.L.distr_mlp_eval_bf16x6, 'f16x3': engine.ctx.L.distr_mlp_eval_f16x3}[arith]
    engine.ctx.check(fn(engine.ctx.h, p(lat), p(x), n, -1.0 if clamp_dist is None else float(clamp_dist), p(out), p(ws), ws.numel(),
                        engine.ctx.stream()))
    return out


class DecodeSdfFunction(torch.autograd.Function):
    """decode_sdf with autograd (decoder_utils.py:53-74 called without no_grad): (latent (1,256), points (n,3)) -> (n,1);
    backward = distr_mlp_backward (one fused forward-recompute + dX chain per point, latent gradient from the delta sums)."""

    @staticmethod
    def forward(ctx, latent, points, engine, clamp_dist):
        out = mlp_eval(engine, latent, points, clamp_dist)
        ctx.engine, ctx.clamp = engine, clamp_dist
        ctx.save_for_backward(latent.detach(), points.detach())
        ctx.need = (latent.requires_grad, points.requires_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        engine = ctx.engine
        dev = engine.device
        latent, points = ctx.saved_tensors
        lat = _f32c(latent, dev).reshape(-1)
        x = _f32c(points, dev).reshape(-1, 3)
        n = x.shape[0]
        gs = _f32c(g, dev).reshape(-1)
        g_x = torch.empty(n, 3, dtype=torch.float32, device=dev) if ctx.need[1] else None
        g_l = torch.empty(256, dtype=torch.float32, device=dev) if ctx.need[0] else None
        ws = torch.empty(engine.ctx.L.distr_mlp_backward_workspace_bytes(n), dtype=torch.uint8, device=dev)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_mlp_backward(engine.ctx.h, p(lat), p(x), n, p(gs), -1.0 if ctx.clamp is None else float(ctx.clamp),
                                                        p(g_x), p(g_l), p(ws), ws.numel(), engine.ctx.stream()))
        return (None if g_l is None else g_l.reshape(latent.shape)), (None if g_x is None else g_x.reshape(points.shape)), None, None


def mlp_eval_autograd(engine, latent, points, clamp_dist=None):
    return DecodeSdfFunction.apply(latent, points, engine, clamp_dist)


def mlp_grad(engine, latent, points):
    """(sdf (n,), d sdf/d xyz (n,3)) of the unclamped decoder."""
    dev = engine.device
    lat = _f32c(latent, dev).reshape(-1)
    x = _f32c(points, dev).reshape(-1, 3)
    n = x.shape[0]
    sdf = torch.empty(n, dtype=torch.float32, device=dev)
    g = torch.empty(n, 3, dtype=torch.float32, device=dev)
    ws = torch.empty(engine.ctx.L.distr_mlp_workspace_bytes(n), dtype=torch.uint8, device=dev)
    p = binding.ptr
    engine.ctx.check(engine.ctx.L.distr_mlp_grad(engine.ctx.h, p(lat), p(x), n, p(sdf), p(g), p(ws), ws.numel(), engine.ctx.stream()))
    return sdf, g


def debug_mlp_layer(engine, latent, points, layer):
    """Test aid: post-activation of hidden layer `layer` -> (n,512)."""
    dev = engine.device
    lat = _f32c(latent, dev).reshape(-1)
    x = _f32c(points, dev).reshape(-1, 3)
    n = x.shape[0]
    out = torch.empty(n, 512, dtype=torch.float32, device=dev)
    ws = torch.empty(engine.ctx.L.distr_mlp_workspace_bytes(n), dtype=torch.uint8, device=dev)
    p = binding.ptr
    engine.ctx.check(engine.ctx.L.distr_debug_mlp_layer(engine.ctx.h, p(lat), p(x), n, int(layer), p(out), p(ws), ws.numel(),
                                                       engine.ctx.stream()))
    return out


def debug_tile_timing(engine, latent, points, tile):
    """Test aid: (sdf (n,), stamps (tiles, 20, 2) int64 [shader clock, 100 MHz wall clock]) of the decoder tile phases."""
    dev = engine.device
    lat = _f32c(latent, dev).reshape(-1)
    x = _f32c(points, dev).reshape(-1, 3)
    n = x.shape[0]
    tiles = (n + tile - 1) // tile
    sdf = torch.empty(n, dtype=torch.float32, device=dev)
    ts = torch.zeros(tiles, 20, 2, dtype=torch.int64, device=dev)
    ws = torch.empty(engine.ctx.L.distr_mlp_workspace_bytes(n), dtype=torch.uint8, device=dev)
    p = binding.ptr
    engine.ctx.check(engine.ctx.L.distr_debug_tile_timing(engine.ctx.h, p(lat), p(x), n, p(sdf), p(ts), p(ws), ws.numel(),
                                                         engine.ctx.stream()))
    return sdf, ts


# ------------------------------------------------------------------------------------------ fused losses (rows f2, f3)
def _u8c(t, device):
    t = t.detach().to(device=device)
    return (t if t.dtype == torch.uint8 else (t != 0).to(torch.uint8)).contiguous()


def _loss_ws(engine, H, W):
    return torch.empty(engine.ctx.L.distr_loss_workspace_bytes(H, W), dtype=torch.uint8, device=engine.device)


class SingleViewLossFunction(torch.autograd.Function):
    """(depth (H,W), normal (H,W,3), min_sdf (H,W)) + constants -> losses[4] = (mask_gt, mask_out, depth, normal):
    distr_single_loss_forward / _backward (core/utils/loss_utils.py:59-172 in two element-wise kernels)."""

    @staticmethod
    def forward(ctx, depth, normal, min_sdf, engine, mask, gt_depth, gt_normal, gt_mask, threshold):
        dev = engine.device
        H, W = min_sdf.shape[0], min_sdf.shape[1]
        d, n, q = _f32c(depth, dev), _f32c(normal, dev), _f32c(min_sdf, dev)
        m, gm = _u8c(mask, dev), _u8c(gt_mask, dev)
        gd = None if gt_depth is None else _f32c(gt_depth, dev)
        gn = None if gt_normal is None else _f32c(gt_normal, dev)
        out8 = torch.empty(8, dtype=torch.float32, device=dev)
        ws = _loss_ws(engine, H, W)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_single_loss_forward(engine.ctx.h, H, W, p(d), p(n), p(m), p(q), p(gd), p(gn), p(gm),
                                                               float(threshold), p(out8), p(ws), ws.numel(), engine.ctx.stream()))
        ctx.engine, ctx.hw, ctx.threshold = engine, (H, W), float(threshold)
        ctx.saved = (d, n, q, m, gd, gn, gm, out8)
        return out8[:4].clone()

    @staticmethod
    def backward(ctx, g):
        engine = ctx.engine
        d, n, q, m, gd, gn, gm, out8 = ctx.saved
        H, W = ctx.hw
        g4 = g.to(dtype=torch.float32).contiguous()
        g_d, g_n, g_q = torch.empty_like(d), torch.empty_like(n), torch.empty_like(q)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_single_loss_backward(engine.ctx.h, H, W, p(d), p(n), p(m), p(q), p(gd), p(gn), p(gm),
                                                                ctx.threshold, p(out8), p(g4), p(g_d), p(g_n), p(g_q),
                                                                engine.ctx.stream()))
        return g_d, g_n, g_q, None, None, None, None, None, None


def single_view_losses(engine, depth, normal, mask, min_sdf, gt_depth, gt_normal, gt_mask, threshold):
    """-> tensor [mask_gt, mask_out, depth, normal] (terms whose ground truth is None are 0)."""
    return SingleViewLossFunction.apply(depth, normal, min_sdf, engine, mask, gt_depth, gt_normal, gt_mask, threshold)


class WarpLossFunction(torch.autograd.Function):
    """(Zdepth1 (P), R1, T1, R2, T2) + constants -> (loss_color, keep (P) uint8, color1 (H,W,3), color2 (H,W,3)):
    distr_warp_loss_forward / _backward (core/sdfrenderer/renderer_warp.py:18-101)."""

    @staticmethod
    def forward(ctx, z1, R1, T1, R2, T2, engine, wcfg, m1, z2, img1, img2):
        dev = engine.device
        H, W = wcfg.H, wcfg.W
        P = H * W
        z1c, z2c = _f32c(z1, dev).reshape(-1), _f32c(z2, dev).reshape(-1)
        m1c = _u8c(m1, dev).reshape(-1)
        i1, i2 = _f32c(img1, dev).reshape(-1), _f32c(img2, dev).reshape(-1)
        cams = [_f32c(t, dev).reshape(-1) for t in (R1, T1, R2, T2)]
        if z1c.numel() != P or i1.numel() != 3 * P or i2.numel() != 3 * P:
            raise ValueError('render_warp: image / depth sizes do not match img_hw')
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        keep = torch.empty(P, dtype=torch.uint8, device=dev)
        c1, c2 = torch.empty(H, W, 3, dtype=torch.float32, device=dev), torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        ws = _loss_ws(engine, H, W)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_warp_loss_forward(engine.ctx.h, C.byref(wcfg), p(z1c), p(m1c), p(z2c), p(i1), p(i2),
                                                             p(cams[0]), p(cams[1]), p(cams[2]), p(cams[3]), p(out3), p(keep),
                                                             p(c1), p(c2), p(ws), ws.numel(), engine.ctx.stream()))
        ctx.engine, ctx.wcfg = engine, wcfg
        ctx.saved = (z1c, m1c, z2c, i1, i2, cams, out3)
        ctx.shapes = (z1.shape, R1.shape, T1.shape, R2.shape, T2.shape)
        ctx.mark_non_differentiable(keep, c1, c2)
        return out3[0].clone(), keep, c1, c2

    @staticmethod
    def backward(ctx, g_loss, g_keep, g_c1, g_c2):
        engine, wcfg = ctx.engine, ctx.wcfg
        z1c, m1c, z2c, i1, i2, cams, out3 = ctx.saved
        dev = engine.device
        gl = g_loss.to(dtype=torch.float32).reshape(1).contiguous()
        g_z = torch.empty_like(z1c)
        g_cam = torch.empty(24, dtype=torch.float32, device=dev)
        ws = _loss_ws(engine, wcfg.H, wcfg.W)
        p = binding.ptr
        engine.ctx.check(engine.ctx.L.distr_warp_loss_backward(engine.ctx.h, C.byref(wcfg), p(z1c), p(m1c), p(z2c), p(i1), p(i2),
                                                              p(cams[0]), p(cams[1]), p(cams[2]), p(cams[3]), p(out3), p(gl),
                                                              p(g_z), p(g_cam), p(ws), ws.numel(), engine.ctx.stream()))
        zs, r1s, t1s, r2s, t2s = ctx.shapes
        return (g_z.reshape(zs), g_cam[0:9].reshape(r1s), g_cam[9:12].reshape(t1s), g_cam[12:21].reshape(r2s),
                g_cam[21:24].reshape(t2s), None, None, None, None, None, None)


def warp_loss(engine, wcfg, z1, m1, z2, img1, img2, R1, T1, R2, T2):
    return WarpLossFunction.apply(z1, R1, T1, R2, T2, engine, wcfg, m1, z2, img1, img2)
