"""Persistent tail launch (k_tail) against the launch-per-step path (not a pytest file): for a list of render configurations, renders
fwd + bwd with DISTR_TAIL=0 and with the tail launch starting at several steps, checks that every output byte and every gradient bit is
the same, and times 20 back-to-back forwards of each.

    python tests/gpu_diag_tail.py [--quick] [--out gpurun_out/tail.md]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

KEYS = ('zdepth', 'mask', 'min_sdf', 'depth', 'normal', 'g_latent', 'g_R', 'g_T')


def engine_with(env):
    """A fresh context created under `env` (the knobs are read by distr_create)."""
    from distr import fixture, functions
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        Ws, bs, latent = fixture.make_decoder_weights()
        eng = functions.engine_from_weights(Ws, bs, 0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return eng, latent


def time_forward(eng, cfg, latent, R, T, reps=20):
    import ctypes as C
    from distr import binding
    dev = eng.device
    P = cfg.band_rows * cfg.W
    fwd_bytes, _ = eng.ctx.workspace_bytes(cfg)
    ws = torch.empty(fwd_bytes, dtype=torch.uint8, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(-1)).to(dev)
    lat, Rt, Tt = t(latent), t(R), t(T)
    outs = [torch.empty(P, device=dev), torch.empty(P, dtype=torch.uint8, device=dev), torch.empty(P, device=dev), torch.empty(P, device=dev),
            torch.empty(3 * P, device=dev)]
    p = binding.ptr

    def fwd():
        eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]),
                                                   p(outs[4]), p(ws), ws.numel(), eng.ctx.stream()))
    for _ in range(4):
        fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fwd()
    e1.record()
    torch.cuda.synchronize()
    st = eng.ctx.render_stats(cfg, ws)
    return e0.elapsed_time(e1) / reps, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    import helpers
    from distr import binding, fixture
    cases = [(64, 20, 'pyramid_recursive', 3), (137, 100, 'pyramid_recursive', 3), (64, 100, 'recursive', 1), (137, 100, 'recursive', 5)]
    if not args.quick:
        cases += [(224, 100, 'pyramid_recursive', 3), (256, 50, 'pyramid_recursive', 3), (50, 70, 'pyramid_recursive', 3)]
    variants = [('off', {'DISTR_TAIL': 0}), ('default', {}), ('from0', {'DISTR_TAIL_FROM': 0}), ('from3', {'DISTR_TAIL_FROM': 3}), ('from9', {'DISTR_TAIL_FROM': 9}),
                ('from0-nocluster', {'DISTR_TAIL_FROM': 0, 'DISTR_CLUSTER': 0}), ('from0-nosticky', {'DISTR_TAIL_FROM': 0, 'DISTR_STICKY': 0})]
    engines = {name: engine_with(env) for name, env in variants}
    lines = ['# persistent tail launch vs one launch per step (all outputs and gradients compared byte for byte)', '',
             '| render | variant | identical | forward ms | launches | tail_from | steals | fallbacks |', '|---|---|---|---|---|---|---|---|']
    ok = True
    for (size, steps, marcher, bs) in cases:
        H, W = (size, size) if size != 50 else (50, 70)
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(30, 20, 1.6, 10)
        kw = dict(march_step=steps, buffer_size=bs, marcher=marcher, use_depth2normal=True, ratio=1.5)
        ref = None
        for name, _ in variants:
            eng, latent = engines[name]
            # twice: the second render of a configuration is the one that may use the previous render's tail hint
            a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
            a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
            cfg = binding.make_cfg((H, W), K, **kw)
            ms, st = time_forward(eng, cfg, latent, R, T)
            same = True
            if ref is None:
                ref = a
            else:
                for k in KEYS:
                    if not np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(ref[k]).view(np.uint8)):
                        same = False
                        d = np.abs(np.asarray(a[k], np.float64) - np.asarray(ref[k], np.float64))
                        print('MISMATCH %dx%d/%d %s %s: %s max %.3e, %d elements' % (H, W, steps, marcher, name, k, d.max(), int((d > 0).sum())))
            ok = ok and same
            lines.append('| %dx%d / %d / %s / bs %d | %s | %s | %.3f | %d | %d | %d | %d |' % (H, W, steps, marcher, bs, name, 'yes' if same else '**NO**', ms,
                         st['num_march_launches'], st['tail_from'], st['tail_steals'], st['cluster_fallbacks']))
    text = '\n'.join(lines) + '\n'
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, 'w').write(text)
    sys.exit(0 if ok else 1)


if __name__ == '__main__':
    main()
