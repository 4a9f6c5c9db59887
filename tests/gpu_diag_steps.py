"""Per-launch table of the march kernels of ONE forward (not a pytest file): for every march launch of the configured render --
rays evaluated, kernel microseconds (hipEvent bracket on the launch stream), TFLOP/s and fraction of the f32-MFMA peak,
work-proportional time at the dense rate, and which tile class the split rule (fine_range) picks.

    python tests/gpu_diag_steps.py [--size 512] [--march-step 50] [--view 0] [--out gpurun_out/steps.md]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch

FLOP = 3146752
PEAK = 157.3


def tile_class(n, t16=4096, t32=8192):
    full, rem = (n // 16384) * 16384, n % 16384
    if n == 0:
        return 'empty'
    small = '' if rem == 0 else ('16c8' if rem <= 512 else '16c4' if rem <= 1024 else '16c2' if rem <= 2048 else '16' if rem <= t16 else '32' if rem <= t32 else '32+16' if rem <= t32 + t16 else '64')
    return ('%dx64r' % (full // 16384) if full else '') + ('+' if full and small else '') + small


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--march-step', type=int, default=50)
    ap.add_argument('--view', type=int, default=0)
    ap.add_argument('--marcher', default='pyramid_recursive')
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--out', default=None)
    ap.add_argument('--arith', default='f32')
    ap.add_argument('--no-save', action='store_true', help='forward only: no ReLU masks saved for the backward (distr_render_cfg.save_for_backward = 0)')
    args = ap.parse_args()
    import ctypes as C
    import bench
    from distr import binding, fixture, functions
    Ws, bs, latent = fixture.make_decoder_weights()
    eng = functions.engine_from_weights(Ws, bs, 0)
    H = W = args.size
    K = fixture.make_intrinsic(H, W)
    R, T = bench.view_camera(fixture, args.view)
    cfg = binding.make_cfg((H, W), K, march_step=args.march_step, buffer_size=3, ratio=1.5, marcher=args.marcher, use_depth2normal=True, arith=args.arith)
    if args.no_save:
        cfg.save_for_backward = 0
    dev = eng.device
    P = H * W
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
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fwd()
    e1.record()
    torch.cuda.synchronize()
    wall_ms = e0.elapsed_time(e1) / 20
    st0 = eng.ctx.render_stats(cfg, ws)
    times = []
    for _ in range(args.reps):
        eng.ctx.profile_enable(True)
        fwd()
        times.append(eng.ctx.profile_read_list())
        eng.ctx.profile_read()
    eng.ctx.profile_enable(False)
    counts = eng.ctx.live_counts(cfg, ws)
    ms = np.median(np.array(times), axis=0)
    ncoarse = 6 if args.marcher == 'pyramid_recursive' else 0
    st1 = eng.ctx.render_stats(cfg, ws)
    tail_from = int(st1['tail_from'])
    tail_steps = 0
    if len(ms) != len(counts):
        # the persistent tail launch: ONE launch for every full-resolution step from tail_from on (its row sums their evaluations)
        assert len(ms) == ncoarse + tail_from + 1, (len(ms), len(counts), tail_from)
        tail_steps = sum(1 for c in counts[ncoarse + tail_from:] if c > 0)
        counts = list(counts[:ncoarse + tail_from]) + [int(sum(counts[ncoarse + tail_from:]))]
    dense_rate = None
    lines = ['# march launches of one forward: %dx%d, %d steps, %s, view %d (median of %d forwards)' % (H, W, args.march_step, args.marcher, args.view, args.reps), '',
             '| launch | rays evaluated | us | TFLOP/s | frac of %.1f | tile class | us at the dense rate |' % PEAK, '|---|---|---|---|---|---|---|']
    # dense rate = best whole-rounds launch
    for n, m in zip(counts, ms):
        if n >= 65536:
            r = FLOP * n / (m * 1e-3) / 1e12
            dense_rate = r if dense_rate is None else max(dense_rate, r)
    tot_us = tot_prop = 0.0
    classes = {}
    for i, (n, m) in enumerate(zip(counts, ms)):
        us = m * 1e3
        tf = FLOP * n / (m * 1e-3) / 1e12 if m > 0 else 0.0
        prop = FLOP * n / (dense_rate * 1e12) * 1e6 if dense_rate else 0.0
        name = ('coarse %d' % i) if i < ncoarse else ('step %d' % (i - ncoarse))
        cls = tile_class(n) if i >= ncoarse else 'coarse'
        if tail_steps and i == len(ms) - 1:
            name, cls = 'tail launch: steps %d..%d (%d with live rays: %.1f us per step)' % (tail_from, tail_from + tail_steps - 1, tail_steps, us / max(tail_steps, 1)), 'tail'
        lines.append('| %s | %d | %.1f | %.1f | %.3f | %s | %.1f |' % (name, n, us, tf, tf / PEAK, cls, prop))
        tot_us += us
        tot_prop += prop
        key = 'coarse' if i < ncoarse else ('rounds' if n >= 16384 else cls)
        c = classes.setdefault(key, [0, 0.0, 0])
        c[0] += 1; c[1] += us; c[2] += n
    lines += ['', 'whole forward, 20 back-to-back without brackets: %.3f ms each; cluster fallbacks %d; march launches %d (tail launch from step %d; tiles taken over: %d)'
              % (wall_ms, st0['cluster_fallbacks'], st1['num_march_launches'], tail_from, st1['tail_steals'])]
    lines += ['', 'total: %d evaluations, %.2f ms in march kernels = %.1f TFLOP/s = %.3f of peak; at the dense rate (%.1f TFLOP/s) the same evaluations '
              'take %.2f ms' % (sum(counts), tot_us / 1e3, FLOP * sum(counts) / (tot_us * 1e-6) / 1e12, FLOP * sum(counts) / (tot_us * 1e-6) / 1e12 / PEAK,
                               dense_rate or 0.0, tot_prop / 1e3), '',
              '| launch class | launches | total us | evaluations | TFLOP/s | frac |', '|---|---|---|---|---|---|']
    for k, (cnt, us, n) in classes.items():
        tf = FLOP * n / (us * 1e-6) / 1e12 if us > 0 else 0.0
        lines.append('| %s | %d | %.0f | %d | %.1f | %.3f |' % (k, cnt, us, n, tf, tf / PEAK))
    text = '\n'.join(lines) + '\n'
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        open(args.out, 'w').write(text)


if __name__ == '__main__':
    main()
