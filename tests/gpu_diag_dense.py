"""Dense-rate probe (not a pytest file): distr_mlp_eval / distr_mlp_grad on N points, median kernel time over reps, TFLOP/s and
fraction of the f32-MFMA peak; optional in-kernel phase stamps of the 64-ray tile.  python tests/gpu_diag_dense.py [--n 262144]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=262144)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--stamps', action='store_true')
    args = ap.parse_args()
    from distr import fixture, functions
    Ws, bs, latent = fixture.make_decoder_weights()
    eng = functions.engine_from_weights(Ws, bs, 0)
    rs = np.random.RandomState(3)
    pts = torch.from_numpy((rs.rand(args.n, 3) * 1.6 - 0.8).astype(np.float32)).cuda()
    lat = torch.from_numpy(latent).cuda()

    def timeit(fn):
        ts = []
        for _ in range(args.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts)), float(np.min(ts))
    for _ in range(3):
        functions.mlp_eval(eng, lat, pts)
    med, mn = timeit(lambda: functions.mlp_eval(eng, lat, pts))
    print('distr_mlp_eval  n=%d: median %.3f ms (min %.3f) = %.1f TFLOP/s = %.3f of peak' % (args.n, med, mn, FLOP * args.n / med / 1e9, FLOP * args.n / med / 1e9 / PEAK))
    a = functions.mlp_eval(eng, lat, pts)
    for mode, peak_name in (('bf16x6', 'bf16'), ('f16x3', 'f16')):
        for _ in range(3):
            functions.mlp_eval(eng, lat, pts, arith=mode)
        b = functions.mlp_eval(eng, lat, pts, arith=mode)
        d = (a - b).abs()
        medb, mnb = timeit(lambda: functions.mlp_eval(eng, lat, pts, arith=mode))
        print('distr_mlp_eval_%s n=%d: median %.3f ms (min %.3f) = %.1f TFLOP/s-equivalent (algorithmic FLOP counted once; %.3f of the 2500 TFLOP/s '
              '%s peak, %.2f x the f32 kernel); max |sdf - sdf_f32| = %.3e (mean %.3e, non-finite %d) over %d points with sdf in [%.3f, %.3f]'
              % (mode, args.n, medb, mnb, FLOP * args.n / medb / 1e9, FLOP * args.n / medb / 1e9 / 2500.0, peak_name, med / medb, float(d[torch.isfinite(d)].max()),
                 float(d[torch.isfinite(d)].mean()), int((~torch.isfinite(b)).sum()), args.n, float(a.min()), float(a.max())))
    med, mn = timeit(lambda: functions.mlp_grad(eng, lat, pts))
    print('distr_mlp_grad  n=%d: median %.3f ms (min %.3f) = %.1f TFLOP/s = %.3f of peak' % (args.n, med, mn, 2 * FLOP * args.n / med / 1e9, 2 * FLOP * args.n / med / 1e9 / PEAK))
    if args.stamps:
        sdf, ts = functions.debug_tile_timing(eng, lat, pts[:256 * 64 * 4], 64)
        ts = ts.cpu().numpy().astype(np.int64)
        cyc = ts[:, :, 0]
        wall = ts[:, :, 1]
        d = np.median(cyc[:, 1:19] - cyc[:, 0:18], axis=0)
        tot_c = np.median(cyc[:, 18] - cyc[:, 0]); tot_w = np.median(wall[:, 18] - wall[:, 0])
        print('tile total %.0f cycles = %.1f us (clock %.3f GHz)' % (tot_c, tot_w / 100.0, tot_c / (tot_w * 10.0) / 1e0 / 1e0 if tot_w else 0))
        names = ['L0 mfma', 'L0 wb', 'L1 mfma', 'L1 wb', 'L2 mfma', 'L2 wb', 'L3 mfma', 'L3 wb', 'L4 mfma', 'L4 wb', 'L5 mfma', 'L5 wb', 'L6 mfma', 'L6 wb',
                 'L7 mfma', 'L7 wb', 'lin8', 'end']
        print(' '.join('%s:%.0f' % (nm, v) for nm, v in zip(names, d)))


if __name__ == '__main__':
    main()
