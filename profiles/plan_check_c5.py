"""Single-GPU EMULATION of the C5 strong-scaling partition (BASELINE config 5: 4 shapes x 1024x1024 x 100 march steps split over N GPUs in row
bands) -- NOT a scaling measurement: every rank's pieces of the N = 2 / 4 / 8 partition are rendered ALONE on this one GPU (render + dense
loss + backward, the step of `bench.py --workload c5`), one rank after the other. What it shows is how evenly a partition spreads the
COST: predicted efficiency = (N = 1 step / N) / slowest rank. It cannot see the all-reduce or a node's fabric.

Per N: the cost-blind partition (distr.parallel.shard_rows: equal row-unit runs) and up to bench.ROW_FEEDBACK_ROUNDS cost-weighted ones
(shard_rows_plan with the row profile of the rendered masks of every shape -- surface pixel = 1, background pixel = BG_WEIGHT, a fixed
cost per opened band -- each scaled by the measured / predicted share of every rank under the previous cut, refine_row_weights): what
`bench.py --workload c5 --gpus N` does in its calibration before its second timed region.

    python profiles/plan_check_c5.py gpurun_out/r05_plan [--steps 3] [--n 2,4,8]      -> r05_plan_check_c5_n{2,4,8}.md in that directory
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from core.inv_optimizer.optimize_multi import _StreamPool  # noqa: E402
from distr import binding, fixture, functions, parallel  # noqa: E402

H = W = 1024
STEPS_MARCH = 100
N_SHAPES = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('out')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--n', default='2,4,8')
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    dev = torch.device('cuda', 0)
    Ws, bs, latent0 = fixture.make_decoder_weights()
    eng = functions.engine_from_weights(Ws, bs, 0)
    cfg = binding.make_cfg((H, W), fixture.make_intrinsic(H, W), march_step=STEPS_MARCH, buffer_size=bench.BUFFER_SIZE, ratio=bench.RATIO,
                           marcher='pyramid_recursive', use_depth2normal=True)
    lats = [torch.from_numpy(l).to(dev).requires_grad_(True) for l in [latent0] + [fixture.make_latent(1234 + i) for i in range(1, N_SHAPES)]]
    R, T = bench.view_camera(fixture, 0)
    Rt, Tt = torch.from_numpy(R).to(dev).requires_grad_(True), torch.from_numpy(T).to(dev).requires_grad_(True)
    rs = np.random.RandomState(5)
    wd, wq, wn = (torch.from_numpy(rs.rand(*s).astype(np.float32)).to(dev) for s in ((H, W), (H, W), (H, W, 3)))

    def image_loss(outs, r0, r1):
        z, mask, q, depth, normal = outs
        mb = mask.reshape(r1 - r0, W).bool()
        return torch.where(mb, depth * wd[r0:r1], torch.zeros_like(depth)).sum() + (q.reshape(r1 - r0, W) * wq[r0:r1]).sum() + (normal * wn[r0:r1]).sum()

    masks = {}

    def step(items):
        """the step of bench.py --workload c5 for one rank's pieces [(shape, r0, r1)]"""
        for t in lats + [Rt, Tt]:
            t.grad = None
        whole = [it for it in items if (it[1], it[2]) == (0, H)]
        rest = [it for it in items if (it[1], it[2]) != (0, H)]
        losses = []
        if len(whole) >= 2:
            outs = functions.render_batch_call(eng, cfg, torch.cat([lats[s] for (s, _, _) in whole], 0), torch.stack([Rt] * len(whole)), torch.stack([Tt] * len(whole)))
            for b, (s, _, _) in enumerate(whole):
                masks[s] = outs[1][b]
                losses.append(image_loss(tuple(o[b] for o in outs), 0, H))
        else:
            rest = whole + rest
        pool = _StreamPool(min(len(rest), 4) if len(rest) > 1 else 0, dev)

        def one(s, r0, r1):
            o = functions.render_call(eng, cfg, lats[s], Rt, Tt) if (r0, r1) == (0, H) else functions.render_band_call(eng, cfg, lats[s], Rt, Tt, r0, r1)
            if (r0, r1) == (0, H):
                masks[s] = o[1]
            return image_loss(o, r0, r1)
        more = [pool.run(i, lambda it=it: one(*it)) for i, it in enumerate(rest)]
        pool.join(more)
        losses += more
        tot = losses[0]
        for L in losses[1:]:
            tot = tot + L
        tot.backward()

    def time_items(items):
        step(items)                      # warm-up (allocator, band shapes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(items)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / args.steps

    all_items = [(s, 0, H) for s in range(N_SHAPES)]
    n1 = time_items(all_items)
    counts = [masks[s].reshape(H, W).float().sum(1).reshape(-1, 4).sum(1).cpu().numpy().tolist() for s in range(N_SHAPES)]
    weights0 = [parallel.row_weights_from_counts(c, W, 4, H) for c in counts]
    print('N = 1: %.1f ms per step (4 shapes as one batch)' % n1)

    for N in [int(x) for x in args.n.split(',')]:
        lines = ['# C5 partition check, N = %d -- single-GPU EMULATION, not a scaling measurement' % N, '',
                 'Every rank\'s pieces of the N = %d partition of BASELINE config 5 (4 shapes x 1024x1024 x 100 march steps, dense loss, fwd + bwd) rendered' % N,
                 'ALONE on one MI355X, one rank after the other (`profiles/plan_check_c5.py`, %d timed steps per rank after one warm-up). N = 1 (the four' % args.steps,
                 'shapes as one batched launch sequence): **%.1f ms per step**. Predicted efficiency = (N = 1 / N) / slowest rank: what the partition' % n1,
                 'alone would cost on N GPUs -- the all-reduce (one ~4 KiB packed buffer) and the node are not in it.', '']
        fixed = bench.ROW_BAND_FIXED * sum(sum(w) for w in weights0) / N_SHAPES
        summary = []
        weights = None
        plan = [parallel.shard_rows(N_SHAPES, H, r, N) for r in range(N)]
        title = 'cost-blind (`shard_rows`: equal row-unit runs)'
        seen = []
        for rnd in range(bench.ROW_FEEDBACK_ROUNDS + 1):
            loads = [time_items(pieces) for pieces in plan]
            lines += ['## %s' % title, '', '| rank | pieces (shape: rows) | rows | measured ms |', '|---|---|---|---|']
            for r, pieces in enumerate(plan):
                lines.append('| %d | %s | %d | %.1f |' % (r, '; '.join('%d: [%d, %d)' % p_ for p_ in pieces), sum(p_[2] - p_[1] for p_ in pieces), loads[r]))
            slow, mean = max(loads), sum(loads) / N
            lines += ['', 'slowest rank %.1f ms, mean %.1f ms, slowest / mean %.3f; predicted efficiency (%.1f / %d) / %.1f = **%.3f**' % (
                slow, mean, slow / mean, n1, N, slow, (n1 / N) / slow), '']
            summary.append((title.split(' (')[0], slow, mean, (n1 / N) / slow))
            seen.append(plan)
            # what bench.py --workload c5 --gpus N does in its calibration: weights from the rendered masks, scaled by measured / predicted share
            weights = parallel.refine_row_weights(weights0 if weights is None else weights, plan, loads, H)
            plan = parallel.shard_rows_plan(N_SHAPES, H, N, 4, weights, fixed)
            title = 'cost-weighted, feedback round %d (`shard_rows_plan`: row profile of the rendered masks -- surface pixel 1, background %.3f, %.0f %% of an image per opened band -- scaled by measured / predicted share of every rank, `refine_row_weights`)' % (
                rnd + 1, parallel.BG_WEIGHT, 100 * bench.ROW_BAND_FIXED)
            if plan in seen:
                break
        best = min(summary, key=lambda t: t[1])
        lines += ['`bench.py --workload c5 --gpus %d` keeps the cut with the fastest slowest rank among these (here: %s, predicted efficiency %.3f) and times it' % (N, best[0], best[3]),
                  'next to the cost-blind one; `value` switches only when it wins by more than %.0f %%.' % (100 * bench.BALANCE_MARGIN), '']
        lines += ['## summary', '', '| partition | slowest rank ms | mean ms | slowest / mean | predicted efficiency |', '|---|---|---|---|---|']
        for (t, slow, mean, eff) in summary:
            lines.append('| %s | %.1f | %.1f | %.3f | %.3f |' % (t, slow, mean, slow / mean, eff))
        lines += ['', 'Sum of the ranks\' times vs N = 1: the bands of a partition cost more in total than the four whole images as one batch (every band pays',
                  'its own latency-bound march tail and a 4-row depth2normal halo, and a rank\'s pieces do not share launches): mean x N = %.1f ms (cost-blind) against %.1f ms.' % (
                      summary[0][2] * N, n1), '']
        path = os.path.join(args.out, 'r05_plan_check_c5_n%d.md' % N)
        open(path, 'w').write('\n'.join(lines))
        print('\n'.join(lines[-12:]))


if __name__ == '__main__':
    main()
