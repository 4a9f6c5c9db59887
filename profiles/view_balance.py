"""Per-view step times of the eight C4 cameras (bench.py --view-offset v on ONE GPU, written by run_round.sh as view_<v>.json)
-> the row-band plan `bench.py --gpus N` derives from such times at N = 2, 4, 8 (distr.parallel.balance_views) and the estimated
step time with / without it, with UNIFORM row costs (the plan bench.py really uses weighs the rows by the rendered mask and is
checked rank by rank in plan_check.py). An estimate from single-GPU measurements: no multi-GPU run is behind these numbers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
from distr import parallel   # noqa: E402  (pure host logic, no GPU needed)

out = sys.argv[1]
ms = []
for v in range(8):
    with open(os.path.join(out, 'view_%d.json' % v)) as f:
        ms.append(json.loads(f.read().strip().splitlines()[-1])['ms_per_step'])
H = 512
print('# per-view step time (512x512, 50 steps, fwd+loss+bwd, one MI355X), ms\n')
print('| view | ' + ' | '.join(str(v) for v in range(8)) + ' |')
print('|---|' + '---|' * 8)
print('| ms | ' + ' | '.join('%.2f' % m for m in ms) + ' |\n')
print('| N | slowest view (no balancing) | ideal (mean) | plan (view, rows) handed over | estimated slowest rank with the plan | N x view-0 rate kept |')
print('|---|---|---|---|---|---|')
for N in (2, 4, 8):
    t = ms[:N]
    plan = parallel.balance_views(t, H)
    est = []
    for r in range(N):
        e = 0.0
        for i, (v, r0, r1) in enumerate(plan[r]):
            rows = (r1 - r0) + (8 if i > 0 else 0)          # a received band also pays its depth2normal halo
            e += t[v] * rows / H
        est.append(e)
    moved = ['rank %d <- view %d rows [%d, %d)' % (r, v, r0, r1) for r in range(N) for (v, r0, r1) in plan[r][1:]]
    print('| %d | %.2f | %.2f | %s | %.2f | %.3f -> %.3f |' % (N, max(t), sum(t) / N, '; '.join(moved) or 'none', max(est),
                                                          ms[0] / max(t), ms[0] / max(est)))
