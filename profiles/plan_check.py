"""Single-GPU check of the view balancer's cost model (run on the GPU box): renders the eight C4 views once for their masks, builds
the plan distr.parallel.balance_views derives from the per-view step times given on the command line (view_<v>.json of
run_round.sh, or measured here with --measure), and times every rank's work items of that plan through `bench.py --items`.
    python profiles/plan_check.py gpurun_out/r02_final [N = 8]            -> every rank's measured ms under the first and the refined plan"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from distr import binding, fixture, functions, parallel  # noqa: E402


def run_items(items):
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu-baseline', '--steps', '10', '--warmup', '3', '--items',
           ','.join('%d:%d:%d' % it for it in items)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])['ms_per_step']


def main():
    H = W = 512
    src = sys.argv[1]
    times = [json.loads(open(os.path.join(src, 'view_%d.json' % v)).read().strip().splitlines()[-1])['ms_per_step'] for v in range(8)]
    Ws, bs, latent = fixture.make_decoder_weights()
    eng = functions.engine_from_weights(Ws, bs, 0)
    cfg = binding.make_cfg((H, W), fixture.make_intrinsic(H, W), march_step=50, buffer_size=3, ratio=1.5, use_depth2normal=True)
    lat = torch.from_numpy(latent).cuda()
    profs = []
    for v in range(8):
        R, T = bench.view_camera(fixture, v)
        with torch.no_grad():
            _, mask, _, _, _ = functions.render_call(eng, cfg, lat, torch.from_numpy(R).cuda(), torch.from_numpy(T).cuda())
        profs.append(parallel.row_profile(mask.reshape(H, W).cpu().numpy()))
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    print('## N = %d' % N)
    plan = parallel.balance_views(times[:N], H, profs[:N])
    for title in ('first plan (row cost profile from the rendered masks)', 'refined plan (one feedback step: distr.parallel.refine_profiles with the times measured under the first plan)'):
        print('\n### %s\n' % title)
        print('| rank | items (view, rows) | measured ms |')
        print('|---|---|---|')
        loads = []
        for r in range(N):
            whole = len(plan[r]) == 1 and plan[r][0] == (r, 0, H)
            loads.append(times[r] if whole else run_items(plan[r]))
            print('| %d | %s | %.2f |' % (r, 'whole view' if whole else '; '.join('view %d rows [%d, %d)' % it for it in plan[r]), loads[-1]))
        print('\nslowest rank %.2f ms (no balancing: %.2f ms, mean of the views: %.2f ms); %d x view-0 rate kept: %.3f (no balancing: %.3f)' % (
            max(loads), max(times[:N]), sum(times[:N]) / N, N, times[0] / max(loads), times[0] / max(times[:N])))
        profs = parallel.refine_profiles(profs[:N], plan, times[:N], loads, H)
        plan = parallel.balance_views(times[:N], H, profs)


if __name__ == '__main__':
    main()
