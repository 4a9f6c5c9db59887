"""Stress run of the cluster tiles under oversubscription (not a pytest file; tests/test_gpu_parity.py runs it in a subprocess
with GPU_MAX_HW_QUEUES=8): NSTREAM HIP streams through ONE context, every iteration issues two 512x512 renders (dense steps: one
workgroup per compute unit, all 256 CUs busy) and six 64x64 / 96x96 renders whose march is almost all cluster tiles, round-robin
over the streams, forward (+ backward on every other small render). Cluster workgroups of different launches compete for the
compute units, so clusters regularly fail to assemble within their budget -- the lead workgroup then evaluates the tile alone.
Every output of every iteration must be bit-identical to the same render issued alone on an idle GPU.

    python tests/gpu_stress_clusters.py [--iters 200] [--streams 8]     -> prints one JSON line
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--streams', type=int, default=8)
    args = ap.parse_args()
    from distr import binding, fixture, functions
    Ws, bs, latent = fixture.make_decoder_weights()
    eng = functions.engine_from_weights(Ws, bs, 0)
    dev = eng.device
    lat = torch.from_numpy(latent).to(dev).reshape(-1)
    p = binding.ptr
    jobs = []          # (cfg, R, T, with_backward)
    for i in range(2):
        K = fixture.make_intrinsic(512, 512)
        cfg = binding.make_cfg((512, 512), K, march_step=50, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True)
        cfg.save_for_backward = 0
        R, T = fixture.make_camera(40.0 * i, 20.0, 1.6, 0.0)
        jobs.append((cfg, R, T, False))
    for i in range(6):
        s = 64 if i % 2 == 0 else 96
        K = fixture.make_intrinsic(s, s)
        cfg = binding.make_cfg((s, s), K, march_step=60, buffer_size=2, marcher='recursive' if i % 3 else 'pyramid_recursive',
                               use_depth2normal=bool(i & 1))
        bwd = (i % 2 == 0)
        cfg.save_for_backward = 1 if bwd else 0
        R, T = fixture.make_camera(25.0 * i, 5.0 + 3 * i, 1.6, 0.0)
        jobs.append((cfg, R, T, bwd))
    Rs = [torch.from_numpy(R).to(dev).reshape(-1) for _, R, _, _ in jobs]
    Ts = [torch.from_numpy(T).to(dev) for _, _, T, _ in jobs]

    def run(j, stream):
        cfg, _, _, bwd = jobs[j]
        P = cfg.H * cfg.W
        fwd_b, bwd_b = eng.ctx.workspace_bytes(cfg)
        ws = torch.empty(fwd_b, dtype=torch.uint8, device=dev)
        o = [torch.empty(P, device=dev), torch.empty(P, dtype=torch.uint8, device=dev), torch.empty(P, device=dev), torch.empty(P, device=dev),
             torch.empty(3 * P, device=dev)]
        st = C.c_void_p(stream.cuda_stream)
        eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat), p(Rs[j]), p(Ts[j]), p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(o[4]),
                                                   p(ws), ws.numel(), st))
        g = None
        if bwd:
            wsb = torch.empty(bwd_b, dtype=torch.uint8, device=dev)
            g = [torch.empty(256, device=dev), torch.empty(9, device=dev), torch.empty(3, device=dev)]
            gq = gws[j]
            eng.ctx.check(eng.ctx.L.distr_render_backward(eng.ctx.h, C.byref(cfg), p(ws), ws.numel(), None, p(gq), None, None, p(g[0]), p(g[1]), p(g[2]),
                                                        p(wsb), wsb.numel(), st))
            o = o + g
        return o, ws
    gws = [torch.from_numpy(np.random.RandomState(j).rand(c.H * c.W).astype(np.float32)).to(dev) for j, (c, _, _, _) in enumerate(jobs)]
    main_s = torch.cuda.current_stream()
    ref = []
    for j in range(len(jobs)):
        o, ws = run(j, main_s)
        torch.cuda.synchronize()
        assert eng.ctx.render_stats(jobs[j][0], ws)['cluster_fallbacks'] == 0, 'fallback on an idle GPU'
        ref.append([t.cpu().numpy().tobytes() for t in o])
    streams = [torch.cuda.Stream() for _ in range(args.streams)]
    for s in streams:
        s.wait_stream(main_s)
    fallbacks = mismatches = renders = 0
    t0 = time.time()
    for it in range(args.iters):
        outs = []
        for j in range(len(jobs)):
            s = streams[(j + it) % len(streams)]
            with torch.cuda.stream(s):
                outs.append(run(j, s))
        torch.cuda.synchronize()
        for j, (o, ws) in enumerate(outs):
            renders += 1
            fallbacks += eng.ctx.render_stats(jobs[j][0], ws)['cluster_fallbacks']
            got = [t.cpu().numpy().tobytes() for t in o]
            if got != ref[j]:
                mismatches += 1
    print(json.dumps({'iters': args.iters, 'streams': args.streams, 'renders': renders, 'mismatching_renders': mismatches,
                      'cluster_fallbacks': int(fallbacks), 'seconds': time.time() - t0,
                      'GPU_MAX_HW_QUEUES': os.environ.get('GPU_MAX_HW_QUEUES')}))
    return 0 if mismatches == 0 else 1


if __name__ == '__main__':
    sys.exit(main())
