#!/usr/bin/env python
"""bench.py -- rays/sec (fwd+bwd) of the MI355X-native DIST sphere tracer on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one the metric is quoted on): one 512x512 view per GPU, 50 march steps,
DeepSDF 8x512 decoder (seed-defined synthetic weights), pyramid_recursive marcher (the API default), buffer_size 3,
ray_marching_ratio 1.5 ("aggressive"), finite-difference normal channel (use_depth2normal). A step = render (fwd) +
image loss + backward to the latent/camera gradients (+ for N>1 the RCCL all-reduce of the packed latent gradient).
Inputs are resident in HBM before the timed region. Views shard over ranks (view-parallel, weak scaling).

Extra objects on the JSON line:
  roofline     the dominant kernel (fused march/MLP kernel k_march): algorithmic FLOP (3 146 752 per decoder
               evaluation, latent hoisted) / summed hipEvent kernel time on the launch stream, vs the 157.3 TFLOP/s
               f32-MFMA peak (the instruction the kernel uses: v_mfma_f32_32x32x2_f32)
  cpu_baseline the CPU oracle (oracle/, "port") timed on this host's cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_EVAL = 3146752          # 1 573 376 MAC per point with the latent columns hoisted
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"

H = W = 512
MARCH_STEP, BUFFER_SIZE, RATIO = 50, 3, 1.5


def view_camera(fixture, view):
    # 8 cameras on a circle around the object: azimuth 45deg * view, elevation 25deg
    return fixture.make_camera(45.0 * view, 25.0 if view else 0.0, 1.6, 0.0) if view else fixture.make_camera(0, 0, 1.6, 0)


def cpu_baseline(fixture, Ws, bs, latent, budget_s=20.0):
    """Times the CPU oracle (fwd+bwd, all host cores) on the same view at the largest resolution that fits the budget."""
    from oracle import oracle as orc
    import helpers
    orc.build()
    O = orc.Oracle(Ws, bs)
    cores = orc.lib().orc_num_threads()
    R, T = view_camera(fixture, 0)
    size, rate, t = 64, None, None
    while True:
        K = fixture.make_intrinsic(size, size)
        t0 = time.perf_counter()
        helpers.oracle_render(O, orc, size, size, K, R, T, latent, march_step=MARCH_STEP, buffer_size=BUFFER_SIZE,
                              ratio=RATIO, marcher='pyramid_recursive', use_depth2normal=True)
        t = time.perf_counter() - t0
        rate = size * size / t
        if size >= H or t * 4.2 > budget_s:
            break
        size *= 2
    return {'value': rate, 'unit': 'rays/s', 'cores': int(cores), 'kind': 'port',
            'sample': '%dx%d image of view 0 (same camera/decoder/marcher, %d steps), fwd+bwd, %.1f s wall, OpenMP over rays'
                      % (size, size, MARCH_STEP, t)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--marcher', default='pyramid_recursive')
    ap.add_argument('--size', type=int, default=512, help='image side (default 512 = the headline config C3)')
    ap.add_argument('--march-step', type=int, default=50)
    args = ap.parse_args()

    global H, W, MARCH_STEP
    H = W = args.size
    MARCH_STEP = args.march_step
    from distr import binding, fixture, functions, parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback path exists)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    Ws, bs, latent_np = fixture.make_decoder_weights()
    eng = functions.engine_from_weights(Ws, bs, local)
    K = fixture.make_intrinsic(H, W)
    views = parallel.shard_views(args.gpus, rank, world)          # one view per GPU
    cfg = binding.make_cfg((H, W), K, march_step=MARCH_STEP, buffer_size=BUFFER_SIZE, ratio=RATIO, marcher=args.marcher,
                           use_depth2normal=True)
    cams = []
    for v in views:
        R, T = view_camera(fixture, v)
        cams.append((torch.from_numpy(R).to(dev).requires_grad_(True), torch.from_numpy(T).to(dev).requires_grad_(True)))
    lat = torch.from_numpy(latent_np).to(dev).requires_grad_(True)
    rs = np.random.RandomState(5)
    wd, wq, wn = (torch.from_numpy(rs.rand(*s).astype(np.float32)).to(dev) for s in ((H, W), (H, W), (H, W, 3)))
    loss_buf = torch.zeros(1, device=dev)

    def step():
        lat.grad = None
        total = None
        for (Rt, Tt) in cams:
            Rt.grad = None
            Tt.grad = None
            z, mask, q, depth, normal = functions.render_call(eng, cfg, lat, Rt, Tt)
            mb = mask.reshape(H, W).bool()
            L = torch.where(mb, depth * wd, torch.zeros_like(depth)).sum() + (q.reshape(H, W) * wq).sum() + (normal * wn).sum()
            total = L if total is None else total + L
        total.backward()
        loss_buf.copy_(total.detach().reshape(1))
        parallel.allreduce_packed([lat.grad, loss_buf])            # one RCCL all-reduce: [latent grad | loss]
        return total

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    eng.ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    parallel.barrier()
    launches, kernel_ms = eng.ctx.profile_read()
    eng.ctx.profile_enable(False)
    elapsed = parallel.allreduce_max_scalar(elapsed, device=dev)

    # forward / backward split of one step (outside the timed region; hipEvents on the current stream)
    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        return r, e0.elapsed_time(e1)
    Rt0, Tt0 = cams[0]
    outs0, fwd_ms = timed(lambda: functions.render_call(eng, cfg, lat, Rt0, Tt0))
    Lsplit = torch.where(outs0[1].reshape(H, W).bool(), outs0[3] * wd, torch.zeros_like(outs0[3])).sum() + \
        (outs0[2].reshape(H, W) * wq).sum() + (outs0[4] * wn).sum()
    _, bwd_ms = timed(lambda: Lsplit.backward())

    # counters of one forward (identical every step: same inputs)
    with torch.no_grad():
        fwd_bytes, _ = eng.ctx.workspace_bytes(cfg)
    ws = torch.empty(fwd_bytes, dtype=torch.uint8, device=dev)
    outs = [torch.empty(H * W, device=dev), torch.empty(H * W, dtype=torch.uint8, device=dev), torch.empty(H * W, device=dev),
            torch.empty(H, W, device=dev), torch.empty(H, W, 3, device=dev)]
    import ctypes as C
    p = binding.ptr
    Rt, Tt = cams[0]
    eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat.detach().reshape(-1).contiguous()),
                                               p(Rt.detach().reshape(-1).contiguous()), p(Tt.detach().contiguous()),
                                               p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(ws), ws.numel(),
                                               eng.ctx.stream()))
    stats = eng.ctx.render_stats(cfg, ws)

    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'r01_traffic.json')     # PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE), see file
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))['bytes_per_launch']
        except Exception:
            traffic = None
    if rank == 0:
        n_fwd = args.steps * len(cams)
        evals = stats['num_point_evals'] * n_fwd
        flops = FLOP_PER_EVAL * evals
        achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        rays = float(args.gpus) * H * W * args.steps
        out = {
            'metric': 'rays/sec (fwd+bwd) at %dx%d, %d march steps, DeepSDF 8x512' % (H, W, MARCH_STEP),
            'value': rays / elapsed, 'unit': 'rays/s', 'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic (seed-defined geometric-init DeepSDF 8x512 weights, latent seed 1234, synthetic cameras)',
            'config': {'workload': '%s%dx%d, %d march steps, %s marcher, buffer_size %d, ratio %.1f, depth2normal normals, '
                                   'fwd+loss+bwd, 1 view per GPU' % ('C3: ' if (H, MARCH_STEP) == (512, 50) else '', H, W, MARCH_STEP, args.marcher, BUFFER_SIZE, RATIO),
                       'parallelism': 'view-parallel x%d (RCCL all-reduce of packed latent grad)' % args.gpus,
                       'rays_in_sphere': stats['num_in_sphere'], 'valid_px': stats['num_valid'],
                       'decoder_evals_per_forward': stats['num_point_evals'],
                       'march_launches_per_forward': stats['num_march_launches'],
                       'forward_ms_one_view': fwd_ms, 'backward_ms_one_view': bwd_ms,
                       'decoder_evals_per_s_march': evals / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                         'traffic_note': 'fabric-side bytes per march step from a separate rocprofv3 PMC pass (profiles/r01_traffic.json)',
                         'kernel': 'k_march (fused 9-layer decoder + march update), %d launches, %.3f ms total, avg %.1f us'
                                   % (launches, kernel_ms, 1e3 * kernel_ms / max(launches, 1)),
                         'flop_per_eval': FLOP_PER_EVAL, 'evals': evals},
        }
        if not args.no_cpu_baseline and args.gpus == 1:      # reported baseline, rank 0 at N=1 only
            out['cpu_baseline'] = cpu_baseline(fixture, Ws, bs, latent_np)
        print(json.dumps(out))
    parallel.barrier()


if __name__ == '__main__':
    main()
