#!/usr/bin/env python
"""bench.py -- rays/sec (fwd+bwd) of the MI355X-native DIST sphere tracer on BASELINE.json's headline config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[2], the one the metric is quoted on): one 512x512 view per GPU, 50 march steps,
DeepSDF 8x512 decoder (seed-defined synthetic weights), pyramid_recursive marcher (the API default), buffer_size 3,
ray_marching_ratio 1.5 ("aggressive"), finite-difference normal channel (use_depth2normal). A step = render (fwd) +
image loss + backward to the latent/camera gradients (+ for N>1 the RCCL all-reduce of the packed latent gradient).
Inputs are resident in HBM before the timed region. Views shard over ranks (view-parallel, weak scaling).

`--workload c5` runs BASELINE.json configs[4] instead (not the headline metric; the strong-scaling curve of SURVEY.md 8e):
4 shapes (latent seeds 1234..1237) x one 1024x1024 view x 100 march steps, a FIXED total amount of work split over the
ranks shape-major then into row bands (distr.parallel.shard_rows, distr.functions.render_band_call); "scaling": "strong".

N > 1 (one rank per GPU over RCCL; nothing of this runs at N = 1, whose line is unchanged): the run validates itself.
  * two timed regions with the same protocol (W warm-up + exactly K steps between barrier + synchronize): first UNBALANCED (one whole
    view per GPU), then -- after five untimed calibration steps -- BALANCED (slow views hand row bands to fast ranks); `value` is the
    better one, config.unbalanced_ms_per_step / balanced_ms_per_step / value_is report both;
  * config.serial_check: rank 0 renders all N views itself and the all-reduced loss / latent gradients of both regions must equal
    that serial sum (1e-5 / 1e-4 relative) -- `ok: false` and a non-zero exit otherwise;
  * config.rccl: backend (anything but RCCL is refused unless DISTR_DIST_BACKEND overrides it for one-GPU test rigs, which the line
    then flags with scaling_measurement: false), world size, RCCL version, and rank / device / PCI bus id / pid of every rank;
  * config.per_rank: every rank's own GPU milliseconds per step and its all-reduce + wait, per timed mode;
  * the opt-in arithmetic passes and the CPU baselines are skipped.

Extra objects on the JSON line:
  roofline     the dominant kernel (fused march/MLP kernel k_march): algorithmic FLOP (3 146 752 per decoder
               evaluation, latent hoisted) / summed hipEvent kernel time on the launch stream, vs the 157.3 TFLOP/s
               f32-MFMA peak (the instruction the kernel uses: v_mfma_f32_32x32x2_f32); `traffic` = fabric-side bytes per march launch,
               MEASURED BY THIS RUN at N = 1 on the headline configuration (live_traffic: two short child runs under rocprofv3 --pmc
               FETCH_SIZE / --pmc WRITE_SIZE, separate passes, ~6 s; --no-live-traffic or any failure: the committed PMC summary of the
               same kernels, profiles/rNN_traffic.json, quoted only under a matching csrc digest -- `traffic_note` says which)
  split_bf16, split_f16   the same K steps in the two OPT-IN f32-equivalent arithmetics (distr_render_cfg.arith = 1 / 2: six bf16
               products, or three f16 products on LDS-resident planes, per f32 product), same protocol; reported beside `value`,
               never as it: `value` is exact f32, bit-identical to the oracle (--no-split-bf16-pass skips both passes)
  cpu_baseline the CPU oracle (oracle/, "port": C++, OpenMP) timed on this host's cores on one fwd+bwd of the same workload;
               cpu_baseline_torch: the PyTorch-CPU restatement (baseline 2) on a bounded sample + config C1
"""
import argparse
import json
import os
import platform
import sys
import time

# RCCL between processes needs dmabuf IPC on this driver stack (hipIpcGetMemHandle fails in legacy mode); must be in the
# environment before the HIP runtime initialises, i.e. before `import torch`
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_EVAL = 3146752          # 1 573 376 MAC per point with the latent columns hoisted
ROW_BAND_FIXED = 0.05            # c5 cost-weighted cut: cost of opening one more band, as a fraction of a whole image's cost (measured: two half-image
                                 # bands cost 7-14 % more than the image, profiles/r05_plan_check_c5_n8.md)
ROW_FEEDBACK_ROUNDS = 3          # ... and how many measure-and-refine rounds the calibration may take (the best measured cut is kept)
BALANCE_MARGIN = 0.02            # N > 1: the balanced timing becomes `value` only when it beats the unbalanced one by more than this
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md "Peak FP32 (matrix)"

H = W = 512
MARCH_STEP, BUFFER_SIZE, RATIO = 50, 3, 1.5


def view_camera(fixture, view):
    # 8 cameras on a circle around the object: azimuth 45deg * view, elevation 25deg
    return fixture.make_camera(45.0 * view, 25.0 if view else 0.0, 1.6, 0.0) if view else fixture.make_camera(0, 0, 1.6, 0)


def cpu_baseline(fixture, Ws, bs, latent, size, march_step, marcher):
    """Times the CPU oracle (C++ restatement, OpenMP over rays, all host cores) on ONE fwd+bwd of the bench workload itself (same
    image size, camera, decoder, marcher, loss): ~30 s of CPU work at 512x512 on a 128-core host."""
    from oracle import oracle as orc
    import helpers
    orc.build()
    O = orc.Oracle(Ws, bs)
    cores = orc.lib().orc_num_threads()
    R, T = view_camera(fixture, 0)
    K = fixture.make_intrinsic(size, size)
    t0 = time.perf_counter()
    helpers.oracle_render(O, orc, size, size, K, R, T, latent, march_step=march_step, buffer_size=BUFFER_SIZE, ratio=RATIO, marcher=marcher,
                          use_depth2normal=True)
    t = time.perf_counter() - t0
    return {'value': size * size / t, 'unit': 'rays/s', 'cores': int(cores), 'kind': 'port',
            'sample': 'one fwd+bwd of the bench workload itself: %dx%d image of view 0, %d steps, %s, dense loss; %.1f s wall, OpenMP over rays'
                      % (size, size, march_step, marcher, t),
            'host': host_description()}


def collective_info(world, local):
    """What the collective layer itself reports (so that a reader can see RCCL saw N ranks on N distinct GPUs): torch.distributed's
    backend and world size, the RCCL version torch was built against, and -- all-gathered -- the device every rank computes on."""
    import torch.distributed as dist
    info = {'backend': None, 'world_size': 1, 'devices_visible': torch.cuda.device_count()}
    if world > 1 and dist.is_initialized():
        info['backend'] = dist.get_backend()
        info['world_size'] = dist.get_world_size()
        info['backend_is_rccl'] = info['backend'] == 'nccl'       # torch's "nccl" backend IS RCCL on ROCm
        props = torch.cuda.get_device_properties(local)
        mine = {'rank': dist.get_rank(), 'local_rank': int(os.environ.get('LOCAL_RANK', '0')), 'device_index': int(local),
                'device_name': props.name, 'pci_bus_id': getattr(props, 'pci_bus_id', None), 'pid': os.getpid(), 'host': platform.node()}
        ranks = [None] * info['world_size']
        dist.all_gather_object(ranks, mine)
        info['ranks'] = ranks
        info['distinct_devices'] = len({(r.get('host'), r['device_index'], r['pci_bus_id']) for r in ranks})    # (host: one GPU per rank on several nodes)
        # a scaling measurement needs RCCL and one GPU per rank; test rigs that time-share one GPU (DISTR_DIST_BACKEND=gloo) say so here
        info['one_gpu_per_rank'] = bool(info['backend_is_rccl'] and info['distinct_devices'] == info['world_size'])
    try:
        info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:
        info['rccl_version'] = None
    info['launcher'] = 'self-spawned torch.distributed.run' if os.environ.get('DISTR_BENCH_SPAWNED') else \
        ('external launcher' if world > 1 else 'single process')
    return info


def host_description():
    model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'cpu_model': model, 'nproc': os.cpu_count()}


def cpu_baseline_torch(fixture, Ws, bs, latent, march_step, marcher, budget_s=70.0):
    """BASELINE.md section 3, baseline (2): the build's own PyTorch-CPU restatement (oracle/torch_restatement.py: batched ATen ops, one
    decoder evaluation per march step over the live rays, autograd backward -- how the reference itself executes) on the host's cores.
    Config C1 (64x64, 20 steps: BASELINE.json configs[0], 'PyTorch CPU reference path') plus the largest power-of-two size of the bench
    configuration that fits the time budget."""
    import torch as th
    from oracle import torch_restatement as tr
    import helpers
    # ATen's CPU kernels on these shapes (a few thousand rows x 512) stop scaling long before a 256-thread host is full and collapse
    # under oversubscription (64x64/20 steps: ~1 s on 4..32 threads, ~2 minutes on 256): a bounded pool, like a user would set
    threads = max(1, min(os.cpu_count() or 1, 32))
    R, T = view_camera(fixture, 0)

    def run(size, steps, mar):
        K = fixture.make_intrinsic(size, size)
        t0 = time.perf_counter()
        tr.render_fwd_bwd(Ws, bs, latent, size, size, K, R, T, helpers.loss_weights(size, size, 5), marcher=mar, march_step=steps,
                          buffer_size=BUFFER_SIZE, ratio=RATIO, use_depth2normal=True, threads=threads)
        return time.perf_counter() - t0
    run(32, 12, marcher)                                            # warm-up (thread pool, allocator)
    t_c1 = run(64, 20, marcher)
    if t_c1 < 5.0:
        t_c1 = min(t_c1, run(64, 20, marcher))
    size, t, spent = 64, None, 0.0
    while True:
        t = run(size, march_step, marcher)
        spent += t
        if size >= H or t * 4.5 > budget_s - spent:
            break
        size *= 2
    return {'value': size * size / t, 'unit': 'rays/s', 'cores': int(th.get_num_threads()), 'kind': 'port',
            'sample': 'PyTorch-CPU restatement, %dx%d image of view 0, %d steps, %s, fwd+bwd, %.1f s wall' % (size, size, march_step, marcher, t),
            'image_size': size, 'same_size_as_value': bool(size == H),      # (smaller than the bench image only when the time budget ran out)
            'c1_64x64_20steps': {'rays_per_s': 64 * 64 / t_c1, 'seconds': t_c1}}


SMALL_RENDERS = (('c2_256x256_50', 256, 50), ('c1_64x64_20', 64, 20), ('drivers_137x137_100', 137, 100))


def small_renders(eng, functions, binding, fixture, latent_np, dev, marcher, steps=10, warmup=4):
    """extra.small_renders (N = 1, outside the headline's timed region, same protocol: warm-up, then K steps of fwd + dense loss + bwd
    between synchronisations): the small-render regime on the driver's clock -- BASELINE configs C2 (256^2 / 50 steps) and C1 (64^2 / 20)
    and 137^2 / 100 steps, the size the reference's own drivers run (run_single_shape.py:110-117). Per entry: ms per step, rays/s,
    forward-only ms, and the march kernels' fraction of the f32-MFMA peak from one bracketed forward (roofline protocol)."""
    out = {}
    R, T = view_camera(fixture, 0)
    for name, size, march in SMALL_RENDERS:
        K = fixture.make_intrinsic(size, size)
        cfg = binding.make_cfg((size, size), K, march_step=march, buffer_size=BUFFER_SIZE, ratio=RATIO, marcher=marcher, use_depth2normal=True)
        lat = torch.from_numpy(np.asarray(latent_np, np.float32)).to(dev).requires_grad_(True)
        Rt = torch.from_numpy(R).to(dev).requires_grad_(True)
        Tt = torch.from_numpy(T).to(dev).requires_grad_(True)
        rs = np.random.RandomState(5)
        wd, wq, wn = (torch.from_numpy(rs.rand(*sh).astype(np.float32)).to(dev) for sh in ((size, size), (size, size), (size, size, 3)))

        def one(backward=True):
            z, mask, q, depth, normal = functions.render_call(eng, cfg, lat, Rt, Tt)
            if not backward:
                return
            mb = mask.reshape(size, size).bool()
            L = torch.where(mb, depth * wd, torch.zeros_like(depth)).sum() + (q.reshape(size, size) * wq).sum() + (normal * wn).sum()
            lat.grad = Rt.grad = Tt.grad = None
            L.backward()
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / steps
        with torch.no_grad():
            one(False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                one(False)
            torch.cuda.synchronize()
            fwd = (time.perf_counter() - t0) / steps
        # one bracketed forward through the raw C ABI on a workspace of its own (the counters live in the workspace)
        import ctypes as C
        p = binding.ptr
        P = size * size
        ws = torch.empty(eng.ctx.workspace_bytes(cfg)[0], dtype=torch.uint8, device=dev)
        o = [torch.empty(P, device=dev), torch.empty(P, dtype=torch.uint8, device=dev), torch.empty(P, device=dev), torch.empty(P, device=dev),
             torch.empty(3 * P, device=dev)]
        args_ = (eng.ctx.h, C.byref(cfg), p(lat.detach().reshape(-1).contiguous()), p(Rt.detach().reshape(-1).contiguous()), p(Tt.detach().contiguous()),
                 p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(o[4]), p(ws), ws.numel(), eng.ctx.stream())
        eng.ctx.check(eng.ctx.L.distr_render_forward(*args_))
        eng.ctx.profile_enable(True)
        eng.ctx.check(eng.ctx.L.distr_render_forward(*args_))
        launches, kernel_ms = eng.ctx.profile_read()
        eng.ctx.profile_enable(False)
        st = eng.ctx.render_stats(cfg, ws)
        tf = FLOP_PER_EVAL * st['num_point_evals'] / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        out[name] = {'size': size, 'march_steps': march, 'ms_per_step': 1e3 * el, 'rays_per_s': size * size / el, 'forward_only_ms': 1e3 * fwd,
                     'march_launches': int(launches), 'march_kernel_ms': kernel_ms, 'march_tflops': tf, 'march_frac_of_peak': tf / PEAK_F32_MFMA_TFLOPS,
                     'decoder_evals': st['num_point_evals'], 'tail_from': st['tail_from'], 'tail_steals': st['tail_steals'],
                     'cluster_fallbacks': st['cluster_fallbacks'], 'steps': steps, 'warmup': warmup}
    return out


# measured on one MI355X (profiles/r03_view_balance.md, r05_plan_check_c5_n8.md): what --plan-only predicts from without a GPU
VIEW_MS_512 = (52.60, 54.28, 51.20, 50.40, 48.14, 46.74, 49.78, 58.94)     # one 512^2 / 50-step view of the C4 camera circle, fwd + loss + bwd
C5_IMAGE_MS = 202.4                                                          # one 1024^2 / 100-step image of C5 (810 ms for the four as one batch)
C5_LOWER_HALF = 1.048                                                        # a row of the lower image half costs this x the mean row (upper: 0.952)


def live_traffic(timeout_s=150.0):
    """roofline.traffic measured by THIS run (the number used to be read from a committed summary): two short child runs of this
    very script under `rocprofv3 --pmc <counter> --kernel-trace` -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, no other trace domain, from
    /tmp with TMPDIR=/tmp, as MI355X_MICROARCH.md's HBM section prescribes -- over the headline kernels only (3 timed steps + 1 warm-up, no CPU
    baseline, no extra passes). Fabric-side bytes per march launch = (FETCH_SIZE x 2 [gfx950 counts 64 B per 128-B request for 16 B / lane
    streaming loads] + WRITE_SIZE) KiB x 1024 / march launches. Returns (bytes_per_launch, info) or (None, reason): every failure (no rocprofv3,
    a timeout -- the child's whole process group is killed --, an unreadable CSV) leaves the caller with the committed number and says why."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, 'rocprofv3 not found'
    if any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', ''):
        return None, 'this run is itself being profiled (no nested rocprofv3)'
    child = [sys.executable, os.path.abspath(__file__), '--steps', '3', '--warmup', '1', '--no-cpu-baseline', '--no-split-bf16-pass', '--no-small-renders',
             '--no-live-traffic']
    env = dict(os.environ, TMPDIR='/tmp')
    got, launches, t0 = {}, {}, time.time()
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='distr_pmc_', dir='/tmp')
        try:
            pr = subprocess.Popen([exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '--'] + child, cwd='/tmp', env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)       # (the session this Popen started: rocprofv3 and the child bench, nothing else)
                except OSError:
                    pass
                pr.wait()
                return None, 'rocprofv3 --pmc %s pass timed out after %.0f s' % (counter, timeout_s)
            if rc != 0:
                return None, 'rocprofv3 --pmc %s pass exited with %d' % (counter, rc)
            total, n = 0.0, 0
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name') == counter and any(k in r.get('Kernel_Name', '') for k in ('k_step', 'k_march', 'k_tail')):
                        total += float(r['Counter_Value'])
                        n += 1
            if n == 0:
                return None, 'rocprofv3 --pmc %s pass: no march kernel in the counter CSV' % counter
            got[counter], launches[counter] = total, n
        except Exception as e:       # noqa: BLE001 -- a measurement aid must never take the bench line down
            return None, 'rocprofv3 --pmc %s pass failed: %r' % (counter, e)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if launches['FETCH_SIZE'] != launches['WRITE_SIZE']:
        return None, 'the two PMC passes saw different numbers of march launches (%d / %d)' % (launches['FETCH_SIZE'], launches['WRITE_SIZE'])
    n = launches['FETCH_SIZE']
    bpl = (got['FETCH_SIZE'] * 2.0 + got['WRITE_SIZE']) * 1024.0 / n
    return bpl, {'march_launches': n, 'fetch_size_kb_total': got['FETCH_SIZE'], 'write_size_kb_total': got['WRITE_SIZE'], 'wall_s': round(time.time() - t0, 1),
                 'how': 'two child runs of `bench.py --steps 3 --warmup 1` (headline kernels only) under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE '
                        '(+ --kernel-trace, separate passes); (FETCH_SIZE x 2 + WRITE_SIZE) KiB x 1024 / march launches'}


def plan_only(args):
    """`bench.py --gpus N --plan-only [--workload c5]`: the partition an N-rank run starts from (before any in-run calibration) and the
    load it predicts for every rank, as ONE JSON line -- no GPU, no process group. The prediction is a static model fitted to
    single-GPU measurements (constants above): a first multi-GPU run that scales badly is read against it -- one rank's `local_ms`
    (config.per_rank of the real line) far above its prediction is a straggler, all of them above it is the node."""
    from distr import parallel
    n = args.gpus
    c5 = args.workload == 'c5'
    size = 1024 if c5 else 512
    parts, pred = [], []
    if c5:
        for r in range(n):
            pieces = parallel.shard_rows(4, size, r, n)
            parts.append([[img, 0, r0, r1] for (img, r0, r1) in pieces])
            ms = 0.0
            for (img, r0, r1) in pieces:
                up = max(0, min(r1, size // 2) - r0)
                lo = (r1 - r0) - up
                ms += C5_IMAGE_MS * ((2.0 - C5_LOWER_HALF) * up + C5_LOWER_HALF * lo) / size
                ms += 0.0 if (r0, r1) == (0, size) else ROW_BAND_FIXED * C5_IMAGE_MS
            pred.append(ms)
        t1 = 4 * C5_IMAGE_MS
        eff = (t1 / n) / max(pred)
    else:
        for r in range(n):
            views = parallel.shard_views(n, r, n)
            parts.append([[0, (v + args.view_offset) % 8, 0, size] for v in views])
            pred.append(sum(VIEW_MS_512[(v + args.view_offset) % 8] for v in views))
        eff = VIEW_MS_512[args.view_offset % 8] / max(pred)
    line = {'plan_only': True, 'workload': args.workload, 'n_gpus': n, 'scaling': 'strong' if c5 else 'weak',
            'partition': parts, 'partition_note': 'per rank: [shape, view, first row, end row] of every piece it renders per step',
            'predicted_ms_per_rank': pred, 'predicted_slowest_ms': max(pred), 'predicted_mean_ms': sum(pred) / n,
            'predicted_efficiency_vs_n1': eff,
            'cost_model': 'static, from single-GPU measurements (profiles/r03_view_balance.md, profiles/r05_plan_check_c5_n8.md): '
                          + ('rows of the lower image half x %.3f, +%.0f %% of an image per band that is not a whole image' % (C5_LOWER_HALF, 100 * ROW_BAND_FIXED)
                             if c5 else 'per-view step time of the C4 camera circle'),
            'in_run_rebalancing': ('cost-weighted row cut from the rendered masks + measured loads, up to %d rounds, best measured cut kept' % ROW_FEEDBACK_ROUNDS)
                                  if c5 else 'slow views hand 4-row-aligned bands to fast ranks (parallel.balance_views), value switches only beyond a %.0f %% margin' % (100 * BALANCE_MARGIN)}
    print(json.dumps(line), flush=True)
    return 0


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute this very command line under torch.distributed.run with N ranks on
    this node (one per GPU), a free rendezvous port on 127.0.0.1, and the IPC mode RCCL needs. Returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), DISTR_BENCH_SPAWNED='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--marcher', default='pyramid_recursive')
    ap.add_argument('--size', type=int, default=512, help='image side (default 512 = the headline config C3)')
    ap.add_argument('--march-step', type=int, default=50)
    ap.add_argument('--view-offset', type=int, default=0, help='diagnostics: rank r renders view (r + offset) mod 8 of the C4 camera circle')
    ap.add_argument('--loss', default='dense', choices=['dense', 'reference'],
                    help='dense (default): seeded per-pixel weights on depth, normal and min-sdf of EVERY pixel (the loss of the golden '
                         'vectors; the heaviest backward: every in-sphere ray carries a gradient sample); reference: the single-view '
                         'loss of run_single_shape.py:93-98 against a ground truth rendered once from a perturbed latent')
    ap.add_argument('--streams', type=int, default=4, help='HIP streams for several work items on one GPU that cannot share a batch (row bands)')
    ap.add_argument('--no-batch', action='store_true',
                    help='several whole images on one GPU (c5 at N <= 2: a batch of shapes): render them one by one on the stream pool instead '
                         'of as ONE batched launch sequence (distr_render_forward_batch)')
    ap.add_argument('--items', default=None,
                    help='diagnostics (N = 1): the work items of the step as view:r0:r1[,view:r0:r1...] instead of one whole view, e.g. '
                         '"5:0:512,7:480:512" = what rank 5 renders under the N = 8 balance plan (profiles/r02_view_balance.md); with '
                         '--workload c5 as shape:r0:r1[,...] = the pieces one rank of a row-band partition renders (profiles/plan_check_c5.py)')
    ap.add_argument('--no-weighted-rows', action='store_true',
                    help='c5 at N > 1: keep the cost-blind row partition (default: after timing it, the ranks exchange the surface-pixel '
                         'counts of their bands and their step times, every rank computes the same cost-weighted cut '
                         '(distr.parallel.shard_rows_plan) and the step is timed again under it)')
    ap.add_argument('--no-balance', action='store_true',
                    help='c3 at N > 1: keep exactly one whole view per GPU (default: five untimed calibration steps before the warm-up, in which the '
                         'ranks exchange step times and row cost profiles and the slowest views hand row bands to the fastest ranks, '
                         'distr.parallel.balance_views)')
    ap.add_argument('--fixture', default='f1', choices=['f1', 'f2'],
                    help='f1 (default, the headline): the seed-defined geometric-init decoder (a smooth blob); f2: the decoder fitted to a '
                         'non-convex shape (torus pierced by a thin plate, tests/golden/fixture_f2.npz) -- a second data point for '
                         'evaluations per ray and the roofline fraction, not the headline metric')
    ap.add_argument('--arith', default='f32', choices=['f32', 'bf16x6', 'f16x3'],
                    help='f32 (default, the headline): exact f32 decoder evaluations; bf16x6: the opt-in six-product split-bf16 march tiles '
                         '(values within ~1e-6 of the exact ones; reported under its own name, never as the headline metric)')
    ap.add_argument('--no-split-bf16-pass', action='store_true', help='skip the two extra, separately reported passes with the opt-in split-bf16 / split-f16 march tiles')
    ap.add_argument('--workload', default='c3', choices=['c3', 'c5'],
                    help='c3 (default, the headline metric): one 512x512 view per GPU, weak scaling; '
                         'c5: 4 shapes x 1024x1024 x 100 steps split over the GPUs in row bands, strong scaling')
    ap.add_argument('--plan-only', action='store_true',
                    help='print the partition of an N-rank run and the per-rank load a static cost model predicts for it (one JSON line; no GPU, no ranks)')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help='do not measure roofline.traffic in this run (two short child runs under rocprofv3 --pmc, ~1 min at N = 1 on the headline '
                         'configuration); the committed PMC summary of the same kernels is reported instead')
    ap.add_argument('--no-small-renders', action='store_true', help='skip extra.small_renders (C2, C1 and 137^2 / 100 steps timed beside the headline at N = 1)')
    args = ap.parse_args()
    if args.plan_only:
        sys.exit(plan_only(args))
    if args.gpus > 1 and 'RANK' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        if os.environ.get('DISTR_BENCH_SPAWNED'):
            raise SystemExit('bench.py: spawned rank without RANK / WORLD_SIZE in its environment')
        sys.exit(spawn_ranks(args.gpus))                 # no launcher around us: become the launcher (an external torchrun still works)

    # stdout carries exactly ONE line, the JSON: whatever a library writes to fd 1 on the way (gloo prints "[Gloo] Rank i is connected to
    # ..." there) goes to stderr instead
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)

    global H, W, MARCH_STEP
    c5 = args.workload == 'c5'
    if c5 and (args.size, args.march_step) == (512, 50):
        args.size, args.march_step = 1024, 100
    H = W = args.size
    MARCH_STEP = args.march_step
    from distr import binding, fixture, functions, parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus:
        raise SystemExit('--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback path exists)')
    backend = torch.distributed.get_backend() if world > 1 else None
    if world > 1 and torch.distributed.get_world_size() != args.gpus:
        raise SystemExit('--gpus %d but torch.distributed reports world size %d' % (args.gpus, torch.distributed.get_world_size()))
    if world > 1 and backend != 'nccl' and not os.environ.get('DISTR_DIST_BACKEND'):
        raise SystemExit('--gpus %d needs RCCL (torch.distributed backend "nccl"), got %r' % (world, backend))
    if world > 1 and backend == 'nccl' and torch.cuda.device_count() < world:
        raise SystemExit('--gpus %d but only %d HIP device(s) are visible: RCCL ranks cannot share a device (tests that time-share one '
                         'GPU set DISTR_DIST_BACKEND=gloo; such a run is not a scaling measurement)' % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)

    Ws, bs, latent_np = fixture.make_decoder_weights() if args.fixture == 'f1' else fixture.load_fixture_f2()
    eng = functions.engine_from_weights(Ws, bs, local)
    K = fixture.make_intrinsic(H, W)
    cfg = binding.make_cfg((H, W), K, march_step=MARCH_STEP, buffer_size=BUFFER_SIZE, ratio=RATIO, marcher=args.marcher,
                           use_depth2normal=True, arith=args.arith)
    # work items of this rank: (shape, view, r0, r1)
    if c5:
        n_shapes = 4
        items = [(img, 0, r0, r1) for (img, r0, r1) in parallel.shard_rows(n_shapes, H, rank, world)]
        if args.items:
            # single-GPU diagnostic (profiles/plan_check_c5.py): the pieces ONE rank of an N-rank partition renders, as shape:r0:r1[,...]
            if world != 1:
                raise SystemExit('--items is a single-GPU diagnostic')
            items = [(int(a), 0, int(b), int(c)) for (a, b, c) in (it.split(':') for it in args.items.split(','))]
        lats_np = [latent_np] + [fixture.make_latent(1234 + i) for i in range(1, n_shapes)]
    else:
        n_shapes = 1
        items = [(0, (v + args.view_offset) % 8, 0, H) for v in parallel.shard_views(args.gpus, rank, world)]     # one view per GPU
        if args.items:
            if world != 1:
                raise SystemExit('--items is a single-GPU diagnostic')
            items = [(0,) + tuple(int(x) for x in it.split(':')) for it in args.items.split(',')]
        lats_np = [latent_np]
    cams = {}
    for (_, v, _, _) in items:
        if v not in cams:
            R, T = view_camera(fixture, v)
            cams[v] = (torch.from_numpy(R).to(dev).requires_grad_(True), torch.from_numpy(T).to(dev).requires_grad_(True))
    lats = [torch.from_numpy(l).to(dev).requires_grad_(True) for l in lats_np]
    lat = lats[0]
    rs = np.random.RandomState(5)
    wd, wq, wn = (torch.from_numpy(rs.rand(*s).astype(np.float32)).to(dev) for s in ((H, W), (H, W), (H, W, 3)))
    loss_buf = torch.zeros(1, device=dev)
    zero_grad = torch.zeros(1, 256, device=dev)

    gts = {}
    if args.loss == 'reference':
        # ground truth per (shape, view): rendered once (outside the timed region) from a perturbed latent
        for (shape, v, _, _) in items:
            if (shape, v) not in gts:
                pert = torch.from_numpy(0.1 * np.random.RandomState(900 + shape).standard_normal((1, 256)).astype(np.float32)).to(dev)
                with torch.no_grad():
                    _, gm, _, gd, gn = functions.render_call(eng, cfg, lats[shape].detach() + pert, cams[v][0].detach(), cams[v][1].detach())
                gts[(shape, v)] = (gd.clone(), gn.clone(), gm.reshape(H, W).clone())
        lw = torch.tensor([1.0, 1.0, 10.0, 5.0], device=dev)      # mask_gt, mask_out, depth, normal (run_single_shape.py:93-98)

    def image_loss(outs, r0, r1, key=None):
        z, mask, q, depth, normal = outs
        if args.loss == 'reference':
            gd, gn, gm = gts[key]
            terms = functions.single_view_losses(eng, depth, normal, mask.reshape(r1 - r0, W), q.reshape(r1 - r0, W), gd[r0:r1], gn[r0:r1],
                                                 gm[r0:r1], cfg.threshold)
            return (terms * lw).sum()
        mb = mask.reshape(r1 - r0, W).bool()
        return torch.where(mb, depth * wd[r0:r1], torch.zeros_like(depth)).sum() + (q.reshape(r1 - r0, W) * wq[r0:r1]).sum() + \
            (normal * wn[r0:r1]).sum()

    # several independent work items on one GPU (c5 at N <= 4): issue them on a small pool of HIP streams so that one item's
    # latency-bound tail overlaps another item's dense steps (same mechanism as core.inv_optimizer.optimize_multi)
    from core.inv_optimizer.optimize_multi import _StreamPool
    pool = _StreamPool(min(len(items), args.streams) if len(items) > 1 else 0, dev)

    cur = {'cfg': cfg}     # the configuration step() renders with (switched to the split-bf16 twin for the extra, separately reported pass)

    def render_item(shape, v, r0, r1):
        Rt, Tt = cams[v]
        if (r0, r1) == (0, H):
            outs = functions.render_call(eng, cur['cfg'], lats[shape], Rt, Tt)
        else:
            outs = functions.render_band_call(eng, cur['cfg'], lats[shape], Rt, Tt, r0, r1)
        if (r0, r1) == (0, H):
            last_mask[v] = outs[1]
        band_mask[(shape, r0, r1)] = outs[1]
        return image_loss(outs, r0, r1, (shape, v))

    last = {}              # gradients of the most recent step (reported as a norm: lets two runs be compared)
    last_mask = {}         # rendered mask of every whole view of the most recent step (row cost profile of the balancer)
    band_mask = {}         # rendered mask of every piece of the most recent step (c5: row weights of the cost-weighted cut)
    local_ms = []          # (calibration only) GPU milliseconds of this rank's own work of a step, without the wait in the all-reduce

    def step(measure=False):
        if measure:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for l in lats:
            l.grad = None
        for (Rt, Tt) in cams.values():
            Rt.grad = None
            Tt.grad = None
        whole = [it for it in items if (it[2], it[3]) == (0, H)]
        if len(whole) >= 2 and not args.no_batch:
            # a batch of whole images (C5: a batch of shapes): every march step is ONE launch over the live rays of all of them
            # (each image with its own shape code and camera; values identical to rendering them one by one)
            outs = functions.render_batch_call(eng, cur['cfg'], torch.cat([lats[it[0]] for it in whole], 0), torch.stack([cams[it[1]][0] for it in whole]),
                                               torch.stack([cams[it[1]][1] for it in whole]))
            losses = []
            for b, it in enumerate(whole):
                last_mask[it[1]] = outs[1][b]
                band_mask[(it[0], 0, H)] = outs[1][b]
                losses.append(image_loss(tuple(o[b] for o in outs), 0, H, (it[0], it[1])))
            rest = [it for it in items if (it[2], it[3]) != (0, H)]
            more = [pool.run(i, lambda it=it: render_item(*it)) for i, it in enumerate(rest)]
            pool.join(more)
            losses += more
        else:
            losses = [pool.run(i, lambda it=it: render_item(*it)) for i, it in enumerate(items)]
            pool.join(losses)
        total = losses[0]
        for L in losses[1:]:
            total = total + L
        total.backward()
        if measure:
            e1.record()
            e1.synchronize()
            local_ms.append(e0.elapsed_time(e1))
        loss_buf.copy_(total.detach().reshape(1))
        for l in lats:
            if l.grad is None:
                l.grad = zero_grad.clone()
        parallel.allreduce_packed([l.grad for l in lats] + [loss_buf])   # one RCCL all-reduce: [latent grads | loss]
        last['grads'] = [l.grad for l in lats]
        return total

    def timed_region():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides (nothing is recorded inside);
        the MAX over ranks, plus what the all-reduce of the last step left on this rank (loss sum, latent gradients)."""
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        parallel.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        parallel.barrier()
        return parallel.allreduce_max_scalar(el, device=dev), {'loss': float(loss_buf.item()), 'grads': [g.detach().clone() for g in last['grads']]}

    def rank_diagnostics(n=3):
        """(N > 1, outside the timed regions) what every rank spends on ITS OWN work of a step (hipEvents around render + loss +
        backward) and how long it then sits in the all-reduce waiting for the slowest rank -- all-gathered, so that a bad scaling
        curve can be read off the one JSON line: a straggler shows as one large local_ms, a slow fabric as a large wait everywhere."""
        loc, wall = [], []
        for _ in range(n):
            torch.cuda.synchronize()
            parallel.barrier()
            t0 = time.perf_counter()
            step(measure=True)
            torch.cuda.synchronize()
            wall.append(1e3 * (time.perf_counter() - t0))
            loc.append(local_ms[-1])
        l_all = parallel.allgather_scalar(float(np.median(loc)), device=dev)
        w_all = parallel.allgather_scalar(float(np.median(wall)), device=dev)
        return {'local_ms': l_all, 'step_wall_ms': w_all, 'allreduce_and_wait_ms': [w - l for w, l in zip(w_all, l_all)]}

    # ---- the timed region(s). N = 1, c5, --no-balance: ONE region (W warm-up + exactly K timed steps).
    # View-parallel runs (c3, N > 1, dense loss) time the step TWICE with the same protocol: first unbalanced (exactly one whole view
    # per GPU), then -- after five untimed calibration steps -- balanced: the eight cameras differ by up to 25 % in cost and every step
    # ends in the all-reduce, so the slowest view paces the job. In the calibration the ranks all-gather their own step times and the
    # row cost profile of their view (from its rendered mask), every rank computes the same plan (distr.parallel.balance_views), and
    # the slowest views hand row bands (bit-identical to the same rows of the full render, gradients sum exactly) to the fastest
    # ranks; the plan is refined once from the times measured under it (distr.parallel.refine_profiles). Total work is unchanged:
    # N whole views per step. Five steps: 0-1 measure the whole views, 2 warms the first plan (new band shapes: allocator), 3
    # measures it, 4 warms the refined plan. BOTH timings are reported (config.unbalanced_ms_per_step / balanced_ms_per_step);
    # `value` is the better of the two and the passes after it run in that mode.
    balance = (not c5) and world > 1 and not args.no_balance and args.loss == 'dense' and not args.items
    plan = None
    view_of = lambda r: (r + args.view_offset) % 8
    fake = os.environ.get('DISTR_BENCH_FAKE_TIMES')                  # tests: force the times on a box where the ranks share one GPU

    def apply_plan(pl):
        items[:] = [(0, view_of(v), r0, r1) for (v, r0, r1) in pl[rank]]
        for (_, v, _, _) in items:
            if v not in cams:
                R, T = view_camera(fixture, v)
                cams[v] = (torch.from_numpy(R).to(dev).requires_grad_(True), torch.from_numpy(T).to(dev).requires_grad_(True))
        pool.streams = _StreamPool(min(len(items), args.streams) if len(items) > 1 else 0, dev).streams

    calibration_steps = 0
    elapsed, snap = timed_region()
    timing = {'unbalanced': None, 'balanced': None}
    diag = {}
    snaps = {'chosen': snap}
    if world > 1:
        diag['unbalanced' if not c5 else 'row_bands'] = rank_diagnostics()
    if balance:
        timing['unbalanced'] = elapsed
        snaps['unbalanced'] = snap
        plan0 = [[(r, 0, H)] for r in range(world)]
        step(measure=True)
        step(measure=True)
        times = [float(x) for x in fake.split(',')] if fake else parallel.allgather_scalar(local_ms[-1], device=dev)
        m = last_mask.get(view_of(rank))
        prof = parallel.row_profile(m.reshape(H, W).detach().cpu().numpy()) if m is not None else [1.0] * ((H + 3) // 4)
        profiles = parallel.allgather_vector(prof, device=dev)
        plan = parallel.balance_views(times, H, profiles)
        apply_plan(plan)
        step()
        calibration_steps = 3
        if not fake:                                                 # (forced times cannot be re-measured)
            step(measure=True)
            loads = parallel.allgather_scalar(local_ms[-1], device=dev)
            profiles = parallel.refine_profiles(profiles, plan, times, loads, H)
            plan = parallel.balance_views(times, H, profiles)
            apply_plan(plan)
            step()
            calibration_steps = 5
        if all(len(p) == 1 for p in plan):
            plan = None                                              # nothing moved: plain one-view-per-GPU run, already timed
            apply_plan(plan0)
        else:
            el_b, snap_b = timed_region()
            timing['balanced'] = el_b
            snaps['balanced'] = snap_b
            diag['balanced'] = rank_diagnostics()
            # `value` switches to the balanced region only when it wins by more than the run-to-run noise (2 %): a plain best-of-two
            # would bias the headline upward by that noise
            if el_b < (1.0 - BALANCE_MARGIN) * elapsed or fake:      # (forced times: the tests want to see the plan's run)
                elapsed, snaps['chosen'] = el_b, snap_b
            else:
                plan_tried, plan = plan, None                        # the balancer did not pay on this node: report it, run unbalanced
                apply_plan(plan0)
                timing['balance_plan_tried'] = plan_tried

    # ---- c5 at N > 1: the cost-weighted cut. The region above timed the cost-blind partition (equal row-unit
    # runs). Now: every rank counts the surface pixels of its pieces per 4-row unit, one packed all-reduce makes the profile of all
    # four images known everywhere, the ranks all-gather what their pieces cost them, and every rank computes the same weighted cut
    # (distr.parallel.shard_rows_plan: cuts stay multiples of 4 rows, bands keep their 4-row halo) -- refined once from the times
    # measured under it. Same total work, timed with the same protocol; BOTH timings are reported, `value` switches to the weighted
    # one only when it wins by more than BALANCE_MARGIN.
    row_plan = None
    if c5 and world > 1 and not args.no_weighted_rows and not args.items:
        timing['uniform'] = elapsed
        snaps['uniform'] = snap
        upi = (H + 3) // 4
        plan_u = [parallel.shard_rows(n_shapes, H, r, world) for r in range(world)]

        def apply_rows(pl):
            items[:] = [(img, 0, r0, r1) for (img, r0, r1) in pl[rank]]
            pool.streams = _StreamPool(min(len(items), args.streams) if len(items) > 1 else 0, dev).streams

        def gathered_weights():
            cnt = torch.zeros(n_shapes * upi, device=dev)
            for (shape, _, r0, r1) in items:
                m = band_mask.get((shape, r0, r1))
                if m is None:
                    continue
                rows = m.reshape(r1 - r0, W).float().sum(1)
                pad = (-(r1 - r0)) % 4
                if pad:
                    rows = torch.cat([rows, torch.zeros(pad, device=dev)])
                cnt[shape * upi + r0 // 4: shape * upi + r0 // 4 + rows.numel() // 4] += rows.reshape(-1, 4).sum(1)
            parallel.allreduce_packed([cnt])
            c = cnt.cpu().numpy()
            return [parallel.row_weights_from_counts(c[i * upi:(i + 1) * upi].tolist(), W, 4, H) for i in range(n_shapes)]

        step(measure=True)
        step(measure=True)
        loads_u = [float(x) for x in fake.split(',')] if fake else parallel.allgather_scalar(local_ms[-1], device=dev)
        weights = parallel.refine_row_weights(gathered_weights(), plan_u, loads_u, H)
        fixed = ROW_BAND_FIXED * sum(sum(w) for w in weights) / n_shapes          # cost of opening one more band (a band's latency-bound tail)
        best = (max(loads_u), None)                                               # (slowest rank's ms, plan); None = the cost-blind cut
        calibration_steps = 2
        for _ in range(1 if fake else ROW_FEEDBACK_ROUNDS):
            cand = parallel.shard_rows_plan(n_shapes, H, world, 4, weights, fixed)
            if cand == plan_u:
                break
            apply_rows(cand)
            step()                                                                # new band shapes: allocator
            step(measure=True)
            calibration_steps += 2
            loads_w = [float(x) for x in fake.split(',')] if fake else parallel.allgather_scalar(local_ms[-1], device=dev)
            if max(loads_w) < best[0] or fake:
                best = (max(loads_w), cand)
            weights = parallel.refine_row_weights(weights, cand, loads_w, H)
        row_plan = best[1] if best[1] is not None else plan_u
        apply_rows(row_plan)
        step()
        calibration_steps += 1
        if row_plan == plan_u:
            row_plan = None                                          # nothing moved
        else:
            el_w, snap_w = timed_region()
            timing['weighted'] = el_w
            snaps['weighted'] = snap_w
            diag['weighted_rows'] = rank_diagnostics()
            if el_w < (1.0 - BALANCE_MARGIN) * elapsed or fake:
                elapsed, snaps['chosen'] = el_w, snap_w
            else:
                timing['row_plan_tried'], row_plan = row_plan, None
                apply_rows(plan_u)

    # ---- serial check (N > 1, outside the timed regions): rank 0 renders ALL the job's images itself, one whole image after the
    # other (the sum the reference accumulates serially before its single backward(), core/inv_optimizer/optimize_multi.py:62-81),
    # and compares loss and shape-code gradients with what the all-reduce left after the timed steps. A sharded run that dropped a
    # band, double-counted a view or reduced over the wrong ranks fails here, in the run that produces the number.
    serial_check = None
    if world > 1:
        serial_check = {'images': (n_shapes if c5 else args.gpus), 'tolerance': {'loss_rel': 1e-5, 'grad_rel': 1e-4}}
        if rank == 0:
            for l in lats:
                l.grad = None
            tot = 0.0
            todo = [(sh, 0) for sh in range(n_shapes)] if c5 else [(0, view_of(r)) for r in range(world)]
            for (sh, v) in todo:
                if v not in cams:
                    Rv, Tv = view_camera(fixture, v)
                    cams[v] = (torch.from_numpy(Rv).to(dev).requires_grad_(True), torch.from_numpy(Tv).to(dev).requires_grad_(True))
                if args.loss == 'reference' and (sh, v) not in gts:
                    pert = torch.from_numpy(0.1 * np.random.RandomState(900 + sh).standard_normal((1, 256)).astype(np.float32)).to(dev)
                    with torch.no_grad():
                        _, gm, _, gd, gn = functions.render_call(eng, cfg, lats[sh].detach() + pert, cams[v][0].detach(), cams[v][1].detach())
                    gts[(sh, v)] = (gd.clone(), gn.clone(), gm.reshape(H, W).clone())
                Ls = image_loss(functions.render_call(eng, cfg, lats[sh], cams[v][0], cams[v][1]), 0, H, (sh, v))
                Ls.backward()
                tot += float(Ls.detach())
            gser = [(l.grad if l.grad is not None else zero_grad).detach().clone() for l in lats]
            gmax = max(float(g.abs().max()) for g in gser)
            for name, sn in snaps.items():
                serial_check[name] = {'loss_rel': abs(sn['loss'] - tot) / max(abs(tot), 1e-30),
                                      'grad_rel': max(float((a - b).abs().max()) for a, b in zip(sn['grads'], gser)) / max(gmax, 1e-30)}
            serial_check['loss_serial_rank0'] = tot
            serial_check['ok'] = all(v['loss_rel'] <= 1e-5 and v['grad_rel'] <= 1e-4 for k, v in serial_check.items() if k in snaps)
            for l in lats:
                l.grad = None
            for (Rt, Tt) in cams.values():
                Rt.grad = None
                Tt.grad = None
        parallel.barrier()

    # ---- second pass (not part of `value`): per-step wall times -> median (SURVEY.md 8d asks for the median of >= 20 iterations)
    per_step = []
    for _ in range(max(args.steps, 1)):
        torch.cuda.synchronize()
        ts = time.perf_counter()
        step()
        torch.cuda.synchronize()
        per_step.append(time.perf_counter() - ts)
    median_s = parallel.allreduce_max_scalar(float(np.median(per_step)), device=dev)

    def forward_stats(cfg0):
        """counters of one forward of every work item of this rank (identical every step: same inputs)"""
        import ctypes as C
        p = binding.ptr
        stats = None
        for (shape, v, r0, r1) in items:
            icfg = cfg0 if (r0, r1) == (0, H) else functions.band_cfg(cfg0, r0, r1)[0]
            n = icfg.band_rows * W
            fwd_bytes, _ = eng.ctx.workspace_bytes(icfg)
            ws = torch.empty(fwd_bytes, dtype=torch.uint8, device=dev)
            outs = [torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, device=dev),
                    torch.empty(n, device=dev), torch.empty(n, 3, device=dev)]
            Rt, Tt = cams[v]
            eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(icfg), p(lats[shape].detach().reshape(-1).contiguous()),
                                                       p(Rt.detach().reshape(-1).contiguous()), p(Tt.detach().contiguous()),
                                                       p(outs[0]), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(ws), ws.numel(),
                                                       eng.ctx.stream()))
            st = eng.ctx.render_stats(icfg, ws)
            stats = st if stats is None else {k: stats[k] + st[k] for k in st}
        return stats

    # ---- extra passes (not part of `value`, reported under their own keys): the same K steps with the OPT-IN split-bf16 and split-f16
    # march tiles (distr_render_cfg.arith = DISTR_ARITH_BF16X6 / _F16X3: six bf16 / three f16 products per f32 product, f32 accumulation;
    # values within ~1e-6 of the exact ones, parity against the reference's goldens at the 1e-4 bar: tests). Same protocol: warm-up,
    # barrier + synchronize on both sides, max over ranks. The headline stays the exact-f32 number above.
    split_modes = {}
    if args.arith == 'f32' and not args.no_split_bf16_pass and world == 1:     # (N > 1: the scaling run times the exact path only)
        for mode in ('bf16x6', 'f16x3'):
            cfg_m = cfg.clone()
            cfg_m.arith = binding.ARITH[mode]
            cur['cfg'] = cfg_m
            for _ in range(max(2, min(args.warmup, 5))):
                step()
            torch.cuda.synchronize()
            parallel.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tot_m = step()
            torch.cuda.synchronize()
            el_m = time.perf_counter() - t0
            parallel.barrier()
            el_m = parallel.allreduce_max_scalar(el_m, device=dev)
            g_m = float(sum(float(g.norm()) for g in last['grads']))
            split_modes[mode] = {'ms_per_step': 1e3 * el_m / args.steps, 'loss_rank0': float(tot_m.detach()), 'latent_grad_norm_all_ranks': g_m, 'elapsed_s': el_m}
            if mode == 'f16x3':
                split_modes[mode]['f16_overflows'] = forward_stats(cfg_m)['f16_overflows']
        cur['cfg'] = cfg
        tot_f32 = step()                                   # back on the exact path (also what the passes below measure)
        for m in split_modes.values():
            m['loss_rank0_exact_f32'] = float(tot_f32.detach())

    # ---- third pass: roofline of the march kernels. Every march launch is bracketed by hipEvents on the launch stream (the
    # brackets serialise a little, which is why they are not in the timed region); algorithmic FLOP / summed kernel time
    ROOF_STEPS = min(5, max(args.steps, 1))
    saved_streams, pool.streams = pool.streams, []       # one stream: brackets of concurrent streams would overlap and double count
    eng.ctx.profile_enable(True)
    for _ in range(ROOF_STEPS):
        step()
    launches, kernel_ms = eng.ctx.profile_read()
    eng.ctx.profile_enable(False)
    pool.streams = saved_streams

    grad_norm = float(sum(float(g.norm()) for g in last['grads']))

    # forward / backward split of one step (outside the timed region; hipEvents on the current stream)
    def timed(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        torch.cuda.synchronize()
        return r, e0.elapsed_time(e1)
    shape0, v0, r00, r01 = items[0]
    Rt0, Tt0 = cams[v0]
    one = (lambda: functions.render_call(eng, cfg, lats[shape0], Rt0, Tt0)) if (r00, r01) == (0, H) else \
        (lambda: functions.render_band_call(eng, cfg, lats[shape0], Rt0, Tt0, r00, r01))
    outs0, fwd_ms = timed(one)
    Lsplit = image_loss(outs0, r00, r01, (shape0, v0))
    _, bwd_ms = timed(lambda: Lsplit.backward())

    stats = forward_stats(cfg)

    traffic = traffic_hbm = None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_traffic.json')))   # newest round's PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)
    tpath = cands[-1] if cands else os.path.join(ROOT, 'profiles', 'r02_traffic.json')
    traffic_digest, traffic_stale = None, None
    from distr import binding as _binding
    built_from = _binding.source_digest()
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic_digest = tj.get('csrc_sha256')
            # the PMC pass describes the kernels it ran: if csrc/ has changed since, the number is not this library's
            traffic_stale = traffic_digest != built_from
            traffic = None if traffic_stale else tj['bytes_per_launch']
            traffic_hbm = tj.get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
    traffic_static, traffic_live_info = traffic, None
    if world == 1 and not args.no_live_traffic and not c5 and not args.items and args.arith == 'f32' and (H, MARCH_STEP) == (512, 50) \
            and args.marcher == 'pyramid_recursive' and args.loss == 'dense' and args.fixture == 'f1':
        torch.cuda.synchronize()
        live, traffic_live_info = live_traffic()
        if live is not None:
            traffic = live
    rccl_info = collective_info(world, local)           # (collective: every rank takes part in the all-gather)
    partition = None
    if world > 1:
        partition = [None] * world
        torch.distributed.all_gather_object(partition, [list(it) for it in items])
    small = None
    if world == 1 and not args.no_small_renders and not c5 and not args.items and args.arith == 'f32' and (H, MARCH_STEP) == (512, 50):
        small = small_renders(eng, functions, binding, fixture, latent_np, dev, args.marcher)
    if rank == 0:
        evals = stats['num_point_evals'] * ROOF_STEPS
        flops = FLOP_PER_EVAL * evals
        achieved = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
        rays = (4.0 if c5 else float(args.gpus)) * H * W * args.steps     # c5: fixed total work (4 images); c3: one view per GPU
        out = {
            'metric': 'rays/sec (fwd+bwd) at %dx%d, %d march steps, DeepSDF 8x512' % (H, W, MARCH_STEP),
            'value': rays / elapsed, 'unit': 'rays/s', 'n_gpus': args.gpus, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'median_ms_per_step': 1e3 * median_s,
            'value_at_median': rays / args.steps / median_s, 'higher_is_better': True, 'scaling': 'strong' if c5 else 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.arith == 'f32' else 'f32 emulated by six bf16 products per f32 product (f32 accumulation; opt-in, not the headline)', 'data': 'synthetic (seed-defined geometric-init DeepSDF 8x512 weights, latent seed 1234, synthetic cameras)' if args.fixture == 'f1' else
                                    'synthetic (fixture F2: DeepSDF 8x512 decoder fitted to an analytic torus + thin plate, synthetic cameras)',
            'config': {'workload': '%s%dx%d, %d march steps, %s marcher, buffer_size %d, ratio %.1f, depth2normal normals, '
                                   '%s loss, fwd+loss+bwd, %s' % ('C5: 4 shapes x ' if c5 else ('C3: ' if (H, MARCH_STEP) == (512, 50) else ''), H, W, MARCH_STEP,
                                                         args.marcher, BUFFER_SIZE, RATIO,
                                                         'dense per-pixel' if args.loss == 'dense' else 'reference single-view',
                                                         ('fixed total work split shape-major then in row bands' + (', whole images of a rank as one batched launch sequence' if not args.no_batch else '')) if c5 else
                                                         ('N views per step on N GPUs (C4 camera circle): one view per GPU, slow views hand row bands to fast '
                                                          'ranks' if plan else '1 view per GPU')),
                       'parallelism': ('shape/row-band-parallel x%d' if c5 else 'view-parallel x%d') % args.gpus + (' with row-band load balancing' if plan else '') +
                                      ((' (%s all-reduce of packed latent grad)' % ('RCCL' if rccl_info.get('backend') == 'nccl' else str(rccl_info.get('backend')) + ' [NOT RCCL: test rig]')) if world > 1 else ' (single process, no collective)'),
                       'rccl': rccl_info,
                       'rank0_items': [list(it) for it in items], 'balance_plan': plan, 'calibration_steps_before_warmup': calibration_steps, 'loss_sum_all_ranks': float(loss_buf.item()),
                       'latent_grad_norm_all_ranks': grad_norm,
                       'rays_in_sphere': stats['num_in_sphere'], 'valid_px': stats['num_valid'], 'fixture': args.fixture, 'arith': args.arith,
                       'decoder_evals_per_image_ray': stats['num_point_evals'] / float(len(items) * H * W) if not c5 else None,
                       'decoder_evals_per_step_rank0': stats['num_point_evals'],
                       'march_launches_per_step_rank0': stats['num_march_launches'],
                       'forward_ms_one_item': fwd_ms, 'backward_ms_one_item': bwd_ms, 'cluster_fallbacks': stats['cluster_fallbacks'],
                       'decoder_evals_per_s_march': evals / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS if args.arith == 'f32' else 2500.0, 'unit': 'TFLOP/s',
                         'frac': achieved / (PEAK_F32_MFMA_TFLOPS if args.arith == 'f32' else 2500.0), 'traffic': traffic,
                         'peak_note': 'f32-MFMA peak' if args.arith == 'f32' else 'bf16 / f16 MFMA dense peak; algorithmic FLOP counted once (the six bf16 / three f16 products per f32 product are not counted several times)',
                         'traffic_csrc_sha256': traffic_digest, 'csrc_sha256': built_from,
                         'traffic_note': ('LIVE: measured by this run -- ' + traffic_live_info['how'] + '; fabric-side bytes (L2 misses: Infinity Cache + HBM), algorithmic bytes are ~32 B per decoder evaluation + the weights once, the excess is the 6.3 MB weight stream each XCD re-fetches per tile round') if isinstance(traffic_live_info, dict) else ('null: the committed PMC pass (%s) was taken on other kernels (csrc digest differs); re-run profiles/promote.sh' % os.path.relpath(tpath, ROOT)) if traffic_stale else 'STATIC: not measured by this run. Fabric-side bytes per march launch (FETCH_SIZE x 2 + WRITE_SIZE) read from the committed summary of a separate rocprofv3 --pmc pass over this same command (%s); algorithmic bytes are ~32 B per decoder evaluation, the excess is the 6.3 MB weight stream each XCD re-fetches from L2 / Infinity Cache per tile round' % os.path.relpath(tpath, ROOT),
                         'traffic_live': traffic_live_info if isinstance(traffic_live_info, dict) else None,
                         'traffic_live_failed': traffic_live_info if isinstance(traffic_live_info, str) else None,
                         'traffic_static': traffic_static,
                         'traffic_hbm': traffic_hbm,
                         'traffic_hbm_note': 'null: not observable -- rocprofv3 on this stack exposes the L2\'s memory-side request counters only (every read request counts as '
                                             '"destined for DRAM", TCC_EA0_RDREQ_DRAM = TCC_EA0_RDREQ); Infinity-Cache hits are not separated from HBM reads. HBM traffic <= `traffic`; '
                                             'the render\'s working set (14.5 MB of weights + ray state) is far below the 256 MiB Infinity Cache (profiles/r06_traffic.json: memory_side_reads)',
                         'kernel': 'k_march / k_step / k_tail (fused 9-layer decoder + march update; one hipEvent bracket per march launch, separate pass of %d steps), %d launches, %.3f ms total, avg %.1f us'
                                   % (ROOF_STEPS, launches, kernel_ms, 1e3 * kernel_ms / max(launches, 1)),
                         'flop_per_eval': FLOP_PER_EVAL, 'evals': evals},
        }
        notes = {'bf16x6': 'opt-in arithmetic (distr_render_cfg.arith = 1, `--arith bf16x6`): every f32 product of the seven wide decoder layers '
                           'as six bf16 products with f32 accumulation; same workload, same K steps, same timing protocol; NOT the headline '
                           '(the headline `value` is exact f32, bit-identical to the oracle). Parity of this mode: reference goldens at the 1e-4 '
                           'bar with 0 mask flips (tests/test_gpu_parity.py::test_render_matches_reference_goldens[*-bf16x6]).',
                 'f16x3': 'opt-in arithmetic (distr_render_cfg.arith = 2, `--arith f16x3`): three f16 products per f32 product on two f16 planes per '
                          'operand (scaled by 64), activations kept as planes in LDS; backward = the split-bf16 dX chain. Limited to decoders '
                          'whose weights and activations stay below 1023 (checked; f16_overflows must be 0). Same protocol as split_bf16; NOT the headline.'}
        for mode, key in (('bf16x6', 'split_bf16'), ('f16x3', 'split_f16')):
            if mode in split_modes:
                m = split_modes[mode]
                m.update(value=rays / m.pop('elapsed_s'), unit='rays/s', speedup_vs_exact_f32=(elapsed / args.steps) / (m['ms_per_step'] * 1e-3), note=notes[mode])
                out[key] = m
        if small is not None:
            out['extra'] = {'small_renders': small,
                            'small_renders_note': 'not part of `value`: the same fwd + dense loss + bwd step at BASELINE configs C2 / C1 and at the size the '
                                                  "reference's drivers run, timed after the headline with the same protocol (warm-up, K steps between synchronisations)"}
        if world > 1:
            c = out['config']
            c['partition'] = partition
            c['serial_check'] = serial_check
            c['per_rank'] = diag
            c['scaling_measurement'] = bool(rccl_info.get('one_gpu_per_rank'))       # false on test rigs whose ranks time-share one GPU over gloo
            if balance:
                c['unbalanced_ms_per_step'] = 1e3 * timing['unbalanced'] / args.steps
                c['balanced_ms_per_step'] = (1e3 * timing['balanced'] / args.steps) if timing['balanced'] is not None else None
                c['value_is'] = 'balanced' if plan else 'unbalanced'
                c['value_switch_margin'] = BALANCE_MARGIN
                if timing.get('balance_plan_tried'):
                    c['balance_plan_tried'] = timing['balance_plan_tried']
            if c5 and timing.get('uniform') is not None:
                c['uniform_rows_ms_per_step'] = 1e3 * timing['uniform'] / args.steps
                c['weighted_rows_ms_per_step'] = (1e3 * timing['weighted'] / args.steps) if timing.get('weighted') is not None else None
                c['value_is'] = 'weighted_rows' if row_plan else 'uniform_rows'
                c['value_switch_margin'] = BALANCE_MARGIN
                c['row_plan'] = row_plan if row_plan else timing.get('row_plan_tried')
            # which of the timed modes is the N = 1 protocol run on N GPUs (one whole view per GPU, nothing moved between ranks): efficiency
            # against the N = 1 line compares like with like only through this mode's number
            c['n1_protocol_equivalent'] = 'unbalanced' if not c5 else 'row_bands (strong scaling: no N = 1 analogue per rank; compare total ms_per_step)'
        if not args.no_cpu_baseline and args.gpus == 1:      # reported baselines, rank 0 at N=1 only (~30 s + ~20 s of CPU work)
            out['cpu_baseline'] = cpu_baseline(fixture, Ws, bs, latent_np, H, MARCH_STEP, args.marcher)
            out['cpu_baseline_torch'] = cpu_baseline_torch(fixture, Ws, bs, latent_np, MARCH_STEP, args.marcher)
        print(json.dumps(out), file=json_out, flush=True)
    parallel.barrier()
    if world > 1:
        ok = parallel.allreduce_max_scalar(0.0 if (rank != 0 or serial_check.get('ok')) else 1.0, device=dev) == 0.0
        if not ok:
            raise SystemExit('bench.py: the all-reduced loss / gradients do not match the serial sum over all images (config.serial_check)')


if __name__ == '__main__':
    main()
