"""View-parallel multi-GPU helpers: one process per GPU, views sharded over ranks, ONE small RCCL all-reduce of the
packed [latent-gradient | scalars] buffer per optimiser step.

The reference has no distributed code at all; the natural data-parallel axis is the sum over view pairs that
optimize_multi.py:62-81 accumulates before a single backward(). Rays never need to be exchanged: every rank holds
the (1.8 M parameter) decoder and renders its own views, so the only collective is the gradient sum -- ~1 KiB,
latency-bound on xGMI, which is why everything is packed into a single call.
"""
import os

import torch
import torch.distributed as dist


class LaunchError(RuntimeError):
    """The job was launched in a way the one-process-per-GPU design cannot run (e.g. more RCCL ranks than GPUs on the node)."""


def init_from_env(backend=None):
    """Initialises torch.distributed from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank). backend: 'nccl' (= RCCL on ROCm) when CUDA/HIP is available, else 'gloo'.
    Under RCCL every rank needs its OWN GPU: LOCAL_RANK >= device_count is refused (LaunchError) before any communicator exists -- a
    mis-launched job (nproc-per-node above the GPU count, or a HIP_VISIBLE_DEVICES mask the launcher did not see) must not quietly
    time-share GPUs. Only the test rigs that run several ranks on one GPU over gloo (DISTR_DIST_BACKEND=gloo) wrap the index."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    have_gpu = torch.cuda.is_available()
    if backend is None:
        backend = os.environ.get('DISTR_DIST_BACKEND') or ('nccl' if have_gpu else 'gloo')
    if have_gpu:
        ndev = torch.cuda.device_count()
        if local >= ndev:
            if backend == 'nccl' and world > 1:
                raise LaunchError('LOCAL_RANK %d (rank %d of %d) but only %d GPU(s) visible to this process: RCCL ranks cannot share a device. '
                                  'Launch one process per GPU (--nproc-per-node <= %d), or set DISTR_DIST_BACKEND=gloo for a protocol test '
                                  'that time-shares GPUs' % (local, rank, world, ndev, ndev))
            local = local % ndev                       # (gloo test rigs with fewer GPUs than ranks)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        kw = {}
        if have_gpu:
            torch.cuda.set_device(local)
            if backend == 'nccl':
                # bind the communicator to THIS rank's GPU at creation: RCCL then initialises eagerly on that device and
                # barrier() / the first collective do not have to guess it from "the device under the current context"
                kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_views(num_views, rank, world_size):
    """Round-robin view assignment: rank r renders views r, r+world, ... (C4: 8 views on 8 GPUs = 1 view each)."""
    return list(range(rank, num_views, world_size))


def shard_rows(num_images, H, rank, world_size, align=4, weights=None, fixed=0.0):
    """Strong-scaling partition of `num_images` H-row images over the ranks (shape-major, then row
    bands whose height is a multiple of `align` = 4 px so the 4x4 pyramid parents stay intact, core/sdfrenderer/renderer.py:732-749;
    depth2normal reaches across a cut through a 4-row halo, core/utils/render_utils.py:31-37): the num_images*ceil(H/4)
    row units are cut into world_size contiguous runs. Returns this rank's [(image, r0, r1), ...]; over all ranks the
    pieces tile every image exactly once. Each piece is rendered with distr.functions.render_band_call.
    weights = None: runs of equal unit COUNT (cost-blind). weights[image][unit] = relative cost of every row unit (row_profile of a
    calibration render of every image; row_weights_from_counts): runs of (nearly) equal total COST -- the rows of an image are far from
    equally expensive (the object sits in the middle; top and bottom rows only hold rays that cross the sphere without a surface).
    `fixed` = cost of opening one more band, in the unit of `weights` (a band's latency-bound march tail does not shrink with
    the band). A pure function of its arguments: every rank computes the same partition."""
    if weights is None:
        upi = (H + align - 1) // align                    # row units per image
        total = num_images * upi
        lo, hi = (rank * total) // world_size, ((rank + 1) * total) // world_size
        return _units_to_pieces(lo, hi, upi, H, align)
    return shard_rows_plan(num_images, H, world_size, align, weights, fixed)[rank]


def _units_to_pieces(lo, hi, upi, H, align):
    out = []
    u = lo
    while u < hi:
        img = u // upi
        end = min(hi, (img + 1) * upi)
        out.append((img, (u - img * upi) * align, min(H, (end - img * upi) * align)))
        u = end
    return out


def shard_rows_plan(num_images, H, world_size, align=4, weights=None, fixed=0.0):
    """plan[rank] = [(image, r0, r1), ...] of the cost-weighted cut (see shard_rows). Greedy left to right: rank r takes units until it
    holds its share of what is left (remaining cost / remaining ranks, rounded to the nearer unit boundary; every opened band adds
    `fixed`), never so many that a later rank would get none. Contiguous, 4-row aligned, every unit exactly once."""
    upi = (H + align - 1) // align
    total_units = num_images * upi
    if weights is None:
        return [_units_to_pieces((r * total_units) // world_size, ((r + 1) * total_units) // world_size, upi, H, align) for r in range(world_size)]
    cost = []
    for img in range(num_images):
        w = list(weights[img]) if (weights[img] is not None and len(weights[img]) == upi) else [1.0] * upi
        cost.extend(max(float(x), 0.0) for x in w)
    if sum(cost) <= 0.0:
        cost = [1.0] * total_units
    remaining = sum(cost)
    cuts = [0]
    u = 0
    for r in range(world_size - 1):
        ranks_left = world_size - r
        # a rank's load = the cost of its units + `fixed` per piece; pieces still to be opened ~ one per remaining rank + one per image
        # boundary ahead (a run that crosses it opens a second band)
        bounds_ahead = (total_units - 1) // upi - u // upi
        target = (remaining + fixed * (ranks_left + bounds_ahead)) / ranks_left
        must_leave = min(ranks_left - 1, total_units - u)       # a unit for each later rank while units last
        acc = 0.0
        start = u
        while u < total_units - must_leave:
            opens = (u == start) or (u % upi == 0)              # the first unit of a piece: a new band
            c = cost[u] + (fixed if opens else 0.0)
            if u > start and acc + 0.5 * c > target:
                break
            acc += c
            u += 1
        # a cut within 2 % of the rank's share of an image boundary moves onto it (no band of a few rows; symmetric images cut in halves
        # stay exact halves instead of drifting by a unit per rank)
        for b in ((u // upi) * upi, (u // upi + 1) * upi):
            if start < b <= total_units - must_leave and b != u:
                lo_, hi_ = min(b, u), max(b, u)
                if sum(cost[lo_:hi_]) <= 0.02 * target:
                    u = b
                    break
        remaining -= sum(cost[start:u])
        cuts.append(u)
    cuts.append(total_units)
    return [_units_to_pieces(cuts[r], cuts[r + 1], upi, H, align) for r in range(world_size)]


def row_weights_from_counts(counts, W, align=4, H=None):
    """counts[unit] = surface pixels (rendered mask == 1) in every `align`-row unit of one image -> relative cost per unit, the same
    model as row_profile: a surface pixel counts 1, a background pixel BG_WEIGHT. (`H`: the last unit may hold fewer rows.)"""
    out = []
    n = len(counts)
    for u, c in enumerate(counts):
        rows = align if (H is None or (u + 1) * align <= H) else max(H - u * align, 0)
        out.append(BG_WEIGHT * W * rows + (1.0 - BG_WEIGHT) * float(c))
    return out


def refine_row_weights(weights, plan, loads, H, align=4):
    """One feedback step of the cost-weighted cut: `loads[r]` = what rank r measured for its pieces under `plan`. The weights of every
    rank's units are scaled by (measured share / predicted share), clamped to [0.5, 2]: bands that hold the object's silhouette rows
    (grazing rays: long marches) cost more per surface pixel than the profile says, and every band pays a latency-bound tail the
    profile does not know. Returns new weights; shard_rows_plan(..., weights=new) is the refined cut."""
    upi = (H + align - 1) // align
    new = [list(w) if w is not None and len(w) == upi else [1.0] * upi for w in weights]
    pred = []
    for r, pieces in enumerate(plan):
        pred.append(sum(sum(new[img][r0 // align:(r1 + align - 1) // align]) for (img, r0, r1) in pieces))
    tp, tl = sum(pred), sum(loads)
    if tp <= 0 or tl <= 0:
        return new
    for r, pieces in enumerate(plan):
        if pred[r] <= 0:
            continue
        k = min(2.0, max(0.5, (loads[r] / tl) / (pred[r] / tp)))
        for (img, r0, r1) in pieces:
            for u in range(r0 // align, (r1 + align - 1) // align):
                new[img][u] *= k
    return new


def is_distributed(group=None):
    """True when torch.distributed is up with more than one rank (the optimisation loops then shard their work items)."""
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank_world(group=None):
    if not is_distributed(group):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


class RemoteRankError(RuntimeError):
    """Another rank failed inside this step (it reported so through the step's all-reduce): every rank leaves the step together."""


def _reduction_device(group, params):
    """Device the packed buffer must live on: the current HIP device for RCCL ('nccl' only moves device memory), else wherever the
    first parameter lives (gloo takes host or device tensors)."""
    if dist.get_backend(group) == 'nccl':
        return torch.device('cuda', torch.cuda.current_device())
    return params[0].device if params else torch.device('cpu')


_pending_flags = {}          # group -> (host copy of a step's error flag, event after the copy): read before the NEXT collective


def reset_pending_errors(group=None):
    """Forgets a deferred error flag of `group` without reading it: called where an optimisation loop STARTS, so that a flag left behind
    by a loop that ended through some other exception cannot surface as a stale RemoteRankError in the first step of the next run."""
    _pending_flags.pop(group, None)


def check_pending_errors(group=None):
    """Reads the error flag of the previous allreduce_grads of `group`, if that call deferred it (device buffers: the flag is copied
    to pinned host memory behind the all-reduce and read here, i.e. before the next collective or at the end of the loop, when the
    copy has long completed -- the optimisation loop never stops to wait for its own all-reduce). Raises RemoteRankError if another
    rank reported a failure in that step; every rank raises it at the same point, before entering another collective."""
    item = _pending_flags.pop(group, None)
    if item is None:
        return
    host, event = item
    if event is not None:
        event.synchronize()
    n = float(host[0])
    if n > 0.0:
        raise RemoteRankError('%d rank(s) failed inside the previous step; leaving the loop on every rank' % int(round(n)))


def allreduce_grads(params, scalars=(), group=None, error=None, lazy=None):
    """The one collective of an optimisation step: the gradients of `params` (shape code, sim(3) parameters, camera
    tensor ...) and the detached loss `scalars` are packed into ONE flat buffer [g_latent | g_sim3 / g_cam | loss ... | error flag]
    and summed over the ranks with a single all-reduce (RCCL on GPUs, gloo on CPU); every rank then holds identical gradients and
    applies the identical optimiser step -- no parameter broadcast. A parameter whose .grad is None on this rank (it rendered nothing
    that depends on it) contributes zeros and receives the sum. Parameters may live on different devices (a host-resident camera
    tensor next to a CUDA shape code) and in different float types: every piece is moved to the reduction device for the
    collective and copied back in its own device / dtype (the buffer is f64 as soon as one piece is, so f64 gradients are not rounded
    through f32). `error`: an exception this rank caught while computing its share of the step (or None). The flag travels in the
    same buffer, so a failure on one rank does not leave the others waiting in a collective: the failing rank re-raises its own
    exception after the all-reduce, every other rank raises RemoteRankError. Returns the reduced scalars as 0-d tensors.
    `lazy` (default: on for device buffers): the flag of THIS step is not read here (a device -> host sync in every optimiser
    step) but by check_pending_errors() at the start of the next call / the end of the loop: the failing rank still raises at
    once (it knows), the others one step later and before they enter another collective, so nobody is left waiting either.
    What the healthy ranks do with the failed step in between: the failing rank contributed ZEROS (never a half-accumulated .grad), and
    the reduced gradients are multiplied ON THE DEVICE by (flag == 0) -- so the optimiser step the healthy ranks still take on that
    step sees a zero gradient (Adam: its momentum decays, nothing of the failed step enters the parameters), the callbacks of that one
    step see the state before it plus that momentum-only update, and RemoteRankError follows before anything else is reduced. A caller
    that checkpoints in an `except RemoteRankError` handler saves a valid, slightly stale state -- not a corrupted one."""
    params = [p for p in params if p is not None]
    if not is_distributed(group):
        if error is not None:
            raise error
        return [s.detach() if torch.is_tensor(s) else torch.tensor(float(s)) for s in scalars]
    check_pending_errors(group)                     # a failure another rank reported in the previous step: leave before the collective
    dev = _reduction_device(group, params)
    grads = [p.grad if p.grad is not None else None for p in params]
    wide = any((g if g is not None else p).dtype == torch.float64 for g, p in zip(grads, params)) or \
        any(torch.is_tensor(s) and s.dtype == torch.float64 for s in scalars)
    dt = torch.float64 if wide else torch.float32
    pieces = []
    for p, g in zip(params, grads):
        pieces.append(torch.zeros(p.numel(), dtype=dt, device=dev) if (g is None or error is not None)
                      else g.detach().reshape(-1).to(device=dev, dtype=dt))
    for s_ in scalars:
        if error is not None:
            pieces.append(torch.zeros(1, dtype=dt, device=dev))
        else:
            pieces.append(s_.detach().reshape(1).to(device=dev, dtype=dt) if torch.is_tensor(s_) else torch.tensor([float(s_)], dtype=dt, device=dev))
    pieces.append(torch.tensor([0.0 if error is None else 1.0], dtype=dt, device=dev))
    flat = torch.cat(pieces)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if error is not None:
        raise error
    if lazy is None:
        lazy = flat.is_cuda
    if lazy:
        flat[:-1] *= (flat[-1:] == 0).to(flat.dtype)      # a step some rank failed in applies NO gradient anywhere (no host sync: a device-side gate)
        if flat.is_cuda:
            host = torch.empty(1, dtype=flat.dtype, pin_memory=True)
            host.copy_(flat[-1:], non_blocking=True)
            event = torch.cuda.Event()
            event.record()
        else:
            host, event = flat[-1:].clone(), None
        _pending_flags[group] = (host, event)
    elif float(flat[-1]) > 0.0:
        raise RemoteRankError('%d rank(s) failed inside this step; leaving the step on every rank' % int(round(float(flat[-1]))))
    off = 0
    for p in params:
        n = p.numel()
        red = flat[off:off + n].reshape(p.shape).to(device=p.device, dtype=(p.grad.dtype if p.grad is not None else p.dtype))
        if p.grad is None:
            p.grad = red.clone()
        else:
            p.grad.copy_(red)
        off += n
    out = []
    for _ in scalars:
        out.append(flat[off].clone())
        off += 1
    return out


BG_WEIGHT = 0.123      # cost of a background pixel (in the unit sphere, no surface) relative to a surface pixel: measured on the C4
                       # views, 0.086 us against 0.70 us per pixel of a 512^2 / 50-step fwd+bwd (profiles/r02_band_cost.log)
BAND_FIXED = 0.02      # fixed cost of rendering one extra band, as a fraction of the mean view time (~1 ms of 51: the band's own
                       # latency-bound launches; measured with `bench.py --items`)


def row_profile(mask, align=4):
    """Relative cost of every `align`-row unit of a view from its rendered mask (H, W) (any array-like of 0 / 1): a surface pixel
    counts 1, a background pixel BG_WEIGHT. The rows of a view are far from equally expensive -- the object sits in the middle of
    the image, the top and bottom rows hold only rays that cross the sphere without finding a surface."""
    import numpy as np
    m = np.asarray(mask, dtype=np.float64)
    H, W = m.shape
    per_row = BG_WEIGHT * W + (1.0 - BG_WEIGHT) * m.sum(1)
    U = (H + align - 1) // align
    out = np.zeros(U)
    for u in range(U):
        out[u] = per_row[u * align:(u + 1) * align].sum()
    return out.tolist()


def balance_views(times, H, profiles=None, align=4, min_rows=16, halo_rows=4, tolerance=0.02, fixed=BAND_FIXED):
    """Row-band load balancing of a view-parallel step. `times[r]` = time rank r needs for ITS view (one H-row view per rank);
    the views are not equally expensive (surface pixels, grazing rays), and every step ends in the gradient all-reduce, so the
    slowest view paces the job. Slow ranks hand the BOTTOM rows of their view to fast ranks as row bands (multiples of `align` = 4
    rows: a band renders bit-identically to the same rows of the full image, distr.functions.render_band_call).
    `profiles[r]` = relative cost of every 4-row unit of view r (row_profile of its rendered mask; None: uniform) -- rows are
    converted to time through it, which matters: the bottom 52 rows of a C4 view are 10 % of its rows and 4 % of its time.
    Cost model of a plan (checked against single-GPU measurements, profiles/r02_band_cost.log): a rank pays the profile time of
    the rows it renders, `halo_rows` more on every cut edge (depth2normal halo), and `fixed` x mean time per received band.
    The plan minimises the slowest rank: the lowest level L (searched upward from the mean in 0.25 % steps) at which every rank
    above L can shed its excess to ranks below L. Pure function of its arguments, so every rank computes the same plan from the
    all-gathered inputs. Returns plan[r] = [(view, r0, r1), ...] (first entry: what is left of the rank's own view); every row of
    every view appears exactly once. Nothing moves if the slowest view is within `tolerance` of the mean."""
    N = len(times)
    plan0 = [[(r, 0, H)] for r in range(N)]
    if N < 2 or min(times) <= 0:
        return plan0
    mean = sum(times) / N
    if max(times) - mean <= tolerance * mean:
        return plan0
    U = (H + align - 1) // align
    prof = []
    for r in range(N):
        p = list(profiles[r]) if (profiles is not None and profiles[r] is not None and len(profiles[r]) == U) else [1.0] * U
        tot = sum(p)
        prof.append([times[r] * x / tot for x in p] if tot > 0 else [times[r] / U] * U)        # time of every unit of view r
    hu = (halo_rows + align - 1) // align                                                        # halo in units
    band_fixed = fixed * mean
    min_units = (min_rows + align - 1) // align

    pre = [[0.0] * (U + 1) for _ in range(N)]            # prefix sums of the unit times
    for r in range(N):
        for u in range(U):
            pre[r][u + 1] = pre[r][u] + prof[r][u]

    def span(d, lo, hi):                                 # time of units [lo, hi) of view d (clipped to the image)
        lo, hi = max(0, lo), min(U, hi)
        return pre[d][hi] - pre[d][lo] if hi > lo else 0.0

    def own_cost(d, top):                                # view d cut at unit `top`: its rows + the halo below the cut
        return span(d, 0, top) + (span(d, top, top + hu) if top < U else 0.0)

    def band_cost(d, lo, hi):                            # a received band [lo, hi) of view d: its rows + halo on every cut edge
        return span(d, lo, hi) + span(d, lo - hu, lo) + (span(d, hi, hi + hu) if hi < U else 0.0)

    def attempt(L):
        top = [U] * N                       # view d keeps units [0, top[d])
        load = list(times)
        extra = [[] for _ in range(N)]
        for d in sorted((r for r in range(N) if times[r] > L), key=lambda r: (-times[r], r)):
            for r in sorted((r for r in range(N) if times[r] < L), key=lambda r: (load[r], r)):
                if load[d] <= L or top[d] - min_units < U // 2:
                    break
                room = L - load[r] - band_fixed
                lo = top[d]
                while lo - 1 >= U // 2 and band_cost(d, lo - 1, top[d]) <= room and own_cost(d, lo) > L:
                    lo -= 1
                if top[d] - lo < min_units:
                    continue
                extra[r].append((d, lo * align, min(H, top[d] * align)))
                load[r] += band_cost(d, lo, top[d]) + band_fixed
                top[d] = lo
                load[d] = own_cost(d, lo)
        return all(load[r] <= L * 1.0000001 for r in range(N)), [[(r, 0, min(H, top[r] * align))] + extra[r] for r in range(N)], max(load)

    best = None
    L = mean
    while L < max(times):
        ok, plan, worst = attempt(L)
        if best is None or worst < best[1]:
            best = (plan, worst)
        if ok:
            break
        L *= 1.0025
    if best is None or best[1] >= max(times) * (1.0 - 1e-9):
        return plan0
    return best[0]


def refine_profiles(profiles, plan, times, loads, H, align=4, halo_rows=4):
    """One feedback step of the balancer. `loads[r]` = what rank r measured for its work under `plan`; for a donor (its own
    view cut at row r1 < H) `times[d] - loads[d]` is the REAL cost of the rows it gave away, while the profile predicted
    `times[d] - own_cost`. The given-away part of the profile is scaled by their ratio (clamped to [0.5, 2.5]): the rows at the
    lower edge of the object carry the grazing rays and cost more per surface pixel than the view's average. Returns the
    corrected profiles; balance_views(times, H, corrected) is the refined plan."""
    U = (H + align - 1) // align
    hu = (halo_rows + align - 1) // align
    out = []
    for d, p in enumerate(profiles):
        p = list(p) if (p is not None and len(p) == U) else [1.0] * U
        top = (plan[d][0][2] + align - 1) // align
        if top < U and sum(p) > 0:
            tot = sum(p)
            own = times[d] * (sum(p[:top]) + sum(p[top:top + hu])) / tot
            pred, real = times[d] - own, times[d] - loads[d]
            if pred > 1e-9:
                k = min(2.5, max(0.5, real / pred))
                p = p[:top] + [x * k for x in p[top:]]
        out.append(p)
    return out


def allgather_vector(vec, device=None, group=None):
    """[vector of rank 0, ..., vector of rank N-1] (equal lengths) on every rank."""
    v = [float(x) for x in vec]
    if not is_distributed(group):
        return [v]
    if str(dist.get_backend(group)) != 'nccl':
        device = None
    t = torch.tensor(v, dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [[float(x) for x in o.tolist()] for o in out]


def allgather_scalar(value, device=None, group=None):
    """[value of rank 0, ..., value of rank N-1] on every rank (one tiny all-gather)."""
    if not is_distributed(group):
        return [float(value)]
    if str(dist.get_backend(group)) != 'nccl':        # gloo has no device all-gather
        device = None
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t, group=group)
    return [float(o.item()) for o in out]


def allreduce_packed(tensors, group=None):
    """Sums every tensor of `tensors` over all ranks with ONE all-reduce of a packed flat f32 buffer (in place)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.detach().copy_(flat[off:off + n].reshape(t.shape).to(t.dtype))
        off += n
    return tensors


def allreduce_max_scalar(value, device=None, group=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def barrier(group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group=group)
