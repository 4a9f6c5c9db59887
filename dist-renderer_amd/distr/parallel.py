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


def init_from_env(backend=None):
    """Initialises torch.distributed from torchrun's env (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world_size, local_rank). backend: 'nccl' (= RCCL on ROCm) when CUDA/HIP is available, else 'gloo'."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()     # (test rigs with fewer GPUs than ranks; one GPU per rank otherwise)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('DISTR_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_views(num_views, rank, world_size):
    """Round-robin view assignment: rank r renders views r, r+world, ... (C4: 8 views on 8 GPUs = 1 view each)."""
    return list(range(rank, num_views, world_size))


def shard_rows(num_images, H, rank, world_size, align=4):
    """Strong-scaling partition of `num_images` H-row images over the ranks (shape-major, then row
    bands whose height is a multiple of `align` = 4 px so the 4x4 pyramid parents stay intact): the num_images*ceil(H/4)
    row units are cut into world_size contiguous runs. Returns this rank's [(image, r0, r1), ...]; over all ranks the
    pieces tile every image exactly once. Each piece is rendered with distr.functions.render_band_call."""
    upi = (H + align - 1) // align                    # row units per image
    total = num_images * upi
    lo, hi = (rank * total) // world_size, ((rank + 1) * total) // world_size
    out = []
    u = lo
    while u < hi:
        img = u // upi
        end = min(hi, (img + 1) * upi)
        out.append((img, (u - img * upi) * align, min(H, (end - img * upi) * align)))
        u = end
    return out


def is_distributed(group=None):
    """True when torch.distributed is up with more than one rank (the optimisation loops then shard their work items)."""
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def rank_world(group=None):
    if not is_distributed(group):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def allreduce_grads(params, scalars=(), group=None):
    """The one collective of an optimisation step: the gradients of `params` (shape code, sim(3) parameters, camera
    tensor ...) and the detached loss `scalars` are packed into ONE flat f32 buffer [g_latent | g_sim3 / g_cam | loss ...] and summed
    over the ranks with a single all-reduce (RCCL on GPUs, gloo on CPU); every rank then holds identical gradients and applies the
    identical optimiser step -- no parameter broadcast. A parameter whose .grad is None on this rank (it rendered nothing that
    depends on it) contributes zeros and receives the sum. Returns the reduced scalars as a list of 0-d tensors."""
    params = [p for p in params if p is not None]
    if not is_distributed(group):
        return [s.detach() if torch.is_tensor(s) else torch.tensor(float(s)) for s in scalars]
    dev = params[0].device if params else torch.device('cpu')
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    sc = [(s.detach().to(dev, torch.float32).reshape(1) if torch.is_tensor(s) else torch.tensor([float(s)], dtype=torch.float32, device=dev))
          for s in scalars]
    flat = torch.cat([g.detach().reshape(-1).to(torch.float32) for g in grads] + sc) if (grads or sc) else torch.zeros(0, device=dev)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        red = flat[off:off + n].reshape(g.shape).to(g.dtype)
        if p.grad is None:
            p.grad = red.clone()
        else:
            p.grad.copy_(red)
        off += n
    out = []
    for _ in sc:
        out.append(flat[off].clone())
        off += 1
    return out


def balance_views(times, H, align=4, min_rows=16, halo_rows=8, tolerance=0.02):
    """Row-band load balancing of a view-parallel step. `times[r]` = seconds rank r needs for ITS view (one H-row view per rank);
    the views are not equally expensive (number of rays that graze the surface), and every step ends in the gradient all-reduce,
    so the slowest view paces the job. Ranks above the mean hand the bottom rows of their view to ranks below it, as row bands
    (multiples of `align` = 4 rows: a band renders bit-identically to the same rows of the full image, distr.functions.
    render_band_call; the receiver also pays `halo_rows` of depth2normal halo). Pure function of `times`, so every rank computes the
    same plan from the all-gathered times. Returns plan[r] = [(view, r0, r1), ...] (first entry: what is left of the rank's own view);
    every row of every view appears exactly once. Views within `tolerance` of the mean are left alone."""
    N = len(times)
    plan = [[(r, 0, H)] for r in range(N)]
    if N < 2 or min(times) <= 0:
        return plan
    mean = sum(times) / N
    excess = [t - mean for t in times]
    if max(excess) <= tolerance * mean:
        return plan
    rows_left = [H] * N
    donors = sorted((r for r in range(N) if excess[r] > 0), key=lambda r: (-excess[r], r))
    receivers = sorted((r for r in range(N) if excess[r] < 0), key=lambda r: (excess[r], r))
    deficit = {r: -excess[r] for r in receivers}
    extra = [[] for _ in range(N)]
    for d in donors:
        per_row = times[d] / H
        ex = excess[d]
        for r in receivers:
            if ex < min_rows * per_row:
                break
            k = int(min(ex, deficit[r] - halo_rows * per_row) / per_row) // align * align
            k = min(k, rows_left[d] - H // 2)          # a view keeps at least half of its rows
            if k < min_rows:
                continue
            extra[r].append((d, rows_left[d] - k, rows_left[d]))
            rows_left[d] -= k
            ex -= k * per_row
            deficit[r] -= (k + halo_rows) * per_row
    return [[(r, 0, rows_left[r])] + extra[r] for r in range(N)]


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
