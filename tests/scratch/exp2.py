import os, sys, ctypes as C
sys.path.insert(0, 'tests'); sys.path.insert(0, 'dist-renderer_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import gpu_diag_tail as g
from distr import binding, fixture
def run(name, env, size, steps, marcher, bs):
    eng, latent = g.engine_with(env)
    H = W = size
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(30, 20, 1.6, 10)
    cfg = binding.make_cfg((H, W), K, march_step=steps, buffer_size=bs, marcher=marcher, use_depth2normal=True, ratio=1.5)
    dev = eng.device
    P = H * W
    ws = torch.empty(eng.ctx.workspace_bytes(cfg)[0], dtype=torch.uint8, device=dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32).reshape(-1)).to(dev)
    lat, Rt, Tt = t(latent), t(R), t(T)
    o = [torch.empty(P, device=dev), torch.empty(P, dtype=torch.uint8, device=dev), torch.empty(P, device=dev), torch.empty(P, device=dev), torch.empty(3 * P, device=dev)]
    p = binding.ptr
    def fwd():
        eng.ctx.check(eng.ctx.L.distr_render_forward(eng.ctx.h, C.byref(cfg), p(lat), p(Rt), p(Tt), p(o[0]), p(o[1]), p(o[2]), p(o[3]), p(o[4]), p(ws), ws.numel(), eng.ctx.stream()))
    for _ in range(4): fwd()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng.ctx.profile_enable(True); fwd(); ts.append(eng.ctx.profile_read_list()); eng.ctx.profile_read()
    eng.ctx.profile_enable(False)
    ms = np.median(np.array(ts), axis=0) * 1e3
    counts = eng.ctx.live_counts(cfg, ws)
    st = eng.ctx.render_stats(cfg, ws)
    print('== %s %d/%d/%s: %d launches, sum %.1f us, tail_from %d' % (name, size, steps, marcher, len(ms), ms.sum(), st['tail_from']))
    print('   us:', ' '.join('%.0f' % x for x in ms[:40]), '...', ' '.join('%.0f' % x for x in ms[-3:]))
    print('   live:', ' '.join(str(c) for c in counts[:40]))
    return ms, counts
for (size, steps, marcher, bs) in [(64, 100, 'recursive', 1), (137, 100, 'pyramid_recursive', 3)]:
    for name, env in [('off', {'DISTR_TAIL': 0}), ('from9', {'DISTR_TAIL_FROM': 9}), ('from12', {'DISTR_TAIL_FROM': 12}), ('from30', {'DISTR_TAIL_FROM': 30}), ('from30-nosticky', {'DISTR_TAIL_FROM': 30, 'DISTR_STICKY': 0}), ('off-nosticky', {'DISTR_TAIL': 0, 'DISTR_STICKY': 0})]:
        run(name, env, size, steps, marcher, bs)
