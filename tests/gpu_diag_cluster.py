"""Diagnostic (not a test): phase timing inside the cluster tile (DISTR_XCHG_TS=1) for a 64x64 render's last march step."""
import ctypes as C
import os
import sys
os.environ['DISTR_XCHG_TS'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from distr import binding, fixture, functions  # noqa: E402

Ws, bs, latent = fixture.make_decoder_weights()
eng = functions.engine_from_weights(Ws, bs, 0)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = fixture.make_intrinsic(size, size)
R, T = fixture.make_camera(30.0, 20.0, 1.6, 0.0)
cfg = binding.make_cfg((size, size), K, march_step=int(sys.argv[2]) if len(sys.argv) > 2 else 99, buffer_size=3, use_depth2normal=True)
lat = torch.from_numpy(latent).cuda().requires_grad_(True)
Rt, Tt = torch.from_numpy(R).cuda(), torch.from_numpy(T).cuda()
for _ in range(3):
    functions.render_call(eng, cfg, lat, Rt, Tt)
out = (C.c_int64 * 64)()
eng.ctx.check(eng.ctx.L.distr_debug_xchg_ts(eng.ctx.h, eng.ctx.stream(), out))
ts = np.array(list(out), np.int64)
t0 = ts[0]
print('lin0 + assembly done: %.2f us' % ((ts[1] - t0) / 100.0))
# stamps of cluster 0 / member 0 (distr_mlp.hpp, DISTR_XTS): 4l = end of layer l's k-loop, 4l+1 = own slice stored + next layer's input
# requested, 4l+3 = first 128 rows of layer l's output staged and visible in LDS (start of layer l+1's k-loop)
prev = ts[1]
for l in range(1, 8):
    a, b = ts[4 * l], ts[4 * l + 1]
    d = ts[4 * l + 3] if l < 7 else ts[32]
    print('layer %d: k-loop %.2f | relu+store+request %.2f | hand-off (until %s) %.2f   (us)' % (l, (a - prev) / 100.0, (b - a) / 100.0,
          'unit 0 staged' if l < 7 else 'all of h7 staged', (d - b) / 100.0))
    prev = d
print('total to lin8 start: %.2f us' % ((ts[32] - t0) / 100.0))
print('shader clock between the first and the last stamp: %.2f GHz (%.0f cycles in %.2f us)' % ((ts[41] - ts[40]) / ((ts[32] - ts[0]) * 10.0) , ts[41] - ts[40], (ts[32] - ts[0]) / 100.0))
if ts[42] > 0:     # a -DDISTR_XTS_UNITS build: layer 2, per unit: end of statement A (+ staging of the next unit), end of statement B
    p = ts[7]
    for u in range(4):
        print('layer 2 unit %d: A (+ staging, barrier) %.2f | B %.2f us' % (u, (ts[42 + 2 * u] - p) / 100.0, (ts[43 + 2 * u] - ts[42 + 2 * u]) / 100.0))
        p = ts[43 + 2 * u]
