import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'dist-renderer_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import gpu_diag_tail as g
import helpers
from distr import binding, fixture
KEYS = g.KEYS
H = W = 64
K = fixture.make_intrinsic(H, W)
R, T = fixture.make_camera(33, 12, 1.6, 0)
ref_eng, latent = g.engine_with({'DISTR_TAIL': 0})
for marcher in ('recursive', 'pyramid_recursive'):
    kw = dict(march_step=60, buffer_size=3, marcher=marcher, use_depth2normal=True, ratio=1.5)
    ref = helpers.hip_render(ref_eng, H, W, K, R, T, latent, **kw)
    cfg = binding.make_cfg((H, W), K, **kw)
    for env in [{'DISTR_CLUSTER_TEST_ABORT': 2, 'DISTR_TAIL': 0}, {'DISTR_CLUSTER_TEST_ABORT': 2, 'DISTR_TAIL': 0, 'DISTR_STICKY': 0}, {'DISTR_CLUSTER_TEST_ABORT': 2, 'DISTR_TAIL': 0, 'DISTR_STICKY': 0, 'DISTR_CLUSTER': 4},
                {'DISTR_CLUSTER_TEST_ABORT': 2, 'DISTR_TAIL': 0, 'DISTR_STICKY': 0, 'DISTR_CLUSTER_MIN': 4, 'DISTR_CLUSTER': 8},
                {'DISTR_CLUSTER_TEST_ABORT': 1, 'DISTR_TAIL': 0}, {'DISTR_CLUSTER_TEST_ABORT': 2, 'DISTR_TAIL_FROM': 0}, {'DISTR_CLUSTER_TEST_ABORT': 2, 'DISTR_TAIL': 0, 'DISTR_SAVE_MASKS': 0}]:
        eng, _ = g.engine_with(env)
        a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
        bad = [k for k in KEYS if not np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(ref[k]).view(np.uint8))]
        ms, st = g.time_forward(eng, cfg, latent, R, T, reps=2)
        rel = float(np.abs(a['g_latent'] - ref['g_latent']).max() / np.abs(ref['g_latent']).max())
        print(marcher, env, 'BAD %s g_latent rel %.2e' % (bad, rel) if bad else 'ok', 'fwd %.2f ms fallbacks %d' % (ms, st['cluster_fallbacks']), flush=True)
