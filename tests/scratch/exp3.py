import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'dist-renderer_amd'); sys.path.insert(0, '.')
import numpy as np, torch
import gpu_diag_tail as g
import helpers
from distr import binding, fixture
KEYS = g.KEYS
H = W = 72; steps = 30
K = fixture.make_intrinsic(H, W)
R, T = fixture.make_camera(30, 20, 1.6, 10)
kw = dict(march_step=steps, buffer_size=3, marcher='pyramid_recursive', use_depth2normal=True, ratio=1.5)
ref_eng, latent = g.engine_with({'DISTR_TAIL': 0})
ref = helpers.hip_render(ref_eng, H, W, K, R, T, latent, **kw)
cfg = binding.make_cfg((H, W), K, **kw)
ms, st = g.time_forward(ref_eng, cfg, latent, R, T, reps=2)
print('live counts:', ref_eng.ctx.live_counts(cfg, torch.empty(1)) if False else '', flush=True)
for env in [{'DISTR_TAIL_TEST_ABSENT': a, 'DISTR_TAIL_FROM': f, **extra} for a in (64, 128, 192, 200, 248) for f in (4,) for extra in ({}, {'DISTR_STICKY': 0}, {'DISTR_CLUSTER': 0})] + \
           [{'DISTR_TAIL_TEST_ABSENT': 200, 'DISTR_TAIL_FROM': 4, 'DISTR_SAVE_MASKS': 0}, {'DISTR_TAIL_TEST_ABSENT': 200, 'DISTR_TAIL_FROM': 12}, {'DISTR_TAIL_TEST_ABSENT': 200, 'DISTR_TAIL_FROM': 0}]:
    eng, _ = g.engine_with(env)
    a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
    bad = [k for k in KEYS if not np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(ref[k]).view(np.uint8))]
    ms, st = g.time_forward(eng, cfg, latent, R, T, reps=2)
    nm = int((a['mask'] != ref['mask']).sum())
    print(env, 'BAD %s mask flips %d' % (bad, nm) if bad else 'ok', 'fwd %.2f ms steals %d fallbacks %d' % (ms, st['tail_steals'], st['cluster_fallbacks']), flush=True)
