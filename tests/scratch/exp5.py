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
for env in [{'DISTR_TAIL_TEST_ABSENT': 64, 'DISTR_TAIL_FROM': 4}, {'DISTR_TAIL_TEST_ABSENT': 64, 'DISTR_TAIL_FROM': 4, 'DISTR_STICKY': 0}, {'DISTR_TAIL_TEST_ABSENT': 64, 'DISTR_TAIL_FROM': 4, 'DISTR_CLUSTER': 0},
            {'DISTR_TAIL_TEST_ABSENT': 0, 'DISTR_TAIL_FROM': 4}, {'DISTR_TAIL_TEST_ABSENT': 8, 'DISTR_TAIL_FROM': 4}, {'DISTR_TAIL_TEST_ABSENT': 64, 'DISTR_TAIL_FROM': 14},
            {'DISTR_TAIL_TEST_ABSENT': 64, 'DISTR_TAIL_FROM': 4, 'DISTR_CLUSTER_MIN': 8}, {'DISTR_TAIL_TEST_ABSENT': 64, 'DISTR_TAIL_FROM': 4, 'DISTR_CLUSTER': 4}]:
    eng, _ = g.engine_with(env)
    nb = 0; fl = 0
    for rep in range(10):
        a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
        bad = [k for k in KEYS if not np.array_equal(np.asarray(a[k]).view(np.uint8), np.asarray(ref[k]).view(np.uint8))]
        nb += bool(bad); fl += int((a['mask'] != ref['mask']).sum())
    print(env, 'bad renders %d/10, mask flips %d' % (nb, fl), flush=True)
