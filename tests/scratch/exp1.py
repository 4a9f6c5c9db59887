import os, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, 'dist-renderer_amd'); sys.path.insert(0, '.')
import gpu_diag_tail as g
from distr import binding, fixture
variants = [('off', {'DISTR_TAIL': 0}), ('from0', {'DISTR_TAIL_FROM': 0}), ('from0-norel', {'DISTR_TAIL_FROM': 0, 'DISTR_TAIL_DBG': 1}),
            ('from0-noacq', {'DISTR_TAIL_FROM': 0, 'DISTR_TAIL_DBG': 2}), ('from0-slowpoll', {'DISTR_TAIL_FROM': 0, 'DISTR_TAIL_DBG': 4}),
            ('from0-none', {'DISTR_TAIL_FROM': 0, 'DISTR_TAIL_DBG': 7}), ('from0-nosticky', {'DISTR_TAIL_FROM': 0, 'DISTR_STICKY': 0}),
            ('from0-nosticky-none', {'DISTR_TAIL_FROM': 0, 'DISTR_STICKY': 0, 'DISTR_TAIL_DBG': 7}), ('off-nosticky', {'DISTR_TAIL': 0, 'DISTR_STICKY': 0})]
for size, steps, marcher, bs in [(64, 100, 'recursive', 1), (64, 20, 'pyramid_recursive', 3)]:
    H = W = size
    K = fixture.make_intrinsic(H, W)
    R, T = fixture.make_camera(30, 20, 1.6, 10)
    for save in (1, 0):
        for name, env in variants:
            eng, latent = g.engine_with(env)
            cfg = binding.make_cfg((H, W), K, march_step=steps, buffer_size=bs, marcher=marcher, use_depth2normal=True, ratio=1.5)
            cfg.save_for_backward = save
            ms, st = g.time_forward(eng, cfg, latent, R, T)
            print('%d/%d/%s save=%d %-22s %.3f ms launches %d' % (size, steps, marcher, save, name, ms, st['num_march_launches']), flush=True)
