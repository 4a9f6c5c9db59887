"""TEST INFRASTRUCTURE -- golden G23: which view pairs the REFERENCE's optimize_multi_view (core/inv_optimizer/optimize_multi.py:34-110)
renders, round by round, for several (number of images, num_views_per_round, sep_dist) -- the loop itself is the reference's, run on CPU
with its compute_loss_color_warp replaced by a recorder (build container only; no reference source copied):

    python oracle/gen_golden_schedule.py        # writes tests/golden/g23_multi_view_schedule.npz

Integer logic only (rot_freq = N / views, idx stride sep_dist, the wrap-around rule of :62-65), but it decides which gradients a round
sums -- and the drop-in loop shards exactly this list over the ranks (pairs[rank::world]).
"""
import os
import sys
import tempfile

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, '..', 'dist-renderer_amd'))
sys.path.insert(0, _HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(_HERE, '..', 'tests', 'golden')
COMBOS = [(24, 8, 1), (24, 8, 2), (20, 8, 1), (10, 4, 3), (7, 8, 1), (36, 8, 1), (9, 2, 2), (16, 16, 1)]


class Ev(object):
    def latent_vec_to_points(self, *a, **k):
        return None


def main():
    rh.install_shims()
    rh.install_device_shims()
    rh.reference_modules()
    import core.inv_optimizer.optimize_multi as om
    assert os.path.abspath(om.__file__).startswith(rh.REFERENCE_ROOT)
    out = {'combos': np.array(COMBOS)}
    for (n, v, sep) in COMBOS:
        rounds, cur = [], []
        code = torch.zeros(1, 4, requires_grad=True)

        def fake(renderer, shape_code, images, cameras, idx1, idx2, weight_list, sim3=None, sim3_scale=None, visualizer=None):
            cur.append((idx1, idx2))
            return shape_code.sum() * 0.0 + 1.0, {'color': torch.tensor(0.0), 'l2reg': torch.tensor(0.0)}
        om.compute_loss_color_warp = fake
        opt = torch.optim.SGD([code], lr=0.0)
        step = opt.step

        def rec_step(*a, **k):
            rounds.append(list(cur))
            del cur[:]
            return step(*a, **k)
        opt.step = rec_step
        om.optimize_multi_view(None, Ev(), code, opt, [None] * n, [None] * n, {}, num_views_per_round=v, num_iters=2, sep_dist=sep,
                               test_step=1000, vis_dir=tempfile.mkdtemp(), vis_flag=False)
        arr = np.array(rounds)
        out['pairs_%d_%d_%d' % (n, v, sep)] = arr
        print((n, v, sep), arr.shape, arr[0].tolist()[:4], '...', arr[-1].tolist()[-2:])
    np.savez_compressed(os.path.join(OUT, 'g23_multi_view_schedule.npz'), **out)


if __name__ == '__main__':
    main()
