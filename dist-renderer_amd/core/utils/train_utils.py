"""sim(3) parameterisation used by the multi-view optimisation (reference: core/utils/train_utils.py:155-177, which
credits PMO, CVPR'19): a similarity transform is {'rot': so(3) vector (3,), 'scale': log-scale (), 'trans': (3,)}.
"""
import torch


def get_lie_rotation_matrix(r, terms=19):
    """exp([r]_x) by its power series up to `terms` (the reference sums 19 terms, train_utils.py:173-176: exact to float
    precision for the small rotations optimised here and, unlike the closed form, differentiable at r = 0).
    Evaluated in Horner form: I + A (I + A/2 (I + A/3 (...)))."""
    zero = r.new_zeros(())
    skew = torch.stack([torch.stack([zero, -r[2], r[1]]),
                        torch.stack([r[2], zero, -r[0]]),
                        torch.stack([-r[1], r[0], zero])])
    eye = torch.eye(3, dtype=r.dtype, device=r.device)
    acc = eye
    for k in range(terms, 0, -1):
        acc = eye + skew.matmul(acc) / float(k)
    return acc


def params_to_mtrx(sim3):
    """{'rot','scale','trans'} -> (3,4) matrix [exp(scale) * R | trans] (train_utils.py:157-160)."""
    R = get_lie_rotation_matrix(sim3['rot'])
    return torch.cat([sim3['scale'].exp() * R, sim3['trans'][:, None]], dim=1)
