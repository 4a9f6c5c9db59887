from .loss_single import compute_all_loss
from .optimize_single import optimize_single_view

__all__ = ['compute_all_loss', 'optimize_single_view']
