from .loss_single import compute_all_loss
from .optimize_single import optimize_single_view
from .loss_multi import compute_loss_color_warp
from .optimize_multi import optimize_multi_view, multi_view_round

__all__ = ['compute_all_loss', 'optimize_single_view', 'compute_loss_color_warp', 'optimize_multi_view', 'multi_view_round']
