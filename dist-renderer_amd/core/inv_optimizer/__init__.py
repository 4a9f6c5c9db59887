"""core.inv_optimizer: the optimisation loops around the renderer (reference: core/inv_optimizer/__init__.py)."""
from core import _dropin

__path__ = _dropin.extend(__path__, __name__)

from .loss_single import compute_all_loss                             # noqa: E402
from .optimize_single import optimize_single_view                     # noqa: E402
from .loss_multi import compute_loss_color_warp                       # noqa: E402
from .optimize_multi import optimize_multi_view, multi_view_round     # noqa: E402

__all__ = ['compute_all_loss', 'optimize_single_view', 'compute_loss_color_warp', 'optimize_multi_view', 'multi_view_round']
