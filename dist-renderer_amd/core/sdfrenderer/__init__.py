"""core.sdfrenderer: SDFRenderer / SDFRenderer_warp / SDFRenderer_color on the MI355X kernels (reference:
core/sdfrenderer/__init__.py; SDFRenderer_deepsdf is the reference's own training-time wrapper and is not mirrored)."""
from core import _dropin

__path__ = _dropin.extend(__path__, __name__)

from .renderer import SDFRenderer              # noqa: E402
from .renderer_rgb import SDFRenderer_color    # noqa: E402
from .renderer_warp import SDFRenderer_warp    # noqa: E402

__all__ = ['SDFRenderer', 'SDFRenderer_color', 'SDFRenderer_warp']
