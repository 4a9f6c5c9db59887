from .renderer import SDFRenderer
from .renderer_warp import SDFRenderer_warp

__all__ = ['SDFRenderer', 'SDFRenderer_warp']
