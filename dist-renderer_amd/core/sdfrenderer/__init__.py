from .renderer import SDFRenderer
from .renderer_rgb import SDFRenderer_color
from .renderer_warp import SDFRenderer_warp

__all__ = ['SDFRenderer', 'SDFRenderer_color', 'SDFRenderer_warp']
