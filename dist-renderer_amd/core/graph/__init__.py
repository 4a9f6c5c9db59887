"""core.graph: the DeepSDF decoder module (reference: core/graph/__init__.py)."""
from core import _dropin

__path__ = _dropin.extend(__path__, __name__)

from .deep_sdf_decoder import Decoder    # noqa: E402

__all__ = ['Decoder']
