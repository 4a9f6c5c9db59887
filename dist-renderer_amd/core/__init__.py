"""Drop-in mirror of the reference's `core` package for the hot path only:

    from core.sdfrenderer import SDFRenderer, SDFRenderer_warp      (run_single_shape.py:14, run_multi_pmodata.py:14)
    from core.utils.decoder_utils import load_decoder, decode_sdf, decode_sdf_gradient
    from core.graph.deep_sdf_decoder import Decoder
    from core.inv_optimizer import optimize_single_view, optimize_multi_view

Put `dist-renderer_amd/` IN FRONT OF the reference checkout on sys.path: every module that exists here runs on the MI355X
through libdistr.so (include/distr.h); everything else of the reference's `core` (datasets, visualiser, evaluator, ...: out of
scope) keeps resolving from the reference checkout, because this package and its sub-packages extend
their __path__ over it (core/_dropin.py). Without a reference checkout only the mirrored modules exist.
"""
from . import _dropin

__path__ = _dropin.extend(__path__, __name__)
_dropin.publish_flat_dirs(list(__path__))
