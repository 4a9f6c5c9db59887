"""Drop-in mirror of the reference's `core` package for the hot path only:

    from core.sdfrenderer import SDFRenderer, SDFRenderer_warp      (run_single_shape.py:14, run_multi_pmodata.py:14)
    from core.utils.decoder_utils import load_decoder, decode_sdf, decode_sdf_gradient
    from core.graph.deep_sdf_decoder import Decoder

Put `dist-renderer_amd/` on sys.path instead of the reference checkout. Everything below runs on the
MI355X through libdistr.so (include/distr.h); datasets, meshing, visualisation and the CLI drivers of the
reference are out of scope.
"""
