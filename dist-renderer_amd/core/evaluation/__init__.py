"""core.evaluation: bulk SDF-grid evaluation / mesh extraction on the fused decoder kernel. The reference's
Evaluator, eval_func and transforms (CPU tooling: chamfer distance, point sampling; core/evaluation/__init__.py:3-5) are
re-exported when a reference checkout is importable next to this build -- they then call THIS create_mesh / decode_sdf through
their flat imports (core/evaluation/transforms.py:5-6)."""
import os

from core import _dropin

__path__ = _dropin.extend(__path__, __name__)

from .create_mesh import (create_mesh, create_mesh_speedup, create_sdf_grid, create_sdf_grid_speedup, get_samples,   # noqa: E402
                          infer_samples)

_dropin.absorb(globals(), __name__, ('evaluator', 'eval_func', 'transforms'), os.path.dirname(os.path.abspath(__file__)))
__all__ = [n for n in globals() if not n.startswith('_') and n not in ('os',)]
