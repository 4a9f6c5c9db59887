"""How this mirror coexists with a checkout of the reference on sys.path.

Only the hot-path modules are mirrored here; `core.dataset`, `core.visualize`, most of
`core.evaluation` ... stay the reference's own code. The reference's drivers import all of them through the ONE package
name `core` (run_single_shape.py:6-13, run_multi_pmodata.py:8-15), so with `dist-renderer_amd/` first on sys.path this
package must not hide the rest of the reference's `core`:

  * every package of the mirror extends its __path__ over the later sys.path entries that hold a package of the same name
    (pkgutil.extend_path): `core.dataset`, `core.visualize.vis_utils`, `core.evaluation.evaluator` ... resolve from the
    reference checkout, while every module that exists here (core.sdfrenderer.*, core.utils.decoder_utils ...) resolves here;
  * the reference's packages import their own files as TOP-LEVEL modules after appending the package directory to sys.path
    (`from decoder_utils import decode_sdf`, core/evaluation/create_mesh.py:8; `from create_mesh import ...`,
    core/evaluation/transforms.py:5; core/utils/__init__.py:2 ...). The same directories are appended here, this build's first,
    so that those flat names also land on the MI355X modules wherever one exists;
  * `from core.evaluation import *` must still deliver the reference's Evaluator etc.: `absorb` re-exports the public names of
    sibling modules that only exist in the reference checkout.
"""
import importlib
import os
import pkgutil
import sys

SUBPACKAGES = ('sdfrenderer', 'utils', 'dataset', 'evaluation', 'graph', 'visualize', 'inv_optimizer')
errors = {}          # module name -> ImportError text of a sibling module that could not be absorbed (missing third-party deps)


def extend(path, name):
    """__path__ of package `name`, extended over same-named packages later on sys.path."""
    return pkgutil.extend_path(path, name)


def publish_flat_dirs(core_paths):
    """Appends <root>/<subpackage> of every `core` root to sys.path (this build's root is core_paths[0]: it wins)."""
    for root in core_paths:
        for sub in SUBPACKAGES:
            d = os.path.join(root, sub)
            if os.path.isdir(d) and d not in sys.path:
                sys.path.append(d)


def absorb(namespace, package, modules, own_dir):
    """`from <package>.<module> import *` for every module that resolves OUTSIDE this build (a reference checkout next to it)."""
    for mod in modules:
        full = package + '.' + mod
        try:
            spec = importlib.util.find_spec(full)
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin or os.path.dirname(os.path.abspath(spec.origin)) == own_dir:
            continue
        try:
            m = importlib.import_module(full)
        except ImportError as e:          # e.g. trimesh / plyfile / cv2 missing in this environment
            errors[full] = str(e)
            continue
        names = getattr(m, '__all__', None) or [n for n in vars(m) if not n.startswith('_')]
        for n in names:
            namespace.setdefault(n, getattr(m, n))
