"""Run through `python -m distr.launch tests/real_driver_imports.py <reference driver.py>` IN THE BUILD CONTAINER ONLY (the reference
checkout never travels): executes the IMPORT BLOCK of a real, unmodified driver of the reference (every top-level statement before
its first def / class: run_single_shape.py:1-15, run_multi_pmodata.py:1-23) under the driver's own __file__, exactly as
distr.launch would when running the driver itself, and reports where every imported name came from. Nothing of the reference is
copied: the block is read from the checkout at run time. Third-party modules the container lacks (cv2, trimesh, ...) are stubbed by
the oracle's harness (test infrastructure, oracle/ref_harness.py) -- they are not on the hot path.
"""
import ast
import inspect
import json
import os
import sys
import types

driver = os.path.abspath(sys.argv[1])
here = os.path.dirname(os.path.abspath(__file__))
for n in ['cv2', 'trimesh', 'plyfile', 'easydict', 'mathutils', 'skimage', 'skimage.measure', 'torch_scatter', 'OpenEXR', 'Imath']:
    if n not in sys.modules:
        try:
            __import__(n)
        except Exception:           # noqa: BLE001  absent here: an empty stand-in (the names are only used inside functions)
            sys.modules[n] = types.ModuleType(n)
if isinstance(sys.modules.get('skimage'), types.ModuleType) and 'skimage.measure' in sys.modules:
    sys.modules['skimage'].measure = sys.modules['skimage.measure']
if not hasattr(sys.modules['easydict'], 'EasyDict'):
    sys.modules['easydict'].EasyDict = dict
if not hasattr(sys.modules['torch_scatter'], 'scatter_max'):
    sys.modules['torch_scatter'].scatter_max = lambda *a, **k: (_ for _ in ()).throw(RuntimeError('stub'))

src = open(driver).read()
tree = ast.parse(src)
first_def = min(n.lineno for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)))
# decorators / comments directly above the first def belong to it; the import block is every statement that ENDS before it
block = [n for n in tree.body if getattr(n, 'end_lineno', n.lineno) < first_def]
mod = ast.Module(body=block, type_ignores=[])
g = {'__file__': driver, '__name__': 'driver_import_block'}
# what `python driver.py` / distr.launch guarantee before the first statement runs: the script's directory is importable
if os.path.dirname(driver) not in sys.path:
    sys.path.append(os.path.dirname(driver))
exec(compile(mod, driver, 'exec'), g)


def where(obj):
    try:
        return os.path.abspath(inspect.getsourcefile(obj))
    except TypeError:
        return os.path.abspath(getattr(sys.modules.get(getattr(obj, '__module__', ''), None), '__file__', '') or '')


names = {k: where(v) for k, v in g.items() if not k.startswith('__') and (inspect.isclass(v) or inspect.isfunction(v) or inspect.ismodule(v))
         and getattr(v, '__module__', getattr(v, '__name__', '')).split('.')[0] not in ('builtins',)}
mods = {k: os.path.abspath(v.__file__) for k, v in sys.modules.items() if (k == 'core' or k.startswith('core.') or k in (
    'create_mesh', 'decoder_utils', 'render_utils', 'renderer', 'renderer_warp', 'loss_utils', 'deep_sdf_decoder', 'optimize_single',
    'optimize_multi', 'loss_single', 'loss_multi', 'train_utils', 'evaluator', 'visualizer', 'vis_utils')) and getattr(v, '__file__', None)}
print(json.dumps({'driver': driver, 'statements': len(block), 'names': names, 'modules': mods}))
