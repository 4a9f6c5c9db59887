"""STUB (test infrastructure, none of the reference's code): same flat-import style as core/dataset of this stub checkout."""
import importlib as _importlib
import pathlib as _pathlib
import sys as _sys

_HERE = str(_pathlib.Path(__file__).resolve().parent)
if _HERE not in _sys.path:
    _sys.path.append(_HERE)
_vis = _importlib.import_module('visualizer')                                      # top-level name, found through sys.path
Visualizer, print_loss_pack, print_loss_pack_color = _vis.Visualizer, _vis.print_loss_pack, _vis.print_loss_pack_color
globals().update({_k: _v for _k, _v in vars(_importlib.import_module('vis_utils')).items() if not _k.startswith('_')})
