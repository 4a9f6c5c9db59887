"""STUB (test infrastructure, none of the reference's code): a package that makes its own directory importable and then pulls its
modules in as TOP-LEVEL names -- the import style the drop-in has to coexist with (tests/test_host_logic.py)."""
import importlib as _importlib
import pathlib as _pathlib
import sys as _sys

_HERE = str(_pathlib.Path(__file__).resolve().parent)
if _HERE not in _sys.path:
    _sys.path.append(_HERE)
for _name in ('mesh_dataset', 'loader_single', 'loader_multi_pmodata'):          # flat module names, resolved through sys.path
    _mod = _importlib.import_module(_name)
    globals().update({_k: _v for _k, _v in vars(_mod).items() if not _k.startswith('_')})
