"""Runs one of the reference's drivers UNCHANGED on this build:

    PYTHONPATH=/path/to/repo/dist-renderer_amd python -m distr.launch /path/to/DIST-Renderer/run_single_shape.py --gpu 0 ...

A script started as `python run_single_shape.py` gets its own directory as sys.path[0], in front of PYTHONPATH, so the
reference's `core` would win. This launcher starts the script with runpy instead: sys.path[0] is this build's package root, the
script's directory follows (the drivers append it themselves, run_single_shape.py:5), and `core.*` resolves as described in
core/_dropin.py -- mirrored modules here, everything else in the reference checkout.

    python -m distr.launch --arith f16x3 /path/to/run_single_shape.py ...

selects one of the opt-in arithmetics for every SDFRenderer the driver constructs without an `arith=` of its own (it sets DISTR_ARITH,
core/sdfrenderer/renderer.py::default_arith); without the option the drivers run in exact f32.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) >= 2 and argv[0] == '--arith':
        from distr import binding
        if argv[1] not in binding.ARITH:
            raise SystemExit('--arith must be one of %s' % sorted(binding.ARITH))
        os.environ['DISTR_ARITH'] = argv[1]
        argv = argv[2:]
    if not argv:
        raise SystemExit('usage: python -m distr.launch [--arith f32|bf16x6|f16x3] <driver.py> [driver arguments]')
    script = os.path.abspath(argv[0])
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:] = [pkg_root] + [p for p in sys.path if os.path.abspath(p or os.getcwd()) != pkg_root]
    script_dir = os.path.dirname(script)
    if script_dir not in sys.path:
        sys.path.append(script_dir)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
