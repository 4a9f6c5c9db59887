"""ctypes binding of libdistr.so (C ABI: include/distr.h) + per-device context cache.

PyTorch is used only for device memory (tensors own every buffer handed to the library) and for
the current HIP stream. There is NO CPU / PyTorch fallback: if the shared library is missing, or no
MI355X is visible, every entry point raises.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, '..', 'csrc')
LIB_PATH = os.path.abspath(os.path.join(CSRC, 'libdistr.so'))

MARCHERS = {'trivial': 0, 'recursive': 1, 'pyramid_recursive': 2}
ARITH = {'f32': 0, 'bf16x6': 1, 'f16x3': 2}
EXPORTS = ['distr_version', 'distr_abi_version', 'distr_create_abi', 'distr_destroy', 'distr_last_error', 'distr_set_decoder',
           'distr_workspace_bytes', 'distr_render_forward', 'distr_render_backward', 'distr_render_normal',
           'distr_mlp_workspace_bytes', 'distr_mlp_eval', 'distr_mlp_grad', 'distr_get_render_stats',
           'distr_profile_enable', 'distr_profile_read', 'distr_debug_mlp_layer', 'distr_debug_tile_timing',
           'distr_loss_workspace_bytes', 'distr_single_loss_forward', 'distr_single_loss_backward',
           'distr_warp_loss_forward', 'distr_warp_loss_backward', 'distr_set_color_decoder', 'distr_color_eval', 'distr_debug_xchg_ts', 'distr_mlp_backward_workspace_bytes', 'distr_mlp_backward',
           'distr_profile_read_list', 'distr_get_live_counts', 'distr_color_backward',
           'distr_render_forward_batch', 'distr_render_backward_batch', 'distr_render_normal_batch', 'distr_mlp_eval_bf16x6', 'distr_mlp_eval_f16x3']

ABI_VERSION = 6                                   # DISTR_ABI_VERSION of include/distr.h this mirror was written against
MAX_VIEWS = 64                                    # DISTR_MAX_VIEWS
VIEW_GRAD_DEPTH, VIEW_GRAD_MASK, VIEW_GRAD_CAMERA = 1, 2, 4      # DISTR_VIEW_GRAD_*


class DistrError(RuntimeError):
    pass


class _Sized(C.Structure):
    """Boundary structs start with `uint32_t struct_size` = sizeof(struct) (ABI handshake, include/distr.h): set on construction."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(type(self))


class DecoderDesc(_Sized):
    _fields_ = [('struct_size', C.c_uint32), ('latent_size', C.c_int32), ('hidden', C.c_int32), ('num_linear', C.c_int32), ('latent_in', C.c_int32)]


class RenderCfg(_Sized):
    _fields_ = [
        ('struct_size', C.c_uint32),
        ('H', C.c_int32), ('W', C.c_int32),
        ('K_inv', C.c_float * 9),
        ('fx', C.c_float), ('fy', C.c_float),
        ('M', C.c_float * 9),
        ('M_normal', C.c_float * 9),
        ('march_step', C.c_int32), ('buffer_size', C.c_int32),
        ('ratio', C.c_float), ('threshold', C.c_float), ('radius', C.c_float), ('clamp_dist', C.c_float),
        ('marcher', C.c_int32),
        ('coarse_steps', C.c_int32 * 2),
        ('use_depth2normal', C.c_int32), ('normalize_normal', C.c_int32), ('want_normal', C.c_int32),
        ('grad_depth', C.c_int32), ('grad_mask', C.c_int32), ('grad_camera', C.c_int32),
        ('save_for_backward', C.c_int32),
        ('row0', C.c_int32), ('rows', C.c_int32),
        ('arith', C.c_int32),
        ('concurrent', C.c_int32),
        ('num_levels', C.c_int32),
        ('level_scale', C.c_int32 * 4),
        ('level_steps', C.c_int32 * 4),
    ]

    @property
    def band_rows(self):
        """Number of image rows this cfg renders (H unless a row band is set)."""
        return self.rows if self.rows > 0 else self.H

    def clone(self):
        c = RenderCfg()
        C.memmove(C.byref(c), C.byref(self), C.sizeof(RenderCfg))
        return c


class WarpCfg(_Sized):
    _fields_ = [('struct_size', C.c_uint32), ('H', C.c_int32), ('W', C.c_int32), ('K', C.c_float * 9), ('K_inv', C.c_float * 9), ('thres_depth', C.c_float)]


def make_warp_cfg(img_hw, intrinsic, thres_depth):
    cfg = WarpCfg()
    cfg.H, cfg.W = int(img_hw[0]), int(img_hw[1])
    K = np.asarray(intrinsic, dtype=np.float64)
    cfg.K = (C.c_float * 9)(*K.astype(np.float32).reshape(-1))
    cfg.K_inv = (C.c_float * 9)(*np.linalg.inv(K).astype(np.float32).reshape(-1))
    cfg.thres_depth = float(thres_depth)
    return cfg


class RenderStats(_Sized):
    _fields_ = [('struct_size', C.c_uint32), ('reserved', C.c_uint32), ('num_in_sphere', C.c_int64), ('num_march_launches', C.c_int64), ('num_point_evals', C.c_int64),
                ('num_valid', C.c_int64), ('num_grad_samples', C.c_int64), ('cluster_fallbacks', C.c_int64), ('f16_overflows', C.c_int64),
                ('tail_from', C.c_int64), ('tail_steals', C.c_int64)]


SOURCES = ('distr_api.hip', 'distr_inst.hip', 'distr_inst.hpp', 'distr_kernels.hpp', 'distr_mlp.hpp', 'distr_mlp_b6.hpp', 'distr_mlp_h3.hpp', 'distr_losses.hpp',
           'distr_dense_asm.hpp')
INST_GROUPS = 6            # distr_inst.hpp: DISTR_NUM_INST_GROUPS translation units of explicit kernel instantiations
HIPCC_FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fPIC']


def source_digest():
    """sha256 over the native sources libdistr.so is built from (csrc/ + include/distr.h, fixed order, name + bytes). Evidence files
    that describe the kernels (profiles/rNN_traffic.json) record it; bench.py only quotes them for the same digest."""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(CSRC, f) for f in SOURCES] + [os.path.join(_HERE, '..', '..', 'include', 'distr.h')]:
        h.update(os.path.basename(path).encode() + b'\0')
        with open(path, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def build_commands(obj_dir=None):
    """[(label, argv, output)] of the compile steps and the link step of libdistr.so: distr_api.hip (host code, launch sequences, the small
    kernels) and one translation unit per group of explicit instantiations of the big template kernels (distr_inst.hip with
    -DDISTR_INST_GROUP=n) -- independent of each other, compiled side by side -- then one link."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    obj_dir = obj_dir or os.path.join(CSRC, '_obj')
    # every unit in a directory of its own, with -save-temps=obj: the device assembly the objects were made from stays next to them
    # (<unit>/<source>-hip-amdgcn-amd-amdhsa-gfx950.s) and is what check_generated_code() reads -- the checked code IS the shipped code
    flags = HIPCC_FLAGS + ['-save-temps=obj']
    steps = [('api', [hipcc] + flags + ['-c', os.path.join(CSRC, 'distr_api.hip'), '-o', os.path.join(obj_dir, 'api', 'distr_api.o')],
              os.path.join(obj_dir, 'api', 'distr_api.o'))]
    for g in range(1, INST_GROUPS + 1):
        o = os.path.join(obj_dir, 'inst%d' % g, 'distr_inst.o')
        steps.append(('inst%d' % g, [hipcc] + flags + ['-DDISTR_INST_GROUP=%d' % g, '-c', os.path.join(CSRC, 'distr_inst.hip'), '-o', o], o))
    link = [hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB_PATH] + [s[2] for s in steps]
    return steps, link


# kernels that contain the cluster tile (hand-counted vmcnt waits, live data in fixed registers between asm statements): unit, symbol part
CLUSTER_KERNELS = (('inst1', 'k_stepILb1ELi0'), ('inst1', 'k_stepILb0ELi0'), ('inst2', 'k_tailILb1'), ('inst2', 'k_tailILb0'),
                   ('inst3', 'k_march16ILi1ELb1'), ('inst3', 'k_march16ILi1ELb0'))
NO_SCRATCH = ('k_step', 'k_march', 'k_bwd', 'k_tail')          # kernels that must not carry scratch (private segment 0): name parts


def check_generated_code(verbose=False, obj_dir=None):
    """Static checks of the code hipcc generated for THIS build (the cluster tile passes live data between asm statements in
    hard-coded registers, protected only by clobber lists -- so every build proves the compiler stayed out of them):
      * profiles/tools/vm_hazard_scan.py: no instruction touches a register an asm-issued load has not delivered (hand-counted vmcnt waits);
      * profiles/tools/check_fixed_regs.py: no compiler-generated instruction names a fixed register inside the cluster code;
      * code-object notes of libdistr.so: the march / backward kernels carry no scratch (private_segment_fixed_size 0).
    Raises DistrError with the findings; returns the per-kernel resource table."""
    import contextlib
    import importlib.util
    import io
    tools = os.path.abspath(os.path.join(_HERE, '..', '..', 'profiles', 'tools'))

    def tool(name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(tools, name + '.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m
    obj_dir = obj_dir or os.path.join(CSRC, '_obj')
    problems, log = [], []
    scan, fixed = tool('vm_hazard_scan'), tool('check_fixed_regs')
    for unit, sym in CLUSTER_KERNELS:
        asm = os.path.join(obj_dir, unit, 'distr_inst-hip-amdgcn-amd-amdhsa-gfx950.s')
        if not os.path.exists(asm):
            problems.append('%s: no device assembly at %s (build_library writes it)' % (sym, asm))
            continue
        for fn, args in ((scan.main, [asm, sym, '10']), (fixed.main, [asm, sym])):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                n = fn(args)
            log.append(buf.getvalue().strip())
            if n:
                problems.append(buf.getvalue().strip())
    res = tool('kernel_resources').resources(LIB_PATH)
    checked = [n for n in res if any(k in n for k in NO_SCRATCH)]        # (names are mangled when llvm-cxxfilt is missing: substring match)
    for name in sorted(checked):
        r = res[name]
        if r['scratch'] != 0 or r['vgpr_spill'] != 0:
            problems.append('%s: %d bytes of scratch per lane, %d spilled VGPRs (must be 0)' % (name.split('(')[0], r['scratch'], r['vgpr_spill']))
    if verbose:
        print('\n'.join(log))
        print('scratch: %d kernels of k_step* / k_march* / k_bwd* / k_tail* checked: none carries scratch' % len(checked) if not problems else 'scratch: see below')
    if problems:
        raise DistrError('generated-code checks failed:\n' + '\n'.join(problems))
    return res


def build_library(force=False, verbose=False, jobs=None, only=None):
    """Compiles csrc/ for gfx950 with hipcc (cross-compiles without a GPU): the translation units of build_commands() in parallel (`jobs`
    at a time: default min(cores, 7); DISTR_BUILD_JOBS overrides), then the link. Returns the .so path. only = labels to recompile (the other
    objects are reused as they are: local iteration on one kernel group; never used by build())."""
    srcs = [os.path.join(CSRC, f) for f in SOURCES]
    srcs.append(os.path.join(_HERE, '..', '..', 'include', 'distr.h'))
    if not force and not only and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    steps, link = build_commands()
    for st in steps:
        os.makedirs(os.path.dirname(st[2]), exist_ok=True)
    if only:
        missing = [s[0] for s in steps if s[0] not in only and not os.path.exists(s[2])]
        if missing:
            raise DistrError('build_library(only=%r): no object yet for %s' % (only, ', '.join(missing)))
        todo = [s for s in steps if s[0] in only]
    else:
        todo = list(steps)
    jobs = int(os.environ.get('DISTR_BUILD_JOBS', jobs or max(1, min(os.cpu_count() or 1, 7))))
    running, failed = [], []
    import time as _time
    t0 = _time.time()
    while todo or running:
        while todo and len(running) < jobs:
            label, cmd, out = todo.pop(0)
            if verbose:
                print(' '.join(cmd), flush=True)
            running.append((label, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        for item in list(running):
            label, pr = item
            if pr.poll() is None:
                continue
            out = pr.communicate()[0]
            running.remove(item)
            if pr.returncode != 0:
                failed.append((label, out))
                for _, other in running:
                    other.kill()
                todo = []
            elif verbose:
                print('[%s done after %.0f s]%s' % (label, _time.time() - t0, ('\n' + out) if out.strip() else ''), flush=True)
        _time.sleep(0.2)
    if failed:
        raise DistrError('hipcc failed on %s:\n%s' % (failed[0][0], failed[0][1][-6000:]))
    if verbose:
        print(' '.join(link), flush=True)
    subprocess.check_call(link, cwd=CSRC)
    return LIB_PATH


_lib = None
_lib_lock = threading.Lock()


def lib():
    """Loads libdistr.so (must have been built: __graft_entry__.build() / distr.binding.build_library())."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise DistrError('libdistr.so not found at %s -- build it with `python __graft_entry__.py build` '
                                 '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
            L = C.CDLL(LIB_PATH)
            vp, fp, u8p = C.c_void_p, C.c_void_p, C.c_void_p   # device pointers travel as integers
            L.distr_version.restype = C.c_char_p
            try:
                L.distr_abi_version.restype = C.c_uint32
                got = int(L.distr_abi_version())
            except AttributeError:
                got = None
            if got != ABI_VERSION:      # a stale libdistr.so next to newer Python (or the reverse): refuse before any struct crosses
                raise DistrError('libdistr.so at %s implements ABI %s, this binding was written for ABI %d: rebuild it '
                                 '(python __graft_entry__.py build)' % (LIB_PATH, got, ABI_VERSION))
            L.distr_create_abi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_uint32]
            L.distr_destroy.argtypes = [vp]
            L.distr_destroy.restype = None
            L.distr_last_error.argtypes = [vp]
            L.distr_last_error.restype = C.c_char_p
            L.distr_set_decoder.argtypes = [vp, C.POINTER(DecoderDesc), C.POINTER(C.c_float), C.c_size_t]
            L.distr_workspace_bytes.argtypes = [vp, C.POINTER(RenderCfg), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
            L.distr_render_forward.argtypes = [vp, C.POINTER(RenderCfg), fp, fp, fp, fp, u8p, fp, fp, fp, vp, C.c_size_t, vp]
            L.distr_render_backward.argtypes = [vp, C.POINTER(RenderCfg), vp, C.c_size_t, fp, fp, fp, fp, fp, fp, fp, vp, C.c_size_t, vp]
            L.distr_render_normal.argtypes = [vp, C.POINTER(RenderCfg), fp, fp, fp, fp, u8p, fp, vp, C.c_size_t, vp]
            L.distr_render_forward_batch.argtypes = [vp, C.POINTER(RenderCfg), C.c_int32, C.POINTER(C.c_int32), fp, C.c_int64, fp, fp,
                                                     fp, u8p, fp, fp, fp, vp, C.c_size_t, vp]
            L.distr_render_backward_batch.argtypes = [vp, C.POINTER(RenderCfg), C.c_int32, vp, C.c_size_t, fp, fp, fp, fp, fp, fp, fp,
                                                      vp, C.c_size_t, vp]
            L.distr_render_normal_batch.argtypes = [vp, C.POINTER(RenderCfg), C.c_int32, fp, C.c_int64, fp, fp, fp, u8p, fp, vp,
                                                    C.c_size_t, vp]
            L.distr_mlp_workspace_bytes.argtypes = [C.c_int64]
            L.distr_mlp_workspace_bytes.restype = C.c_size_t
            L.distr_mlp_eval.argtypes = [vp, fp, fp, C.c_int64, C.c_float, fp, vp, C.c_size_t, vp]
            L.distr_mlp_eval_bf16x6.argtypes = [vp, fp, fp, C.c_int64, C.c_float, fp, vp, C.c_size_t, vp]
            L.distr_mlp_eval_f16x3.argtypes = [vp, fp, fp, C.c_int64, C.c_float, fp, vp, C.c_size_t, vp]
            L.distr_mlp_grad.argtypes = [vp, fp, fp, C.c_int64, fp, fp, vp, C.c_size_t, vp]
            L.distr_debug_mlp_layer.argtypes = [vp, fp, fp, C.c_int64, C.c_int, fp, vp, C.c_size_t, vp]
            L.distr_debug_tile_timing.argtypes = [vp, fp, fp, C.c_int64, fp, vp, vp, C.c_size_t, vp]
            L.distr_get_render_stats.argtypes = [vp, C.POINTER(RenderCfg), vp, C.POINTER(RenderStats), vp]
            L.distr_profile_enable.argtypes = [vp, C.c_int]
            L.distr_profile_read.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), vp]
            L.distr_profile_read_list.argtypes = [vp, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64), vp]
            L.distr_get_live_counts.argtypes = [vp, C.POINTER(RenderCfg), vp, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), vp]
            L.distr_loss_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
            L.distr_loss_workspace_bytes.restype = C.c_size_t
            L.distr_single_loss_forward.argtypes = [vp, C.c_int32, C.c_int32, fp, fp, u8p, fp, fp, fp, u8p, C.c_float, fp, vp, C.c_size_t, vp]
            L.distr_single_loss_backward.argtypes = [vp, C.c_int32, C.c_int32, fp, fp, u8p, fp, fp, fp, u8p, C.c_float, fp, fp, fp, fp, fp, vp]
            L.distr_warp_loss_forward.argtypes = [vp, C.POINTER(WarpCfg), fp, u8p, fp, fp, fp, fp, fp, fp, fp, fp, u8p, fp, fp, vp, C.c_size_t, vp]
            L.distr_warp_loss_backward.argtypes = [vp, C.POINTER(WarpCfg), fp, u8p, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, fp, vp, C.c_size_t, vp]
            L.distr_set_color_decoder.argtypes = [vp, C.POINTER(DecoderDesc), C.POINTER(C.c_float), C.c_size_t]
            L.distr_color_eval.argtypes = [vp, fp, fp, C.c_int64, fp, vp, C.c_size_t, vp]
            L.distr_color_backward.argtypes = [vp, fp, fp, C.c_int64, fp, fp, fp, vp, C.c_size_t, vp]
            L.distr_debug_xchg_ts.argtypes = [vp, vp, C.POINTER(C.c_int64)]
            L.distr_mlp_backward_workspace_bytes.argtypes = [C.c_int64]
            L.distr_mlp_backward_workspace_bytes.restype = C.c_size_t
            L.distr_mlp_backward.argtypes = [vp, fp, fp, C.c_int64, fp, C.c_float, fp, fp, vp, C.c_size_t, vp]
            _lib = L
    return _lib


_concurrent = threading.local()


def concurrent_renders():
    """True inside a `concurrent_section()` of this thread: the caller is issuing several renders on different streams at once."""
    return getattr(_concurrent, 'depth', 0) > 0


class concurrent_section(object):
    """`with binding.concurrent_section():` around renders that are issued on a pool of streams -- every cfg built inside carries
    `concurrent = 1` (distr_render_cfg.concurrent: the march tails of those renders do not go sticky, they share the chip)."""

    def __enter__(self):
        _concurrent.depth = getattr(_concurrent, 'depth', 0) + 1
        return self

    def __exit__(self, *exc):
        _concurrent.depth -= 1
        return False


def make_cfg(img_hw, intrinsic, march_step=50, buffer_size=5, ratio=1.5, threshold=5e-5, radius=1.0, clamp_dist=0.1,
             marcher='pyramid_recursive', coarse_steps=(3, 3), transform_matrix=None, use_transform=True,
             use_depth2normal=False, normalize_normal=True, want_normal=True,
             grad_depth=True, grad_mask=True, grad_camera=True, band=None, arith='f32', scale_list=None, march_step_list=None):
    """Host-side part of SDFRenderer.__init__ (core/sdfrenderer/renderer.py:13-59) as a C struct. scale_list / march_step_list (the
    reference's keywords, coarsest level first): the general pyramid (num_levels / level_scale / level_steps of include/distr.h); without
    them coarse_steps describes the default [4, 2, 1] (or [2, 1]: coarse_steps = (s, 0))."""
    cfg = RenderCfg()
    cfg.H, cfg.W = int(img_hw[0]), int(img_hw[1])
    K = np.asarray(intrinsic, dtype=np.float64)
    Kinv = np.linalg.inv(K).astype(np.float32)
    cfg.K_inv = (C.c_float * 9)(*Kinv.reshape(-1))
    cfg.fx, cfg.fy = float(np.float32(K[0, 0])), float(np.float32(K[1, 1]))
    if transform_matrix is None:
        transform_matrix = np.array([[1., 0., 0.], [0., 0., -1.], [0., 1., 0.]])
    Mm = np.asarray(transform_matrix, dtype=np.float32)
    if Mm.shape != (3, 3):
        raise NotImplementedError('only 3x3 transform matrices are supported (the reference\'s 3x4 sim3 inverse path '
                                  'hits an un-imported pdb.set_trace(), renderer.py:116)')
    # use_transform=False removes the matrix from the sample points only; render_normal transforms the normals unconditionally
    # (renderer.py:895 vs :899, golden G24)
    cfg.M_normal = (C.c_float * 9)(*Mm.reshape(-1))
    if not use_transform:
        Mm = np.eye(3, dtype=np.float32)
    cfg.M = (C.c_float * 9)(*Mm.reshape(-1))
    cfg.march_step, cfg.buffer_size = int(march_step), int(buffer_size)
    cfg.ratio, cfg.threshold, cfg.radius, cfg.clamp_dist = float(ratio), float(threshold), float(radius), float(clamp_dist)
    if marcher not in MARCHERS:
        raise ValueError('Error! Invalid type of ray marching: {}.'.format(marcher))
    cfg.marcher = MARCHERS[marcher]
    cfg.coarse_steps = (C.c_int32 * 2)(int(coarse_steps[0]), int(coarse_steps[1]))
    if scale_list is not None:
        sl, ms = [int(v) for v in scale_list], [int(v) for v in (march_step_list if march_step_list is not None else [])]
        if any(float(v) != int(v) for v in scale_list) or not 2 <= len(sl) <= 4 or len(ms) != len(sl):
            raise NotImplementedError('pyramid scale_list=%r / march_step_list=%r: 2..4 integer scales with one step count each' % (scale_list, march_step_list))
        cfg.num_levels = len(sl)
        cfg.level_scale = (C.c_int32 * 4)(*(sl + [0] * (4 - len(sl))))
        cfg.level_steps = (C.c_int32 * 4)(*(ms[:-1] + [0] * (5 - len(sl))))
    cfg.use_depth2normal, cfg.normalize_normal, cfg.want_normal = int(use_depth2normal), int(normalize_normal), int(want_normal)
    cfg.grad_depth, cfg.grad_mask, cfg.grad_camera = int(grad_depth), int(grad_mask), int(grad_camera)
    cfg.save_for_backward = 1
    if band is not None:       # (row0, rows): render only these image rows (strong scaling of one view, include/distr.h)
        cfg.row0, cfg.rows = int(band[0]), int(band[1])
    if arith not in ARITH:
        raise ValueError("arith must be one of %s" % sorted(ARITH))
    cfg.concurrent = 1 if concurrent_renders() else 0      # inside a stream pool (concurrent_section): no sticky tail launches
    cfg.arith = ARITH[arith]     # 'f32': exact (default); 'bf16x6' / 'f16x3': split-bf16 / split-f16 march tiles (DISTR_ARITH_*)
    return cfg


class Context(object):
    """One distr_ctx (device-bound handle holding the packed decoder)."""

    def __init__(self, device_index=0):
        import torch
        if not torch.cuda.is_available():
            raise DistrError('no HIP device visible: the MI355X kernels cannot run (no CPU fallback exists)')
        self.device_index = int(device_index)
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.distr_create_abi(C.byref(h), self.device_index, ABI_VERSION)
        self.h = h
        if rc != 0:
            msg = self.L.distr_last_error(h).decode() if h else 'distr_create failed'
            raise DistrError(msg)
        self._decoder_key = None

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.L.distr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise DistrError('libdistr error %d: %s' % (rc, self.L.distr_last_error(self.h).decode()))

    def set_decoder(self, flat_weights):
        w = np.ascontiguousarray(flat_weights, dtype=np.float32)
        desc = DecoderDesc(latent_size=256, hidden=512, num_linear=9, latent_in=4)
        self.check(self.L.distr_set_decoder(self.h, C.byref(desc), w.ctypes.data_as(C.POINTER(C.c_float)), w.size))

    def set_color_decoder(self, flat_weights, latent_size):
        w = np.ascontiguousarray(flat_weights, dtype=np.float32)
        desc = DecoderDesc(latent_size=int(latent_size), hidden=512, num_linear=9, latent_in=4)
        self.check(self.L.distr_set_color_decoder(self.h, C.byref(desc), w.ctypes.data_as(C.POINTER(C.c_float)), w.size))

    def workspace_bytes(self, cfg):
        f, b = C.c_size_t(), C.c_size_t()
        self.check(self.L.distr_workspace_bytes(self.h, C.byref(cfg), C.byref(f), C.byref(b)))
        return f.value, b.value

    def stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device_index).cuda_stream)

    def profile_enable(self, on=True):
        self.check(self.L.distr_profile_enable(self.h, 1 if on else 0))

    def profile_read(self):
        n, ms = C.c_int64(), C.c_double()
        self.check(self.L.distr_profile_read(self.h, C.byref(n), C.byref(ms), self.stream()))
        return n.value, ms.value

    def profile_read_list(self, cap=8192):
        """Per-launch kernel ms of the bracketed march launches since the last profile_read (launch order)."""
        buf, n = (C.c_float * cap)(), C.c_int64()
        self.check(self.L.distr_profile_read_list(self.h, buf, cap, C.byref(n), self.stream()))
        return [buf[i] for i in range(min(cap, n.value))]

    def live_counts(self, cfg, ws, cap=4096):
        """Rays evaluated by every march launch of the forward that used `ws` (launch order)."""
        buf, n = (C.c_int32 * cap)(), C.c_int32()
        self.check(self.L.distr_get_live_counts(self.h, C.byref(cfg), C.c_void_p(ws.data_ptr()), buf, cap, C.byref(n), self.stream()))
        return [buf[i] for i in range(min(cap, n.value))]

    def render_stats(self, cfg, ws):
        st = RenderStats()
        self.check(self.L.distr_get_render_stats(self.h, C.byref(cfg), C.c_void_p(ws.data_ptr()), C.byref(st), self.stream()))
        return {k: getattr(st, k) for k, _ in RenderStats._fields_ if k not in ('struct_size', 'reserved')}


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())
