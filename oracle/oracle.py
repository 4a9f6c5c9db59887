"""TEST INFRASTRUCTURE -- ctypes wrapper around oracle/_build/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (dist-renderer_amd/) must never import it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'liboracle.so')

MARCHERS = {'trivial': 0, 'recursive': 1, 'pyramid_recursive': 2}


class OrcCfg(C.Structure):
    _fields_ = [
        ('H', C.c_int32), ('W', C.c_int32),
        ('K_inv', C.c_float * 9),
        ('fx', C.c_float), ('fy', C.c_float),
        ('M', C.c_float * 9),
        ('Mn', C.c_float * 9),
        ('march_step', C.c_int32), ('buffer_size', C.c_int32),
        ('ratio', C.c_float), ('threshold', C.c_float), ('radius', C.c_float), ('clamp_dist', C.c_float),
        ('marcher', C.c_int32),
        ('coarse_steps', C.c_int32 * 2),
        ('use_depth2normal', C.c_int32), ('normalize_normal', C.c_int32), ('want_normal', C.c_int32),
        ('grad_depth', C.c_int32), ('grad_mask', C.c_int32), ('grad_camera', C.c_int32),
        ('num_levels', C.c_int32), ('level_scale', C.c_int32 * 4), ('level_steps', C.c_int32 * 4),
    ]


def build(force=False):
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, 'distr_oracle.cpp')):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [fp, C.c_int64]
        L.orc_create_color.restype = C.c_void_p
        L.orc_create_color.argtypes = [fp, C.c_int64, C.c_int]
        L.orc_color_eval.argtypes = [C.c_void_p, fp, fp, C.c_int64, fp]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_latent_consts.argtypes = [C.c_void_p, fp, fp, fp]
        L.orc_mlp_eval.argtypes = [C.c_void_p, fp, fp, C.c_int64, C.c_float, fp]
        L.orc_mlp_grad.argtypes = [C.c_void_p, fp, fp, C.c_int64, fp, fp]
        L.orc_mlp_layer.argtypes = [C.c_void_p, fp, fp, C.c_int64, C.c_int, fp]
        L.orc_tanh.restype = C.c_float
        L.orc_tanh.argtypes = [C.c_float]
        L.orc_render_forward.restype = C.c_void_p
        L.orc_render_forward.argtypes = [C.c_void_p, C.POINTER(OrcCfg), fp, fp, fp, fp, C.POINTER(C.c_uint8), fp, fp, fp]
        L.orc_state_free.argtypes = [C.c_void_p]
        L.orc_state_num_evals.restype = C.c_int64
        L.orc_state_num_evals.argtypes = [C.c_void_p]
        L.orc_state_num_steps.restype = C.c_int64
        L.orc_state_num_steps.argtypes = [C.c_void_p]
        L.orc_state_live_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.orc_state_num_inside.restype = C.c_int64
        L.orc_state_num_inside.argtypes = [C.c_void_p]
        L.orc_state_init_z.argtypes = [C.c_void_p, fp]
        L.orc_render_backward.restype = C.c_int64
        L.orc_render_backward.argtypes = [C.c_void_p, C.c_void_p, fp, fp, fp, fp, fp, fp, fp]
        _lib = L
    return _lib


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def make_cfg(H, W, intrinsic, march_step=50, buffer_size=5, ratio=1.5, threshold=5e-5, radius=1.0, clamp_dist=0.1,
             marcher='pyramid_recursive', coarse_steps=(3, 3), transform_matrix=None, use_transform=True,
             use_depth2normal=False, normalize_normal=True, want_normal=True,
             grad_depth=True, grad_mask=True, grad_camera=True, scale_list=None, march_step_list=None):
    cfg = OrcCfg()
    cfg.H, cfg.W = H, W
    K = np.asarray(intrinsic, dtype=np.float64)
    Kinv = np.linalg.inv(K).astype(np.float32)        # renderer.py:161-164
    cfg.K_inv = (C.c_float * 9)(*Kinv.reshape(-1))
    cfg.fx, cfg.fy = float(np.float32(K[0, 0])), float(np.float32(K[1, 1]))
    if transform_matrix is None:
        transform_matrix = np.array([[1., 0., 0.], [0., 0., -1.], [0., 1., 0.]])   # renderer.py:45
    Mfull = np.asarray(transform_matrix, dtype=np.float32)
    Mm = Mfull if use_transform else np.eye(3, dtype=np.float32)      # the points' (inverse) transform: renderer.py:895, 202-223
    cfg.M = (C.c_float * 9)(*Mm.reshape(-1))
    cfg.Mn = (C.c_float * 9)(*Mfull.reshape(-1))                      # the normals' transform is unconditional: renderer.py:899 (golden G24)
    cfg.march_step, cfg.buffer_size = march_step, buffer_size
    cfg.ratio, cfg.threshold, cfg.radius, cfg.clamp_dist = ratio, threshold, radius, clamp_dist
    cfg.marcher = MARCHERS[marcher]
    cfg.coarse_steps = (C.c_int32 * 2)(*coarse_steps)
    if scale_list is not None:       # the general pyramid (renderer.py:713-805), coarsest level first like the reference's keywords
        sl, ms = [int(v) for v in scale_list], [int(v) for v in march_step_list]
        cfg.num_levels = len(sl)
        cfg.level_scale = (C.c_int32 * 4)(*(sl + [0] * (4 - len(sl))))
        cfg.level_steps = (C.c_int32 * 4)(*(ms[:-1] + [0] * (5 - len(sl))))
    cfg.use_depth2normal, cfg.normalize_normal, cfg.want_normal = int(use_depth2normal), int(normalize_normal), int(want_normal)
    cfg.grad_depth, cfg.grad_mask, cfg.grad_camera = int(grad_depth), int(grad_mask), int(grad_camera)
    return cfg


def flatten_decoder(Ws, bs):
    return np.concatenate([np.concatenate([np.asarray(W, np.float32).reshape(-1), np.asarray(b, np.float32).reshape(-1)])
                           for W, b in zip(Ws, bs)]).astype(np.float32)


class RenderState(object):
    def __init__(self, oracle, ptr, cfg):
        self.oracle, self.ptr, self.cfg = oracle, ptr, cfg

    def __del__(self):
        try:
            if self.ptr:
                lib().orc_state_free(self.ptr)
                self.ptr = None
        except Exception:      # interpreter shutdown
            pass

    @property
    def num_evals(self):
        return lib().orc_state_num_evals(self.ptr)

    @property
    def num_inside(self):
        return lib().orc_state_num_inside(self.ptr)

    @property
    def live_counts(self):
        n = lib().orc_state_num_steps(self.ptr)
        out = np.zeros(n, dtype=np.int64)
        if n:
            lib().orc_state_live_counts(self.ptr, out.ctypes.data_as(C.POINTER(C.c_int64)))
        return out

    @property
    def init_z(self):
        out = np.zeros(self.cfg.H * self.cfg.W, dtype=np.float32)
        lib().orc_state_init_z(self.ptr, _fp(out))
        return out

    def backward(self, g_zdepth=None, g_min_sdf=None, g_depth=None, g_normal=None):
        gz, gq, gd, gn = _f32(g_zdepth), _f32(g_min_sdf), _f32(g_depth), _f32(g_normal)
        g_lat = np.zeros(256, np.float32)
        g_R = np.zeros(9, np.float32)
        g_T = np.zeros(3, np.float32)
        ns = lib().orc_render_backward(self.oracle.h, self.ptr, _fp(gz), _fp(gq), _fp(gd), _fp(gn), _fp(g_lat), _fp(g_R), _fp(g_T))
        return g_lat.reshape(1, 256), g_R.reshape(3, 3), g_T, ns


class Oracle(object):
    """CPU restatement of Decoder.inference + SDFRenderer.render*."""

    def __init__(self, Ws, bs):
        flat = flatten_decoder(Ws, bs)
        self.h = lib().orc_create(_fp(flat), flat.size)
        if not self.h:
            raise ValueError('decoder is not the DeepSDF 8x512 / latent_in=[4] architecture')

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                lib().orc_destroy(self.h)
                self.h = None
        except Exception:      # interpreter shutdown
            pass

    def latent_consts(self, latent):
        lat = _f32(latent).reshape(-1)
        c0, c4 = np.zeros(512, np.float32), np.zeros(512, np.float32)
        lib().orc_latent_consts(self.h, _fp(lat), _fp(c0), _fp(c4))
        return c0, c4

    def decode_sdf(self, latent, xyz, clamp_dist=None):
        lat, x = _f32(latent).reshape(-1), _f32(xyz).reshape(-1, 3)
        out = np.zeros(x.shape[0], np.float32)
        lib().orc_mlp_eval(self.h, _fp(lat), _fp(x), x.shape[0], -1.0 if clamp_dist is None else float(clamp_dist), _fp(out))
        return out

    def decode_sdf_and_gradient(self, latent, xyz):
        lat, x = _f32(latent).reshape(-1), _f32(xyz).reshape(-1, 3)
        sdf = np.zeros(x.shape[0], np.float32)
        g = np.zeros((x.shape[0], 3), np.float32)
        lib().orc_mlp_grad(self.h, _fp(lat), _fp(x), x.shape[0], _fp(sdf), _fp(g))
        return sdf, g

    def layer_activations(self, latent, xyz, layer):
        lat, x = _f32(latent).reshape(-1), _f32(xyz).reshape(-1, 3)
        out = np.zeros((x.shape[0], 512), np.float32)
        lib().orc_mlp_layer(self.h, _fp(lat), _fp(x), x.shape[0], int(layer), _fp(out))
        return out

    def render(self, cfg, latent, R, T):
        """Returns dict(zdepth, mask, min_sdf, depth, normal) as numpy arrays + 'state' for backward."""
        P = cfg.H * cfg.W
        lat, Rr, Tt = _f32(latent).reshape(-1), _f32(R).reshape(-1), _f32(T).reshape(-1)
        zdepth = np.zeros(P, np.float32)
        mask = np.zeros(P, np.uint8)
        min_sdf = np.zeros(P, np.float32)
        depth = np.zeros(P, np.float32)
        normal = np.zeros(P * 3, np.float32)
        ptr = lib().orc_render_forward(self.h, C.byref(cfg), _fp(lat), _fp(Rr), _fp(Tt), _fp(zdepth),
                                       mask.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(min_sdf), _fp(depth), _fp(normal))
        return dict(zdepth=zdepth, mask=mask, min_sdf=min_sdf, depth=depth.reshape(cfg.H, cfg.W),
                    normal=normal.reshape(cfg.H, cfg.W, 3), state=RenderState(self, ptr, cfg))


class ColorOracle(object):
    """CPU restatement of decode_color (core/utils/decoder_utils.py:94-112) on the colour decoder
    (latent = 256 + color_size, last_dim = 3)."""

    def __init__(self, Ws, bs):
        flat = flatten_decoder(Ws, bs)
        self.nlat = int(np.asarray(Ws[0]).shape[1]) - 3
        self.h = lib().orc_create_color(_fp(flat), flat.size, self.nlat)
        if not self.h:
            raise ValueError('not a DeepSDF 8x512-shaped colour decoder')

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                lib().orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def decode_color(self, color_code, shape_code, xyz):
        lat = np.concatenate([_f32(shape_code).reshape(-1), _f32(color_code).reshape(-1)])
        assert lat.size == self.nlat
        x = _f32(xyz).reshape(-1, 3)
        out = np.zeros((x.shape[0], 3), np.float32)
        lib().orc_color_eval(self.h, _fp(lat), _fp(x), x.shape[0], _fp(out))
        return out
