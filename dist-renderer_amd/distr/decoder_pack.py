"""DeepSDF decoder -> canonical flat f32 weight buffer for `distr_set_decoder` (include/distr.h).

Host-side counterpart of `load_decoder` (core/utils/decoder_utils.py:7-51) + the layer
construction of `Decoder.__init__` (core/graph/deep_sdf_decoder.py:19-73): takes a module or a
state_dict, folds weight-norm (`lin{l}.weight_g/.weight_v` -> W = g * v / ||v||_row), validates that
the architecture is the one the HIP kernels are specialised for (DeepSDF '8x512', latent 256,
latent_in=[4], ReLU, final tanh, no LayerNorm / xyz_in_all / use_tanh; latent_dropout only in eval mode, where it is the identity), and
returns one contiguous float32 array: for l in 0..8: W_l row-major (out,in) followed by b_l.
The LDS/MFMA-fragment packing itself is done natively inside the library.
"""
import numpy as np

from . import fixture

_SHAPES = fixture.layer_shapes()


class UnsupportedDecoder(NotImplementedError):
    pass


def _np(t):
    if hasattr(t, 'detach'):
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def effective_weights(state_dict):
    """state_dict (possibly 'module.'-prefixed, possibly weight-normed) -> ([W_l f32], [b_l f32])."""
    sd = {}
    for k, v in state_dict.items():
        k = k[len('module.'):] if k.startswith('module.') else k
        sd[k] = _np(v)
    if any(k.startswith('bn') for k in sd):
        raise UnsupportedDecoder('LayerNorm decoders (weight_norm=False with norm_layers) are not supported')
    Ws, bs = [], []
    l = 0
    while ('lin%d.bias' % l) in sd:
        if ('lin%d.weight_v' % l) in sd:
            v = sd['lin%d.weight_v' % l].astype(np.float32)
            g = sd['lin%d.weight_g' % l].astype(np.float32).reshape(-1, 1)
            nrm = np.sqrt((v.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)
            W = (v * (g / nrm)).astype(np.float32)          # torch._weight_norm: v * (g / ||v||)
        elif ('lin%d.parametrizations.weight.original1' % l) in sd:   # new-style parametrization
            v = sd['lin%d.parametrizations.weight.original1' % l].astype(np.float32)
            g = sd['lin%d.parametrizations.weight.original0' % l].astype(np.float32).reshape(-1, 1)
            nrm = np.sqrt((v ** 2).sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)
            W = (v * (g / nrm)).astype(np.float32)
        else:
            W = sd['lin%d.weight' % l].astype(np.float32)
        Ws.append(np.ascontiguousarray(W))
        bs.append(np.ascontiguousarray(sd['lin%d.bias' % l].astype(np.float32)))
        l += 1
    return Ws, bs


def validate(Ws, bs):
    if len(Ws) != len(_SHAPES):
        raise UnsupportedDecoder('expected %d linear layers, got %d' % (len(_SHAPES), len(Ws)))
    for l, (W, b) in enumerate(zip(Ws, bs)):
        if tuple(W.shape) != _SHAPES[l] or b.shape != (_SHAPES[l][0],):
            raise UnsupportedDecoder('lin%d has shape %s, kernels are specialised for %s (DeepSDF 8x512, '
                                     'latent 256, latent_in=[4], last_dim=1)' % (l, W.shape, _SHAPES[l]))


def validate_color(Ws, bs):
    """Colour decoder (load_decoder(color_size=cs), decoder_utils.py:16-24): returns its latent length 256 + cs."""
    if len(Ws) != 9:
        raise UnsupportedDecoder('expected 9 linear layers, got %d' % len(Ws))
    cs = Ws[0].shape[1] - 3 - fixture.LATENT_SIZE
    if cs <= 0:
        raise UnsupportedDecoder('colour decoder must take latent = 256 + color_size (> 256), got %d' % (Ws[0].shape[1] - 3))
    want = fixture.color_layer_shapes(cs)
    for l, (W, b) in enumerate(zip(Ws, bs)):
        if tuple(W.shape) != want[l] or b.shape != (want[l][0],):
            raise UnsupportedDecoder('colour lin%d has shape %s, expected %s (DeepSDF 8x512 with latent 256+%d, latent_in=[4], '
                                     'last_dim=3)' % (l, W.shape, want[l], cs))
    return fixture.LATENT_SIZE + cs


def flatten_color(Ws, bs):
    """-> (flat f32 array for distr_set_color_decoder, latent length)."""
    nlat = validate_color(Ws, bs)
    parts = []
    for W, b in zip(Ws, bs):
        parts.append(np.asarray(W, np.float32).reshape(-1))
        parts.append(np.asarray(b, np.float32).reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32), nlat


def pack_color_module(decoder_color):
    check_module_flags(decoder_color)
    Ws, bs = effective_weights(decoder_color.state_dict())
    return flatten_color(Ws, bs)


def check_module_flags(decoder):
    """Rejects constructor options of core/graph/deep_sdf_decoder.py:19-73 that change the math."""
    d = decoder.module if hasattr(decoder, 'module') else decoder
    if getattr(d, 'xyz_in_all', None):
        raise UnsupportedDecoder('xyz_in_all decoders are not supported')
    if getattr(d, 'use_tanh', False):
        raise UnsupportedDecoder('use_tanh decoders are not supported')
    if getattr(d, 'latent_dropout', False) and getattr(d, 'training', False):
        # F.dropout(latent, training=self.training), deep_sdf_decoder.py:84-87: the identity in eval mode (what every driver runs:
        # SDFRenderer(is_eval=True) calls decoder.eval()), stochastic in training mode
        raise UnsupportedDecoder('latent_dropout decoder in training mode: the fused kernels evaluate the deterministic (eval) network only '
                                 '-- call decoder.eval() (SDFRenderer(is_eval=True) does)')
    li = tuple(getattr(d, 'latent_in', (4,)))
    if li != (4,):
        raise UnsupportedDecoder('latent_in=%s is not supported (only [4])' % (li,))
    if not hasattr(d, 'th'):
        raise UnsupportedDecoder('decoder without the final tanh is not supported')


def flatten(Ws, bs):
    validate(Ws, bs)
    parts = []
    for W, b in zip(Ws, bs):
        parts.append(np.asarray(W, np.float32).reshape(-1))
        parts.append(np.asarray(b, np.float32).reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


def pack_module(decoder):
    """nn.Module (optionally DataParallel-wrapped) -> flat f32 array."""
    check_module_flags(decoder)
    Ws, bs = effective_weights(decoder.state_dict())
    return flatten(Ws, bs)


def fixture_state_dict(Ws, bs, weight_norm=False):
    """state_dict (numpy) of the fixture decoder in plain or DeepSDF weight_norm form."""
    sd = {}
    for l, (W, b) in enumerate(zip(Ws, bs)):
        if weight_norm and l < 8:
            sd['lin%d.weight_v' % l] = W
            sd['lin%d.weight_g' % l] = np.sqrt((W.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32)).astype(np.float32)
        else:
            sd['lin%d.weight' % l] = W
        sd['lin%d.bias' % l] = b
    return sd
