"""decode_sdf / decode_sdf_gradient / load_decoder with the reference's signatures
(core/utils/decoder_utils.py:7-92), evaluated by the fused MFMA decoder kernel instead of nine ATen GEMMs.
No (n,259) latent-concatenated input is ever materialised (decoder_utils.py:61-62) and MAX_POINTS chunking is
unnecessary (accepted and ignored).
"""
import json
import os

import torch

from distr import functions


def _engine(decoder, ref_tensor):
    dev = ref_tensor.device
    if dev.type != 'cuda':
        raise RuntimeError('decode_sdf: tensors must be on the GPU (no CPU path in this build)')
    return functions.get_engine(decoder, dev.index if dev.index is not None else torch.cuda.current_device())


def load_decoder(experiment_directory, checkpoint_num=None, color_size=None, experiment_directory_color=None, parallel=True):
    """specs.json + ModelParameters/<ckpt>.pth -> Decoder (reference: decoder_utils.py:7-51). With `color_size` the
    colour decoder is built instead: latent = CodeLength + color_size, dims[3] += color_size, last_dim = 3, weights from
    `experiment_directory_color` (saved without the DataParallel 'module.' prefix, decoder_utils.py:35-42)."""
    from core.graph.deep_sdf_decoder import Decoder
    specs_filename = os.path.join(experiment_directory, 'specs.json')
    if not os.path.isfile(specs_filename):
        raise Exception('The experiment directory does not include specifications file "specs.json"')
    with open(specs_filename) as f:
        specs = json.load(f)
    net = dict(specs['NetworkSpecs'])
    if color_size is not None:
        net['dims'] = list(net['dims'])
        net['dims'][3] = net['dims'][3] + color_size
        decoder = Decoder(specs['CodeLength'] + color_size, last_dim=3, **net)
    else:
        decoder = Decoder(specs['CodeLength'], **net)
    if parallel:
        decoder = torch.nn.DataParallel(decoder)
    if checkpoint_num is not None:
        root = experiment_directory_color if color_size is not None else experiment_directory
        state = torch.load(os.path.join(root, 'ModelParameters', checkpoint_num + '.pth'), map_location='cpu')
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state['model_state_dict'].items()}
        if parallel:
            sd = {'module.' + k: v for k, v in sd.items()}
        decoder.load_state_dict(sd)
    return decoder


def decode_sdf(decoder, latent_vector, points, clamp_dist=0.1, MAX_POINTS=100000, no_grad=False, arith='f32'):
    """(n,3) points -> (n,1) SDF, optionally clamped (decoder_utils.py:53-74). Differentiable w.r.t. the latent code and
    the points unless `no_grad` (fused backward: distr_mlp_backward); the decoder weights are frozen. `arith` (not in the
    reference): 'f32' = exact f32 MFMA (default); 'bf16x6' / 'f16x3' = split-bf16 / split-f16 arithmetic, forward only, f32-equivalent
    but not bit-identical (distr_mlp_eval_bf16x6 / distr_mlp_eval_f16x3; the latter needs activations below 1023 and returns NaN otherwise)."""
    if latent_vector is None:
        raise NotImplementedError('latent_vector=None (decoder_utils.py:58-59) is not supported')
    eng = _engine(decoder, points)
    if (not no_grad) and torch.is_grad_enabled() and (latent_vector.requires_grad or points.requires_grad):
        if arith != 'f32':
            raise NotImplementedError("arith=%r is forward-only: call decode_sdf(..., no_grad=True)" % arith)
        return functions.mlp_eval_autograd(eng, latent_vector, points, clamp_dist)
    return functions.mlp_eval(eng, latent_vector, points, clamp_dist, arith=arith)


def decode_sdf_gradient(decoder, latent_vector, points, clamp_dist=0.1, MAX_POINTS=100000, no_grad=False):
    """d sdf / d points, (n,3). Keeps the reference's value semantics: the clamp inside decode_sdf zeroes the
    gradient where |f| > clamp_dist, and the (n,3) grad_outputs of decoder_utils.py:84 made torch 1.1 return 3x
    the gradient. Returned detached (second-order terms vanish for ReLU decoders after normalisation)."""
    sdf, g = functions.mlp_grad(_engine(decoder, points), latent_vector, points)
    g = 3.0 * g
    if clamp_dist is not None:
        g = g * (sdf.abs() <= clamp_dist).to(g.dtype)[:, None]
    return g


def decode_color(decoder, color_code, shape_code, points, MAX_POINTS=100000, no_grad=False):
    """(n,3) surface points -> (n,3) rgb of the colour decoder (decoder_utils.py:94-112); differentiable w.r.t. the colour code,
    the shape code and the points unless `no_grad` (distr_color_backward). MAX_POINTS chunking is unnecessary (accepted, ignored)."""
    dev = points.device
    if dev.type != 'cuda':
        raise RuntimeError('decode_color: tensors must be on the GPU (no CPU path in this build)')
    eng = functions.get_color_engine(decoder, dev.index if dev.index is not None else torch.cuda.current_device())
    needs = (not no_grad) and torch.is_grad_enabled() and any(getattr(t, 'requires_grad', False) for t in (color_code, shape_code, points))
    if needs:
        return functions.color_eval_autograd(eng, color_code, shape_code, points)
    return functions.color_eval(eng, color_code, shape_code, points)
