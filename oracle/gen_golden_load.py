"""TEST INFRASTRUCTURE -- golden G22: the reference's load_decoder (core/utils/decoder_utils.py:7-51) on a synthetic DeepSDF experiment
directory (specs.json + ModelParameters/2000.pth in DeepSDF's own layout: weight-normed 8 x 512 decoder saved from DataParallel; a colour
decoder saved without the 'module.' prefix), run by the REFERENCE itself on CPU (build container only; shims in oracle/ref_harness.py):

    python oracle/gen_golden_load.py        # writes tests/golden/g22_load_decoder.npz

Records what the drivers rely on after `load_decoder(experiment_directory, checkpoint)` (run_single_shape.py:62-63): the wrapper class,
the state-dict keys and shapes, the constructor attributes the spec file sets, and decoder.module.inference on fixed inputs in eval mode.
The experiment directory is rebuilt from the seed-defined fixture by tests/helpers.py::write_deepsdf_experiment on both sides.
"""
import os
import sys
import tempfile

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, '..', 'dist-renderer_amd'))
sys.path.insert(0, os.path.join(_HERE, '..', 'tests'))
sys.path.insert(0, _HERE)
from distr import fixture, decoder_pack  # noqa: E402
import helpers  # noqa: E402
import ref_harness as rh  # noqa: E402

OUT = os.path.join(_HERE, '..', 'tests', 'golden')
ATTRS = ('num_layers', 'norm_layers', 'latent_in', 'latent_dropout', 'xyz_in_all', 'weight_norm', 'use_tanh', 'dropout_prob', 'dropout')


def describe(dec, x):
    m = dec.module
    m.eval()
    with torch.no_grad():
        y = m.inference(torch.from_numpy(x)).numpy()
    sd = dec.state_dict()
    return dict(cls=type(dec).__name__, inner=type(m).__name__, keys=np.array(sorted(sd.keys())), shapes=np.array([list(sd[k].shape) + [0] * (2 - sd[k].dim()) for k in sorted(sd.keys())]),
                attrs=np.array([repr(getattr(m, a)) for a in ATTRS]), y=y)


def main():
    rh.install_shims()
    du = rh.reference_modules()[3]
    assert os.path.abspath(du.__file__).startswith(rh.REFERENCE_ROOT)
    Ws, bs, latent = fixture.make_decoder_weights()
    cWs, cbs = fixture.make_color_decoder_weights(color_size=16)[:2]
    tmp = tempfile.mkdtemp()
    d_shape = helpers.write_deepsdf_experiment(os.path.join(tmp, 'sofas'), decoder_pack.fixture_state_dict(Ws, bs, True), '2000', module_prefix=True)
    d_color = helpers.write_deepsdf_experiment(os.path.join(tmp, 'sofas_color'), decoder_pack.fixture_state_dict(cWs, cbs, True), 'latest', module_prefix=False)
    rs = np.random.RandomState(22)
    x = np.concatenate([np.repeat(latent, 32, 0), (rs.rand(32, 3) * 1.6 - 0.8).astype(np.float32)], 1).astype(np.float32)
    xc = np.concatenate([np.repeat(latent, 32, 0), 0.1 * rs.standard_normal((32, 16)).astype(np.float32), x[:, -3:]], 1).astype(np.float32)
    out = {}
    a = describe(du.load_decoder(d_shape, '2000'), x)
    out.update({'shape_' + k: v for k, v in a.items()})
    # colour decoder (renderer_rgb / run_multi_*: specs from the SHAPE experiment, weights from the colour experiment, no 'module.' prefix)
    c = describe(du.load_decoder(d_shape, 'latest', color_size=16, experiment_directory_color=d_color), xc)
    out.update({'color_' + k: v for k, v in c.items()})
    u = du.load_decoder(d_shape, None)                              # no checkpoint: architecture only (random init)
    out['nockpt_keys'] = np.array(sorted(u.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, 'g22_load_decoder.npz'), x=x, xc=xc, attr_names=np.array(ATTRS), **out)
    print('g22', a['cls'], a['inner'], len(a['keys']), 'keys', a['attrs'].tolist(), 'y', a['y'][:3, 0], '| colour', c['y'][:2])


if __name__ == '__main__':
    main()
