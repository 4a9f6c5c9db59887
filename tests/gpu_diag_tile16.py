"""Diagnostic (not a test): duration of the bare 16-ray decoder tile (distr_mlp_eval with DISTR_TILE_RB=-1: trivial
prologue / epilogue) for comparison with the 16-ray march step (k_march16<FINE>) under rocprofv3 --kernel-trace."""
import os
import sys
os.environ['DISTR_TILE_RB'] = '-1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-renderer_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from distr import fixture, functions  # noqa: E402

Ws, bs, latent = fixture.make_decoder_weights()
eng = functions.engine_from_weights(Ws, bs, 0)
lat = torch.from_numpy(latent).cuda()
for n in (16, 256, 4096):
    pts = torch.from_numpy((np.random.RandomState(1).rand(n, 3) - 0.5).astype(np.float32)).cuda()
    for _ in range(20):
        functions.mlp_eval(eng, lat, pts)
torch.cuda.synchronize()
print('done')
