"""Row f1 timing (not a pytest file): the 256^3 SDF grid the reference's create_mesh evaluates for marching cubes (16.8 M points,
core/evaluation/create_mesh.py:56-68; there: 512 batches of 32^3 points with a host round trip each), plain and coarse-to-fine.
    python tests/gpu_diag_grid.py [arith]      (arith: f32 (default) | bf16x6 | f16x3, passed to create_sdf_grid / _speedup)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch


def main():
    from core.evaluation import create_sdf_grid, create_sdf_grid_speedup
    from core.graph.deep_sdf_decoder import Decoder
    from distr import fixture
    Ws, bs, latent = fixture.make_decoder_weights()
    dec = Decoder(256, [512] * 8, norm_layers=(), latent_in=[4])
    dec.load_state_dict({('lin%d.%s' % (l, n)): torch.from_numpy(a) for l, (W, b) in enumerate(zip(Ws, bs)) for n, a in (('weight', W), ('bias', b))})
    dec = dec.cuda()
    lat = torch.from_numpy(latent).cuda()
    arith = sys.argv[1] if len(sys.argv) > 1 else 'f32'
    import functools
    create_sdf_grid = functools.partial(create_sdf_grid, arith=arith)
    create_sdf_grid_speedup = functools.partial(create_sdf_grid_speedup, arith=arith)
    print('arith', arith)
    for name, fn in (('create_sdf_grid', create_sdf_grid), ('create_sdf_grid_speedup', create_sdf_grid_speedup)):
        for N in (128, 256):
            fn(dec, lat, N)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                g = fn(dec, lat, N)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            print('%-24s N=%d: %.1f ms for %d grid points (%.1f M points/s, %.1f TFLOP/s on the evaluated points incl. sample generation); surface voxels %d'
                  % (name, N, t * 1e3, N ** 3, N ** 3 / t / 1e6, 3146752 * N ** 3 / t / 1e12 if name == 'create_sdf_grid' else float('nan'),
                     int((g.abs() < 2.0 / (N - 1)).sum())))


if __name__ == '__main__':
    main()
