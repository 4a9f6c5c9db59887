"""CPU-only checks of the host side: the C-ABI library builds for gfx950, loads without a GPU and exports every
symbol include/distr.h declares; the decoder packer; the config struct mirror; the product never imports the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def libdistr():
    from distr import binding
    binding.build_library()
    return binding.lib()


def test_library_exports_every_declared_symbol(libdistr):
    hdr = open(os.path.join(ROOT, 'include', 'distr.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(distr_[a-z_]+)\s*\(', hdr))
    assert {'distr_create', 'distr_render_forward', 'distr_render_backward', 'distr_mlp_eval'} <= declared
    for name in declared:
        assert hasattr(libdistr, name), name
    from distr import binding
    assert set(binding.EXPORTS) == declared


def test_cfg_struct_matches_header():
    from distr import binding
    hdr = open(os.path.join(ROOT, 'include', 'distr.h')).read()
    body = re.search(r'typedef struct distr_render_cfg \{(.*?)\} distr_render_cfg;', hdr, flags=re.S).group(1)
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for decl in body.split(';'):
        decl = decl.strip()
        if not decl:
            continue
        for n in decl.split(None, 1)[1].split(','):
            names.append(n.strip().split('[')[0])
    assert names == [f[0] for f in binding.RenderCfg._fields_]
    assert C.sizeof(binding.RenderCfg) == 4 * (2 + 9 + 2 + 9 + 2 + 4 + 1 + 2 + 3 + 3 + 1 + 2)


def test_create_without_gpu_fails_loudly(libdistr):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = C.c_void_p()
    rc = libdistr.distr_create(C.byref(h), 0)
    assert rc != 0
    assert b'device' in libdistr.distr_last_error(h)
    libdistr.distr_destroy(h)
    from distr import binding
    with pytest.raises(binding.DistrError):
        binding.Context(0)


def test_no_cpu_fallback_and_no_oracle_in_product():
    """The product package must not import / call anything under oracle/ (nor the reference)."""
    pkg = os.path.join(ROOT, 'dist-renderer_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), (dirpath, f)
                assert 'liboracle' not in src and '/root/reference' not in src, (dirpath, f)


def test_decoder_pack_validates(fixture_decoder):
    from distr import decoder_pack
    Ws, bs, _ = fixture_decoder
    flat = decoder_pack.flatten(Ws, bs)
    assert flat.dtype == np.float32 and flat.size == sum(W.size + b.size for W, b in zip(Ws, bs))
    with pytest.raises(decoder_pack.UnsupportedDecoder):
        decoder_pack.flatten(Ws[:-1], bs[:-1])
    bad = [W.copy() for W in Ws]
    bad[3] = np.zeros((256, 512), np.float32)
    with pytest.raises(decoder_pack.UnsupportedDecoder):
        decoder_pack.flatten(bad, bs)
    sd = decoder_pack.fixture_state_dict(Ws, bs, weight_norm=True)
    sd = {'module.' + k: v for k, v in sd.items()}          # DataParallel prefix (decoder_utils.py:29-30)
    We, be = decoder_pack.effective_weights(sd)
    assert max(np.abs(a - b).max() for a, b in zip(We, Ws)) <= 2e-7


def test_module_flags_rejected():
    import torch
    from core.graph.deep_sdf_decoder import Decoder
    from distr import decoder_pack
    d = Decoder(256, [512] * 8, latent_in=[4], xyz_in_all=True)
    with pytest.raises(decoder_pack.UnsupportedDecoder):
        decoder_pack.pack_module(d)
    d = Decoder(256, [512] * 8, latent_in=[4], norm_layers=list(range(8)), weight_norm=False)   # LayerNorm variant
    with pytest.raises(decoder_pack.UnsupportedDecoder):
        decoder_pack.pack_module(d)
    d = Decoder(256, [512] * 8, latent_in=[4])
    assert decoder_pack.pack_module(d).size == 1839358
    x = torch.randn(5, 259)
    assert d.inference(x).shape == (5, 1)


def test_renderer_requires_gpu(fixture_decoder):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from core.sdfrenderer import SDFRenderer
    from core.graph.deep_sdf_decoder import Decoder
    from distr import fixture
    d = Decoder(256, [512] * 8, latent_in=[4])
    with pytest.raises(ValueError):
        SDFRenderer(d, fixture.make_intrinsic(32, 32))
    with pytest.raises(ValueError):
        SDFRenderer(d, fixture.make_intrinsic(32, 32), use_gpu=False)


def test_fragment_packing_layout():
    """Restates pack_fragments (csrc/distr_api.hip) and checks the index map the kernel's dense() loop relies on:
    every weight appears exactly once and lane (i,h) of wave w / block ob / group g / step s holds W[o][8g+2s+h]."""
    K, O = 16, 256
    NOB, NG = O // 128, K // 8
    W = np.arange(O * K, dtype=np.float32).reshape(O, K)
    dst = np.full(O * K, -1, np.float32)
    for g in range(NG):
        for w in range(4):
            for ob in range(NOB):
                for lane in range(64):
                    o, h = w * 32 * NOB + 32 * ob + (lane & 31), lane >> 5
                    base = ((((g * 4 + w) * NOB + ob) * 64) + lane) * 4
                    for s in range(4):
                        dst[base + s] = W[o, 8 * g + 2 * s + h]
    assert sorted(dst.tolist()) == sorted(W.reshape(-1).tolist())
    # MFMA 32x32x2 semantics: D[i][j] += sum_{h} A[i][h] * B[h][j]; the loop visits k = 8g+2s+h in natural order
    order = [8 * g + 2 * s + h for g in range(NG) for s in range(4) for h in range(2)]
    assert order == list(range(K))


def test_view_sharding():
    from distr import parallel
    assert parallel.shard_views(8, 3, 8) == [3]
    assert parallel.shard_views(8, 1, 4) == [1, 5]
    assert sorted(sum((parallel.shard_views(8, r, 3) for r in range(3)), [])) == list(range(8))


def test_row_sharding_tiles_every_image_once():
    from distr import parallel
    for (n, H, world) in [(4, 1024, 8), (4, 1024, 1), (4, 1024, 3), (1, 70, 4), (3, 64, 16), (2, 512, 2)]:
        cover = np.zeros((n, H), np.int32)
        for r in range(world):
            for (img, r0, r1) in parallel.shard_rows(n, H, r, world):
                assert r0 % 4 == 0 and (r1 % 4 == 0 or r1 == H) and r0 < r1
                cover[img, r0:r1] += 1
        assert (cover == 1).all()
    assert parallel.shard_rows(4, 1024, 5, 8) == [(2, 512, 1024)]
    assert parallel.shard_rows(4, 1024, 1, 2) == [(2, 0, 1024), (3, 0, 1024)]


def test_c_abi_from_plain_c(libdistr, tmp_path):
    """include/distr.h compiles as C (gcc -std=c99 -pedantic) and a plain-C program resolves every entry point from the
    shared library and runs the GPU-free calls."""
    import shutil
    import subprocess
    from distr import binding
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    exe = str(tmp_path / 'abi_check')
    src = os.path.join(ROOT, 'tests', 'c_abi', 'abi_check.c')
    subprocess.check_call(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), '-o', exe, src, '-ldl'])
    out = subprocess.run([exe, binding.LIB_PATH], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert 'symbols=26' in out.stdout and 'version="distr' in out.stdout
    assert ('sizeof(cfg)=%d' % C.sizeof(binding.RenderCfg)) in out.stdout
    import torch
    if not torch.cuda.is_available():
        assert 'create_rc=0' not in out.stdout and 'device' in out.stdout
