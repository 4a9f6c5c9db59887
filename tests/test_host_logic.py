"""CPU-only checks of the host side: the C-ABI library builds for gfx950, loads without a GPU and exports every
symbol include/distr.h declares; the decoder packer; the config struct mirror; the product never imports the oracle."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT


@pytest.fixture(scope='module')
def libdistr():
    from distr import binding
    binding.build_library()
    return binding.lib()


def test_library_exports_every_declared_symbol(libdistr):
    hdr = open(os.path.join(ROOT, 'include', 'distr.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    hdr = re.sub(r'^\s*#define.*$', '', hdr, flags=re.M)        # distr_create is a macro over distr_create_abi (ABI handshake)
    declared = set(re.findall(r'\b(distr_[a-z0-9_]+)\s*\(', hdr))
    assert {'distr_create_abi', 'distr_abi_version', 'distr_render_forward', 'distr_render_backward', 'distr_mlp_eval'} <= declared
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
    assert C.sizeof(binding.RenderCfg) == 4 * (1 + 2 + 9 + 2 + 9 + 9 + 2 + 4 + 1 + 2 + 3 + 3 + 1 + 2 + 1 + 1 + 1 + 4 + 4)


def test_create_without_gpu_fails_loudly(libdistr):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    h = C.c_void_p()
    from distr import binding
    rc = libdistr.distr_create_abi(C.byref(h), 0, binding.ABI_VERSION)
    assert rc != 0
    assert b'device' in libdistr.distr_last_error(h)
    libdistr.distr_destroy(h)
    with pytest.raises(binding.DistrError):
        binding.Context(0)


def test_abi_handshake(libdistr):
    """A caller built against another include/distr.h is refused before any field is interpreted (needs no GPU): wrong ABI version
    at distr_create_abi, wrong / unset struct_size on every struct that crosses the boundary."""
    from distr import binding
    assert libdistr.distr_abi_version() == binding.ABI_VERSION
    assert ('ABI %d' % binding.ABI_VERSION).encode() in libdistr.distr_version()
    hdr = open(os.path.join(ROOT, 'include', 'distr.h')).read()
    assert int(re.search(r'#define DISTR_ABI_VERSION (\d+)u', hdr).group(1)) == binding.ABI_VERSION
    h = C.c_void_p()
    assert libdistr.distr_create_abi(C.byref(h), 0, binding.ABI_VERSION - 1) == -1            # DISTR_ERR_INVALID_ARG
    assert b'ABI' in libdistr.distr_last_error(h)
    libdistr.distr_destroy(h)
    h = C.c_void_p()
    libdistr.distr_create_abi(C.byref(h), 0, binding.ABI_VERSION)                               # (fails for lack of a device; the context still answers)
    cfg = binding.make_cfg((64, 64), np.array([[64., 0, 32], [0, 64., 32], [0, 0, 1]]), march_step=20, buffer_size=3)
    assert cfg.struct_size == C.sizeof(binding.RenderCfg)
    f, b = C.c_size_t(), C.c_size_t()
    assert libdistr.distr_workspace_bytes(h, C.byref(cfg), C.byref(f), C.byref(b)) == 0 and f.value > 0
    assert cfg.clone().struct_size == cfg.struct_size
    for bad in (0, cfg.struct_size - 4, cfg.struct_size + 8, 64):     # zeroed struct, round 2's struct (no `arith`), a future one, an old H
        cfg.struct_size = bad
        assert libdistr.distr_workspace_bytes(h, C.byref(cfg), C.byref(f), C.byref(b)) == -1
        assert b'struct_size' in libdistr.distr_last_error(h)
    libdistr.distr_destroy(h)
    for cls in (binding.DecoderDesc, binding.RenderStats, binding.WarpCfg):
        assert cls().struct_size == C.sizeof(cls) and cls._fields_[0][0] == 'struct_size'


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
    d = Decoder(256, [512] * 8, latent_in=[4], use_tanh=True)
    with pytest.raises(decoder_pack.UnsupportedDecoder):
        decoder_pack.pack_module(d)
    # latent_dropout (deep_sdf_decoder.py:84-87) is the identity in eval mode: accepted there, refused in training mode
    d = Decoder(256, [512] * 8, latent_in=[4], latent_dropout=True)
    with pytest.raises(decoder_pack.UnsupportedDecoder):
        decoder_pack.pack_module(d.train())
    plain = Decoder(256, [512] * 8, latent_in=[4])
    plain.load_state_dict(d.state_dict())
    assert np.array_equal(decoder_pack.pack_module(d.eval()), decoder_pack.pack_module(plain))
    xx = torch.randn(7, 259)
    assert torch.equal(d.inference(xx), plain.eval().inference(xx))
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


def test_weighted_row_sharding_tiles_every_image_once_and_balances_cost():
    """the cost-weighted C5 cut (distr.parallel.shard_rows(..., weights=)): for N in 1..8 and 16 the pieces tile every
    image exactly once, every cut is a multiple of 4 rows (the 4x4 pyramid parents, core/sdfrenderer/renderer.py:732-749), the cut is a
    pure function of its arguments (every rank computes it), it balances the COST it was given (slowest / mean <= 1.03 where the
    cost-blind cut is off by tens of per cent), equal weights reproduce the image-boundary cuts of the cost-blind partition, and one
    feedback step (refine_row_weights) moves rows away from a rank that measured slower than predicted."""
    from distr import parallel
    n, H, W, align = 4, 1024, 1024, 4
    upi = H // align
    rs = np.random.RandomState(3)
    weights = []
    for i in range(n):           # an object in the middle rows of every image, different size / position per shape
        rows = np.arange(upi)
        counts = 4 * W * 0.6 * np.exp(-((rows - (110 + 15 * i)) / (38.0 + 6 * i)) ** 2)
        weights.append(parallel.row_weights_from_counts(counts.tolist(), W, align, H))

    def load(plan, w):
        return [sum(sum(w[img][r0 // align:(r1 + align - 1) // align]) for (img, r0, r1) in pieces) for pieces in plan]

    for world in list(range(1, 9)) + [16]:
        plan = parallel.shard_rows_plan(n, H, world, align, weights)
        assert plan == parallel.shard_rows_plan(n, H, world, align, [list(w) for w in weights])          # deterministic
        cover = np.zeros((n, H), np.int32)
        for r, pieces in enumerate(plan):
            assert pieces == parallel.shard_rows(n, H, r, world, align, weights)
            assert len(pieces) >= 1
            for (img, r0, r1) in pieces:
                assert r0 % 4 == 0 and (r1 % 4 == 0 or r1 == H) and r0 < r1
                cover[img, r0:r1] += 1
        assert (cover == 1).all(), world
        lw = load(plan, weights)
        assert max(lw) <= 1.03 * (sum(lw) / world), (world, lw)
        if world in (2, 8, 16):
            blind = [parallel.shard_rows(n, H, r, world) for r in range(world)]
            lb = load(blind, weights)
            assert max(lb) / (sum(lb) / world) >= max(lw) / (sum(lw) / world) - 1e-9
    # the cost-blind cut of 8 ranks over these images is visibly unbalanced (what the weights are for)
    lb = load([parallel.shard_rows(n, H, r, 8) for r in range(8)], weights)
    assert max(lb) / (sum(lb) / 8) > 1.10
    # equal weights: the same partition as the cost-blind one wherever that one cuts on unit boundaries
    eq = [[1.0] * upi for _ in range(n)]
    for world in (1, 2, 4, 8):
        assert parallel.shard_rows_plan(n, H, world, align, eq) == [parallel.shard_rows(n, H, r, world) for r in range(world)]
    # odd image height (last unit short) and fewer units than ranks
    for (n2, H2, world) in [(1, 70, 4), (3, 64, 16), (2, 10, 8)]:
        w2 = [list(rs.rand((H2 + 3) // 4) + 0.1) for _ in range(n2)]
        cover = np.zeros((n2, H2), np.int32)
        for pieces in parallel.shard_rows_plan(n2, H2, world, 4, w2):
            for (img, r0, r1) in pieces:
                assert r0 % 4 == 0 and (r1 % 4 == 0 or r1 == H2) and r0 < r1
                cover[img, r0:r1] += 1
        assert (cover == 1).all()
    # feedback: rank 2 measured 40 % more than its share -> its pieces shrink in the refined cut
    plan = parallel.shard_rows_plan(n, H, 8, align, weights)
    lw = load(plan, weights)
    measured = [x * (1.4 if r == 2 else 1.0) for r, x in enumerate(lw)]
    w2 = parallel.refine_row_weights(weights, plan, measured, H, align)
    plan2 = parallel.shard_rows_plan(n, H, 8, align, w2)
    rows = lambda pieces: sum(r1 - r0 for (_, r0, r1) in pieces)
    assert rows(plan2[2]) < rows(plan[2])
    l2 = load(plan2, w2)
    assert max(l2) <= 1.03 * (sum(l2) / 8)


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
    assert ('symbols=%d' % len(binding.EXPORTS)) in out.stdout and 'version="distr' in out.stdout
    assert ('sizeof(cfg)=%d' % C.sizeof(binding.RenderCfg)) in out.stdout
    # the ctypes mirrors against the C structs, field by field: name, offset, size (a drift here corrupts arguments silently)
    mirrors = {'distr_render_cfg': binding.RenderCfg, 'distr_decoder_desc': binding.DecoderDesc, 'distr_render_stats': binding.RenderStats,
               'distr_warp_cfg': binding.WarpCfg}
    seen = {k: [] for k in mirrors}
    for line in out.stdout.splitlines():
        if line.startswith('offset '):
            _, name, off, size = line.split()
            st, field = name.split('.')
            seen[st].append(field)
            fd = getattr(mirrors[st], field)
            assert (fd.offset, fd.size) == (int(off), int(size)), (name, fd.offset, fd.size, off, size)
        elif line.startswith('sizeof distr_'):
            _, st, size = line.split()
            assert C.sizeof(mirrors[st]) == int(size), st
    for st, cls in mirrors.items():
        assert seen[st] == [f for f, _ in cls._fields_], (st, seen[st])          # same fields, same order, none missing
    import torch
    if not torch.cuda.is_available():
        assert 'create_rc=0' not in out.stdout and 'device' in out.stdout


# ---------------------------------------------------------------------------------------------------------------- drop-in
STUB = os.path.join(ROOT, 'tests', 'dropin_stub', 'refcheckout')


def _run_driver(cmd, env_extra=None):
    import json
    import subprocess
    env = dict(os.environ, PYTHONPATH=PKG)
    env.update(env_extra or {})
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])


def _check_driver_resolution(j):
    ours = os.path.join(PKG, 'core')
    for k in ('optimize_single_view', 'optimize_multi_view', 'SDFRenderer', 'SDFRenderer_warp', 'load_decoder',
              'downsize_camera_intrinsic', 'get_tensor_from_camera', 'create_mesh_speedup', 'Evaluator.create_mesh', 'Evaluator.decode_sdf'):
        assert j[k].startswith(ours), (k, j[k])
    for k in ('LoaderSingle', 'LoaderMultiPMO', 'Visualizer', 'project_points', 'Evaluator', 'pytorch_ssim'):
        assert j[k].startswith(os.path.join(STUB, 'core')), (k, j[k])


def test_dropin_coexists_with_reference_checkout():
    """with this build first on sys.path the reference drivers' import lists (run_single_shape.py:6-13,
    run_multi_pmodata.py:8-15) must still reach the reference's own core.dataset / core.visualize / core.evaluation.evaluator,
    while every mirrored module -- also through the reference's FLAT imports (`from create_mesh import ...`,
    `from decoder_utils import decode_sdf`) -- resolves to this build. Reference checkout = tests/dropin_stub (stubs)."""
    j = _run_driver([sys.executable, '-m', 'distr.launch', os.path.join(STUB, 'run_driver.py'), '--gpu', '0'])
    assert j['argv'] == ['--gpu', '0'] and j['default_arith'] == 'f32'
    _check_driver_resolution(j)
    # the launcher's own option: an opt-in arithmetic for renderers the driver constructs without `arith=`; the driver's arguments pass through
    j = _run_driver([sys.executable, '-m', 'distr.launch', '--arith', 'f16x3', os.path.join(STUB, 'run_driver.py'), '--gpu', '0'], {'DISTR_ARITH': ''})
    assert j['argv'] == ['--gpu', '0'] and j['default_arith'] == 'f16x3'
    import subprocess
    bad = subprocess.run([sys.executable, '-m', 'distr.launch', '--arith', 'fp8', os.path.join(STUB, 'run_driver.py')], env=dict(os.environ, PYTHONPATH=PKG),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert bad.returncode != 0 and 'arith' in bad.stderr
    bad = subprocess.run([sys.executable, '-m', 'distr.launch', os.path.join(STUB, 'run_driver.py')], env=dict(os.environ, PYTHONPATH=PKG, DISTR_ARITH='fp8'),
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert bad.returncode != 0 and 'DISTR_ARITH' in bad.stderr          # an unknown name is an error, not a silent fallback to f32


REFERENCE = '/root/reference'


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason='the reference checkout exists in the build container only (it never travels)')
@pytest.mark.parametrize('driver', ['run_single_shape.py', 'run_multi_pmodata.py', 'run_single_camera.py', 'run_multi_realdata.py'])
def test_real_reference_driver_imports_resolve(driver):
    """The REAL drivers of the reference, unmodified, in the real checkout (run_single_shape.py:5-14, run_multi_pmodata.py:8-15):
    their import blocks, executed under distr.launch, resolve every hot-path module to THIS build (also through the reference's flat
    imports, e.g. core/evaluation/evaluator.py's `from create_mesh import ...`) and everything else -- datasets, visualiser,
    evaluator -- to the reference. Container-only: nothing is copied, nothing ships (tests/real_driver_imports.py reads the import
    block at run time)."""
    path = os.path.join(REFERENCE, driver)
    if not os.path.exists(path):
        pytest.skip('driver not in this checkout')
    j = _run_driver([sys.executable, '-m', 'distr.launch', os.path.join(ROOT, 'tests', 'real_driver_imports.py'), path])
    ours, ref = os.path.join(PKG, 'core') + os.sep, os.path.join(REFERENCE, 'core') + os.sep
    mods, names = j['modules'], j['names']
    assert j['statements'] >= 8
    # packages and modules of the hot path -> this build
    for m in ('core', 'core.sdfrenderer', 'core.sdfrenderer.renderer', 'core.utils.decoder_utils', 'core.utils.render_utils', 'core.inv_optimizer',
              'core.evaluation', 'core.utils'):
        assert m in mods and mods[m].startswith(ours), (m, mods.get(m))
    assert all(v.startswith(ours) or v.startswith(ref) for v in mods.values()), mods          # nothing from anywhere else
    if 'create_mesh' in mods:                                   # the reference's flat import of the meshing module lands here too
        assert mods['create_mesh'].startswith(ours), mods['create_mesh']
    # the rest of the reference's `core` -> the reference checkout
    for m in ('core.dataset', 'core.visualize', 'core.visualize.vis_utils'):
        assert m in mods and mods[m].startswith(ref), (m, mods.get(m))
    want_ours = {'run_single_shape.py': ['SDFRenderer', 'optimize_single_view', 'load_decoder'],
                 'run_single_camera.py': ['SDFRenderer', 'load_decoder'],
                 'run_multi_pmodata.py': ['SDFRenderer_warp', 'optimize_multi_view', 'load_decoder'],
                 'run_multi_realdata.py': ['SDFRenderer_warp', 'load_decoder']}[driver]
    for n in want_ours:
        assert names[n].startswith(ours), (n, names[n])
    for n in ('Visualizer', 'Evaluator'):
        if n in names:
            assert names[n].startswith(ref), (n, names[n])
    loaders = [n for n in names if n.startswith('Loader')]
    assert loaders and all(names[n].startswith(ref) for n in loaders), loaders
    if 'create_mesh_speedup' in names:
        assert names['create_mesh_speedup'].startswith(ours)


def test_dropin_with_explicit_sys_path_order():
    """Same resolution when a caller orders sys.path itself (this build first, the reference checkout later)."""
    code = ("import sys, runpy; sys.path.insert(0, %r); sys.argv = ['run_driver.py']; "
            "runpy.run_path(%r, run_name='__main__')" % (PKG, os.path.join(STUB, 'run_driver.py')))
    _check_driver_resolution(_run_driver([sys.executable, '-c', code], {'PYTHONPATH': ''}))


def test_render_utils_camera_helpers():
    import torch
    from core.utils.render_utils import downsize_camera_intrinsic, get_camera_from_tensor, get_tensor_from_camera
    from distr import fixture
    K = np.array([[300.0, 1.5, 112.0], [0.0, 310.0, 112.0], [0.0, 0.0, 1.0]])
    K2 = downsize_camera_intrinsic(K, 2)
    assert np.allclose(K2, [[150.0, 0.75, 56.0], [0.0, 155.0, 56.0], [0.0, 0.0, 1.0]])      # whole first two rows, incl. the skew
    with pytest.raises(ValueError):
        downsize_camera_intrinsic(K, 10)          # 22.4 px: rejected (like the reference, only a fractional part below one half is)
    for cam in ((30, 20, 1.6, 10), (170, -80, 2.0, 95), (0, 0, 1.6, 0), (-120, 45, 1.3, 180)):
        R, T = fixture.make_camera(*cam)
        RT = np.concatenate([R, T[:, None]], 1)
        q = get_tensor_from_camera(RT)
        assert q.shape == (7,) and abs(float(q[:4].norm()) - 1) < 1e-6 and float(q[0]) >= 0
        back = get_camera_from_tensor(q)
        assert np.abs(back.numpy() - RT).max() < 2e-6
        assert get_tensor_from_camera(torch.from_numpy(RT)).dtype == torch.float32


class _FakeRenderer(object):
    """CPU stand-in with SDFRenderer.render's output contract (the loops only call render / get_threshold)."""

    def __init__(self, hw):
        self.hw = hw

    def get_threshold(self):
        return 5e-5

    def render(self, latent, R, T, **kw):
        import torch
        h, w = self.hw
        s = latent.sum()
        depth = torch.full((h, w), 1.5) + 0.01 * s
        normal = torch.zeros(h, w, 3) + torch.tensor([0.0, 0.0, -1.0]) * (1 + 0 * s)
        mask = torch.zeros(h, w, dtype=torch.uint8)
        mask[2:6, 2:6] = 1
        q = torch.full((h, w), 1e-3) + 1e-3 * s
        return depth, normal, mask, q


def test_optimize_single_view_hooks(tmp_path, capsys):
    """evaluator / visualizer / test_step / vis_folder of optimize_single.py:87-101 are honoured (VERDICT r1 missing #7,
    ADVICE r1: visualizer.reset_data / add_loss_from_pack in compute_all_loss)."""
    import torch
    from core.inv_optimizer import optimize_single_view

    class Vis(object):
        def __init__(self):
            self.resets = self.packs = self.dumps = self.curves = 0
            self.losses, self.chamfers, self.keys = [], [], set()

        def reset_data(self): self.resets += 1
        def add_data(self, name, src, mask=None): self.keys.add(name)
        def add_loss_from_pack(self, pack): self.packs += 1
        def add_loss(self, loss): self.losses.append(float(loss))
        def add_chamfer(self, d): self.chamfers.append(d)
        def show_loss_curve(self, fname): self.curves += 1
        def show_all_data(self, fname): pass
        def dump_all_data(self, fname): self.dumps += 1

    class Eval(object):
        calls = []

        def latent_vec_to_points(self, code, fname=None, silent=True):
            self.calls.append(fname)
            return np.zeros((4, 3))

        def compute_chamfer_distance(self, a, b):
            return 0.002
    h = w = 8
    gt = {'depth': torch.full((h, w), 1.4), 'normal': torch.zeros(h, w, 3) + torch.tensor([0.0, 0.0, -1.0]),
          'silhouette': torch.zeros(h, w, dtype=torch.uint8)}
    gt['silhouette'][3:7, 3:7] = 1
    lat = torch.zeros(1, 256, requires_grad=True)
    opt = torch.optim.Adam([lat], lr=1e-2)
    wd = dict(w_depth=10.0, w_normal=5.0, w_mask_gt=1.0, w_mask_out=1.0, w_l2reg=1.0)
    vis, ev = Vis(), Eval()
    folder = str(tmp_path / 'vis')
    out, _ = optimize_single_view([_FakeRenderer((h, w)), _FakeRenderer((h, w))], ev, opt, lat, torch.eye(3, 4), gt, wd, num_iters=6,
                                  points_gt=np.zeros((4, 3)), test_step=3, visualizer=vis, silent=False, vis_folder=folder)
    assert out is lat and os.path.isdir(folder)
    assert vis.resets == 6 and vis.packs == 6 and vis.dumps == 6 and vis.curves == 6 and len(vis.losses) == 6     # first renderer only
    assert {'mask_output', 'mask_gt', 'depth_output', 'depth_gt', 'loss_depth', 'normal_output', 'loss_normal'} <= vis.keys
    assert len(ev.calls) == 2 and ev.calls[0].endswith('output_2.ply') and vis.chamfers == [0.002, 0.002]
    printed = capsys.readouterr().out
    assert printed.count('NAME = [') == 6 and 'CHAMFER DISTANCE: 2.000' in printed
    assert float(lat.detach().abs().max()) > 0
    # silent: nothing printed, no evaluator / visualiser traffic beyond the loss images
    vis2, ev2 = Vis(), Eval()
    ev2.calls = []
    optimize_single_view([_FakeRenderer((h, w))], ev2, opt, lat, torch.eye(3, 4), gt, wd, num_iters=3, points_gt=np.zeros((4, 3)),
                         test_step=1, visualizer=vis2, silent=True, vis_folder=folder)
    assert capsys.readouterr().out == '' and ev2.calls == [] and vis2.dumps == 0 and vis2.packs == 3


def test_compute_loss_color_and_mask_visualizer():
    import torch
    from core.utils.loss_utils import compute_loss_color, compute_loss_mask, normalize_vectors
    rs = np.random.RandomState(0)
    a, b = torch.from_numpy(rs.rand(6, 7, 3).astype(np.float32)), torch.from_numpy(rs.rand(6, 7, 3).astype(np.float32))
    m1 = torch.from_numpy((rs.rand(6, 7) > 0.3)); m2 = torch.from_numpy((rs.rand(6, 7) > 0.3))
    loss, vis = compute_loss_color(a, m1, b, m2.to(torch.uint8))
    both = (m1 & m2).numpy()
    assert vis is None and abs(float(loss) - np.abs(a.numpy()[both] - b.numpy()[both]).mean()) < 1e-7
    v = normalize_vectors(torch.tensor([[3.0, 0.0, 4.0]]), dim=1)
    assert np.allclose(v.numpy(), [[0.6, 0.0, 0.8]])
    q = torch.from_numpy(rs.randn(6, 7).astype(np.float32) * 1e-3)
    lg, lo, _ = compute_loss_mask(q, m1, m2)
    miss, extra = (m2 & ~m1).numpy(), (m1 & ~m2).numpy()
    assert abs(float(lg) - np.maximum(q.numpy()[miss] - 5e-5, 0).mean()) < 1e-9
    assert abs(float(lo) - np.maximum(5e-5 - q.numpy()[extra], 0).mean()) < 1e-9


def test_balance_views_plan():
    """distr.parallel.balance_views: a pure function of the gathered step times (+ row cost profiles); every row of every view is
    assigned exactly once, bands are 4-row aligned and at least 16 rows, the modelled slowest rank drops, balanced inputs are left
    alone; row_profile weighs surface rows above background rows; refine_profiles moves the plan the way a measurement says."""
    from distr import parallel
    H = 512
    times = [52.71, 54.40, 51.19, 50.40, 47.98, 46.50, 49.57, 58.67]          # the eight C4 cameras, ms per step (profiles/r02_view_balance.md)

    def check(plan, N, Hh):
        cover = np.zeros((N, Hh), np.int32)
        for r, items in enumerate(plan):
            assert items[0][0] == r and items[0][1] == 0 and items[0][2] >= Hh // 2
            for (v, r0, r1) in items:
                assert r0 % 4 == 0 and (r1 % 4 == 0 or r1 == Hh) and r1 - r0 >= 16
                cover[v, r0:r1] += 1
        assert (cover == 1).all()

    plan = parallel.balance_views(times, H)
    check(plan, 8, H)
    assert plan == parallel.balance_views(list(times), H) and plan[7][0][2] < H and len(plan[5]) == 2      # view 7 gives, the cheapest rank takes
    est = [sum((r1 - r0 + (4 if i else 0) + (4 if (r1 < H) else 0)) * times[v] / H for i, (v, r0, r1) in enumerate(items)) for items in plan]
    assert max(est) < 0.95 * max(times)
    assert parallel.balance_views([50.0, 50.5], H) == [[(0, 0, H)], [(1, 0, H)]]
    assert parallel.balance_views([60.0], H) == [[(0, 0, H)]]
    two = parallel.balance_views([70.0, 40.0], 192)
    check(two, 2, 192)
    assert two[0] == [(0, 0, 152)] and two[1] == [(1, 0, 192), (0, 152, 192)]
    # row profile: object in rows 110..436 -> the bottom rows are background and cheap, so more of them move
    mask = np.zeros((H, H), np.uint8)
    mask[110:436, 128:384] = 1
    prof = parallel.row_profile(mask)
    assert len(prof) == H // 4 and prof[0] == prof[-1] and prof[60] > 4 * prof[0]
    pl2 = parallel.balance_views(times, H, [prof] * 8)
    check(pl2, 8, H)
    assert pl2[7][0][2] < plan[7][0][2]
    # feedback: the donors measured a larger saving than the model predicted -> their lower rows cost more -> fewer rows move
    U = H // 4
    loads = []
    for r in range(8):
        top = pl2[r][0][2] // 4
        own = times[r] * (sum(prof[:top]) + sum(prof[top:top + 1])) / sum(prof)
        loads.append(own - 0.5 * (times[r] - own) if top < U else times[r])
    pl3 = parallel.balance_views(times, H, parallel.refine_profiles([prof] * 8, pl2, times, loads, H))
    check(pl3, 8, H)
    assert pl3[7][0][2] > pl2[7][0][2]


def test_balance_views_properties_random():
    """Randomised property check of the view balancer (the N > 1 path of bench.py runs for the first time on the driver's multi-GPU
    node, so the planner is exercised here on everything it could meet): for random step times, row cost profiles and image heights
    the plan tiles every view exactly once with aligned bands of >= 16 rows, every view keeps at least half of its rows, the plan is
    a pure function of its inputs, and under the planner's own cost model the slowest rank never gets slower."""
    from distr import parallel
    rs = np.random.RandomState(7)
    moved = 0
    for trial in range(300):
        N = int(rs.randint(2, 9))
        H = int(rs.choice([64, 128, 192, 256, 512, 1024])) + (int(rs.choice([0, 0, 2, 6])) if trial % 5 == 0 else 0)
        U = (H + 3) // 4
        times = (50.0 * (1.0 + 0.3 * rs.rand(N))).tolist()
        if trial % 7 == 0:
            times[int(rs.randint(N))] *= 1.5
        profs = None
        if trial % 3:
            profs = []
            for r in range(N):
                c, w = rs.uniform(0.3, 0.7) * U, rs.uniform(0.1, 0.4) * U
                profs.append((0.123 + np.exp(-0.5 * ((np.arange(U) - c) / w) ** 2) * rs.uniform(0.5, 3.0)).tolist())
        plan = parallel.balance_views(times, H, profs)
        assert plan == parallel.balance_views(list(times), H, profs)
        cover = np.zeros((N, H), np.int32)
        load = np.zeros(N)
        for r, items in enumerate(plan):
            assert items[0][0] == r and items[0][1] == 0 and items[0][2] >= (U // 2) * 4
            for i, (v, r0, r1) in enumerate(items):
                assert 0 <= r0 < r1 <= H and r0 % 4 == 0 and (r1 % 4 == 0 or r1 == H)
                assert i == 0 or r1 - r0 >= 16 or r1 == H
                cover[v, r0:r1] += 1
                p = np.asarray(profs[v]) if profs is not None else np.ones(U)
                t = times[v] * p / p.sum()
                lo, hi = r0 // 4, (r1 + 3) // 4
                load[r] += t[lo:hi].sum() + (t[max(0, lo - 1):lo].sum() if i > 0 else 0.0) + (t[hi:hi + 1].sum() if hi < U else 0.0)
                load[r] += parallel.BAND_FIXED * np.mean(times) if i > 0 else 0.0
        assert (cover == 1).all()
        assert load.max() <= max(times) * (1.0 + 1e-9)
        moved += sum(len(p) > 1 for p in plan) > 0
    assert moved > 100          # the sweep really exercises plans that move rows


def test_split_study_pieces_reconstruct_their_operand():
    """The splitting the numerics study (oracle/study_split_bf16.py) and the kernels' host-side packers are built on: three bf16
    pieces carry a f32 value exactly; two f16 pieces of the x64-scaled value carry 22 bits (relative error <= 2^-21) down to where
    the second piece enters the f16 denormals (|x| ~ 2e-3), below that an absolute error of half a denormal step (2^-25 / 64 = 4.7e-10)
    -- over the magnitudes the decoder has (weights ~1e-4 .. 0.5, activations ~1e-4 .. 10)."""
    import torch
    from oracle.study_split_bf16 import pieces, pieces16
    g = torch.Generator().manual_seed(3)
    x = torch.cat([torch.randn(20000, generator=g) * s for s in (1e-4, 1e-2, 0.5, 10.0)])
    p3 = pieces(x, 3)
    assert torch.equal(p3[0] + p3[1] + p3[2], x)
    for q in p3:
        assert torch.equal(q.detach().to(torch.bfloat16).to(torch.float32), q.detach())      # every piece is a bf16 number
    h2 = pieces16(x, 2)
    err = ((h2[0] + h2[1]) - x).abs()
    assert bool((err <= torch.maximum(x.abs() * 2.0 ** -21, torch.full_like(x, 2.0 ** -25 / 64))).all())
    for q in h2:
        assert torch.equal((q.detach() * 64).to(torch.float16).to(torch.float32) / 64, q.detach())


def test_python_helpers_match_reference_golden():
    """G21 (oracle/gen_golden_helpers.py): the reference's own render_utils / loss_utils / train_utils helpers evaluated on seeded
    inputs, against the restatements the drop-in package ships next to the kernels -- values, in-place side effects and autograd
    gradients. CPU tensors: these helpers are plain PyTorch on whatever device their inputs live on."""
    import torch
    from conftest import GOLDEN
    from core.utils import render_utils as ru, loss_utils as lu, train_utils as tu
    assert os.path.abspath(ru.__file__).startswith(PKG)
    g = dict(np.load(os.path.join(GOLDEN, 'g21_python_helpers.npz')))
    T = torch.from_numpy
    # depth2normal: values, the in-place zeroing of its input (render_utils.py:24-25), gradient
    d = T(g['d2n_depth_in'].copy()).requires_grad_(True)
    dd = d * 1.0
    n = ru.depth2normal(dd, float(g['d2n_fx']), float(g['d2n_fy']))
    (n * T(g['d2n_wn'])).sum().backward()
    assert np.abs(n.detach().numpy() - g['d2n_normal']).max() <= 1e-6
    assert np.array_equal(dd.detach().numpy(), g['d2n_depth_after'])
    assert np.abs(d.grad.numpy() - g['d2n_grad']).max() <= 1e-5 * np.abs(g['d2n_grad']).max()
    assert np.abs(ru.depth2normal(T(g['d2n_depth_in'].copy()), 40.0).numpy() - g['d2n_normal_single_f']).max() <= 1e-6
    # quaternion -> rotation, camera tensor -> [R | T] (single and batched) + gradient
    assert np.abs(ru.quad2rotation(T(g['quat'])).numpy() - g['quat_rot']).max() <= 1e-6
    assert np.abs(ru.get_camera_from_tensor(T(g['cam'])).numpy() - g['cam_RT_batch']).max() <= 1e-6
    single = ru.get_camera_from_tensor(T(g['cam'][2]))
    assert single.shape == (3, 4) and np.abs(single.numpy() - g['cam_RT_single']).max() <= 1e-6
    c = T(g['cam'][1].copy()).requires_grad_(True)
    (ru.get_camera_from_tensor(c) * T(g['cam_wRT'])).sum().backward()
    assert np.abs(c.grad.numpy() - g['cam_grad']).max() <= 1e-5 * np.abs(g['cam_grad']).max()
    # and back: the inverse the reference takes from Blender's mathutils
    for i in range(g['cam'].shape[0]):
        back = ru.get_tensor_from_camera(g['cam_RT_batch'][i]).numpy()
        q = g['cam'][i, :4] * (1.0 if g['cam'][i, 0] >= 0 else -1.0)
        assert np.abs(back[:4] - q).max() <= 2e-6 and np.abs(back[4:] - g['cam'][i, 4:]).max() <= 1e-6
    # intrinsics of a downsized image
    assert np.allclose(ru.downsize_camera_intrinsic(g['K'], 2), g['K_half'], atol=0) and np.allclose(ru.downsize_camera_intrinsic(g['K'], 4), g['K_quarter'], atol=0)
    assert bool(g['K_fifth_raises'])
    with pytest.raises(ValueError):
        ru.downsize_camera_intrinsic(g['K'], 5)
    # image / mask downsizing (loss_utils.py:27-57): masks survive only where the whole block is set
    for f in (2, 4):
        assert np.abs(lu.downsize_img_tensor(T(g['ds_img']), f).numpy() - g['ds_img_%d' % f]).max() <= 1e-6
        assert np.abs(lu.downsize_img_tensor(T(g['ds_img3']), f).numpy() - g['ds_img3_%d' % f]).max() <= 1e-6
        mk = lu.downsize_img_tensor(T(g['ds_mask']), f)
        assert mk.dtype == torch.uint8 and np.array_equal(mk.numpy(), g['ds_mask_%d' % f])
        assert np.array_equal(lu.downsize_img_tensor(T(g['ds_mask']).bool(), f).numpy() != 0, g['ds_mask_%d' % f] != 0)
    # bilinear sampling at pixel coordinates, zero padding outside
    assert np.abs(lu.grid_sample_on_img(T(g['gs_img']), T(g['gs_xy'])).numpy() - g['gs_out']).max() <= 1e-6
    # colour loss + gradient
    co = T(g['lc_out'].copy()).requires_grad_(True)
    lc, _ = lu.compute_loss_color(co, T(g['lc_m1']), T(g['lc_gt']), T(g['lc_m2']))
    lc.backward()
    assert abs(float(lc) - float(g['lc_loss'])) <= 1e-6 and np.abs(co.grad.numpy() - g['lc_grad']).max() <= 1e-7
    # sim(3): exp of a so(3) vector by the 19-term series, [exp(s) R | t], gradients
    sim3 = {'rot': T(g['sim3_rot'].copy()).requires_grad_(True), 'scale': torch.tensor(float(g['sim3_scale']), requires_grad=True),
            'trans': T(g['sim3_trans'].copy()).requires_grad_(True)}
    M = tu.params_to_mtrx(sim3)
    (M * T(g['sim3_w'])).sum().backward()
    assert np.abs(M.detach().numpy() - g['sim3_mtrx']).max() <= 1e-6
    assert np.abs(sim3['rot'].grad.numpy() - g['sim3_g_rot']).max() <= 1e-5 and abs(float(sim3['scale'].grad) - float(g['sim3_g_scale'])) <= 1e-5
    assert np.abs(sim3['trans'].grad.numpy() - g['sim3_g_trans']).max() <= 1e-6
    assert np.abs(tu.get_lie_rotation_matrix(torch.tensor([1.3, -0.8, 2.1])).numpy() - g['lie_big']).max() <= 2e-5      # (a large rotation: the truncated series itself)


def test_load_decoder_matches_reference_golden(fixture_decoder, tmp_path):
    """G22 (oracle/gen_golden_load.py): the reference's load_decoder (decoder_utils.py:7-51) on a DeepSDF experiment directory -- the
    first call of every driver (run_single_shape.py:62-63: `load_decoder(dir, checkpoint)` then `.module.cuda()`). The drop-in one must
    hand back the same wrapper, state-dict keys / shapes, constructor attributes and eval-mode outputs, for the shape decoder (saved from
    DataParallel: 'module.' prefix) and for a colour decoder (specs of the shape experiment, weights of the colour experiment, no
    prefix); the packer must accept the loaded module (weight-norm folded)."""
    import torch
    import helpers
    from conftest import GOLDEN
    from core.utils.decoder_utils import load_decoder
    from distr import fixture, decoder_pack
    g = dict(np.load(os.path.join(GOLDEN, 'g22_load_decoder.npz')))
    Ws, bs, latent = fixture_decoder
    cWs, cbs = fixture.make_color_decoder_weights(color_size=16)[:2]
    d_shape = helpers.write_deepsdf_experiment(str(tmp_path / 'sofas'), decoder_pack.fixture_state_dict(Ws, bs, True), '2000', module_prefix=True)
    d_color = helpers.write_deepsdf_experiment(str(tmp_path / 'sofas_color'), decoder_pack.fixture_state_dict(cWs, cbs, True), 'latest', module_prefix=False)

    def check(dec, pre, x):
        assert type(dec).__name__ == str(g[pre + 'cls']) and type(dec.module).__name__ == str(g[pre + 'inner'])
        sd = dec.state_dict()
        assert sorted(sd.keys()) == [str(k) for k in g[pre + 'keys']]
        assert [list(sd[k].shape) + [0] * (2 - sd[k].dim()) for k in sorted(sd.keys())] == g[pre + 'shapes'].tolist()
        for a, want in zip(g['attr_names'], g[pre + 'attrs']):
            assert repr(getattr(dec.module, str(a))) == str(want), (a, getattr(dec.module, str(a)), want)
        dec.module.eval()
        with torch.no_grad():
            y = dec.module.inference(torch.from_numpy(x)).numpy()
        assert np.abs(y - g[pre + 'y']).max() <= 1e-6
    dec = load_decoder(d_shape, '2000')
    check(dec, 'shape_', g['x'])
    check(load_decoder(d_shape, 'latest', color_size=16, experiment_directory_color=d_color), 'color_', g['xc'])
    assert sorted(load_decoder(d_shape, None).state_dict().keys()) == [str(k) for k in g['nockpt_keys']]
    with pytest.raises(Exception):
        load_decoder(str(tmp_path / 'nowhere'), '2000')
    # what the renderer does with it: fold the weight norm, hand effective weights to the packer -- they are the fixture's
    Wse, bse = decoder_pack.effective_weights({k: v.numpy() for k, v in dec.module.state_dict().items()})
    for W, We in zip(Ws, Wse):
        assert np.abs(W - We).max() <= 2e-7 * max(1.0, np.abs(W).max())
    assert decoder_pack.pack_module(dec.module).size == sum(W.size + b.size for W, b in zip(Ws, bs))


def test_multi_view_pair_schedule_matches_reference_golden(monkeypatch):
    """G23 (oracle/gen_golden_schedule.py): the view pairs the reference's optimize_multi_view renders, round by round
    (optimize_multi.py:48-65: rot_freq, the sep_dist stride, the wrap-around rule), recorded from the reference's own loop -- against the
    pairs the drop-in loop hands to its round (and shards over the ranks: pairs[rank::world])."""
    import torch
    from conftest import GOLDEN
    import core.inv_optimizer.optimize_multi as om
    assert os.path.abspath(om.__file__).startswith(PKG)
    g = dict(np.load(os.path.join(GOLDEN, 'g23_multi_view_schedule.npz')))
    for (n, v, sep) in g['combos'].tolist():
        rounds = []

        def fake_round(renderer, shape_code, images, cameras, pairs, weight_list, **kw):
            rounds.append([list(p) for p in pairs])
            return shape_code.sum() * 0.0 + 1.0, {'color': torch.tensor(0.0), 'l2reg': torch.tensor(0.0)}
        monkeypatch.setattr(om, 'multi_view_round', fake_round)
        monkeypatch.setattr(om, '_StreamPool', lambda n_, dev: None)
        code = torch.zeros(1, 4, requires_grad=True)
        om.optimize_multi_view(None, None, code, torch.optim.SGD([code], lr=0.0), [None] * n, [None] * n, {}, num_views_per_round=v, num_iters=2,
                               sep_dist=sep, test_step=1000, distributed=False)
        assert rounds == g['pairs_%d_%d_%d' % (n, v, sep)].tolist(), (n, v, sep)


def test_generated_kernel_text_matches_its_generators():
    """The two hand-scheduled instruction streams in csrc/ are GENERATED text kept in the tree: distr_dense_asm.hpp (the 64-ray tile's
    k-loop) is the output of gen_dense_asm.py, the CL8_*_TEXT macro block of distr_mlp.hpp (the 8-member cluster tile's unit statements)
    the output of gen_cl8_units.py. Editing one side without the other must fail here."""
    import subprocess
    from conftest import ROOT
    csrc = os.path.join(ROOT, 'dist-renderer_amd', 'csrc')
    out = subprocess.run([sys.executable, os.path.join(csrc, 'gen_dense_asm.py')], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-500:]
    assert out.stdout == open(os.path.join(csrc, 'distr_dense_asm.hpp')).read()
    chk = subprocess.run([sys.executable, os.path.join(csrc, 'gen_cl8_units.py'), '--check', os.path.join(csrc, 'distr_mlp.hpp')],
                         capture_output=True, text=True, timeout=120)
    assert chk.returncode == 0, chk.stdout + chk.stderr


def test_cluster_members_own_disjoint_words_of_a_mask_block():
    """The arithmetic behind csrc/distr_kernels.hpp::store_own_mask_words (round 5): in store_mask_chunk's format (restated from
    distr_mlp.hpp::mask_nibble_put: 16-row block rb of `layer`, rows 4 kq .. 4 kq + 3 -> 32-bit word (idxu >> 1) + 16 (kq & 1) of the ray's
    128-word block, bits [shift, shift + 4), idxu = 64 w + 4 layer + ob) the rows a member of a cluster of CL computes are, per layer and
    half h, whole words -- except lin3 (256 rows) with 8 members, where two members share a word, 16 bits each. The words the members
    store (plus the zero words of lin3) cover all 128 words of the block exactly once, for CL = 8, 4, 2."""
    def put(layer, rb, wr_log, kq):
        R = 16 * rb
        w, ob = R >> wr_log, (R & ((1 << wr_log) - 1)) >> 5
        idxu = 64 * w + 4 * layer + ob
        return (idxu >> 1) + 16 * (kq & 1), 8 * (rb & 1) + 16 * (idxu & 1) + 4 * (kq >> 1)

    for CL in (8, 4, 2):
        written = {}                                   # (word, first bit, bits) -> member, as store_own_mask_words stores them
        for m in range(CL):
            for layer in range(8):
                for h in (0, 1):
                    if layer != 3:
                        for gi in range(8 // CL):
                            g = m * (8 // CL) + gi
                            written[(32 * (g >> 1) + 2 * layer + (g & 1) + 16 * h, 0, 32)] = m
                    elif CL == 8:
                        wi = 32 * (m >> 1) + 6 + 16 * h
                        written[(wi, 16 * (m & 1), 16)] = m
                        if m & 1:
                            written[(wi + 1, 0, 32)] = m
                    else:
                        for wq in range(4 // CL):
                            wi = 32 * (m * (4 // CL) + wq) + 6 + 16 * h
                            written[(wi, 0, 32)] = m
                            written[(wi + 1, 0, 32)] = m
        bits = np.zeros((128, 32), np.int32)
        for (word, lo, n), m in written.items():
            bits[word, lo:lo + n] += 1
        assert (bits == 1).all(), CL                    # every bit of the 512-byte block stored exactly once
        # the bits of every row land in a piece stored by the member that computes the row
        for layer in range(8):
            nblk, wr_log = (16, 6) if layer == 3 else (32, 7)
            for rb in range(nblk):
                owner = rb // (nblk // CL)
                for kq in range(4):
                    word, shift = put(layer, rb, wr_log, kq)
                    piece = [k for k in written if k[0] == word and k[1] <= shift < k[1] + k[2]]
                    assert len(piece) == 1 and written[piece[0]] == owner, (CL, layer, rb, kq)


def test_rccl_launch_with_more_ranks_than_gpus_is_refused(monkeypatch):
    """under backend 'nccl' (= RCCL) a LOCAL_RANK beyond the visible GPUs is a mis-launched job and is REFUSED before any
    communicator exists -- not wrapped onto a shared GPU. The wrap stays for the gloo test rigs that time-share one GPU on purpose."""
    import torch
    from distr import parallel
    monkeypatch.setattr(torch.cuda, 'is_available', lambda: True)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 2)
    monkeypatch.setenv('WORLD_SIZE', '8')
    monkeypatch.setenv('RANK', '5')
    monkeypatch.setenv('LOCAL_RANK', '5')
    with pytest.raises(parallel.LaunchError) as e:
        parallel.init_from_env(backend='nccl')
    assert 'RCCL ranks cannot share a device' in str(e.value) and 'only 2 GPU' in str(e.value)
    monkeypatch.delenv('DISTR_DIST_BACKEND', raising=False)
    with pytest.raises(parallel.LaunchError):
        parallel.init_from_env()                     # (default backend with a GPU visible: nccl)
    # a single process, or the gloo rig: wrapped as before (no process group is created at world size 1)
    monkeypatch.setenv('WORLD_SIZE', '1')
    monkeypatch.setenv('RANK', '0')
    assert parallel.init_from_env(backend='gloo') == (0, 1, 1)
    assert parallel.init_from_env(backend='nccl') == (0, 1, 1)


@pytest.mark.parametrize('workload,n', [('c3', 8), ('c5', 8), ('c5', 2), ('c3', 4)])
def test_bench_plan_only(workload, n):
    """`bench.py --gpus N --plan-only`: the partition and the predicted per-rank load of an N-rank run as ONE JSON line, without a GPU and
    without starting ranks. The partition is the one the real run starts from (parallel.shard_views / shard_rows)."""
    import json
    import subprocess
    from distr import parallel
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--plan-only', '--workload', workload], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['plan_only'] is True and d['n_gpus'] == n and d['workload'] == workload and len(d['partition']) == n == len(d['predicted_ms_per_rank'])
    if workload == 'c5':
        assert d['scaling'] == 'strong'
        rows = {}
        for r, pieces in enumerate(d['partition']):
            assert [tuple(p[:1] + p[2:]) for p in pieces] == [tuple(x) for x in parallel.shard_rows(4, 1024, r, n)]
            for (img, _, r0, r1) in pieces:
                assert r0 % 4 == 0
                rows.setdefault(img, []).append((r0, r1))
        for img in range(4):          # every image tiled exactly once
            segs = sorted(rows[img])
            assert segs[0][0] == 0 and segs[-1][1] == 1024 and all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
    else:
        assert d['scaling'] == 'weak' and [p[0][1] for p in d['partition']] == list(range(n))
    assert max(d['predicted_ms_per_rank']) == d['predicted_slowest_ms'] and 0.5 < d['predicted_efficiency_vs_n1'] <= 1.0


def test_build_is_split_into_parallel_units_and_checked():
    """libdistr.so = distr_api.hip + one translation unit per group of explicit kernel instantiations (distr_inst.hpp), compiled side by side;
    every unit keeps its device assembly, and build() runs the generated-code checks on it (the checkers were claimed, not wired)."""
    import re
    from distr import binding
    steps, link = binding.build_commands()
    labels = [s[0] for s in steps]
    assert labels == ['api'] + ['inst%d' % g for g in range(1, binding.INST_GROUPS + 1)]
    hdr = open(os.path.join(binding.CSRC, 'distr_inst.hpp')).read()
    assert int(re.search(r'DISTR_NUM_INST_GROUPS = (\d+)', hdr).group(1)) == binding.INST_GROUPS
    assert sorted(set(int(g) for g in re.findall(r'DISTR_GROUP_ON\((\d+)\)', hdr))) == list(range(1, binding.INST_GROUPS + 1))
    for label, cmd, out in steps:
        assert '--offload-arch=gfx950' in cmd and '-save-temps=obj' in cmd and '-ffp-contract=off' in cmd
        assert ('-DDISTR_INST_GROUP=%s' % label[4:] in cmd) == label.startswith('inst')
    assert all(s[2] in link for s in steps) and link[-len(steps) - 1] == binding.LIB_PATH or binding.LIB_PATH in link
    entry = open(os.path.join(ROOT, '__graft_entry__.py')).read()
    assert 'check_generated_code' in entry
    # every kernel that holds the cluster tile is on the list the checkers walk
    assert {k for _, k in binding.CLUSTER_KERNELS} >= {'k_stepILb1ELi0', 'k_stepILb0ELi0', 'k_tailILb1', 'k_tailILb0', 'k_march16ILi1ELb1', 'k_march16ILi1ELb0'}
    asm = os.path.join(binding.CSRC, '_obj', 'inst1', 'distr_inst-hip-amdgcn-amd-amdhsa-gfx950.s')
    if os.path.exists(asm) and os.path.exists(binding.LIB_PATH) and os.path.getmtime(binding.LIB_PATH) >= os.path.getmtime(asm):
        res = binding.check_generated_code()          # (the build of this checkout, if it is there: raises on a finding)
        assert any('k_step' in n for n in res)


def test_bench_live_traffic_failure_paths(monkeypatch, tmp_path):
    """bench.live_traffic (roofline.traffic measured by the run itself: two child runs under rocprofv3 --pmc) never takes the bench line down:
    without rocprofv3, or inside a profiled run (no nested profiler), it returns (None, reason) and the caller keeps the committed number. A
    fake rocprofv3 that writes the counter CSV shows the arithmetic: (FETCH_SIZE x 2 + WRITE_SIZE) KiB x 1024 / march launches, march kernels only."""
    import stat
    import bench
    monkeypatch.setenv('ROCPROFILER_TEST_MARK', '1')
    v, why = bench.live_traffic()
    assert v is None and 'profiled' in why
    monkeypatch.delenv('ROCPROFILER_TEST_MARK')
    for k in list(os.environ):
        if k.startswith(('ROCPROF', 'ROCP_')):
            monkeypatch.delenv(k)
    monkeypatch.setenv('LD_PRELOAD', '')
    import shutil
    monkeypatch.setattr(shutil, 'which', lambda name: None)
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, 'exists', lambda p: False if p == '/opt/rocm/bin/rocprofv3' else real_exists(p))
    v, why = bench.live_traffic()
    assert v is None and 'not found' in why
    # a stand-in profiler: parses `--pmc <counter> ... -d <dir> --`, writes <dir>/x/1_counter_collection.csv, runs nothing
    fake = tmp_path / 'rocprofv3'
    fake.write_text('#!%s\nimport os, sys\na = sys.argv\nc = a[a.index("--pmc") + 1]\nd = a[a.index("-d") + 1]\n'
                    'os.makedirs(os.path.join(d, "x"))\nv = {"FETCH_SIZE": 100.0, "WRITE_SIZE": 50.0}[c]\n'
                    'rows = ["Kernel_Name,Counter_Name,Counter_Value"] + ["\\"void distr::k_step<true, 0>(A, B)\\",%%s,%%g" %% (c, v)] * 3 + '
                    '["\\"void distr::k_march<1, 2, true, 0>(A, B)\\",%%s,%%g" %% (c, v), "\\"distr::k_bwd<2, 2, 0>(A)\\",%%s,1e9" %% c]\n'
                    'open(os.path.join(d, "x", "1_counter_collection.csv"), "w").write("\\n".join(rows) + "\\n")\n' % sys.executable)
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setattr(shutil, 'which', lambda name: str(fake))
    monkeypatch.setattr(os.path, 'exists', real_exists)
    v, info = bench.live_traffic()
    assert isinstance(info, dict), info
    assert info['march_launches'] == 4 and v == (4 * 100.0 * 2.0 + 4 * 50.0) * 1024.0 / 4
