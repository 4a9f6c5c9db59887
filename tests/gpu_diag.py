"""First-contact diagnostics on the MI355X box (not a pytest file): prints layer-by-layer residuals of the fused
decoder vs the CPU oracle, render residuals per marcher, and a dense-throughput probe. Every stage is isolated so a
failure in one still lets the others report. Usage: python tests/gpu_diag.py  (writes to stdout)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'dist-renderer_amd'), ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch


def stage(name):
    def deco(fn):
        def run(*a, **k):
            print('\n=== %s' % name, flush=True)
            t0 = time.time()
            try:
                fn(*a, **k)
                print('--- %s done in %.2fs' % (name, time.time() - t0), flush=True)
            except Exception:
                traceback.print_exc()
                print('--- %s FAILED' % name, flush=True)
        return run
    return deco


def main():
    print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
    print('cpu cores', os.cpu_count())
    from distr import binding, fixture, functions
    from oracle import oracle as orc
    import helpers
    orc.build()
    Ws, bs, latent = fixture.make_decoder_weights()
    O = orc.Oracle(Ws, bs)
    eng = functions.engine_from_weights(Ws, bs, 0)
    lat_t = torch.from_numpy(latent)
    rs = np.random.RandomState(3)
    pts = (rs.rand(64 * 2 + 9, 3) * 1.8 - 0.9).astype(np.float32)

    @stage('decoder layers vs oracle')
    def layers():
        for layer in range(8):
            got = functions.debug_mlp_layer(eng, lat_t, torch.from_numpy(pts), layer).cpu().numpy()
            ref = O.layer_activations(latent, pts, layer)
            w = 256 if layer == 3 else 512
            d = np.abs(got[:, :w] - ref[:, :w])
            bad = np.argwhere(d > 0)
            print('layer %d: max diff %.3e, mismatches %d / %d, ref absmax %.3f, got absmax %.3f, nan %d' %
                  (layer, d.max(), len(bad), d.size, np.abs(ref[:, :w]).max(), np.nanmax(np.abs(got[:, :w])), np.isnan(got).sum()))
            if len(bad):
                for (r, f) in bad[:6]:
                    print('    ray %d feature %d: got %.8g ref %.8g' % (r, f, got[r, f], ref[r, f]))
                rows = np.unique(bad[:, 1])
                print('    mismatching features (first 40):', rows[:40], 'rays:', np.unique(bad[:, 0])[:20])
    layers()

    @stage('decode_sdf / gradient vs oracle')
    def evals():
        got = functions.mlp_eval(eng, lat_t, torch.from_numpy(pts)).cpu().numpy().reshape(-1)
        ref = O.decode_sdf(latent, pts)
        print('sdf max diff %.3e (ref range %.4f..%.4f) first got %s ref %s' % (np.abs(got - ref).max(), ref.min(), ref.max(), got[:4], ref[:4]))
        s, g = functions.mlp_grad(eng, lat_t, torch.from_numpy(pts))
        s_ref, g_ref = O.decode_sdf_and_gradient(latent, pts)
        print('grad: sdf diff %.3e, grad max diff %.3e (ref absmax %.3f)' % (np.abs(s.cpu().numpy() - s_ref).max(),
              np.abs(g.cpu().numpy() - g_ref).max(), np.abs(g_ref).max()))
        print('  got', g.cpu().numpy()[:3], '\n  ref', g_ref[:3])
    evals()

    @stage('render C1 vs oracle')
    def renders():
        H = W = 64
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(30, 20, 1.6, 10)
        for marcher in ('trivial', 'recursive', 'pyramid_recursive'):
            for d2n in (False, True):
                kw = dict(march_step=20, buffer_size=3, marcher=marcher, use_depth2normal=d2n)
                try:
                    a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
                    b = helpers.oracle_render(O, orc, H, W, K, R, T, latent, **kw)
                    ma, mb = a['mask'].astype(bool), b['mask'].astype(bool)
                    both = (ma & mb).reshape(H, W)
                    print('%s d2n=%d: valid %d/%d flips %d | depth %.2e zdepth %.2e minsdf %.2e normal %.2e | glat %.2e gR %.2e gT %.2e' % (
                        marcher, d2n, ma.sum(), mb.sum(), (ma != mb).sum(),
                        np.abs(a['depth'] - b['depth'])[both].max() if both.any() else -1,
                        np.abs(a['zdepth'] - b['zdepth'])[ma & mb].max() if both.any() else -1,
                        np.abs(a['min_sdf'] - b['min_sdf']).max(),
                        np.abs(a['normal'] - b['normal'])[both].max() if both.any() else -1,
                        np.abs(a['g_latent'] - b['g_latent']).max() / np.abs(b['g_latent']).max(),
                        np.abs(a['g_R'] - b['g_R']).max() / np.abs(b['g_R']).max(),
                        np.abs(a['g_T'] - b['g_T']).max() / np.abs(b['g_T']).max()), flush=True)
                except Exception:
                    traceback.print_exc()
    renders()

    @stage('dense decoder throughput (262144 points = one full 512x512 step)')
    def thr():
        n = 512 * 512
        p = torch.rand(n, 3, device='cuda') * 1.6 - 0.8
        for _ in range(2):
            functions.mlp_eval(eng, lat_t, p)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 5
        for _ in range(reps):
            functions.mlp_eval(eng, lat_t, p)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print('mlp_eval %d pts: %.3f ms -> %.1f TFLOP/s (%.1f%% of 157.3)' % (n, ms, 3146752 * n / ms / 1e9, 3146752 * n / ms / 1e9 / 157.3 * 100))
        s, g = functions.mlp_grad(eng, lat_t, p)
        torch.cuda.synchronize()
        e0.record()
        functions.mlp_grad(eng, lat_t, p)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print('mlp_grad %d pts: %.3f ms -> %.1f TFLOP/s (fwd+bwd dX = 2x flops)' % (n, ms, 2 * 3146752 * n / ms / 1e9))
    thr()

    @stage('cumulative per-layer time (262144 points, 64-ray tiles): layer l - layer l-1 = cost of layer l incl. overheads')
    def layer_times():
        n = 512 * 512
        p = torch.rand(n, 3, device='cuda') * 1.6 - 0.8
        prev = None
        for layer in range(8):
            for _ in range(2):
                functions.debug_mlp_layer(eng, lat_t, p, layer)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                functions.debug_mlp_layer(eng, lat_t, p, layer)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            mac = {0: 8 * 512, 1: 512 * 512, 2: 512 * 512, 3: 512 * 256, 4: 256 * 512, 5: 512 * 512, 6: 512 * 512, 7: 512 * 512}[layer]
            ideal = 2.0 * mac * n / 157.3e12 * 1e3
            print('  up to layer %d: %.3f ms (delta %.3f ms, MFMA-ideal for this layer %.3f ms)' % (layer, ms, ms - (prev or 0), ideal))
            prev = ms
    layer_times()

    @stage('in-kernel phase timing of the decoder tile (wave 0 of each workgroup)')
    def phases():
        import os as _os
        tile = 64
        for n, label in ((512 * 512, 'dense: 4096 tiles of 64'), (tile * 200, '200 tiles (chip mostly idle)')):
            p = torch.rand(n, 3, device='cuda') * 1.6 - 0.8
            functions.debug_tile_timing(eng, lat_t, p, tile)
            sdf, ts = functions.debug_tile_timing(eng, lat_t, p, tile)
            torch.cuda.synchronize()
            ref = functions.mlp_eval(eng, lat_t, p).reshape(-1)
            print('  [%s] sdf equal to mlp_eval: %s' % (label, bool((sdf == ref).all())))
            t = ts.cpu().numpy().astype(np.float64)
            cyc, wall = t[:, :, 0], t[:, :, 1]
            tot_c, tot_w = cyc[:, 18] - cyc[:, 0], (wall[:, 18] - wall[:, 0]) * 10.0   # wall: 100 MHz -> ns
            print('  [%s] tile total: %.0f cycles, %.1f us  -> shader clock %.3f GHz' % (label, np.median(tot_c), np.median(tot_w) / 1e3, np.median(tot_c / tot_w)))
            names = ['L0 mfma', 'L0 wb', 'L1 mfma', 'L1 wb', 'L2 mfma', 'L2 wb', 'L3 mfma', 'L3 wb', 'L4 mfma', 'L4 wb', 'L5 mfma', 'L5 wb',
                     'L6 mfma', 'L6 wb', 'L7 mfma', 'L7 wb', 'L8', 'tail']
            d = np.diff(cyc[:, :19], axis=1)
            print('   ' + ' | '.join('%s %.0f' % (nm, np.median(d[:, i])) for i, nm in enumerate(names)))
            mf = sum(np.median(d[:, i]) for i in range(0, 16, 2)); wb = sum(np.median(d[:, i]) for i in range(1, 16, 2))
            print('   sum mfma-loop phases %.0f (ideal 787456 at 64 cyc/MFMA for 64-ray tiles), write-back+barrier phases %.0f, L8+tail %.0f' % (mf, wb, np.median(d[:, 16]) + np.median(d[:, 17])))
    phases()

    @stage('C3 fwd/bwd timing')
    def c3():
        H = W = 512
        K = fixture.make_intrinsic(H, W)
        R, T = fixture.make_camera(0, 0, 1.6, 0)
        for marcher in ('pyramid_recursive', 'recursive', 'trivial'):
            kw = dict(march_step=50, buffer_size=3, marcher=marcher, use_depth2normal=True)
            helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a = helpers.hip_render(eng, H, W, K, R, T, latent, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            cfg = a['cfg']
            fwd_bytes, _ = eng.ctx.workspace_bytes(cfg)
            print('%s: fwd+bwd %.1f ms (%.0f rays/s), valid %d, loss %.3f' % (marcher, dt * 1e3, H * W / dt, a['mask'].sum(), a['loss']))
    c3()


if __name__ == '__main__':
    main()
