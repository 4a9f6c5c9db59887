/*
 * distr.h -- C ABI of libdistr.so: MI355X-native differentiable sphere tracing of DeepSDF decoders.
 *
 * Drop-in boundary for the hot path of B1ueber2y/DIST-Renderer (all citations: paths under the
 * reference tree, file:line). The reference has no native interface -- its hot path is Python that
 * issues hundreds of ATen launches per march step -- so each entry point below replaces a Python
 * method; the Python mirror in dist-renderer_amd/core/ calls these through ctypes with
 * tensor.data_ptr() and the current HIP stream (INTEGRATION.md shows the binding).
 *
 * Conventions
 *   - plain C, no C++ / torch types; every pointer named *_dev is DEVICE memory owned by the caller;
 *   - the library never allocates or frees device memory inside forward/backward and never
 *     synchronises the device: all work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - all functions return 0 (DISTR_OK) or a negative error code; distr_last_error() gives text;
 *   - a context is bound to one HIP device; it is not thread-safe, distinct contexts are independent;
 *   - everything is float32 except masks (uint8) -- the same types the reference computes in.
 *
 * ABI handshake (the reference's counterpart is the constructor contract SDFRenderer.__init__, renderer.py:13: a caller that
 * passes the wrong arguments gets a TypeError, not a corrupted render). Two checks, so that a caller compiled against an older
 * header fails loudly instead of passing shifted fields:
 *   - distr_create is a macro over distr_create_abi(out, device, DISTR_ABI_VERSION): the library refuses (DISTR_ERR_INVALID_ARG)
 *     a caller built for another ABI version; a binary that still links the old `distr_create` symbol does not load at all;
 *   - every struct that crosses the boundary starts with `uint32_t struct_size`, which the caller sets to sizeof(the struct)
 *     (DISTR_INIT zeroes a struct and sets it); every entry point that takes the struct checks it (DISTR_ERR_INVALID_ARG), and
 *     distr_get_render_stats never writes past the size the caller announced.
 */
#ifndef DISTR_H_
#define DISTR_H_

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DISTR_ABI_VERSION 6u /* bumped whenever a struct layout or an entry point's signature changes (4: struct_size fields, distr_create_abi;
                                5: distr_render_stats.tail_from / tail_steals; 6: distr_render_cfg.num_levels / level_scale / level_steps) */

/* zero a boundary struct and announce its size: distr_render_cfg cfg; DISTR_INIT(cfg); cfg.H = ...; */
#define DISTR_INIT(s) do { memset(&(s), 0, sizeof(s)); (s).struct_size = (uint32_t)sizeof(s); } while (0)

#define DISTR_OK 0
#define DISTR_ERR_INVALID_ARG (-1)
#define DISTR_ERR_UNSUPPORTED (-2)
#define DISTR_ERR_HIP (-3)
#define DISTR_ERR_WORKSPACE (-4)
#define DISTR_ERR_NO_DECODER (-5)

#define DISTR_MARCH_TRIVIAL 0           /* SDFRenderer.ray_marching_trivial            renderer.py:472 */
#define DISTR_MARCH_RECURSIVE 1         /* SDFRenderer.ray_marching_recursive          renderer.py:512 */
#define DISTR_MARCH_PYRAMID_RECURSIVE 2 /* SDFRenderer.ray_marching_pyramid_recursive  renderer.py:713 */

#define DISTR_ARITH_F32 0    /* exact f32 MFMA (default): bit-identical to a k-ordered fmaf chain */
#define DISTR_ARITH_BF16X6 1 /* six-product split-bf16 (opt-in): f32-equivalent accuracy, ~1.6x the dense rate, not bit-identical */
#define DISTR_ARITH_F16X3 2  /* three-product split-f16 (opt-in): f32-equivalent accuracy for decoders whose weights and activations stay
                                below 1023 in magnitude (f16 range after the x64 scaling; checked: DISTR_ERR_UNSUPPORTED for the weights,
                                distr_render_stats.f16_overflows / NaN for activations), not bit-identical */

#define DISTR_MAX_BUFFER_SIZE 8
#define DISTR_MAX_PYRAMID_LEVELS 4 /* len(scale_list) of the pyramid marcher (distr_render_cfg::num_levels) */
#define DISTR_MAX_VIEWS 64 /* views per batched render (distr_render_forward_batch) */

/* per-view gradient switches of a batched render (view_flags[]): the no_grad_* keyword arguments of render_depth / render
 * (renderer.py:836, 943) may differ between the views of one batch (render_warp renders view 2 with no_grad_depth,
 * renderer_warp.py:108-109) */
#define DISTR_VIEW_GRAD_DEPTH 1
#define DISTR_VIEW_GRAD_MASK 2
#define DISTR_VIEW_GRAD_CAMERA 4

typedef struct distr_ctx distr_ctx;

/* Architecture of the decoder: replaces Decoder.__init__ (core/graph/deep_sdf_decoder.py:19-73) +
 * specs.json handling of load_decoder (core/utils/decoder_utils.py:7-27). The kernels are specialised
 * for DeepSDF "8x512": latent 256, hidden 512, latent_in=[4], ReLU, final tanh, no LayerNorm. */
typedef struct distr_decoder_desc {
  uint32_t struct_size; /* sizeof(distr_decoder_desc) */
  int32_t latent_size;  /* 256 */
  int32_t hidden;       /* 512 */
  int32_t num_linear;   /* 9  (lin0..lin8) */
  int32_t latent_in;    /* 4  */
} distr_decoder_desc;

/* Per-renderer + per-call options: SDFRenderer.__init__ (renderer.py:13-59) and the keyword arguments
 * of render / render_depth (renderer.py:836, 943). */
typedef struct distr_render_cfg {
  uint32_t struct_size;       /* sizeof(distr_render_cfg): checked by every entry point that takes a cfg (ABI handshake, top of this file) */
  int32_t H, W;               /* img_hw                                                     renderer.py:31-35 */
  float K_inv[9];             /* float32(inv(K)), row-major                                 renderer.py:161-164 */
  float fx, fy;               /* K[0,0], K[1,1] (depth2normal)                              renderer.py:973-974 */
  float M[9];                 /* transform_matrix (3x3) applied (inverted) to the sample POINTS; identity when use_transform=False
                                 renderer.py:45, 100-120, 202-223 */
  float M_normal[9];          /* matrix applied to the surface NORMALS: always the constructor's transform_matrix -- render_normal calls
                                 transform_points unconditionally (renderer.py:899), use_transform only switches the points' inverse
                                 transform (:895); golden G24 */
  int32_t march_step;         /* renderer.py:18  */
  int32_t buffer_size;        /* renderer.py:19  (<= DISTR_MAX_BUFFER_SIZE) */
  float ratio;                /* ray_marching_ratio renderer.py:21 */
  float threshold;            /* renderer.py:24  */
  float radius;               /* renderer.py:23  */
  float clamp_dist;           /* clamp_dist kwarg of render()                               renderer.py:943 */
  int32_t marcher;            /* DISTR_MARCH_*   ray_marching_type                          renderer.py:807-834 */
  int32_t coarse_steps[2];    /* march_step_list[0:2] for scale_list=[4,2,1]; {s, 0}: scale_list=[2,1] with march_step_list=[s,-1]
                                 (steps of the coarse levels, coarsest first, 0 = no such level) renderer.py:25-26, 724-765 */
  int32_t use_depth2normal;   /* renderer.py:22, 972-975 */
  int32_t normalize_normal;   /* normalize_normal kwarg                                     renderer.py:898-901 */
  int32_t want_normal;        /* 0: render_depth() only (no depth/normal image outputs) */
  int32_t grad_depth;         /* !no_grad_depth                                             renderer.py:410-411, 876-877 */
  int32_t grad_mask;          /* !no_grad_mask                                              renderer.py:388-389 */
  int32_t grad_camera;        /* !no_grad_camera                                            renderer.py:536-542 */
  int32_t save_for_backward;  /* 1: distr_render_backward will be called on this forward's workspace: the march kernel also
                                 saves the ReLU masks of the rows it keeps (512 B each) so that the backward pass need not
                                 recompute the decoder; 0: inference only (smaller workspace, backward not allowed) */
  int32_t row0, rows;         /* row band (strong scaling of one large view over several GPUs): render only image
                                 rows [row0, row0+rows) of the H x W image; rows == 0: the whole image. row0 must be a multiple
                                 of 4 and rows a multiple of 4 unless the band ends at H (keeps the 4x4 pyramid parents of
                                 renderer.py:732-749 intact), so every ray of the band is bit-identical to the same ray of a
                                 full render. All per-pixel buffers (outputs, upstream gradients, P in the workspace sizes) then
                                 cover rows*W band pixels. depth2normal needs the rows above/below: the first/last band row that
                                 is not an image border row gets an undefined normal -- callers render a halo of 4 rows and
                                 crop it (distr/functions.py::render_band_call). */
  int32_t arith;              /* DISTR_ARITH_*: arithmetic of the decoder evaluations of the MARCH (no counterpart in the reference, which has
                                 one arithmetic: f32). 0 (default, also what a zeroed struct selects): exact f32. 1: every f32 product of
                                 the seven wide layers as six bf16 products with f32 accumulation (csrc/distr_mlp_b6.hpp): values within
                                 ~1e-6 of the exact ones, on 64- / 32-ray tiles only (no cluster tiles: meant for large, dense renders). The
                                 backward pass is the split-bf16 dX chain on the ReLU masks this forward saved. 2: three f16 products per f32
                                 product with the activations kept as two f16 planes in LDS (csrc/distr_mlp_h3.hpp): same accuracy class and
                                 tile set as 1 for decoders inside the f16 range (see DISTR_ARITH_F16X3), backward = that of mode 1. */
  int32_t concurrent;         /* hint, 0 by default: 1 = the caller has OTHER renders in flight on other streams of this device (a pool of
                                 streams over the scales of a multi-scale renderer list or the view pairs of a round). The tail of the march
                                 then does not turn "sticky": a sticky launch keeps up to 248 compute units to itself for milliseconds, which
                                 is the fastest way to finish ONE small render and the slowest way to share the chip. Values never depend on
                                 it (tests/gpu_diag_multiscale.py: three scales on three streams 23.8 -> 20.6 ms per iteration). */
  int32_t num_levels;         /* pyramid marcher, ABI 6. 0 (what a zeroed struct selects): the pyramid is described by coarse_steps above
                                 (scale_list [4,2,1], or [2,1] with coarse_steps = {s, 0}). 2..DISTR_MAX_PYRAMID_LEVELS: the general form
                                 of ray_marching_pyramid_recursive (renderer.py:713-805: one level per scale_list entry) --                */
  int32_t level_scale[4];     /* scale_list, coarsest first; integers, the last one 1, each a multiple of the next (the ratio of two
                                 consecutive levels is the `scale` of get_downscaled_grid_map, renderer.py:604-629, 739)                  */
  int32_t level_steps[4];     /* march_step_list of the coarse levels (1..15 steps each), coarsest first; entry num_levels-1 is ignored:
                                 the full-resolution level marches march_step minus the coarse steps (renderer.py:724-725)               */
} distr_render_cfg;

/* Counters of one forward call (read back with distr_get_render_stats). */
typedef struct distr_render_stats {
  uint32_t struct_size;       /* IN: sizeof(distr_render_stats) of the caller; distr_get_render_stats refuses any other size */
  uint32_t reserved;
  int64_t num_in_sphere;      /* rays hitting the unit sphere (N of renderer.py:473)   */
  int64_t num_march_launches; /* launches of the fused march/MLP kernel                */
  int64_t num_point_evals;    /* decoder evaluations executed by the march kernel      */
  int64_t num_valid;          /* final valid pixels                                    */
  int64_t num_grad_samples;   /* (backward) gradient-carrying samples of the last backward on this workspace */
  int64_t cluster_fallbacks;  /* cluster tiles (16 rays split over 8 / 4 compute units) whose workgroups were not co-resident in time
                                 (compute units held by other streams / ranks) or whose barrier timed out: the tile's lead workgroup
                                 evaluated it alone instead -- same values bit for bit, only slower. 0 on an otherwise idle GPU. */
  int64_t f16_overflows;      /* arith = DISTR_ARITH_F16X3 only: decoder evaluations of the march whose result was not finite because an
                                 activation left the f16 range. Anything but 0 means the render is not to be trusted: use arith 1 or 0. */
  int64_t tail_from;          /* first full-resolution march step that ran inside the persistent tail launch (one launch for ALL remaining
                                 steps, no launch behind the last live ray); = the number of full-resolution steps when there was none.
                                 num_march_launches counts the tail launch once. */
  int64_t tail_steals;        /* tiles of the tail launch evaluated by a workgroup other than their owner because the owner was not resident
                                 in time (compute units held by another stream / process): same values, only slower. 0 on an idle GPU. */
} distr_render_stats;

/* distr_create(&ctx, device): a macro, so that the ABI version the CALLER was compiled with reaches the library. A context is
 * returned even on failure (read distr_last_error, then distr_destroy). */
int distr_create_abi(distr_ctx** out, int hip_device, uint32_t abi_version);
#define distr_create(out, hip_device) distr_create_abi((out), (hip_device), DISTR_ABI_VERSION)
uint32_t distr_abi_version(void); /* DISTR_ABI_VERSION the LIBRARY was built with */
void distr_destroy(distr_ctx* ctx);
const char* distr_last_error(const distr_ctx* ctx);
const char* distr_version(void);

/* Upload + pack decoder weights. `weights_host` = for l in 0..8: W_l row-major (out,in) then b_l (effective
 * weights, weight-norm already folded); replaces load_decoder()'s load_state_dict (decoder_utils.py:33-50). */
int distr_set_decoder(distr_ctx* ctx, const distr_decoder_desc* desc, const float* weights_host, size_t n_floats);

/* Bytes the caller must provide as `ws_dev` to forward (it doubles as the saved-for-backward buffer and must
 * stay untouched until backward has run) and as `ws_bwd_dev` to backward. */
int distr_workspace_bytes(distr_ctx* ctx, const distr_render_cfg* cfg, size_t* forward_bytes, size_t* backward_bytes);

/* SDFRenderer.render_depth (renderer.py:836-878) and, when cfg->want_normal, the rest of
 * SDFRenderer.render (renderer.py:943-999).
 *   latent_dev[256], R_dev[9] row-major, T_dev[3]
 *   zdepth_dev[H*W]   Zdepth   (1e11 where the ray misses the unit sphere)
 *   mask_dev[H*W]     final valid mask (uint8 0/1)
 *   min_sdf_dev[H*W]  min_sdf_sample / min_abs_query
 *   depth_dev[H*W]    (want_normal) depth = Zdepth*calib on valid px, else 1e11 -- 0 when use_depth2normal
 *                     (the reference's depth2normal zeroes the background in place, render_utils.py:24-25)
 *   normal_dev[H*W*3] (want_normal) (h,w,3) normal map
 * Output pointers may be NULL when not wanted. */
int distr_render_forward(distr_ctx* ctx, const distr_render_cfg* cfg, const float* latent_dev, const float* R_dev,
                         const float* T_dev, float* zdepth_dev, uint8_t* mask_dev, float* min_sdf_dev,
                         float* depth_dev, float* normal_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* What loss.backward() does through the reference's autograd tape (optimize_single.py:83): upstream gradients of
 * the forward outputs (any may be NULL) -> gradients of latent (256), R (9), T (3). `ws_dev` is the forward
 * workspace of the SAME cfg/inputs. */
int distr_render_backward(distr_ctx* ctx, const distr_render_cfg* cfg, const void* ws_dev, size_t ws_bytes,
                          const float* g_zdepth_dev, const float* g_min_sdf_dev, const float* g_depth_dev,
                          const float* g_normal_dev, float* g_latent_dev, float* g_R_dev, float* g_T_dev,
                          void* ws_bwd_dev, size_t ws_bwd_bytes, void* stream);

/* Several views in ONE launch sequence: what the reference does view by view in the multi-view round (16 render_depth calls with the
 * same shape code before a single backward(), core/inv_optimizer/optimize_multi.py:62-81, renderer_warp.py:108-109) and shape by
 * shape for a batch of shapes. All `nviews` views share cfg (image size, intrinsics, marcher, steps); each has its own camera
 * R_dev[v][9], T_dev[v][3] and shape code latent_dev + v * latent_stride (floats; 0 = one code shared by all views). Every march
 * step is one launch over the live rays of all views (no tile mixes two views, so every ray's arithmetic -- and every output
 * byte -- is exactly that of a stand-alone distr_render_forward of its view); the latency-bound tail steps of the views overlap
 * instead of queueing behind each other.
 *   outputs     [nviews][H*W] (normal: [nviews][H*W][3]), same meaning as distr_render_forward
 *   view_flags  HOST array [nviews] of DISTR_VIEW_GRAD_* (may only clear bits that cfg's grad_* fields have set), or NULL = cfg's
 *   ws_dev      nviews x forward_bytes of distr_workspace_bytes(cfg); view v's workspace (distr_get_render_stats,
 *               distr_get_live_counts) starts at ws_dev + v * forward_bytes
 * distr_render_backward_batch: upstream gradients [nviews][H*W] (g_normal [nviews][H*W][3], any may be NULL) -> g_latent
 * [nviews][256], g_R [nviews][9], g_T [nviews][3] (per view: a shared shape code's gradient is the sum over v, left to the caller
 * like the reference leaves it to autograd); ws_bwd_dev = nviews x backward_bytes. Every view's gradients are bit-identical to a
 * stand-alone distr_render_backward of that view (each view keeps its own tile decomposition and reduction order). */
int distr_render_forward_batch(distr_ctx* ctx, const distr_render_cfg* cfg, int32_t nviews, const int32_t* view_flags,
                               const float* latent_dev, int64_t latent_stride, const float* R_dev, const float* T_dev,
                               float* zdepth_dev, uint8_t* mask_dev, float* min_sdf_dev, float* depth_dev, float* normal_dev,
                               void* ws_dev, size_t ws_bytes, void* stream);
int distr_render_backward_batch(distr_ctx* ctx, const distr_render_cfg* cfg, int32_t nviews, const void* ws_dev, size_t ws_bytes,
                                const float* g_zdepth_dev, const float* g_min_sdf_dev, const float* g_depth_dev,
                                const float* g_normal_dev, float* g_latent_dev, float* g_R_dev, float* g_T_dev,
                                void* ws_bwd_dev, size_t ws_bwd_bytes, void* stream);

/* SDFRenderer.render_normal (renderer.py:880-910) for caller-provided Zdepth/mask: writes (3, H*W) like the
 * reference (untransformed by R; render() applies R and the x-flip itself). */
int distr_render_normal(distr_ctx* ctx, const distr_render_cfg* cfg, const float* latent_dev, const float* R_dev,
                        const float* T_dev, const float* zdepth_dev, const uint8_t* mask_dev, float* normal3xP_dev,
                        void* ws_dev, size_t ws_bytes, void* stream);
/* The same for nviews views in one launch sequence (zdepth / mask [nviews][H*W] -> normal [nviews][3][H*W]; cameras, shape codes
 * and workspace as in distr_render_forward_batch). */
int distr_render_normal_batch(distr_ctx* ctx, const distr_render_cfg* cfg, int32_t nviews, const float* latent_dev,
                              int64_t latent_stride, const float* R_dev, const float* T_dev, const float* zdepth_dev,
                              const uint8_t* mask_dev, float* normal3xP_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* decode_sdf (core/utils/decoder_utils.py:53-74): n points xyz_dev[n][3] -> sdf_dev[n]; clamp_dist < 0 = no clamp.
 * decode_sdf_gradient (decoder_utils.py:76-92) without the 3x of the torch-1.1 grad_outputs quirk:
 * grad_dev[n][3] = d f / d xyz (callers apply clamp mask / scaling). ws >= distr_mlp_workspace_bytes(n). */
size_t distr_mlp_workspace_bytes(int64_t n);
int distr_mlp_eval(distr_ctx* ctx, const float* latent_dev, const float* xyz_dev, int64_t n, float clamp_dist,
                   float* sdf_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* decode_sdf in SIX-PRODUCT SPLIT-bf16 arithmetic (opt-in): every f32 product of the seven wide layers is replaced by six bf16
 * products (three bf16 planes per operand, f32 accumulation, v_mfma_f32_32x32x16_bf16) -- f32-equivalent accuracy (measured max
 * |delta sdf| against distr_mlp_eval: profiles/r03_dense_b6.log) at a multiple of the f32-MFMA rate, but NOT bit-identical to the
 * exact path, which is why it is a separate entry point and nothing switches to it silently. Same arguments as distr_mlp_eval.
 * Used by the bulk SDF grid evaluation for meshing (core/evaluation/create_mesh.py: create_sdf_grid(..., arith='bf16x6')). */
int distr_mlp_eval_bf16x6(distr_ctx* ctx, const float* latent_dev, const float* xyz_dev, int64_t n, float clamp_dist,
                          float* sdf_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* decode_sdf in THREE-PRODUCT SPLIT-f16 arithmetic (opt-in, DISTR_ARITH_F16X3): two f16 planes per operand (scaled by 64), three
 * products per f32 product on v_mfma_f32_32x32x16_f16, activations kept as planes in LDS. Same accuracy class as the bf16 form at
 * half its MFMA work; limited to decoders whose weights and activations stay below 1023 in magnitude: DISTR_ERR_UNSUPPORTED when a
 * weight does not, NaN in sdf_dev for a point one of whose activations does not. Same arguments as distr_mlp_eval. */
int distr_mlp_eval_f16x3(distr_ctx* ctx, const float* latent_dev, const float* xyz_dev, int64_t n, float clamp_dist,
                         float* sdf_dev, void* ws_dev, size_t ws_bytes, void* stream);
int distr_mlp_grad(distr_ctx* ctx, const float* latent_dev, const float* xyz_dev, int64_t n, float* sdf_dev,
                   float* grad_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* Backward of decode_sdf for callers that differentiate through it (decoder_utils.py:53-74 without no_grad): g_sdf[n] is
 * the upstream gradient of the (clamped) outputs; writes g_xyz[n][3] = g * df/dxyz (may be null) and
 * g_latent[256] = sum_n g * df/dlatent (may be null). clamp >= 0: the gradient is zero where |f| > clamp (torch.clamp). */
size_t distr_mlp_backward_workspace_bytes(int64_t n);
int distr_mlp_backward(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, const float* g_sdf, float clamp,
                       float* g_xyz, float* g_latent, void* ws, size_t ws_bytes, void* stream);

/* Test aid: post-activation of hidden layer `layer` (0..7) for n points -> out_dev[n][512]. */
int distr_debug_mlp_layer(distr_ctx* ctx, const float* latent_dev, const float* xyz_dev, int64_t n, int layer,
                          float* out_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* Test aid: in-kernel phase timing of the decoder tile. ts_dev[tile][40] (int64) receives (shader-clock, 100 MHz wall
 * clock) stamp pairs of wave 0: [0] entry, [2l+1] after layer l's MFMA loop, [2l+2] after its write-back + barriers
 * (l = 0..7), [17] after lin8, [18] kernel end; sdf_dev[n] receives the decoder output. */
int distr_debug_tile_timing(distr_ctx* ctx, const float* latent_dev, const float* xyz_dev, int64_t n, float* sdf_dev,
                            long long* ts_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* Debug / measurement. distr_get_render_stats copies the counters of the forward (and last backward) that used
 * `ws_dev` device->host and synchronises `stream`.
 * Profiling: when enabled, every launch of the fused march/MLP kernel is bracketed by hipEvents on the launch
 * stream; distr_profile_read synchronises `stream` and returns the number of bracketed launches and their summed
 * kernel milliseconds since the last read (reset on read). */
int distr_get_render_stats(distr_ctx* ctx, const distr_render_cfg* cfg, const void* ws_dev, distr_render_stats* out,
                       void* stream);
int distr_profile_enable(distr_ctx* ctx, int enable);
int distr_profile_read(distr_ctx* ctx, int64_t* launches, double* total_ms, void* stream);
/* Per-launch view of the same brackets (does not reset): ms_out[i] = kernel time of the i-th bracketed march launch since the
 * last distr_profile_read, in launch order (coarse-level steps first, then the full-resolution steps of each forward). */
int distr_profile_read_list(distr_ctx* ctx, float* ms_out, int64_t cap, int64_t* n, void* stream);
/* Rays evaluated by every march launch of the forward that used `ws_dev`, in launch order (the live-ray profile of
 * ray_marching_recursive, renderer.py:528-567: what the reference learns from torch.nonzero on the host every step). */
int distr_get_live_counts(distr_ctx* ctx, const distr_render_cfg* cfg, const void* ws_dev, int32_t* out, int32_t cap, int32_t* n,
                          void* stream);
/* Test aid (DISTR_XCHG_TS=1): 64 wall-clock stamps (100 MHz) of the last cluster-tile launch on `stream`: phase boundaries of
 * cluster 0 / member 0 (csrc/distr_mlp.hpp, DISTR_XTS). */
int distr_debug_xchg_ts(distr_ctx* ctx, void* stream, int64_t* out64);

/* ---- Colour decoder: SDFRenderer_color.render_color (core/sdfrenderer/renderer_rgb.py:20-38) evaluates
 * a second DeepSDF-8x512-shaped decoder with latent = [shape code | colour code] (256 + color_size) and last_dim = 3
 * (load_decoder(color_size=...), core/utils/decoder_utils.py:16-24) at the surface points; decode_color
 * (decoder_utils.py:94-112). Weights: same flat layout as
 * distr_set_decoder with lin0 (512, 259+cs), lin4 (512, 512+cs), lin8 (3, 512); desc->latent_size = 256 + cs. */
int distr_set_color_decoder(distr_ctx* ctx, const distr_decoder_desc* desc, const float* weights, size_t n_floats);
/* latent_cat[256+cs] = cat(shape_code, color_code) (device), xyz[n][3] -> rgb[n][3] = tanh(lin8(...)). Workspace:
 * distr_mlp_workspace_bytes(n). */
int distr_color_eval(distr_ctx* ctx, const float* latent_cat, const float* xyz, int64_t n, float* rgb, void* ws, size_t ws_bytes,
                     void* stream);

/* Backward of distr_color_eval (decode_color differentiated, decoder_utils.py:94-112 with no_grad=False): g_rgb[n][3] = upstream
 * gradient of the colours; writes g_xyz[n][3] = d/d points (may be null) and g_latent_cat[256+cs] = d/d [shape code | colour code]
 * (may be null). Workspace: distr_mlp_backward_workspace_bytes(n). */
int distr_color_backward(distr_ctx* ctx, const float* latent_cat, const float* xyz, int64_t n, const float* g_rgb, float* g_xyz,
                         float* g_latent_cat, void* ws, size_t ws_bytes, void* stream);

/* ---- Image-space consumers right after the hot path, fused into a few element-wise
 * kernels. Same rules as above: caller-owned device buffers, everything enqueued on `stream`, no host sync. Scalars
 * (losses, their upstream gradients) live in device memory so that nothing has to be read back. */

/* Workspace of the loss entry points for an H x W image (per-block reduction partials). */
size_t distr_loss_workspace_bytes(int32_t H, int32_t W);

/* f3: the four single-view loss terms of compute_all_loss (core/inv_optimizer/loss_single.py:29-54):
 *   out8[0] mask_gt  = mean over (gt & ~out) of max(min_sdf - threshold, 0)         core/utils/loss_utils.py:75-87
 *   out8[1] mask_out = mean over (out & ~gt) of max(threshold - min_sdf, 0)         core/utils/loss_utils.py:89-101
 *   out8[2] depth    = mean over (out & gt & 0 < gt_depth < 1e5) of |depth - gt_depth|   loss_utils.py:105-133
 *   out8[3] normal   = mean over (out & gt & |normal| != 0) of -cos(normal, gt_normal)   loss_utils.py:140-172
 *   out8[4..7] the four pixel counts (as floats); a term over an empty set is 0.
 * depth/normal/mask/min_sdf are the outputs of distr_render_forward (H*W, row-major); gt_depth / gt_normal may be
 * null (term = 0). */
int distr_single_loss_forward(distr_ctx* ctx, int32_t H, int32_t W, const float* depth, const float* normal,
                              const uint8_t* mask, const float* min_sdf, const float* gt_depth, const float* gt_normal,
                              const uint8_t* gt_mask, float threshold, float* out8, void* ws, size_t ws_bytes, void* stream);
/* Backward of the above: g4 = upstream gradients of out8[0..3] (device); writes the gradient images that
 * distr_render_backward consumes (any of them may be null). */
int distr_single_loss_backward(distr_ctx* ctx, int32_t H, int32_t W, const float* depth, const float* normal,
                               const uint8_t* mask, const float* min_sdf, const float* gt_depth, const float* gt_normal,
                               const uint8_t* gt_mask, float threshold, const float* out8, const float* g4,
                               float* g_depth, float* g_normal, float* g_min_sdf, void* stream);

/* f2: the photometric warp loss of SDFRenderer_warp.render_warp (core/sdfrenderer/renderer_warp.py:18-101): view-1
 * surface points (camera ray * Zdepth1, with gradient) are projected into view 2, view 2's depth (Zdepth2 * calib_map)
 * and colour are sampled bilinearly (grid_sample_on_img, core/utils/loss_utils.py:9-25: align_corners, zero padding),
 * points whose projected depth disagrees by (err^2 >= thres_depth) are dropped, and the loss is the mean L1 colour
 * difference of the kept points. */
typedef struct distr_warp_cfg {
  uint32_t struct_size; /* sizeof(distr_warp_cfg) */
  int32_t H, W;
  float K[9];        /* float32(K), row-major            renderer.py:37-39  */
  float K_inv[9];    /* float32(inv(K))                  renderer.py:161-164 */
  float thres_depth; /* render_warp kwarg                renderer_warp.py:103 */
} distr_warp_cfg;
/* out3 (device) = { loss_color, #kept points, #valid view-1 pixels }; keep[P] (uint8), color1 / color2 [P][3] (the
 * detached visualisation images, zeros off the kept set) may be null. */
int distr_warp_loss_forward(distr_ctx* ctx, const distr_warp_cfg* cfg, const float* zdepth1, const uint8_t* mask1,
                            const float* zdepth2, const float* img1, const float* img2, const float* R1, const float* T1,
                            const float* R2, const float* T2, float* out3, uint8_t* keep, float* color1, float* color2,
                            void* ws, size_t ws_bytes, void* stream);
/* g_loss: upstream gradient of loss_color (device scalar). Writes g_zdepth1[P] (feeds distr_render_backward's
 * g_zdepth) and g_cam[24] = { dR1[9], dT1[3], dR2[9], dT2[3] }. */
int distr_warp_loss_backward(distr_ctx* ctx, const distr_warp_cfg* cfg, const float* zdepth1, const uint8_t* mask1,
                             const float* zdepth2, const float* img1, const float* img2, const float* R1, const float* T1,
                             const float* R2, const float* T2, const float* out3, const float* g_loss, float* g_zdepth1,
                             float* g_cam, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISTR_H_ */
