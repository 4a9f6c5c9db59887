// distr_api.hip -- host side of libdistr.so: C ABI (include/distr.h), weight-fragment packer, workspace carving,
// kernel launch sequences. Built with: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC (-c, next to the distr_inst.hip
// groups; distr.binding.build_library links them into libdistr.so).
// No device allocation / synchronisation happens inside forward/backward (caller-owned workspaces, caller's stream).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "distr_inst.hpp"      // distr_kernels.hpp + the big template kernels as extern templates (their code: distr_inst.hip, per group)
#include "distr_losses.hpp"
#include "distr_mlp_b6.hpp"

using namespace distr;

struct distr_ctx {
  int device = 0;
  std::mutex mu;             // entry points serialise per context (exchange regions, event pool, error text); launches stay async
  std::string err;
  float* dec_buf = nullptr;  // one device allocation holding every packed array
  float* dec_buf_color = nullptr;   // same for the colour decoder (distr_set_color_decoder)
  DecoderDev DC{};
  bool has_color = false;
  DecoderDev D{};
  DecoderDev16 D16{};
  DecoderB6 B6{};                   // split-bf16 weight planes of the shape decoder (distr_mlp_eval_bf16x6), own allocation
  uint32_t* dec_buf_b6 = nullptr;
  DecoderH3 H3{};                   // split-f16 weight planes (distr_mlp_h3.hpp), in the same allocation; h3_ok: the weights fit the f16 range
  bool h3_ok = false;
  bool has_decoder = false;
  bool profiling = false;
  int hybrid_threshold = 8192;  // t32: largest remainder of a march step (rays) that runs on 32-ray tiles (fine_split)
  int tail16_threshold = 4096;  // t16: ... and on 16-ray tiles (16x16x4 MFMA)
  bool save_masks = true;       // save ReLU masks in the forward so that the backward skips the decoder recompute
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  // exchange regions of the cluster tiles (distr_mlp.hpp, "Cluster tile"): granule slots in ordinary (cached) device memory +
  // assembly words in uncached memory, one region per stream that renders through this context (concurrent renders on different
  // streams must not share them)
  struct XRegion { char* buf = nullptr; uint32_t* flags = nullptr; uint32_t epoch = 0; hipStream_t stream = nullptr; bool used = false; uint64_t last_use = 0; };
  uint64_t xr_clock = 0;
  bool xchg_ts = false;         // DISTR_XCHG_TS=1: cluster 0 / member 0 writes phase stamps behind the flag words (distr_debug_xchg_ts)
  static constexpr int NXR = 8;
  XRegion xr[NXR];
  bool cluster = true;          // DISTR_CLUSTER=0: single-workgroup 16-ray tiles only
  int max_cl = 8;               // DISTR_CLUSTER=4|8: largest cluster size
  int min_cl = 2;               // smallest cluster: pair tiles (2 CUs per 16 rays) for 1008 < rays <= 2032 (DISTR_CLUSTER_MIN=4: off)
  int cluster_test_abort = 0;   // DISTR_CLUSTER_TEST_ABORT (tests): 1 = every cluster aborts at assembly (fallback path); 2 = member 0 of every cluster
                                // gives up while staging h7, behind its last slice (the others complete without noticing: Xchg::test_abort)
  bool sticky = true;           // DISTR_STICKY=0: cluster tiles never keep their rays across march steps (sticky_tile16)
  int xchg_sc1 = 0;             // DISTR_XCHG_SC1=1 (tests): write-through slice stores even for clusters on one XCD (the mixed-XCD path)
  int cluster_spread = 0;       // DISTR_CLUSTER_SPREAD=1 (tests): cluster members on consecutive workgroups = different XCDs (Xchg::spread)
  // Persistent tail launch (k_tail): the full-resolution steps from `tail_from` on run inside one launch. tail_from is a HOST decision
  // taken without synchronising: the step at which the PREVIOUS render of the same configuration first had at most tail_rays live rays
  // (k_finalize writes it to a host-mapped word, read here whenever the next render is enqueued: stale is fine, k_tail is correct for any
  // count); no hint yet (a configuration's first render): no tail launch. tail_rays sits just above the 496 rays at which a step's tiles
  // turn sticky: a step inside k_tail costs a few microseconds MORE than a launch of its own (claim + release / acquire fences against a
  // launch boundary, profiles/r06_tail_steps.md), what the tail launch saves is every launch the host would issue behind the last live
  // ray -- so it takes over right where the sticky tiles would.
  bool tail = true;             // DISTR_TAIL=0: one launch per step to the end (rounds 1-5)
  int tail_px = 0;              // DISTR_TAIL_PX: renders of at most this many pixels start the tail launch at step 0 (0: hint only)
  int tail_rays = 640;          // DISTR_TAIL_RAYS
  int tail_force = -1;          // DISTR_TAIL_FROM=n (tests): tail_from = n for every render of the recursive marchers
  int tail_absent = 0;          // DISTR_TAIL_TEST_ABSENT=n (tests): the first n workgroups of the tail launch leave at once (never resident)
  struct TailHint { int32_t key[16]; bool used = false; uint64_t last_use = 0; };
  static constexpr int NHINT = 16;
  TailHint hints[NHINT];
  int32_t* hint_host = nullptr; // NHINT host-mapped words (hipHostMalloc): -1 = nothing written yet
  int32_t* hint_dev = nullptr;
  bool hint_failed = false;
};

namespace {

int fail(distr_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

// Every entry point that launches or allocates runs with the CONTEXT's device current and restores the caller's device on exit
// (a context for a GPU other than the caller's current one must not depend on, or leave behind, a changed current device).
struct EntryGuard {
  std::lock_guard<std::mutex> lock;
  int prev = -1, dev;
  explicit EntryGuard(distr_ctx* ctx) : lock(ctx->mu), dev(ctx->device) {
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    if (prev != dev) (void)hipSetDevice(dev);
  }
  ~EntryGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(ctx, DISTR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define LAUNCH_CHECK(name)                                                                    \
  do {                                                                                        \
    hipError_t e_ = hipGetLastError();                                                        \
    if (e_ != hipSuccess) return fail(ctx, DISTR_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// A-fragment packing for v_mfma_f32_32x32x2_f32 (see distr_mlp.hpp::dense):
//   dst float4 index ((g*4 + w)*NOB + ob)*64 + lane = { W[o][8g+2s+h] : s=0..3 }, o = w*32*NOB + 32*ob + (lane&31), h = lane>>5
void pack_fragments(const float* W, int K, int O, float* dst) {
  const int NOB = O / 128, NG = K / 8;
  for (int g = 0; g < NG; ++g)
    for (int w = 0; w < 4; ++w)
      for (int ob = 0; ob < NOB; ++ob)
        for (int lane = 0; lane < 64; ++lane) {
          const int o = w * 32 * NOB + 32 * ob + (lane & 31), h = lane >> 5;
          float* d = dst + ((((size_t)g * 4 + w) * NOB + ob) * 64 + lane) * 4;
          for (int s = 0; s < 4; ++s) d[s] = W[(size_t)o * K + 8 * g + 2 * s + h];
        }
}

// A-fragments of v_mfma_f32_16x16x4_f32 (distr_mlp.hpp::dense16):
//   float4 index ((g*4 + w)*NB + ob)*64 + lane = { W[w*16*NB + 16*ob + (lane&15)][16g + 4s + (lane>>4)] : s = 0..3 }
void pack_fragments16(const float* W, int K, int O, float* dst) {
  const int NB = O / 64, NG = K / 16;
  for (int g = 0; g < NG; ++g)
    for (int w = 0; w < 4; ++w)
      for (int ob = 0; ob < NB; ++ob)
        for (int lane = 0; lane < 64; ++lane) {
          const int o = w * 16 * NB + 16 * ob + (lane & 15), kq = lane >> 4;
          float* d = dst + ((((size_t)g * 4 + w) * NB + ob) * 64 + lane) * 4;
          for (int s = 0; s < 4; ++s) d[s] = W[(size_t)o * K + 16 * g + 4 * s + kq];
        }
}

// bf16 (round to nearest even) of an f32, as its upper 16 bits
inline uint16_t to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline float from_bf16(uint16_t b) {
  const uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Split-bf16 A-fragment planes of v_mfma_f32_32x32x16_bf16 (distr_mlp_b6.hpp::dense_b6): W = w0 + w1 + w2 with w0 = bf16(W),
// w1 = bf16(W - w0), w2 = bf16(W - w0 - w1); fragment index (((kb * 4 + wave) * NOB + ob) * 3 + plane) * 64 + lane holds the 8 bf16
// W_plane[o][16 kb + 8 h + 0..7], o = wave * 32 NOB + 32 ob + (lane & 31), h = lane >> 5.
void pack_fragments_b6(const float* W, int K, int O, uint16_t* dst) {
  const int NOB = O / 128, NKB = K / 16;
  for (int kb = 0; kb < NKB; ++kb)
    for (int w = 0; w < 4; ++w)
      for (int ob = 0; ob < NOB; ++ob)
        for (int lane = 0; lane < 64; ++lane) {
          const int o = w * 32 * NOB + 32 * ob + (lane & 31), h = lane >> 5;
          for (int i = 0; i < 8; ++i) {
            const float v = W[(size_t)o * K + 16 * kb + 8 * h + i];
            const uint16_t p0 = to_bf16(v);
            const float r1 = v - from_bf16(p0);
            const uint16_t p1 = to_bf16(r1);
            const uint16_t p2 = to_bf16(r1 - from_bf16(p1));
            const uint16_t pl[3] = {p0, p1, p2};
            for (int p = 0; p < 3; ++p)
              dst[((((((size_t)kb * 4 + w) * NOB + ob) * 3 + p) * 64 + lane) * 8) + i] = pl[p];
          }
        }
}

// Split-f16 A-fragment planes of v_mfma_f32_32x32x16_f16 (distr_mlp_h3.hpp::dense_h3): SW W = w0 + w1 with w0 = f16(SW W),
// w1 = f16(SW W - w0) (round to nearest even, denormals kept); fragment index (((kb * 4 + wave) * NOB + ob) * 2 + plane) * 64 + lane
// holds the 8 f16 W_plane[o][16 kb + 8 h + 0..7]. Returns false when a scaled weight leaves the f16 range.
bool pack_fragments_h3(const float* W, int K, int O, uint16_t* dst) {
  const int NOB = O / 128, NKB = K / 16;
  bool ok = true;
  for (int kb = 0; kb < NKB; ++kb)
    for (int w = 0; w < 4; ++w)
      for (int ob = 0; ob < NOB; ++ob)
        for (int lane = 0; lane < 64; ++lane) {
          const int o = w * 32 * NOB + 32 * ob + (lane & 31), h = lane >> 5;
          for (int i = 0; i < 8; ++i) {
            const float v = H3_SW * W[(size_t)o * K + 16 * kb + 8 * h + i];
            if (!(fabsf(v) < 65504.f)) ok = false;
            const _Float16 h0 = (_Float16)v;
            const _Float16 h1 = (_Float16)(v - (float)h0);
            uint16_t pl[2];
            memcpy(&pl[0], &h0, 2); memcpy(&pl[1], &h1, 2);
            for (int p = 0; p < 2; ++p)
              dst[((((((size_t)kb * 4 + w) * NOB + ob) * 2 + p) * 64 + lane) * 8) + i] = pl[p];
          }
        }
  return ok;
}

struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(void* b) : base((char*)b) {}
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? (T*)(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

// The pyramid of a cfg, finest level first (index = LevelView index): scale[0] = 1, steps[l] = dense steps of coarse level l.
// distr_render_cfg::num_levels == 0: the default scale_list [4,2,1] ([2,1] with coarse_steps = {s, 0}) described by coarse_steps;
// else level_scale / level_steps (coarsest first, like scale_list / march_step_list of renderer.py:25-26). Returns why not, or null.
struct Pyramid { int nlev; int scale[MAX_LEVELS]; int steps[MAX_LEVELS]; };
const char* pyramid_of(const distr_render_cfg& c, Pyramid& p) {
  memset(&p, 0, sizeof(p));
  p.nlev = 1; p.scale[0] = 1;
  if (c.marcher != DISTR_MARCH_PYRAMID_RECURSIVE) return nullptr;
  if (c.num_levels == 0) {
    // coarse_steps[1] == 0: the two-level pyramid scale_list=[2,1] (coarse_steps[0] steps at half resolution); a level with no steps
    // does not exist in the reference (ray_marching_trivial concatenates an empty list)
    if (c.coarse_steps[0] < 1 || c.coarse_steps[1] < 0) return "pyramid needs >=1 step per coarse level";
    p.nlev = (c.coarse_steps[1] == 0) ? 2 : 3;
    for (int l = 1; l < p.nlev; ++l) { p.scale[l] = 1 << l; p.steps[l] = c.coarse_steps[p.nlev - 1 - l]; }
  } else {
    if (c.num_levels < 2 || c.num_levels > MAX_LEVELS) return "num_levels must be 0 (coarse_steps) or 2..4";
    p.nlev = c.num_levels;
    for (int l = 0; l < p.nlev; ++l) { p.scale[l] = c.level_scale[p.nlev - 1 - l]; p.steps[l] = l ? c.level_steps[p.nlev - 1 - l] : 0; }
    if (p.scale[0] != 1) return "the last entry of level_scale (scale_list) must be 1 (renderer.py:726)";
    for (int l = 1; l < p.nlev; ++l) {
      if (p.scale[l] <= p.scale[l - 1] || p.scale[l] % p.scale[l - 1] != 0 || p.scale[l] / p.scale[l - 1] > 8)
        return "level_scale: every scale must be an integer multiple (2..8 x) of the next finer one";
    }
  }
  for (int l = 1; l < p.nlev; ++l) {
    if (p.steps[l] < 1) return "pyramid needs >=1 step per coarse level";
    if (p.steps[l] > 15) return "at most 15 steps per coarse level";
  }
  return nullptr;
}

int check_cfg(distr_ctx* ctx, const distr_render_cfg* c) {
  if (!c) return fail(ctx, DISTR_ERR_INVALID_ARG, "cfg is null");
  if (c->struct_size != sizeof(distr_render_cfg))      // ABI handshake (include/distr.h): a caller built against another header
    return fail(ctx, DISTR_ERR_INVALID_ARG, "distr_render_cfg.struct_size is %u, this library (ABI %u) expects %zu: caller compiled against "
                "another include/distr.h, or the struct was not initialised with DISTR_INIT", c->struct_size, DISTR_ABI_VERSION, sizeof(distr_render_cfg));
  if (c->H < 1 || c->W < 1 || (int64_t)c->H * c->W >= (1 << 26)) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad image size %dx%d", c->H, c->W);
  if (c->buffer_size < 1 || c->buffer_size > MAX_BS) return fail(ctx, DISTR_ERR_UNSUPPORTED, "buffer_size %d not in [1,%d]", c->buffer_size, MAX_BS);
  if (c->marcher < 0 || c->marcher > 2) return fail(ctx, DISTR_ERR_INVALID_ARG, "unknown marcher %d", c->marcher);
  int fine = c->march_step, coarse_rows = 0;
  if (c->marcher == DISTR_MARCH_PYRAMID_RECURSIVE) {
    Pyramid py;
    if (const char* why = pyramid_of(*c, py)) return fail(ctx, DISTR_ERR_UNSUPPORTED, "%s", why);
    for (int l = 1; l < py.nlev; ++l) { fine -= py.steps[l]; coarse_rows += py.steps[l]; }
    if (c->rows != 0 && c->rows != c->H && 4 % py.scale[py.nlev - 1] != 0)
      return fail(ctx, DISTR_ERR_UNSUPPORTED, "row bands (row0 a multiple of 4) need a pyramid whose coarsest scale divides 4");
  }
  if (fine < 1 || fine > MAX_STEPS) return fail(ctx, DISTR_ERR_INVALID_ARG, "march_step %d leaves %d full-resolution steps (need 1..%d)", c->march_step, fine, MAX_STEPS);
  // fewer marched rows than selected rows: the reference's torch.topk raises ("selected index k out of range", renderer.py:314-318) unless an
  // early break happens to pad the lists (:562-567) -- refused here, where the reference fails at render time
  if (coarse_rows + fine < c->buffer_size)
    return fail(ctx, DISTR_ERR_INVALID_ARG, "buffer_size %d exceeds the %d rows a ray's march produces (march_step %d): the reference's top-k selection raises there",
                c->buffer_size, coarse_rows + fine, c->march_step);
  if (!(c->radius > 0.f) || !(c->threshold >= 0.f)) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad radius/threshold");
  if (c->arith != DISTR_ARITH_F32 && c->arith != DISTR_ARITH_BF16X6 && c->arith != DISTR_ARITH_F16X3) return fail(ctx, DISTR_ERR_INVALID_ARG, "unknown arith %d", c->arith);
  if (c->arith == DISTR_ARITH_F16X3 && ctx->has_decoder && !ctx->h3_ok)
    return fail(ctx, DISTR_ERR_UNSUPPORTED, "arith f16x3: a decoder weight times %g leaves the f16 range; use bf16x6 or f32 for this decoder", (double)H3_SW);
  if (c->rows != 0) {
    if (c->rows < 0 || c->row0 < 0 || c->row0 + c->rows > c->H) return fail(ctx, DISTR_ERR_INVALID_ARG, "row band [%d,+%d) outside the %d-row image", c->row0, c->rows, c->H);
    if ((c->row0 & 3) || ((c->rows & 3) && c->row0 + c->rows != c->H))
      return fail(ctx, DISTR_ERR_INVALID_ARG, "row band [%d,+%d) must start on a multiple of 4 and span a multiple of 4 rows (or end at H)", c->row0, c->rows);
  }
  return DISTR_OK;
}

// Exchange region of `stream` (allocated on the stream's first render: 32 MiB of granule slots + 128 KiB of uncached assembly
// words; steady state never allocates). Null (-> single-workgroup tiles) when disabled, out of regions, or the allocation fails.
distr_ctx::XRegion* xchg_region(distr_ctx* ctx, hipStream_t stream) {
  if (!ctx->cluster) return nullptr;
  {  // a launch sequence that is being captured into a graph would replay with the epochs of the capture (the barrier words
     // would already match) and must not allocate or query streams: single-workgroup tiles for captured renders
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess) (void)hipGetLastError();
    else if (cs != hipStreamCaptureStatusNone) return nullptr;
  }
  distr_ctx::XRegion* free_slot = nullptr;
  for (auto& r : ctx->xr) {
    if (r.used && r.stream == stream) { r.last_use = ++ctx->xr_clock; return &r; }
    if (!r.used && !free_slot) free_slot = &r;
  }
  if (!free_slot) {
    // all regions taken: hand the least recently used one whose stream has nothing in flight to the new stream
    distr_ctx::XRegion* lru = nullptr;
    for (auto& r : ctx->xr) {
      if (lru && r.last_use >= lru->last_use) continue;
      const hipError_t q = hipStreamQuery(r.stream);
      if (q == hipErrorNotReady) continue;            // still working: its barrier words are live
      if (q != hipSuccess) (void)hipGetLastError();   // stale handle of a destroyed stream
      lru = &r;
    }
    if (!lru) return nullptr;
    lru->stream = stream; lru->last_use = ++ctx->xr_clock;
    return lru;
  }
  constexpr size_t buf_bytes = (size_t)256 * XCLUSTER_BYTES, flag_bytes = (size_t)256 * 128 * sizeof(uint32_t) + 64 * sizeof(long long);
  int cur = -1;
  (void)hipGetDevice(&cur);
  if (cur != ctx->device) (void)hipSetDevice(ctx->device);       // the region must live on the context's device
  struct Restore { int cur, dev; ~Restore() { if (cur >= 0 && cur != dev) (void)hipSetDevice(cur); } } restore{cur, ctx->device};
  // (granule tags start at 8 = epoch 1 << 3: a zeroed slot never validates)
  if (hipMalloc((void**)&free_slot->buf, buf_bytes) != hipSuccess ||
      hipExtMallocWithFlags((void**)&free_slot->flags, flag_bytes, hipDeviceMallocUncached) != hipSuccess ||
      hipMemset(free_slot->buf, 0, buf_bytes) != hipSuccess ||
      hipMemset(free_slot->flags, 0, flag_bytes) != hipSuccess) {
    (void)hipGetLastError();
    if (free_slot->buf) { (void)hipFree(free_slot->buf); free_slot->buf = nullptr; }
    if (free_slot->flags) { (void)hipFree(free_slot->flags); free_slot->flags = nullptr; }
    ctx->cluster = false;
    return nullptr;
  }
  free_slot->used = true; free_slot->stream = stream; free_slot->epoch = 0; free_slot->last_use = ++ctx->xr_clock;
  return free_slot;
}

// Exchange parameters of the next launch on region `r`. `epochs` = barrier epochs the launch may use (1; a step launch whose
// cluster tiles may go sticky uses one per remaining march step). Epochs stay below 2^28 (an arrival word is epoch << 4 | XCC id,
// a granule tag epoch << 3 | layer): before the counter gets there, the assembly words AND the granule slots are cleared on the
// stream (no stale word or tag may equal a future one) and counting restarts at 1.
inline Xchg next_xchg(distr_ctx::XRegion* r, hipStream_t s, bool ts, int max_cl, int test_abort, int min_cl, uint32_t epochs = 1, bool sticky = false,
                      int force_sc1 = 0) {
  Xchg x{nullptr, nullptr, 0, max_cl, min_cl, test_abort, nullptr, 0, 0, 1, force_sc1, 0, 0};
  if (r && ts) x.ts = reinterpret_cast<long long*>(r->flags + 256 * 128);
  if (r) {
    if (r->epoch > 0x0fffffffu - epochs - 1) {
      (void)hipMemsetAsync(r->flags, 0, (size_t)256 * 128 * sizeof(uint32_t), s);
      (void)hipMemsetAsync(r->buf, 0, (size_t)256 * XCLUSTER_BYTES, s);
      r->epoch = 0;
    }
    x.buf = r->buf; x.flags = r->flags; x.epoch = r->epoch + 1; x.epochs = epochs; x.sticky = sticky ? 1 : 0;
    r->epoch += epochs;
  }
  return x;
}

// Slot of this render configuration in the tail-hint table (least recently used entry replaced; its word restarts at -1 = no hint).
// Null when the host-mapped words cannot be had (allocation failed once, or the stream is capturing: no allocation inside a capture).
int32_t* tail_hint_slot(distr_ctx* ctx, const distr_render_cfg& c, int nviews, bool capturing, int32_t** dev_word) {
  *dev_word = nullptr;
  if (ctx->hint_failed) return nullptr;
  if (!ctx->hint_host) {
    if (capturing) return nullptr;
    void* h = nullptr;
    void* d = nullptr;
    if (hipHostMalloc(&h, distr_ctx::NHINT * sizeof(int32_t), hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      if (h) (void)hipHostFree(h);
      ctx->hint_failed = true;
      return nullptr;
    }
    ctx->hint_host = (int32_t*)h; ctx->hint_dev = (int32_t*)d;
    for (int i = 0; i < distr_ctx::NHINT; ++i) ctx->hint_host[i] = -1;
  }
  Pyramid py;
  (void)pyramid_of(c, py);
  const int32_t key[16] = {c.H, c.W, c.row0, c.rows, c.march_step, c.marcher, c.buffer_size, nviews, py.nlev, py.scale[1], py.scale[2], py.scale[3],
                           py.steps[1], py.steps[2], py.steps[3], 0};
  static uint64_t clock = 0;
  int lru = 0;
  for (int i = 0; i < distr_ctx::NHINT; ++i) {
    auto& h = ctx->hints[i];
    if (h.used && memcmp(h.key, key, sizeof(key)) == 0) { h.last_use = ++clock; *dev_word = ctx->hint_dev + i; return ctx->hint_host + i; }
    if (!h.used) { lru = i; break; }
    if (h.last_use < ctx->hints[lru].last_use) lru = i;
  }
  auto& h = ctx->hints[lru];
  if (h.used && capturing) return nullptr;      // (re-using a slot resets its word from the host: not while a capture records device writes to it)
  memcpy(h.key, key, sizeof(key));
  h.used = true; h.last_use = ++clock;
  __atomic_store_n(ctx->hint_host + lru, -1, __ATOMIC_RELAXED);
  *dev_word = ctx->hint_dev + lru;
  return ctx->hint_host + lru;
}

inline int band_rows(const distr_render_cfg& c) { return c.rows > 0 ? c.rows : c.H; }
inline int band_row0(const distr_render_cfg& c) { return c.rows > 0 ? c.row0 : 0; }

// Lays the forward workspace out; with base==nullptr only sizes are computed.
size_t make_view(const distr_render_cfg& c, void* base, View& V, bool save_masks, int nviews = 1) {
  Carver cv(base);
  memset(&V, 0, sizeof(V));
  V.cfg = c;
  V.nviews = nviews;
  V.rows = band_rows(c); V.row0 = band_row0(c);
  V.band = (V.rows != c.H) ? 1 : 0;
  V.P = V.rows * c.W;
  V.pyramid = (c.marcher == DISTR_MARCH_PYRAMID_RECURSIVE) ? 1 : 0;
  Pyramid py;
  (void)pyramid_of(c, py);             // (check_cfg has accepted it)
  V.nlev = py.nlev;
  V.fine_steps = c.march_step;
  for (int l = 1; l < V.nlev; ++l) V.fine_steps -= py.steps[l];
  V.C = cv.take<Consts>(1);
  V.lv[0].h = V.rows; V.lv[0].w = c.W; V.lv[0].scale = 1.f; V.lv[0].off = 0.f;
  V.lv[0].y0 = V.row0; V.lv[0].full_h = c.H; V.lv[0].rdiv = 1; V.lv[0].sdiv = 1;
  for (int l = 1; l < V.nlev; ++l) {     // get_downscaled_grid_map (renderer.py:604-629): ceil(h / ratio) cells, centres scale * i + (scale - 1) / 2
    const int r = py.scale[l] / py.scale[l - 1];
    V.lv[l].rdiv = r; V.lv[l].sdiv = py.scale[l];
    V.lv[l].h = (V.lv[l - 1].h + r - 1) / r;
    V.lv[l].w = (V.lv[l - 1].w + r - 1) / r;
    V.lv[l].y0 = V.lv[l - 1].y0 / r;                 // a band's row0 is a multiple of the coarsest scale: exact
    V.lv[l].full_h = (V.lv[l - 1].full_h + r - 1) / r;
    V.lv[l].scale = (float)py.scale[l];
    V.lv[l].off = (V.lv[l].scale - 1.f) / 2.f;
  }
  for (int l = 0; l < V.nlev; ++l) {
    LevelView& L = V.lv[l];
    L.n = L.h * L.w;
    L.steps = py.steps[l];
    L.valid = cv.take<uint8_t>(L.n);
    L.list = cv.take<int32_t>(L.n);
    if (l > 0) {
      L.cinit = cv.take<float>(L.n);
      L.cm = cv.take<float>(L.n);
      L.rs = cv.take<float>((size_t)L.steps * L.n);
      L.rzb = cv.take<float>((size_t)L.steps * L.n);
      L.rza = cv.take<float>((size_t)L.steps * L.n);
    }
  }
  const size_t P = V.P, bs = c.buffer_size;
  V.live[0] = cv.take<int32_t>(P);
  V.live[1] = cv.take<int32_t>(P);
  V.m = cv.take<float>(P); V.init_now = cv.take<float>(P); V.maxbound = cv.take<float>(P);
  V.minabs = cv.take<float>(P); V.first_sdf = cv.take<float>(P);
  V.tk_s = cv.take<float>(bs * P); V.tk_zb = cv.take<float>(bs * P); V.tk_za = cv.take<float>(bs * P);
  V.tk_src = cv.take<int32_t>(bs * P);
  V.tk_slot = cv.take<int32_t>(bs * P);
  V.tclaim = cv.take<int32_t>(P / 16 + 2);
  V.tail_from = V.fine_steps;
  V.zdepth_s = cv.take<float>(P); V.depth_pre = cv.take<float>(P); V.nrm_t = cv.take<float>(3 * P);
  V.mask_s = cv.take<uint8_t>(P);
  V.nlist = cv.take<int32_t>(P); V.n_sdf = cv.take<float>(P); V.n_g = cv.take<float>(3 * P);
  V.save_masks = (save_masks && c.save_for_backward && (c.grad_depth || c.grad_mask)) ? 1 : 0;
  if (V.save_masks) {
    V.mfine = (int64_t)P * (bs + 1);
    int64_t off = 0;
    for (int l = 1; l < V.nlev; ++l) { V.moff[l] = off; off += (int64_t)V.lv[l].steps * V.lv[l].n; }
    V.morigin = V.mfine + off;
    V.mstore = cv.take<uint4>((size_t)(V.morigin + 1) * 32);
  }
  V.vstride = (int64_t)((cv.off + 255) & ~(size_t)255);    // view b of a batch: the same layout b * vstride bytes further
  return (size_t)V.vstride;
}

constexpr int BWD_CHUNK = 64;   // tiles per reduction chunk

size_t bwd_bytes(const distr_render_cfg& c) {
  const size_t P = (size_t)band_rows(c) * c.W;
  const size_t smax = P * c.buffer_size + 1;
  const size_t tiles = (smax + 31) / 32 + 256;
  const size_t nblk = (P + 255) / 256, nchunks = (tiles + BWD_CHUNK - 1) / BWD_CHUNK;
  const size_t b = ((smax * sizeof(Sample) + 255) & ~(size_t)255) + tiles * PSTRIDE * sizeof(float) + nchunks * PSTRIDE * sizeof(float) +
                   nblk * (2 * sizeof(int32_t) + 16 * sizeof(float)) + 2048;
  return (b + 255) & ~(size_t)255;
}

inline dim3 grid1(int64_t n, int per = 256) { return dim3((unsigned)((n + per - 1) / per)); }

struct MarchTimer {  // optional hipEvent bracket around the march kernel launches
  distr_ctx* ctx; hipStream_t s; bool on;
  MarchTimer(distr_ctx* c, hipStream_t st) : ctx(c), s(st), on(c->profiling) {}
  void begin() {
    if (!on) return;
    if (ctx->ev_used == ctx->ev_pool.size()) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      ctx->ev_pool.emplace_back(a, b);
    }
    (void)hipEventRecord(ctx->ev_pool[ctx->ev_used].first, s);
  }
  void end() {
    if (!on) return;
    (void)hipEventRecord(ctx->ev_pool[ctx->ev_used].second, s);
    ctx->ev_used++;
  }
};

}  // namespace

extern "C" {

const char* distr_version(void) { return "distr 0.6 (ABI 6; gfx950, f32 MFMA)"; }

uint32_t distr_abi_version(void) { return DISTR_ABI_VERSION; }

int distr_create_abi(distr_ctx** out, int hip_device, uint32_t abi_version) {
  if (!out) return DISTR_ERR_INVALID_ARG;
  *out = nullptr;
  distr_ctx* ctx = new distr_ctx();
  ctx->device = hip_device;
  if (abi_version != DISTR_ABI_VERSION) {
    ctx->err = "caller was built for distr ABI " + std::to_string(abi_version) + ", this library implements ABI " +
               std::to_string(DISTR_ABI_VERSION) + ": rebuild the caller against this include/distr.h";
    *out = ctx;
    return DISTR_ERR_INVALID_ARG;
  }
  if (const char* e = getenv("DISTR_HYBRID_THRESHOLD")) ctx->hybrid_threshold = atoi(e);
  if (const char* e = getenv("DISTR_TAIL16_THRESHOLD")) ctx->tail16_threshold = atoi(e);
  if (const char* e = getenv("DISTR_CLUSTER")) { ctx->cluster = atoi(e) != 0; if (atoi(e) >= 4) ctx->max_cl = atoi(e); }
  if (const char* e = getenv("DISTR_CLUSTER_MIN")) ctx->min_cl = atoi(e);
  if (const char* e = getenv("DISTR_XCHG_TS")) ctx->xchg_ts = atoi(e) != 0;
  if (const char* e = getenv("DISTR_CLUSTER_TEST_ABORT")) ctx->cluster_test_abort = atoi(e);     // 1: abort at assembly; 2: member 0 drops out behind its last slice
  if (const char* e = getenv("DISTR_SAVE_MASKS")) ctx->save_masks = atoi(e) != 0;
  if (const char* e = getenv("DISTR_STICKY")) ctx->sticky = atoi(e) != 0;
  if (const char* e = getenv("DISTR_XCHG_SC1")) ctx->xchg_sc1 = atoi(e) != 0;
  if (const char* e = getenv("DISTR_CLUSTER_SPREAD")) ctx->cluster_spread = atoi(e) != 0;
  if (const char* e = getenv("DISTR_TAIL")) ctx->tail = atoi(e) != 0;
  if (const char* e = getenv("DISTR_TAIL_PX")) ctx->tail_px = atoi(e);
  if (const char* e = getenv("DISTR_TAIL_RAYS")) ctx->tail_rays = atoi(e);
  if (const char* e = getenv("DISTR_TAIL_FROM")) ctx->tail_force = atoi(e);
  if (const char* e = getenv("DISTR_TAIL_TEST_ABSENT")) ctx->tail_absent = atoi(e);
  {
    // invariants of the tile-size split (fine_split and the host-side grid sizes rely on them): multiples of 64,
    // 64 <= t16 <= t32, and t16 + t32 below one full round (16384 rays) so that "remainder" ranges never reach a round
    const int t32 = ctx->hybrid_threshold, t16 = std::min(ctx->tail16_threshold, ctx->hybrid_threshold);
    if (t32 < 64 || t16 < 64 || (t32 & 63) || (ctx->tail16_threshold & 63) || t32 + t16 >= 16384) {
      ctx->err = "DISTR_HYBRID_THRESHOLD / DISTR_TAIL16_THRESHOLD must be multiples of 64 with 64 <= tail16, 64 <= hybrid and "
                 "min(tail16, hybrid) + hybrid < 16384 (got hybrid " + std::to_string(ctx->hybrid_threshold) + ", tail16 " +
                 std::to_string(ctx->tail16_threshold) + ")";
      *out = ctx;
      return DISTR_ERR_INVALID_ARG;
    }
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || hip_device < 0 || hip_device >= n) {
    // keep the context so the caller can read the message
    ctx->err = std::string("no usable HIP device ") + std::to_string(hip_device) + " (" + (e == hipSuccess ? "count=" + std::to_string(n) : hipGetErrorString(e)) + ")";
    *out = ctx;
    return DISTR_ERR_HIP;
  }
  *out = ctx;
  return DISTR_OK;
}

void distr_destroy(distr_ctx* ctx) {
  if (!ctx) return;
  if (ctx->dec_buf) { (void)hipSetDevice(ctx->device); (void)hipFree(ctx->dec_buf); }
  if (ctx->dec_buf_color) { (void)hipSetDevice(ctx->device); (void)hipFree(ctx->dec_buf_color); }
  if (ctx->dec_buf_b6) { (void)hipSetDevice(ctx->device); (void)hipFree(ctx->dec_buf_b6); }
  for (auto& r : ctx->xr) { if (r.buf) (void)hipFree(r.buf); if (r.flags) (void)hipFree(r.flags); }
  if (ctx->hint_host) (void)hipHostFree(ctx->hint_host);
  for (auto& p : ctx->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  delete ctx;
}

const char* distr_last_error(const distr_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// Packs one DeepSDF-8x512-shaped decoder for the tile kernels. nlat = latent length (the latent columns of lin0 / lin4 are
// folded into per-call constants, so the tile itself never sees them); nout = rows of lin8 (1: SDF, 3: colour).
static int build_decoder(distr_ctx* ctx, int nlat, int nout, const float* w, size_t n_floats, float** dev_buf, DecoderDev& D,
                         DecoderDev16* D16, DecoderB6* B6 = nullptr, uint32_t** dev_buf_b6 = nullptr, DecoderH3* H3 = nullptr,
                         bool* h3_ok = nullptr) {
  const int in0 = nlat + 3, in4 = 256 + nlat;
  const int OUT[9] = {512, 512, 512, 253, 512, 512, 512, 512, nout};
  const int IN[9] = {in0, 512, 512, 512, in4, 512, 512, 512, 512};
  size_t need = 0;
  for (int l = 0; l < 9; ++l) need += (size_t)OUT[l] * IN[l] + OUT[l];
  if (n_floats != need) return fail(ctx, DISTR_ERR_INVALID_ARG, "weight buffer has %zu floats, expected %zu", n_floats, need);
  const float* W[9]; const float* b[9];
  const float* p = w;
  for (int l = 0; l < 9; ++l) { W[l] = p; p += (size_t)OUT[l] * IN[l]; b[l] = p; p += OUT[l]; }

  // padded dense matrices [O][K] of the eight MFMA layers
  const int Kp[8] = {8, 512, 512, 512, 256, 512, 512, 512};
  const int Op[8] = {512, 512, 512, 256, 512, 512, 512, 512};
  std::vector<std::vector<float>> Wp(8);
  for (int l = 0; l < 8; ++l) Wp[l].assign((size_t)Op[l] * Kp[l], 0.f);
  for (int o = 0; o < 512; ++o) for (int k = 0; k < 3; ++k) Wp[0][(size_t)o * 8 + k] = W[0][(size_t)o * in0 + nlat + k];
  for (int l : {1, 2, 5, 6, 7}) memcpy(Wp[l].data(), W[l], sizeof(float) * 512 * 512);
  for (int o = 0; o < 253; ++o) memcpy(&Wp[3][(size_t)o * 512], &W[3][(size_t)o * 512], sizeof(float) * 512);
  for (int o = 0; o < 512; ++o) {
    for (int k = 0; k < 253; ++k) Wp[4][(size_t)o * 256 + k] = W[4][(size_t)o * in4 + k];
    for (int k = 0; k < 3; ++k) Wp[4][(size_t)o * 256 + 253 + k] = W[4][(size_t)o * in4 + 253 + nlat + k];
  }

  std::vector<float> host;
  auto reserve = [&](size_t n) { size_t off = (host.size() + 63) & ~(size_t)63; host.resize(off + n, 0.f); return off; };
  size_t offWf[8], offWb[8] = {0}, offB[8] = {0};
  for (int l = 0; l < 8; ++l) {
    offWf[l] = reserve(Wp[l].size());
    pack_fragments(Wp[l].data(), Kp[l], Op[l], host.data() + offWf[l]);
  }
  for (int l = 1; l < 8; ++l) {
    std::vector<float> Wt((size_t)Kp[l] * Op[l]);
    for (int o = 0; o < Op[l]; ++o) for (int k = 0; k < Kp[l]; ++k) Wt[(size_t)k * Op[l] + o] = Wp[l][(size_t)o * Kp[l] + k];
    offWb[l] = reserve(Wt.size());
    pack_fragments(Wt.data(), /*K'=*/Op[l], /*O'=*/Kp[l], host.data() + offWb[l]);
  }
  size_t offW16[8];
  for (int l = 0; l < 8; ++l) {
    const int K16 = (l == 0) ? 16 : Kp[l];
    std::vector<float> W16((size_t)Op[l] * K16, 0.f);
    for (int o = 0; o < Op[l]; ++o) for (int k = 0; k < Kp[l]; ++k) W16[(size_t)o * K16 + k] = Wp[l][(size_t)o * Kp[l] + k];
    offW16[l] = reserve(W16.size());
    pack_fragments16(W16.data(), K16, Op[l], host.data() + offW16[l]);
  }
  for (int l : {1, 2, 3, 5, 6, 7}) {
    offB[l] = reserve(Op[l]);
    memcpy(host.data() + offB[l], b[l], sizeof(float) * OUT[l]);
  }
  const size_t o_W0lat_t = reserve((size_t)nlat * HID), o_W4lat_t = reserve((size_t)nlat * HID);
  const size_t o_W0lat = reserve((size_t)HID * nlat), o_W4lat = reserve((size_t)HID * nlat);
  for (int o = 0; o < HID; ++o) for (int k = 0; k < nlat; ++k) {
    const float v0 = W[0][(size_t)o * in0 + k], v4 = W[4][(size_t)o * in4 + 253 + k];
    host[o_W0lat_t + (size_t)k * HID + o] = v0; host[o_W0lat + (size_t)o * nlat + k] = v0;
    host[o_W4lat_t + (size_t)k * HID + o] = v4; host[o_W4lat + (size_t)o * nlat + k] = v4;
  }
  const size_t o_b0 = reserve(HID), o_b4 = reserve(HID), o_w8 = reserve((size_t)nout * HID), o_W0x = reserve(3 * HID);
  memcpy(host.data() + o_b0, b[0], sizeof(float) * HID);
  memcpy(host.data() + o_b4, b[4], sizeof(float) * HID);
  memcpy(host.data() + o_w8, W[8], sizeof(float) * nout * HID);
  for (int o = 0; o < HID; ++o) for (int k = 0; k < 3; ++k) host[o_W0x + (size_t)k * HID + o] = W[0][(size_t)o * in0 + nlat + k];

  if (*dev_buf) { HIP_TRY(hipFree(*dev_buf)); *dev_buf = nullptr; }       // (the entry point's guard made ctx->device current)
  HIP_TRY(hipMalloc((void**)dev_buf, host.size() * sizeof(float)));
  HIP_TRY(hipMemcpy(*dev_buf, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  const float* d = *dev_buf;
  for (int l = 0; l < 8; ++l) { D.Wf[l] = d + offWf[l]; D.Wb[l] = l ? d + offWb[l] : nullptr; D.bias[l] = (l == 0 || l == 4) ? nullptr : d + offB[l]; }
  D.W0lat_t = d + o_W0lat_t; D.W4lat_t = d + o_W4lat_t; D.W0lat = d + o_W0lat; D.W4lat = d + o_W4lat;
  D.b0 = d + o_b0; D.b4 = d + o_b4; D.w8 = d + o_w8; D.W0x = d + o_W0x;
  D.b8 = b[8][0];
  D.b8x[0] = nout > 1 ? b[8][1] : 0.f; D.b8x[1] = nout > 2 ? b[8][2] : 0.f;
  D.nlat = nlat;
  if (D16) for (int l = 0; l < 8; ++l) D16->Wf[l] = d + offW16[l];
#ifdef DISTR_DIAG
  // DIAGNOSTICS BUILDS ONLY (-DDISTR_DIAG; values are wrong): every 512 x 512 layer of the 16-ray / cluster tiles reads lin1's fragments, so
  // that their weight stream (2 MB instead of 6.3 MB) stays in one XCD's 4 MiB L2 -- separates "waiting for weights" from "issuing
  // instructions" in the tail. Not in the product library: an environment variable must never be able to change a render's values.
  if (D16 && getenv("DISTR_DEBUG_ALIAS_WEIGHTS")) {
    fprintf(stderr, "distr: DISTR_DEBUG_ALIAS_WEIGHTS is set -- the 16-ray / cluster tiles compute with the WRONG weights (timing diagnostics only)\n");
    for (int l : {2, 5, 6, 7}) D16->Wf[l] = D16->Wf[1];
  }
#endif
  if (B6) {   // split-bf16 planes of lin1..lin7 for the opt-in arithmetic mode (distr_mlp_eval_bf16x6): 9.4 MB, own allocation
    std::vector<uint16_t> hb;
    size_t offb[8] = {0}, offbt[8] = {0};
    for (int l = 1; l < 8; ++l) {
      offb[l] = (hb.size() + 127) & ~(size_t)127;
      hb.resize(offb[l] + Wp[l].size() * 3, 0);
      pack_fragments_b6(Wp[l].data(), Kp[l], Op[l], hb.data() + offb[l]);
    }
    for (int l = 1; l < 8; ++l) {   // transposed matrices (backward dX chain): K' = Op[l], O' = Kp[l]
      std::vector<float> Wt((size_t)Kp[l] * Op[l]);
      for (int o = 0; o < Op[l]; ++o) for (int k = 0; k < Kp[l]; ++k) Wt[(size_t)k * Op[l] + o] = Wp[l][(size_t)o * Kp[l] + k];
      offbt[l] = (hb.size() + 127) & ~(size_t)127;
      hb.resize(offbt[l] + Wt.size() * 3, 0);
      pack_fragments_b6(Wt.data(), /*K'=*/Op[l], /*O'=*/Kp[l], hb.data() + offbt[l]);
    }
    size_t offh[8] = {0}, offht[8] = {0};
    if (H3) {   // split-f16 planes of lin1..lin7 and of their transposes: 2 x 6.3 MB
      *h3_ok = true;
      for (int l = 1; l < 8; ++l) {
        offh[l] = (hb.size() + 127) & ~(size_t)127;
        hb.resize(offh[l] + Wp[l].size() * 2, 0);
        if (!pack_fragments_h3(Wp[l].data(), Kp[l], Op[l], hb.data() + offh[l])) *h3_ok = false;
      }
      for (int l = 1; l < 8; ++l) {
        std::vector<float> Wt((size_t)Kp[l] * Op[l]);
        for (int o = 0; o < Op[l]; ++o) for (int k = 0; k < Kp[l]; ++k) Wt[(size_t)k * Op[l] + o] = Wp[l][(size_t)o * Kp[l] + k];
        offht[l] = (hb.size() + 127) & ~(size_t)127;
        hb.resize(offht[l] + Wt.size() * 2, 0);
        (void)pack_fragments_h3(Wt.data(), /*K'=*/Op[l], /*O'=*/Kp[l], hb.data() + offht[l]);
      }
    }
    if (*dev_buf_b6) { HIP_TRY(hipFree(*dev_buf_b6)); *dev_buf_b6 = nullptr; }
    HIP_TRY(hipMalloc((void**)dev_buf_b6, hb.size() * sizeof(uint16_t)));
    HIP_TRY(hipMemcpy(*dev_buf_b6, hb.data(), hb.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    B6->Wp[0] = nullptr; B6->Wb[0] = nullptr;
    for (int l = 1; l < 8; ++l) {
      B6->Wp[l] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(*dev_buf_b6) + offb[l]);
      B6->Wb[l] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(*dev_buf_b6) + offbt[l]);
    }
    if (H3) {
      H3->Wp[0] = nullptr; H3->Wb[0] = nullptr;
      for (int l = 1; l < 8; ++l) {
        H3->Wp[l] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(*dev_buf_b6) + offh[l]);
        H3->Wb[l] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(*dev_buf_b6) + offht[l]);
      }
    }
  }
  return DISTR_OK;
}

int distr_set_decoder(distr_ctx* ctx, const distr_decoder_desc* desc, const float* w, size_t n_floats) {
  if (!ctx || !desc || !w) return fail(ctx, DISTR_ERR_INVALID_ARG, "null argument");
  EntryGuard guard_(ctx);
  if (desc->struct_size != sizeof(distr_decoder_desc)) return fail(ctx, DISTR_ERR_INVALID_ARG, "distr_decoder_desc.struct_size is %u, expected %zu", desc->struct_size, sizeof(distr_decoder_desc));
  if (desc->latent_size != LAT || desc->hidden != HID || desc->num_linear != 9 || desc->latent_in != 4)
    return fail(ctx, DISTR_ERR_UNSUPPORTED, "decoder (latent %d, hidden %d, %d linears, latent_in %d) unsupported: kernels are "
                "specialised for DeepSDF 8x512 (latent 256, latent_in=[4])", desc->latent_size, desc->hidden, desc->num_linear, desc->latent_in);
  int rc = build_decoder(ctx, LAT, 1, w, n_floats, &ctx->dec_buf, ctx->D, &ctx->D16, &ctx->B6, &ctx->dec_buf_b6, &ctx->H3, &ctx->h3_ok);
  if (rc) return rc;
  ctx->has_decoder = true;
  return DISTR_OK;
}

int distr_set_color_decoder(distr_ctx* ctx, const distr_decoder_desc* desc, const float* w, size_t n_floats) {
  if (!ctx || !desc || !w) return fail(ctx, DISTR_ERR_INVALID_ARG, "null argument");
  EntryGuard guard_(ctx);
  if (desc->struct_size != sizeof(distr_decoder_desc)) return fail(ctx, DISTR_ERR_INVALID_ARG, "distr_decoder_desc.struct_size is %u, expected %zu", desc->struct_size, sizeof(distr_decoder_desc));
  if (desc->latent_size <= LAT || desc->latent_size > 4096 || desc->hidden != HID || desc->num_linear != 9 || desc->latent_in != 4)
    return fail(ctx, DISTR_ERR_UNSUPPORTED, "colour decoder (latent %d, hidden %d, %d linears, latent_in %d) unsupported: expected the "
                "DeepSDF 8x512 shape with latent = 256 + color_size and last_dim = 3", desc->latent_size, desc->hidden, desc->num_linear, desc->latent_in);
  int rc = build_decoder(ctx, desc->latent_size, 3, w, n_floats, &ctx->dec_buf_color, ctx->DC, nullptr);
  if (rc) return rc;
  ctx->has_color = true;
  return DISTR_OK;
}

int distr_color_eval(distr_ctx* ctx, const float* latent_cat, const float* xyz, int64_t n, float* rgb, void* ws, size_t ws_bytes,
                     void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_color) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_color_decoder has not been called");
  if (n < 0 || !latent_cat || (n > 0 && (!xyz || !rgb)) || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  if (ws_bytes < distr_mlp_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "colour workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* c0c4 = (float*)ws;
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->DC, latent_cat);
  LAUNCH_CHECK("k_latent_consts");
  if (n > 0) {
    hipLaunchKernelGGL(k_color, dim3((unsigned)((n + 63) / 64)), dim3(NTHREADS), 0, s, xyz, n, (const float*)c0c4, rgb, ctx->DC);
    LAUNCH_CHECK("k_color");
  }
  return DISTR_OK;
}


int distr_color_backward(distr_ctx* ctx, const float* latent_cat, const float* xyz, int64_t n, const float* g_rgb, float* g_xyz,
                         float* g_latent_cat, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_color) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_color_decoder has not been called");
  if (n < 0 || !latent_cat || (n > 0 && (!xyz || !g_rgb)) || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  if (ws_bytes < distr_mlp_backward_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "colour backward workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (g_latent_cat) HIP_TRY(hipMemsetAsync(g_latent_cat, 0, (size_t)ctx->DC.nlat * sizeof(float), s));
    return DISTR_OK;
  }
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* partial = (float*)(((uintptr_t)(c0c4 + 2 * HID) + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->DC, latent_cat);
  LAUNCH_CHECK("k_latent_consts");
  const unsigned tiles = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(k_color_bwd, dim3(tiles), dim3(NTHREADS), 0, s, xyz, n, (const float*)c0c4, g_rgb, g_xyz, partial, ctx->DC);
  LAUNCH_CHECK("k_color_bwd");
  if (g_latent_cat) {
    hipLaunchKernelGGL(k_points_latent_grad, dim3(1), dim3(256), 0, s, (const float*)partial, (int)tiles, ctx->DC, g_latent_cat);
    LAUNCH_CHECK("k_points_latent_grad");
  }
  return DISTR_OK;
}

int distr_workspace_bytes(distr_ctx* ctx, const distr_render_cfg* cfg, size_t* fwd, size_t* bwd) {
  int rc = check_cfg(ctx, cfg);
  if (rc) return rc;
  View V;
  if (fwd) *fwd = make_view(*cfg, nullptr, V, ctx->save_masks);
  if (bwd) *bwd = bwd_bytes(*cfg);
  return DISTR_OK;
}

namespace {

inline int32_t cfg_view_flags(const distr_render_cfg& c) {
  return (c.grad_depth ? VF_GRAD_DEPTH : 0) | (c.grad_mask ? VF_GRAD_MASK : 0) | (c.grad_camera ? VF_GRAD_CAMERA : 0);
}

// per-view gradient switches of a batch: null -> every view uses cfg's; a view may only switch OFF what cfg has on (the
// workspace layout, e.g. whether ReLU masks are saved, follows cfg)
int make_view_flags(distr_ctx* ctx, const distr_render_cfg& c, int nviews, const int32_t* view_flags, ViewFlags& vf) {
  memset(&vf, 0, sizeof(vf));
  const int32_t all = cfg_view_flags(c);
  for (int b = 0; b < nviews; ++b) {
    const int32_t f = view_flags ? view_flags[b] : all;
    if (f & ~all) return fail(ctx, DISTR_ERR_INVALID_ARG, "view_flags[%d] = %d enables a gradient that cfg (flags %d) has off", b, f, all);
    vf.f[b] = (uint8_t)f;
  }
  return DISTR_OK;
}

inline int64_t pad_to(int64_t v, int64_t g) { return (v + g - 1) / g * g; }

int render_forward_impl(distr_ctx* ctx, const distr_render_cfg* cfg, int nviews, const int32_t* view_flags, const float* latent,
                        int64_t lat_stride, const float* R, const float* T, float* zdepth, uint8_t* mask, float* min_sdf,
                        float* depth, float* normal, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  int rc = check_cfg(ctx, cfg);
  if (rc) return rc;
  if (nviews < 1 || nviews > DISTR_MAX_VIEWS) return fail(ctx, DISTR_ERR_INVALID_ARG, "nviews %d not in [1, %d]", nviews, DISTR_MAX_VIEWS);
  if (lat_stride != 0 && lat_stride < LAT) return fail(ctx, DISTR_ERR_INVALID_ARG, "latent_stride must be 0 (shared shape code) or >= %d", LAT);
  if (!latent || !R || !T || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  View V;
  const size_t single = make_view(*cfg, ws, V, ctx->save_masks, nviews);
  if (ws_bytes < single * nviews) return fail(ctx, DISTR_ERR_WORKSPACE, "forward workspace too small: %zu < %zu", ws_bytes, single * nviews);
  ViewFlags vf;
  rc = make_view_flags(ctx, *cfg, nviews, view_flags, vf);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const DecoderDev& D = ctx->D;
  const int P = V.P;
  const unsigned NV = (unsigned)nviews;
  auto gridv = [&](int64_t n) { return dim3((unsigned)((n + 255) / 256), NV); };   // (blocks of 256, view)

  // persistent tail launch: from which full-resolution step on (see distr_ctx::tail)
  int32_t* hint_dev = nullptr;
  if (cfg->marcher != DISTR_MARCH_TRIVIAL && cfg->arith == DISTR_ARITH_F32 && ctx->tail && !cfg->concurrent && ctx->tail16_threshold > 0) {
    bool capturing = false;
    {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &cs) != hipSuccess) (void)hipGetLastError();
      else capturing = cs != hipStreamCaptureStatusNone;
    }
    int32_t* hint = tail_hint_slot(ctx, *cfg, nviews, capturing, &hint_dev);
    if (ctx->tail_force >= 0) V.tail_from = std::min(ctx->tail_force, V.fine_steps);
    else if ((int64_t)nviews * P <= ctx->tail_px) V.tail_from = 0;
    else if (hint) {
      const int32_t h = __atomic_load_n(hint, __ATOMIC_RELAXED);
      if (h >= 0 && h + 2 <= V.fine_steps) V.tail_from = h;      // (nothing to gain from a tail of one step)
    }
  }

  hipLaunchKernelGGL(k_prep, dim3(4, NV), dim3(256), 0, s, V, D, latent, lat_stride, R, T, vf);
  LAUNCH_CHECK("k_prep");
  for (int l = 0; l < V.nlev; ++l) {
    hipLaunchKernelGGL(k_setup_level, gridv(V.lv[l].n), dim3(256), 0, s, V, l);
    LAUNCH_CHECK("k_setup_level");
    if (V.band) {
      hipLaunchKernelGGL(k_maxinit_full, gridv((int64_t)V.lv[l].full_h * V.lv[l].w), dim3(256), 0, s, V, l);
      LAUNCH_CHECK("k_maxinit_full");
    }
  }
  MarchTimer timer(ctx, s);
  MarchArgs A;
  memset(&A, 0, sizeof(A));
  A.V = V;
  A.B6 = ctx->B6;
  A.H3 = ctx->H3;
  const bool b6 = cfg->arith != DISTR_ARITH_F32;                   // split-bf16 / split-f16 tiles: 64- and 32-ray roles only, no cluster tiles
  const bool h3 = cfg->arith == DISTR_ARITH_F16X3;
  const bool recursive = cfg->marcher != DISTR_MARCH_TRIVIAL;     // live-ray lists + tile-size split (fine_split)
  const int t32 = ctx->hybrid_threshold, t16 = b6 ? 0 : std::min(ctx->tail16_threshold, ctx->hybrid_threshold);
  A.t16 = recursive ? t16 : 0; A.t32 = recursive ? t32 : 0; A.which = 64;
  // f(origin) of every view (sample point of padded rows) is evaluated by nviews extra workgroups of ONE march launch: for the
  // recursive marchers they ride on the 16-ray role of the last step (free: a tail step); 'trivial' puts them on its first launch
  distr_ctx::XRegion* xr = (recursive && !b6) ? xchg_region(ctx, s) : nullptr;
  auto up8 = [](int64_t v) { return (int32_t)((v + 7) / 8 * 8); };
  for (int l = V.nlev - 1; l >= 1; --l) {
    hipLaunchKernelGGL(k_coarse_init, gridv(V.lv[l].n), dim3(256), 0, s, V, l);
    LAUNCH_CHECK("k_coarse_init");
    for (int st = 0; st < V.lv[l].steps; ++st) {
      A.lvl = l; A.step = st; A.origin_tile = 0;
      // tile size of a coarse level from the (host-known) pixel count of all views: a level that fits one round of 16- / 32-ray
      // tiles runs on those (small images: 111 / 212 us per step instead of 380 us)
      const int64_t ln = V.lv[l].n;
      const bool c16 = !b6 && nviews * pad_to(ln, 16) <= t16;
      const int crb = (nviews * pad_to(ln, 32) <= t32) ? 1 : 2;
      const int ctile = c16 ? 16 : 32 * crb;
      unsigned tiles = NV * (unsigned)((ln + ctile - 1) / ctile);
      A.xc = next_xchg(c16 ? xr : nullptr, s, ctx->xchg_ts, ctx->max_cl, ctx->cluster_test_abort, ctx->min_cl, 1, false, ctx->xchg_sc1);
      A.xc.spread = ctx->cluster_spread;
      if (c16 && xr) tiles = std::max(tiles, 256u);      // cluster tiles: up to 8 workgroups per 16 rays
      timer.begin();
      if (c16) {
        if (V.save_masks) hipLaunchKernelGGL((k_march16<MODE_COARSE, true>), dim3(tiles), dim3(NTHREADS), 0, s, A, D, ctx->D16);
        else hipLaunchKernelGGL((k_march16<MODE_COARSE, false>), dim3(tiles), dim3(NTHREADS), 0, s, A, D, ctx->D16);
      } else if (h3) {
        if (V.save_masks) {
          if (crb == 1) hipLaunchKernelGGL((k_march<MODE_COARSE, 1, true, 2>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
          else hipLaunchKernelGGL((k_march<MODE_COARSE, 2, true, 2>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        } else {
          if (crb == 1) hipLaunchKernelGGL((k_march<MODE_COARSE, 1, false, 2>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
          else hipLaunchKernelGGL((k_march<MODE_COARSE, 2, false, 2>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        }
      } else if (b6) {
        if (V.save_masks) {
          if (crb == 1) hipLaunchKernelGGL((k_march<MODE_COARSE, 1, true, 1>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
          else hipLaunchKernelGGL((k_march<MODE_COARSE, 2, true, 1>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        } else {
          if (crb == 1) hipLaunchKernelGGL((k_march<MODE_COARSE, 1, false, 1>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
          else hipLaunchKernelGGL((k_march<MODE_COARSE, 2, false, 1>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        }
      } else if (V.save_masks) {
        if (crb == 1) hipLaunchKernelGGL((k_march<MODE_COARSE, 1, true>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        else hipLaunchKernelGGL((k_march<MODE_COARSE, 2, true>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
      } else {
        if (crb == 1) hipLaunchKernelGGL((k_march<MODE_COARSE, 1, false>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        else hipLaunchKernelGGL((k_march<MODE_COARSE, 2, false>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
      }
      timer.end();
      LAUNCH_CHECK("k_march<coarse>");
    }
  }
  hipLaunchKernelGGL(k_fine_init, gridv(P), dim3(256), 0, s, V);
  LAUNCH_CHECK("k_fine_init");
  // upper bounds of a step's live rays in the virtual concatenation of the views (every view padded to 16 / 64 rays)
  const int64_t N64 = (int64_t)nviews * pad_to(P, 64);
  for (int st = 0; st < V.fine_steps; ++st) {
    A.lvl = 0; A.step = st;
    timer.begin();
    if (st == V.tail_from) {
      // every remaining step inside this launch (k_tail): 256 workgroups, one per compute unit
      A.origin_tile = 1;
      A.xc = next_xchg(xr, s, ctx->xchg_ts, ctx->max_cl, ctx->cluster_test_abort, ctx->min_cl, (uint32_t)(V.fine_steps - st), ctx->sticky, ctx->xchg_sc1);
      A.xc.spread = ctx->cluster_spread;
      A.xc.t_go = TAIL_T_GO;
      A.tail_absent = ctx->tail_absent;
      if (V.save_masks) hipLaunchKernelGGL((k_tail<true>), dim3(256), dim3(NTHREADS), 0, s, A, D, ctx->D16);
      else hipLaunchKernelGGL((k_tail<false>), dim3(256), dim3(NTHREADS), 0, s, A, D, ctx->D16);
      timer.end();
      LAUNCH_CHECK("k_tail");
      break;
    }
    if (!recursive) {
      // 'trivial': every in-sphere ray, every step, on 64-ray tiles
      A.origin_tile = (st == 0) ? 1 : 0;
      const unsigned tiles = NV * (unsigned)((P + 63) / 64) + (A.origin_tile ? NV : 0u);
      if (h3) {
        if (V.save_masks) hipLaunchKernelGGL((k_march<MODE_FINE, 2, true, 2>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        else hipLaunchKernelGGL((k_march<MODE_FINE, 2, false, 2>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
      } else if (b6) {
        if (V.save_masks) hipLaunchKernelGGL((k_march<MODE_FINE, 2, true, 1>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
        else hipLaunchKernelGGL((k_march<MODE_FINE, 2, false, 1>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
      } else if (V.save_masks) hipLaunchKernelGGL((k_march<MODE_FINE, 2, true>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
      else hipLaunchKernelGGL((k_march<MODE_FINE, 2, false>), dim3(tiles), dim3(NTHREADS), 0, s, A, D);
      timer.end();
      LAUNCH_CHECK("k_march<fine>");
      continue;
    }
    // one launch per step: the three tile sizes are roles of the same grid (k_step, fine_split). Roles that are provably empty
    // from the pixel count alone get no workgroups: with N64 <= bound the remainder rules never reach the 64-ray role (bound <
    // one round, distr_create), with N64 <= t16 the 32-ray role stays empty too.
    const int64_t bound = (t16 < t32) ? (int64_t)t32 + t16 : t32;
    const bool skip64 = N64 <= bound;
    const bool skip32 = t32 <= t16 || N64 <= t16;
    StepGrid G;
    G.n64 = skip64 ? 0 : std::min(up8(N64 / 64), 256);            // persistent: at most one 64-ray workgroup per CU
    A.origin_tile = (st == V.fine_steps - 1) ? 1 : 0;
    if (b6) {
      // split-bf16: no 16-ray role; the views' origin tiles (last step) close the 32-ray role's grid
      G.n32 = up8(std::min<int64_t>(N64, t32) / 32 + (A.origin_tile ? nviews : 0));
      G.n16 = 0;
      A.xc = next_xchg(nullptr, s, false, ctx->max_cl, 0, ctx->min_cl);
      const unsigned grid = (unsigned)(G.n64 + G.n32);
      if (h3) {
        if (V.save_masks) hipLaunchKernelGGL((k_step<true, 2>), dim3(grid), dim3(NTHREADS), 0, s, A, D, ctx->D16, G);
        else hipLaunchKernelGGL((k_step<false, 2>), dim3(grid), dim3(NTHREADS), 0, s, A, D, ctx->D16, G);
      } else if (V.save_masks) hipLaunchKernelGGL((k_step<true, 1>), dim3(grid), dim3(NTHREADS), 0, s, A, D, ctx->D16, G);
      else hipLaunchKernelGGL((k_step<false, 1>), dim3(grid), dim3(NTHREADS), 0, s, A, D, ctx->D16, G);
    } else {
      G.n32 = skip32 ? 0 : up8(std::min<int64_t>(N64, t32) / 32);
      A.xc = next_xchg(xr, s, ctx->xchg_ts, ctx->max_cl, ctx->cluster_test_abort, ctx->min_cl, (uint32_t)(V.fine_steps - st), ctx->sticky && !cfg->concurrent, ctx->xchg_sc1);
      A.xc.spread = ctx->cluster_spread;
      unsigned n16 = (unsigned)(std::min<int64_t>(N64, t16) / 16) + (A.origin_tile ? NV : 0u);
      if (xr) n16 = std::max(n16, 256u);                             // cluster tiles: 8 / 4 / 2 workgroups per tile of at most 32 / 64 / 128
      G.n16 = up8(n16);
      const unsigned grid = (unsigned)(G.n64 + G.n32 + G.n16);
      if (V.save_masks) hipLaunchKernelGGL((k_step<true>), dim3(grid), dim3(NTHREADS), 0, s, A, D, ctx->D16, G);
      else hipLaunchKernelGGL((k_step<false>), dim3(grid), dim3(NTHREADS), 0, s, A, D, ctx->D16, G);
    }
    timer.end();
    LAUNCH_CHECK("k_step");
  }
  hipLaunchKernelGGL(k_finalize, gridv(P), dim3(256), 0, s, V, zdepth, mask, min_sdf, depth, hint_dev, (int32_t)ctx->tail_rays);
  LAUNCH_CHECK("k_finalize");
  if (cfg->want_normal) {
    if (cfg->use_depth2normal) {
      hipLaunchKernelGGL(k_depth2normal, gridv(P), dim3(256), 0, s, V, depth, normal);
      LAUNCH_CHECK("k_depth2normal");
    } else {
      if (normal) HIP_TRY(hipMemsetAsync(normal, 0, (size_t)nviews * P * 3 * sizeof(float), s));
      BwdArgs B;
      memset(&B, 0, sizeof(B));
      B.V = V; B.zdepth = V.zdepth_s; B.zstride = V.vstride;
      hipLaunchKernelGGL((k_bwd<BWD_POINTGRAD, 2>), dim3(NV * (unsigned)((P + 63) / 64)), dim3(NTHREADS), 0, s, B, D);
      LAUNCH_CHECK("k_bwd<pointgrad>");
      hipLaunchKernelGGL(k_normal_finish, gridv(P), dim3(256), 0, s, V, normal, (float*)nullptr, 1);
      LAUNCH_CHECK("k_normal_finish");
    }
  }
  return DISTR_OK;
}

int render_backward_impl(distr_ctx* ctx, const distr_render_cfg* cfg, int nviews, const void* ws, size_t ws_bytes, const float* g_zdepth,
                         const float* g_min_sdf, const float* g_depth, const float* g_normal, float* g_latent, float* g_R,
                         float* g_T, void* ws_bwd, size_t ws_bwd_bytes, void* stream) {
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  int rc = check_cfg(ctx, cfg);
  if (rc) return rc;
  if (nviews < 1 || nviews > DISTR_MAX_VIEWS) return fail(ctx, DISTR_ERR_INVALID_ARG, "nviews %d not in [1, %d]", nviews, DISTR_MAX_VIEWS);
  if (!ws || !ws_bwd) return fail(ctx, DISTR_ERR_INVALID_ARG, "null workspace");
  if (!cfg->save_for_backward) return fail(ctx, DISTR_ERR_INVALID_ARG, "forward was run with save_for_backward=0");
  View V;
  const size_t single = make_view(*cfg, const_cast<void*>(ws), V, ctx->save_masks, nviews);
  if (ws_bytes < single * nviews) return fail(ctx, DISTR_ERR_WORKSPACE, "forward workspace too small: %zu < %zu", ws_bytes, single * nviews);
  const size_t bsingle = bwd_bytes(*cfg);
  if (ws_bwd_bytes < bsingle * nviews) return fail(ctx, DISTR_ERR_WORKSPACE, "backward workspace too small: %zu < %zu", ws_bwd_bytes, bsingle * nviews);
  hipStream_t s = (hipStream_t)stream;
  const DecoderDev& D = ctx->D;
  const int P = V.P;
  const unsigned NV = (unsigned)nviews;
  const size_t smax = (size_t)P * cfg->buffer_size + 1;
  constexpr int TILE = 64;
  Carver cv(ws_bwd);
  BwdWs W;
  W.bstride = (int64_t)bsingle;
  W.samples = cv.take<Sample>(smax);
  const unsigned tiles = (unsigned)((smax + TILE - 1) / TILE);
  const unsigned prows = (unsigned)((smax + 31) / 32) + 256;     // partial rows (sized for 32-sample tiles: bwd_bytes)
  W.partial = cv.take<float>((size_t)prows * PSTRIDE);
  const int nblk = (P + 255) / 256;
  const unsigned nchunks = (prows + BWD_CHUNK - 1) / BWD_CHUNK;
  W.chunk_part = cv.take<float>((size_t)nchunks * PSTRIDE);
  W.BB.cnt = cv.take<int32_t>(nblk); W.BB.off = cv.take<int32_t>(nblk); W.BB.acc = cv.take<float>((size_t)nblk * 16);
  hipLaunchKernelGGL(k_bwd_prep<false>, dim3(nblk, NV), dim3(256), 0, s, V, g_zdepth, g_min_sdf, g_depth, g_normal, W);
  LAUNCH_CHECK("k_bwd_prep<count>");
  hipLaunchKernelGGL(k_bwd_scan, dim3(NV), dim3(256), 0, s, V, W, nblk);
  LAUNCH_CHECK("k_bwd_scan");
  hipLaunchKernelGGL(k_bwd_prep<true>, dim3(nblk, NV), dim3(256), 0, s, V, g_zdepth, g_min_sdf, g_depth, g_normal, W);
  LAUNCH_CHECK("k_bwd_prep<emit>");
  BwdArgs B;
  memset(&B, 0, sizeof(B));
  B.V = V; B.samples = W.samples; B.partial = W.partial; B.bstride = W.bstride; B.B6 = ctx->B6; B.H3 = ctx->H3;
  // tile-size split of every view's sample list (bwd_range): full rounds on 64-sample tiles, a small remainder on 32-sample tiles
  const bool bsplit = V.save_masks != 0;
  if (bsplit) {
    B.split = 1;
    if (cfg->arith == DISTR_ARITH_F16X3) {      // the dX chain in the arithmetic of the forward it differentiates
      hipLaunchKernelGGL((k_bwd<BWD_SAVED, 2, 2>), dim3(NV * (unsigned)((smax + 63) / 64)), dim3(NTHREADS), 0, s, B, D);
      hipLaunchKernelGGL((k_bwd<BWD_SAVED, 1, 2>), dim3(NV * (unsigned)((std::min<size_t>(smax, 8192) + 31) / 32)), dim3(NTHREADS), 0, s, B, D);
    } else if (cfg->arith == DISTR_ARITH_BF16X6) {
      hipLaunchKernelGGL((k_bwd<BWD_SAVED, 2, 1>), dim3(NV * (unsigned)((smax + 63) / 64)), dim3(NTHREADS), 0, s, B, D);
      hipLaunchKernelGGL((k_bwd<BWD_SAVED, 1, 1>), dim3(NV * (unsigned)((std::min<size_t>(smax, 8192) + 31) / 32)), dim3(NTHREADS), 0, s, B, D);
    } else {
      hipLaunchKernelGGL((k_bwd<BWD_SAVED, 2>), dim3(NV * (unsigned)((smax + 63) / 64)), dim3(NTHREADS), 0, s, B, D);
      hipLaunchKernelGGL((k_bwd<BWD_SAVED, 1>), dim3(NV * (unsigned)((std::min<size_t>(smax, 8192) + 31) / 32)), dim3(NTHREADS), 0, s, B, D);
    }
  } else {
    hipLaunchKernelGGL((k_bwd<BWD_FULL, 2>), dim3(NV * tiles), dim3(NTHREADS), 0, s, B, D);     // DISTR_SAVE_MASKS=0: recompute the forward
  }
  LAUNCH_CHECK("k_bwd");
  hipLaunchKernelGGL(k_bwd_reduce, dim3((2 * HID + 12 + 255) / 256, nchunks, NV), dim3(256), 0, s, V, W, BWD_CHUNK, bsplit ? -1 : TILE);
  LAUNCH_CHECK("k_bwd_reduce");
  hipLaunchKernelGGL(k_bwd_final, dim3(NV), dim3(256), 0, s, V, D, W, (int)nchunks, BWD_CHUNK, bsplit ? -1 : TILE, g_latent, g_R, g_T);
  LAUNCH_CHECK("k_bwd_final");
  return DISTR_OK;
}

int render_normal_impl(distr_ctx* ctx, const distr_render_cfg* cfg, int nviews, const float* latent, int64_t lat_stride, const float* R,
                       const float* T, const float* zdepth, const uint8_t* mask, float* normal3xP, void* ws, size_t ws_bytes,
                       void* stream) {
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  int rc = check_cfg(ctx, cfg);
  if (rc) return rc;
  if (nviews < 1 || nviews > DISTR_MAX_VIEWS) return fail(ctx, DISTR_ERR_INVALID_ARG, "nviews %d not in [1, %d]", nviews, DISTR_MAX_VIEWS);
  if (lat_stride != 0 && lat_stride < LAT) return fail(ctx, DISTR_ERR_INVALID_ARG, "latent_stride must be 0 (shared shape code) or >= %d", LAT);
  if (!latent || !R || !T || !zdepth || !mask || !normal3xP || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  View V;
  const size_t single = make_view(*cfg, ws, V, ctx->save_masks, nviews);
  if (ws_bytes < single * nviews) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small: %zu < %zu", ws_bytes, single * nviews);
  hipStream_t s = (hipStream_t)stream;
  const DecoderDev& D = ctx->D;
  const int P = V.P;
  const unsigned NV = (unsigned)nviews;
  ViewFlags vf;
  rc = make_view_flags(ctx, *cfg, nviews, nullptr, vf);
  if (rc) return rc;
  hipLaunchKernelGGL(k_prep, dim3(4, NV), dim3(256), 0, s, V, D, latent, lat_stride, R, T, vf);
  LAUNCH_CHECK("k_prep");
  HIP_TRY(hipMemsetAsync(normal3xP, 0, (size_t)nviews * P * 3 * sizeof(float), s));
  hipLaunchKernelGGL(k_mask_list, dim3((unsigned)((P + 255) / 256), NV), dim3(256), 0, s, V, mask);
  LAUNCH_CHECK("k_mask_list");
  BwdArgs B;
  memset(&B, 0, sizeof(B));
  B.V = V; B.zdepth = zdepth; B.zstride = (int64_t)P * sizeof(float);
  hipLaunchKernelGGL((k_bwd<BWD_POINTGRAD, 2>), dim3(NV * (unsigned)((P + 63) / 64)), dim3(NTHREADS), 0, s, B, D);
  LAUNCH_CHECK("k_bwd<pointgrad>");
  hipLaunchKernelGGL(k_normal_finish, dim3((unsigned)((P + 255) / 256), NV), dim3(256), 0, s, V, (float*)nullptr, normal3xP, 0);
  LAUNCH_CHECK("k_normal_finish");
  return DISTR_OK;
}

}  // namespace

int distr_render_forward(distr_ctx* ctx, const distr_render_cfg* cfg, const float* latent, const float* R, const float* T,
                         float* zdepth, uint8_t* mask, float* min_sdf, float* depth, float* normal, void* ws, size_t ws_bytes,
                         void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  return render_forward_impl(ctx, cfg, 1, nullptr, latent, 0, R, T, zdepth, mask, min_sdf, depth, normal, ws, ws_bytes, stream);
}

int distr_render_forward_batch(distr_ctx* ctx, const distr_render_cfg* cfg, int32_t nviews, const int32_t* view_flags,
                               const float* latent, int64_t latent_stride, const float* R, const float* T, float* zdepth,
                               uint8_t* mask, float* min_sdf, float* depth, float* normal, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  return render_forward_impl(ctx, cfg, nviews, view_flags, latent, latent_stride, R, T, zdepth, mask, min_sdf, depth, normal, ws,
                             ws_bytes, stream);
}

int distr_render_backward(distr_ctx* ctx, const distr_render_cfg* cfg, const void* ws, size_t ws_bytes, const float* g_zdepth,
                          const float* g_min_sdf, const float* g_depth, const float* g_normal, float* g_latent, float* g_R,
                          float* g_T, void* ws_bwd, size_t ws_bwd_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  return render_backward_impl(ctx, cfg, 1, ws, ws_bytes, g_zdepth, g_min_sdf, g_depth, g_normal, g_latent, g_R, g_T, ws_bwd,
                              ws_bwd_bytes, stream);
}

int distr_render_backward_batch(distr_ctx* ctx, const distr_render_cfg* cfg, int32_t nviews, const void* ws, size_t ws_bytes,
                                const float* g_zdepth, const float* g_min_sdf, const float* g_depth, const float* g_normal,
                                float* g_latent, float* g_R, float* g_T, void* ws_bwd, size_t ws_bwd_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  return render_backward_impl(ctx, cfg, nviews, ws, ws_bytes, g_zdepth, g_min_sdf, g_depth, g_normal, g_latent, g_R, g_T, ws_bwd,
                              ws_bwd_bytes, stream);
}

int distr_render_normal(distr_ctx* ctx, const distr_render_cfg* cfg, const float* latent, const float* R, const float* T,
                        const float* zdepth, const uint8_t* mask, float* normal3xP, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  return render_normal_impl(ctx, cfg, 1, latent, 0, R, T, zdepth, mask, normal3xP, ws, ws_bytes, stream);
}

int distr_render_normal_batch(distr_ctx* ctx, const distr_render_cfg* cfg, int32_t nviews, const float* latent, int64_t latent_stride,
                              const float* R, const float* T, const float* zdepth, const uint8_t* mask, float* normal3xP, void* ws,
                              size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  return render_normal_impl(ctx, cfg, nviews, latent, latent_stride, R, T, zdepth, mask, normal3xP, ws, ws_bytes, stream);
}

size_t distr_mlp_workspace_bytes(int64_t n) { (void)n; return 2 * HID * sizeof(float) + 256; }

int distr_mlp_eval(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, float clamp, float* sdf, void* ws,
                   size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  if (n < 0 || (n > 0 && (!xyz || !sdf)) || !latent || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad argument");
  if (ws_bytes < distr_mlp_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small");
  if (n == 0) return DISTR_OK;
  hipStream_t s = (hipStream_t)stream;
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->D, latent);
  LAUNCH_CHECK("k_latent_consts");
  MarchArgs A;
  memset(&A, 0, sizeof(A));
  A.xyz = xyz; A.sdf_out = sdf; A.c0c4 = c0c4; A.n = n; A.clamp = clamp;
  MarchTimer timer(ctx, s);
  timer.begin();
  // a point list that fits one wave of 16-ray tiles runs on those (107 us instead of a 380 us 64-ray tile: decode_sdf on a few
  // thousand points is latency-bound); same values bit for bit
  if (n <= std::min(ctx->tail16_threshold, ctx->hybrid_threshold))
    hipLaunchKernelGGL((k_march16<MODE_EVAL, false>), dim3((unsigned)((n + 15) / 16)), dim3(NTHREADS), 0, s, A, ctx->D, ctx->D16);
  else hipLaunchKernelGGL((k_march<MODE_EVAL, 2, false>), dim3((unsigned)((n + 63) / 64)), dim3(NTHREADS), 0, s, A, ctx->D);
  timer.end();
  LAUNCH_CHECK("k_march<eval>");
  return DISTR_OK;
}

int distr_mlp_eval_bf16x6(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, float clamp, float* sdf, void* ws,
                          size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  if (n < 0 || (n > 0 && (!xyz || !sdf)) || !latent || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad argument");
  if (ws_bytes < distr_mlp_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small");
  if (n == 0) return DISTR_OK;
  hipStream_t s = (hipStream_t)stream;
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->D, latent);      // (exact f32: the latent columns stay a per-call constant)
  LAUNCH_CHECK("k_latent_consts");
  MarchTimer timer(ctx, s);
  timer.begin();
  hipLaunchKernelGGL(k_eval_b6, dim3((unsigned)((n + 63) / 64)), dim3(NTHREADS), 0, s, xyz, n, (const float*)c0c4, clamp, sdf, ctx->D, ctx->B6);
  timer.end();
  LAUNCH_CHECK("k_eval_b6");
  return DISTR_OK;
}

int distr_mlp_eval_f16x3(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, float clamp, float* sdf, void* ws,
                         size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  if (!ctx->h3_ok) return fail(ctx, DISTR_ERR_UNSUPPORTED, "f16x3: a decoder weight times %g leaves the f16 range; use bf16x6 or f32 for this decoder", (double)H3_SW);
  if (n < 0 || (n > 0 && (!xyz || !sdf)) || !latent || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad argument");
  if (ws_bytes < distr_mlp_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small");
  if (n == 0) return DISTR_OK;
  hipStream_t s = (hipStream_t)stream;
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->D, latent);      // (exact f32: the latent columns stay a per-call constant)
  LAUNCH_CHECK("k_latent_consts");
  MarchTimer timer(ctx, s);
  timer.begin();
  hipLaunchKernelGGL(k_eval_h3, dim3((unsigned)((n + 63) / 64)), dim3(NTHREADS), 0, s, xyz, n, (const float*)c0c4, clamp, sdf, ctx->D, ctx->H3);
  timer.end();
  LAUNCH_CHECK("k_eval_h3");
  return DISTR_OK;
}

int distr_mlp_grad(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, float* sdf, float* grad, void* ws,
                   size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  if (n < 0 || (n > 0 && (!xyz || !sdf || !grad)) || !latent || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad argument");
  if (ws_bytes < distr_mlp_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small");
  if (n == 0) return DISTR_OK;
  hipStream_t s = (hipStream_t)stream;
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->D, latent);
  LAUNCH_CHECK("k_latent_consts");
  BwdArgs B;
  memset(&B, 0, sizeof(B));
  B.n = n; B.xyz = xyz; B.c0c4 = c0c4; B.out_sdf = sdf; B.out_g = grad;
  hipLaunchKernelGGL((k_bwd<BWD_POINTGRAD, 2>), dim3((unsigned)((n + 63) / 64)), dim3(NTHREADS), 0, s, B, ctx->D);
  LAUNCH_CHECK("k_bwd<pointgrad>");
  return DISTR_OK;
}

size_t distr_mlp_backward_workspace_bytes(int64_t n) {
  const size_t tiles = (size_t)((n + 31) / 32);
  return distr_mlp_workspace_bytes(n) + tiles * PSTRIDE * sizeof(float) + 256;
}

int distr_mlp_backward(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, const float* g_sdf, float clamp,
                       float* g_xyz, float* g_latent, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  if (n < 0 || (n > 0 && (!xyz || !g_sdf)) || !latent || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad argument");
  if (ws_bytes < distr_mlp_backward_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    if (g_latent) HIP_TRY(hipMemsetAsync(g_latent, 0, LAT * sizeof(float), s));
    return DISTR_OK;
  }
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  float* partial = (float*)(((uintptr_t)(c0c4 + 2 * HID) + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->D, latent);
  LAUNCH_CHECK("k_latent_consts");
  BwdArgs B;
  memset(&B, 0, sizeof(B));
  B.n = n; B.xyz = xyz; B.c0c4 = c0c4; B.coef = g_sdf; B.clamp = clamp; B.partial = partial; B.out_g = g_xyz;
  const unsigned tiles = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL((k_bwd<BWD_POINTGRAD, 2>), dim3(tiles), dim3(NTHREADS), 0, s, B, ctx->D);
  LAUNCH_CHECK("k_bwd<pointgrad+latent>");
  if (g_latent) {
    hipLaunchKernelGGL(k_points_latent_grad, dim3(1), dim3(256), 0, s, (const float*)partial, (int)tiles, ctx->D, g_latent);
    LAUNCH_CHECK("k_points_latent_grad");
  }
  return DISTR_OK;
}

int distr_debug_mlp_layer(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, int layer, float* out, void* ws,
                          size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  if (n <= 0 || !xyz || !out || !latent || !ws || layer < 0 || layer > 7) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad argument");
  if (ws_bytes < distr_mlp_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->D, latent);
  LAUNCH_CHECK("k_latent_consts");
  hipLaunchKernelGGL((k_debug_layer<2>), dim3((unsigned)((n + 63) / 64)), dim3(NTHREADS), 0, s, xyz, n, (const float*)c0c4, layer, out, ctx->D, (long long*)nullptr);
  LAUNCH_CHECK("k_debug_layer");
  return DISTR_OK;
}

int distr_debug_tile_timing(distr_ctx* ctx, const float* latent, const float* xyz, int64_t n, float* sdf_out, long long* ts_out,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (!ctx->has_decoder) return fail(ctx, DISTR_ERR_NO_DECODER, "distr_set_decoder has not been called");
  if (n <= 0 || !xyz || !sdf_out || !ts_out || !latent || !ws) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad argument");
  if (ws_bytes < distr_mlp_workspace_bytes(n)) return fail(ctx, DISTR_ERR_WORKSPACE, "workspace too small");
  hipStream_t s = (hipStream_t)stream;
  float* c0c4 = (float*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  hipLaunchKernelGGL(k_latent_consts, dim3(4), dim3(256), 0, s, c0c4, ctx->D, latent);
  LAUNCH_CHECK("k_latent_consts");
  hipLaunchKernelGGL((k_debug_layer<2>), dim3((unsigned)((n + 63) / 64)), dim3(NTHREADS), 0, s, xyz, n, (const float*)c0c4, 8, sdf_out, ctx->D, ts_out);
  LAUNCH_CHECK("k_debug_layer<timing>");
  return DISTR_OK;
}

int distr_get_render_stats(distr_ctx* ctx, const distr_render_cfg* cfg, const void* ws, distr_render_stats* out, void* stream) {
  if (!ctx || !ws || !out) return fail(ctx, DISTR_ERR_INVALID_ARG, "null argument");
  EntryGuard guard_(ctx);
  int rc = check_cfg(ctx, cfg);
  if (rc) return rc;
  if (out->struct_size != sizeof(distr_render_stats))   // never write past what the caller allocated
    return fail(ctx, DISTR_ERR_INVALID_ARG, "distr_render_stats.struct_size is %u, expected %zu (set it before the call)", out->struct_size, sizeof(distr_render_stats));
  View V;
  make_view(*cfg, const_cast<void*>(ws), V, ctx->save_masks);
  hipStream_t s = (hipStream_t)stream;
  static thread_local std::vector<char> hostbuf;
  hostbuf.resize(sizeof(Consts));
  HIP_TRY(hipMemcpyAsync(hostbuf.data(), V.C, sizeof(Consts), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const Consts* C = (const Consts*)hostbuf.data();
  memset(out, 0, sizeof(*out));
  out->struct_size = (uint32_t)sizeof(*out);
  out->num_in_sphere = C->cnt_level[0];
  int64_t ev = 0, launches = 0;
  for (int l = 1; l < V.nlev; ++l) { ev += (int64_t)V.lv[l].steps * C->cnt_level[l]; launches += V.lv[l].steps; }
  if (cfg->marcher == DISTR_MARCH_TRIVIAL) ev += (int64_t)V.fine_steps * C->cnt_level[0];
  else for (int t = 0; t < V.fine_steps; ++t) ev += C->cnt_live[t] + C->cnt_sticky[t];
  launches += (C->tail_from < V.fine_steps) ? C->tail_from + 1 : V.fine_steps;     // (steps from tail_from on share one launch: k_tail)
  out->num_point_evals = ev;
  out->num_march_launches = launches;
  out->num_valid = C->cnt_valid;
  out->num_grad_samples = C->cnt_samples;
  out->cluster_fallbacks = C->xchg_err;
  out->f16_overflows = C->f16_overflow;
  out->tail_from = C->tail_from;
  out->tail_steals = C->tail_steals;
  return DISTR_OK;
}

int distr_profile_enable(distr_ctx* ctx, int enable) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  ctx->profiling = enable != 0;
  ctx->ev_used = 0;
  return DISTR_OK;
}

int distr_profile_read(distr_ctx* ctx, int64_t* launches, double* total_ms, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  double tot = 0.0;
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second));
    tot += ms;
  }
  if (launches) *launches = (int64_t)ctx->ev_used;
  if (total_ms) *total_ms = tot;
  ctx->ev_used = 0;
  return DISTR_OK;
}

int distr_profile_read_list(distr_ctx* ctx, float* ms_out, int64_t cap, int64_t* n, void* stream) {
  if (!ctx || !n) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  *n = (int64_t)ctx->ev_used;
  for (size_t i = 0; i < ctx->ev_used && (int64_t)i < cap && ms_out; ++i)
    HIP_TRY(hipEventElapsedTime(&ms_out[i], ctx->ev_pool[i].first, ctx->ev_pool[i].second));
  return DISTR_OK;
}

int distr_get_live_counts(distr_ctx* ctx, const distr_render_cfg* cfg, const void* ws, int32_t* out, int32_t cap, int32_t* n,
                          void* stream) {
  if (!ctx || !ws || !out || !n) return fail(ctx, DISTR_ERR_INVALID_ARG, "null argument");
  EntryGuard guard_(ctx);
  int rc = check_cfg(ctx, cfg);
  if (rc) return rc;
  View V;
  make_view(*cfg, const_cast<void*>(ws), V, ctx->save_masks);
  hipStream_t s = (hipStream_t)stream;
  static thread_local std::vector<char> hostbuf;
  hostbuf.resize(sizeof(Consts));
  HIP_TRY(hipMemcpyAsync(hostbuf.data(), V.C, sizeof(Consts), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const Consts* C = (const Consts*)hostbuf.data();
  int k = 0;
  for (int l = V.nlev - 1; l >= 1; --l)
    for (int st = 0; st < V.lv[l].steps; ++st) { if (k < cap) out[k] = C->cnt_level[l]; ++k; }
  for (int t = 0; t < V.fine_steps; ++t) { if (k < cap) out[k] = (cfg->marcher == DISTR_MARCH_TRIVIAL) ? C->cnt_level[0] : C->cnt_live[t] + C->cnt_sticky[t]; ++k; }
  *n = k;
  return DISTR_OK;
}

int distr_debug_xchg_ts(distr_ctx* ctx, void* stream, int64_t* out64) {
  if (!ctx || !out64) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  for (auto& r : ctx->xr)
    if (r.used && r.stream == (hipStream_t)stream) {
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      HIP_TRY(hipMemcpy(out64, r.flags + 256 * 128, 64 * sizeof(long long), hipMemcpyDeviceToHost));
      return DISTR_OK;
    }
  return fail(ctx, DISTR_ERR_INVALID_ARG, "no exchange region for this stream");
}

// ------------------------------------------------------------------------------------------ f2 / f3: fused losses
size_t distr_loss_workspace_bytes(int32_t H, int32_t W) {
  const size_t nblk = ((size_t)H * W + 255) / 256;
  return nblk * 24 * sizeof(float) + 256;
}

static int loss_args(distr_ctx* ctx, int32_t H, int32_t W, const void* ws, size_t ws_bytes) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  if (H < 1 || W < 1 || (int64_t)H * W >= (1 << 26)) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad image size %dx%d", H, W);
  if (ws_bytes < distr_loss_workspace_bytes(H, W) || !ws) return fail(ctx, DISTR_ERR_WORKSPACE, "loss workspace too small");
  return DISTR_OK;
}

int distr_single_loss_forward(distr_ctx* ctx, int32_t H, int32_t W, const float* depth, const float* normal, const uint8_t* mask,
                              const float* min_sdf, const float* gt_depth, const float* gt_normal, const uint8_t* gt_mask,
                              float threshold, float* out8, void* ws, size_t ws_bytes, void* stream) {
  int rc = loss_args(ctx, H, W, ws, ws_bytes);
  if (rc) return rc;
  EntryGuard guard_(ctx);
  if (!mask || !min_sdf || !gt_mask || !out8 || (gt_depth && !depth) || (gt_normal && !normal))
    return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  hipStream_t s = (hipStream_t)stream;
  SingleLossArgs A{H * W, depth, normal, mask, min_sdf, gt_depth, gt_normal, gt_mask, threshold};
  const int nblk = (A.P + 255) / 256;
  hipLaunchKernelGGL(k_single_loss_partial, dim3(nblk), dim3(256), 0, s, A, (float*)ws);
  LAUNCH_CHECK("k_single_loss_partial");
  hipLaunchKernelGGL(k_single_loss_final, dim3(1), dim3(64), 0, s, (const float*)ws, nblk, out8);
  LAUNCH_CHECK("k_single_loss_final");
  return DISTR_OK;
}

int distr_single_loss_backward(distr_ctx* ctx, int32_t H, int32_t W, const float* depth, const float* normal, const uint8_t* mask,
                               const float* min_sdf, const float* gt_depth, const float* gt_normal, const uint8_t* gt_mask,
                               float threshold, const float* out8, const float* g4, float* g_depth, float* g_normal,
                               float* g_min_sdf, void* stream) {
  if (!ctx) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (H < 1 || W < 1 || (int64_t)H * W >= (1 << 26)) return fail(ctx, DISTR_ERR_INVALID_ARG, "bad image size %dx%d", H, W);
  if (!mask || !min_sdf || !gt_mask || !out8 || !g4 || (gt_depth && !depth) || (gt_normal && !normal))
    return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  hipStream_t s = (hipStream_t)stream;
  SingleLossArgs A{H * W, depth, normal, mask, min_sdf, gt_depth, gt_normal, gt_mask, threshold};
  hipLaunchKernelGGL(k_single_loss_bwd, grid1(A.P), dim3(256), 0, s, A, out8, g4, g_depth, g_normal, g_min_sdf);
  LAUNCH_CHECK("k_single_loss_bwd");
  return DISTR_OK;
}

static WarpArgs warp_args(const distr_warp_cfg* c, const float* z1, const uint8_t* m1, const float* z2, const float* img1,
                          const float* img2, const float* R1, const float* T1, const float* R2, const float* T2) {
  WarpArgs A;
  A.H = c->H; A.W = c->W;
  memcpy(A.K, c->K, sizeof(A.K));
  memcpy(A.K_inv, c->K_inv, sizeof(A.K_inv));
  A.thres_depth = c->thres_depth;
  A.z1 = z1; A.m1 = m1; A.z2 = z2; A.img1 = img1; A.img2 = img2;
  A.R1 = R1; A.T1 = T1; A.R2 = R2; A.T2 = T2;
  return A;
}

int distr_warp_loss_forward(distr_ctx* ctx, const distr_warp_cfg* cfg, const float* zdepth1, const uint8_t* mask1,
                            const float* zdepth2, const float* img1, const float* img2, const float* R1, const float* T1,
                            const float* R2, const float* T2, float* out3, uint8_t* keep, float* color1, float* color2,
                            void* ws, size_t ws_bytes, void* stream) {
  if (!ctx || !cfg) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (cfg->struct_size != sizeof(distr_warp_cfg)) return fail(ctx, DISTR_ERR_INVALID_ARG, "distr_warp_cfg.struct_size is %u, expected %zu", cfg->struct_size, sizeof(distr_warp_cfg));
  int rc = loss_args(ctx, cfg->H, cfg->W, ws, ws_bytes);
  if (rc) return rc;
  if (!zdepth1 || !mask1 || !zdepth2 || !img1 || !img2 || !R1 || !T1 || !R2 || !T2 || !out3)
    return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  hipStream_t s = (hipStream_t)stream;
  const WarpArgs A = warp_args(cfg, zdepth1, mask1, zdepth2, img1, img2, R1, T1, R2, T2);
  const int nblk = (A.H * A.W + 255) / 256;
  hipLaunchKernelGGL(k_warp_fwd, dim3(nblk), dim3(256), 0, s, A, keep, color1, color2, (float*)ws);
  LAUNCH_CHECK("k_warp_fwd");
  hipLaunchKernelGGL(k_warp_final, dim3(1), dim3(64), 0, s, (const float*)ws, nblk, out3);
  LAUNCH_CHECK("k_warp_final");
  return DISTR_OK;
}

int distr_warp_loss_backward(distr_ctx* ctx, const distr_warp_cfg* cfg, const float* zdepth1, const uint8_t* mask1,
                             const float* zdepth2, const float* img1, const float* img2, const float* R1, const float* T1,
                             const float* R2, const float* T2, const float* out3, const float* g_loss, float* g_zdepth1,
                             float* g_cam, void* ws, size_t ws_bytes, void* stream) {
  if (!ctx || !cfg) return DISTR_ERR_INVALID_ARG;
  EntryGuard guard_(ctx);
  if (cfg->struct_size != sizeof(distr_warp_cfg)) return fail(ctx, DISTR_ERR_INVALID_ARG, "distr_warp_cfg.struct_size is %u, expected %zu", cfg->struct_size, sizeof(distr_warp_cfg));
  int rc = loss_args(ctx, cfg->H, cfg->W, ws, ws_bytes);
  if (rc) return rc;
  if (!zdepth1 || !mask1 || !zdepth2 || !img1 || !img2 || !R1 || !T1 || !R2 || !T2 || !out3 || !g_loss || !g_cam)
    return fail(ctx, DISTR_ERR_INVALID_ARG, "null device pointer");
  hipStream_t s = (hipStream_t)stream;
  const WarpArgs A = warp_args(cfg, zdepth1, mask1, zdepth2, img1, img2, R1, T1, R2, T2);
  const int nblk = (A.H * A.W + 255) / 256;
  hipLaunchKernelGGL(k_warp_bwd, dim3(nblk), dim3(256), 0, s, A, out3, g_loss, g_zdepth1, (float*)ws);
  LAUNCH_CHECK("k_warp_bwd");
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(64), 0, s, (const float*)ws, nblk, 24, g_cam);
  LAUNCH_CHECK("k_sum_partials");
  return DISTR_OK;
}

}  // extern "C"
