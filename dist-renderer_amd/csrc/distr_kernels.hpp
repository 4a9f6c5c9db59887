// distr_kernels.hpp -- device-side data layout + all HIP kernels of the DIST hot path (gfx950).
//
// Reference functions restated here (paths under the reference tree):
//   ray setup / unit-sphere init        core/sdfrenderer/renderer.py:171-282
//   trivial / recursive / pyramid march core/sdfrenderer/renderer.py:472-583, 713-805
//   min-|sdf| sample selection          core/sdfrenderer/renderer.py:304-420   (done ONLINE: O(bs) state per ray
//                                        instead of the reference's (steps x N x 5) history tensors)
//   render_depth / render outputs       core/sdfrenderer/renderer.py:836-878, 943-999
//   depth2normal                        core/utils/render_utils.py:9-43
//   autograd backward                   (PyTorch tape in the reference; SURVEY.md Appendix A.6 contract)
#pragma once
#include <type_traits>

#include "distr_mlp.hpp"
#include "distr_mlp_b6.hpp"
#include "distr_mlp_h3.hpp"
#include "../../include/distr.h"

namespace distr {

constexpr int MAX_BS = DISTR_MAX_BUFFER_SIZE;
constexpr int MAX_LEVELS = DISTR_MAX_PYRAMID_LEVELS;     // levels of the pyramid marcher (renderer.py:713-805: one per scale_list entry)
constexpr int MAX_STEPS = 2048;
constexpr int PSTRIDE = 1040;  // floats per backward tile partial: sd0[512] sd4[512] gR[9] gc[3] pad[4]

struct Consts {
  float c0[HID];          // b0 + W0[:, :256] * latent
  float c4[HID];          // b4 + W4[:, 253:509] * latent
  float latent[LAT];
  float R[9], T[3], c[3];
  float cdist;
  int32_t vflags;         // per-view gradient switches (VF_*): a batch renders views with different no_grad_* options in one launch
  int32_t xchg_err;       // cluster tiles that fell back to the single-workgroup path (not assembled in time / barrier timeout); results stay exact
  float f_origin;         // f(0,0,0): sample point of padded history rows (renderer.py:539, 555)
  int32_t origin_done;    // f_origin has been evaluated (by the launch that turned sticky; else the last step evaluates it)
  uint32_t maxinit_bits[MAX_LEVELS];
  int32_t cnt_level[MAX_LEVELS];   // [0] rays hitting the sphere; [l] valid pixels of the coarse grid of level l
  int32_t cnt_valid;
  int32_t cnt_normal;
  int32_t cnt_samples;
  float pad_coef;
  float cam_acc[12];      // backward: gR[9] + g_campos[3] from non-MLP terms
  float red[PSTRIDE];     // backward: reduced tile partials
  int32_t cnt_live[MAX_STEPS + 2];  // live rays entering fine step t (compacted list of that step's launch)
  int32_t cnt_sticky[MAX_STEPS + 2]; // rays evaluated at fine step t by sticky tiles (no list: sticky_tile16); statistics only
  int32_t f16_overflow;   // split-f16 arithmetic (cfg.arith = 2): decoder evaluations whose value left the f16 range (non-finite result)
  // persistent tail kernel (k_tail): fine steps [tail_from, fine_steps) run inside ONE launch (tail_from = fine_steps: no tail launch).
    int32_t tail_from;
  int32_t tail_steals;    // tiles a workgroup other than their owner evaluated after waiting too long (owner not resident); statistics
  // tail_sync[2k] = virtual tiles of step tail_from + k evaluated with their stores complete (view 0's only: the step barrier);
  // tail_sync[2k + 1] = THIS view's live rays entering step tail_from + k + 1 (= cnt_live of that step, kept next to the barrier word so
  // that one 8-byte load answers both "is the step over" and "how many rays does the next one have")
  alignas(8) int32_t tail_sync[2 * (MAX_STEPS + 2)];
};

struct LevelView {
  int32_t h, w, n, steps;
  int32_t y0, full_h;     // row band: first row of this level's grid in the full image's level grid; rows of the full grid
  float scale, off;       // pixel centre = scale*i + off  (renderer.py:616-617)
  int32_t rdiv, sdiv;     // grid of this level = the next finer level's / rdiv (scale_list[l] / scale_list[l-1], renderer.py:739), = the full image's / sdiv
  uint8_t* valid;
  int32_t* list;
  float* cinit;           // start depth of this level
  float* cm;              // marching depth
  float* rs;              // [steps][n] sdf rows
  float* rzb;             // [steps][n] depth before the step
  float* rza;             // [steps][n] depth after the step
};

struct View {
  distr_render_cfg cfg;
  Consts* C;
  LevelView lv[MAX_LEVELS];
  int32_t nlev, P, fine_steps, pyramid;
  int32_t row0, rows, band;   // band: rows [row0, row0+rows) of cfg.H are rendered (band != 0: a proper sub-range)
  int32_t* live[2];
  float *m, *init_now, *maxbound, *minabs, *first_sdf;
  float *tk_s, *tk_zb, *tk_za;   // [bs][P] selected rows: sdf, depth before, depth after (pyramid) / marching depth after
  int32_t* tk_src;               // [bs][P] row source (see src_* helpers) or -1 for a padded row
  int32_t* tk_slot;              // [bs][P] physical ReLU-mask slot (0..bs) of the entry
  int32_t* tclaim;               // [P/16 + 2] k_tail: claim word of the view's 16-ray tiles (last step index + 1 somebody took the tile at)
  int32_t tail_from;             // first full-resolution step of the persistent tail launch (host decision; = fine_steps: none)
  // ReLU masks saved for the backward pass: one 512-byte block (8 layers x 512 bits) per kept row.
  //   block = px*(bs+1) + slot                      rows of the full-resolution march
  //   block = mfine + moff[lvl] + step*n_lvl + ray  rows of the coarse pyramid levels
  uint4* mstore;
  int64_t mfine, moff[MAX_LEVELS], morigin;   // morigin: block of f(origin) (sample point of padded rows)
  int32_t save_masks;
  float *zdepth_s, *depth_pre, *nrm_t;
  uint8_t* mask_s;
  int32_t* nlist;
  float *n_sdf, *n_g;
  // Batch of views (distr_render_forward_batch): `nviews` independent views (own camera, own latent constants, own live lists
  // and counters) share every launch. View b's workspace is this one shifted by b * vstride bytes (view_at); a march tile never
  // mixes views, so per-tile state (camera, c0 / c4 staged in LDS) stays uniform and every ray's arithmetic is exactly the
  // single-view one.
  int32_t nviews;
  int64_t vstride;
};

enum { VF_GRAD_DEPTH = 1, VF_GRAD_MASK = 2, VF_GRAD_CAMERA = 4 };

template <typename T>
__device__ __forceinline__ void adv(T*& p, int64_t d) { if (p) p = reinterpret_cast<T*>(reinterpret_cast<char*>(p) + d); }

// workspace of view b of a batch (b uniform over the workgroup: pure SALU pointer arithmetic, only the members a kernel
// touches are materialised)
__device__ __forceinline__ View view_at(const View& V0, int b) {
  View V = V0;
  if (V0.nviews <= 1) return V;
  const int64_t d = (int64_t)b * V0.vstride;
  adv(V.C, d);
#pragma unroll
  for (int l = 0; l < MAX_LEVELS; ++l) {
    adv(V.lv[l].valid, d); adv(V.lv[l].list, d); adv(V.lv[l].cinit, d); adv(V.lv[l].cm, d);
    adv(V.lv[l].rs, d); adv(V.lv[l].rzb, d); adv(V.lv[l].rza, d);
  }
  adv(V.live[0], d); adv(V.live[1], d);
  adv(V.m, d); adv(V.init_now, d); adv(V.maxbound, d); adv(V.minabs, d); adv(V.first_sdf, d);
  adv(V.tk_s, d); adv(V.tk_zb, d); adv(V.tk_za, d); adv(V.tk_src, d); adv(V.tk_slot, d); adv(V.tclaim, d);
  adv(V.mstore, d);
  adv(V.zdepth_s, d); adv(V.depth_pre, d); adv(V.nrm_t, d); adv(V.mask_s, d);
  adv(V.nlist, d); adv(V.n_sdf, d); adv(V.n_g, d);
  return V;
}

// Members of a (rebased, register-resident) View selected by a run-time index: written as selects, never as indexing into the
// local copy (a dynamically indexed member array would force the whole struct into scratch memory)
template <typename T>
__device__ __forceinline__ T sel3(int l, T a0, T a1, T a2) { return l == 2 ? a2 : (l == 1 ? a1 : a0); }   // operands by VALUE (a ternary of lvalues selects addresses)
template <typename T>
__device__ __forceinline__ T sel4(int l, T a0, T a1, T a2, T a3) { return l == 3 ? a3 : (l == 2 ? a2 : (l == 1 ? a1 : a0)); }
static_assert(MAX_LEVELS == 4, "sel4 / level_sel / moff_sel list the levels");
__device__ __forceinline__ LevelView level_sel(const View& V, int l) {
  LevelView L;
#define DISTR_LSEL(f) L.f = sel4(l, V.lv[0].f, V.lv[1].f, V.lv[2].f, V.lv[3].f)
  DISTR_LSEL(h); DISTR_LSEL(w); DISTR_LSEL(n); DISTR_LSEL(steps); DISTR_LSEL(y0); DISTR_LSEL(full_h); DISTR_LSEL(scale); DISTR_LSEL(off); DISTR_LSEL(rdiv); DISTR_LSEL(sdiv);
  DISTR_LSEL(valid); DISTR_LSEL(list); DISTR_LSEL(cinit); DISTR_LSEL(cm); DISTR_LSEL(rs); DISTR_LSEL(rzb); DISTR_LSEL(rza);
#undef DISTR_LSEL
  return L;
}
__device__ __forceinline__ int64_t moff_sel(const View& V, int l) { return sel4<int64_t>(l, V.moff[0], V.moff[1], V.moff[2], V.moff[3]); }
__device__ __forceinline__ int32_t* live_sel(const View& V, int i) { return sel3<int32_t*>(i & 1, V.live[0], V.live[1], V.live[1]); }

// Virtual concatenation of the views' work lists: view b contributes its count c_b rounded up to a multiple of `g` (so that no
// tile straddles two views); every wavefront computes the prefix sums redundantly (one load + six shuffles; nviews <= 64 = one
// lane per view). `p0` = the counter of view 0, the counter of view b lies b * stride bytes further. A single view (B = 1)
// takes none of the shuffles.
__device__ __forceinline__ int32_t vload(const int32_t* p0, int64_t stride, int B) {   // lane b: count of view b (0 beyond the batch)
  if (B <= 1) return *p0;
  const int lane = threadIdx.x & 63;
  return (lane < B) ? *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p0) + (int64_t)lane * stride) : 0;
}
__device__ __forceinline__ int32_t vpad(int32_t c, int g) { return (c + g - 1) & ~(g - 1); }
__device__ __forceinline__ int32_t vprefix(int32_t c, int B, int g) {                    // inclusive prefix of the padded counts
  int32_t v = vpad(c, g);
  if (B <= 1) return v;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int32_t t = __shfl_up(v, o); if (lane >= o) v += t; }
  return v;
}
__device__ __forceinline__ int32_t vtotal(int32_t incl, int B) { return (B <= 1) ? incl : __shfl(incl, 63); }
// view that holds virtual index `vidx` (< total): its index, the virtual start of its segment and its real count
__device__ __forceinline__ void vfind(int32_t c, int32_t incl, int B, int g, int64_t vidx, int& b, int64_t& start, int32_t& cnt) {
  if (B <= 1) { b = 0; start = 0; cnt = c; return; }
  const int lane = threadIdx.x & 63;
  const int32_t excl = incl - vpad(c, g);
  const unsigned long long hit = __ballot(lane < B && vidx >= excl && vidx < incl);
  b = __builtin_amdgcn_readfirstlane(hit ? (__ffsll((long long)hit) - 1) : 0);
  start = __shfl(excl, b);
  cnt = __shfl(c, b);
}

struct Sample { int32_t src; float zb; float coef; int32_t flags; float sdf; int32_t mblock; int32_t pad0, pad1; };

// row source encoding: fine rows (level 0): the full-resolution march step that produced the row (the pixel is the row's own, k_bwd_prep
// puts it into the Sample); coarse rows: level<<28 | step<<24 | ray (< 2^24)
__device__ __forceinline__ int32_t src_coarse(int lvl, int step, int ray) { return (lvl << 28) | (step << 24) | ray; }
__device__ __forceinline__ int src_level(int32_t src) { return src >> 28; }
__device__ __forceinline__ int src_ray(int32_t src) { return (src >> 28) ? (src & 0x00ffffff) : (src & 0x0fffffff); }
__device__ __forceinline__ int src_step(int32_t src) { return (src >> 24) & 15; }

// ------------------------------------------------------------------------------------------ geometry
struct RayGeo { float d[3], r[3], hx, hy, hz, rn, calib; };
struct Sph { float dist, chord, init_raw, ptq; float v[3]; bool in; };

__device__ __forceinline__ RayGeo make_ray(const float* Ki, const float* R, float px, float py) {
  RayGeo g;
  g.hx = Ki[0] * px + Ki[1] * py + Ki[2];
  g.hy = Ki[3] * px + Ki[4] * py + Ki[5];
  g.hz = Ki[6] * px + Ki[7] * py + Ki[8];
  const float hn = sqrtf(g.hx * g.hx + g.hy * g.hy + g.hz * g.hz);
  g.calib = g.hz / (hn + 1e-12f);
#pragma unroll
  for (int i = 0; i < 3; ++i) g.r[i] = R[0 * 3 + i] * g.hx + R[1 * 3 + i] * g.hy + R[2 * 3 + i] * g.hz;
  g.rn = sqrtf(g.r[0] * g.r[0] + g.r[1] * g.r[1] + g.r[2] * g.r[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) g.d[i] = g.r[i] / (g.rn + 1e-12f);
  return g;
}

__device__ __forceinline__ Sph intersect(float radius, const float* c, float cdist, const float* d) {
  Sph s;
  s.ptq = c[0] * d[0] + c[1] * d[1] + c[2] * d[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) s.v[i] = c[i] - s.ptq * d[i];
  s.dist = sqrtf(s.v[0] * s.v[0] + s.v[1] * s.v[1] + s.v[2] * s.v[2]);
  s.in = s.dist <= radius;
  const float value = radius * radius - s.dist * s.dist;
  s.chord = value >= 0.f ? 2.0f * sqrtf(value) : 0.f;
  s.init_raw = sqrtf(cdist * cdist - s.dist * s.dist) - s.chord / 2.0f;
  return s;
}

__device__ __forceinline__ void make_point(const float* M, const float* c, const float* d, float zd, float* p) {
  float q[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) q[i] = d[i] * zd + c[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = M[0 * 3 + i] * q[0] + M[1 * 3 + i] * q[1] + M[2 * 3 + i] * q[2];
}

__device__ __forceinline__ void level_center(const LevelView& L, int i, float& px, float& py) {
  px = L.scale * (float)(i % L.w) + L.off;
  py = L.scale * (float)(i / L.w + L.y0) + L.off;
}

struct CamRegs { float R[9], c[3], cdist; };
__device__ __forceinline__ CamRegs load_cam(const Consts* C) {
  CamRegs k;
#pragma unroll
  for (int i = 0; i < 9; ++i) k.R[i] = C->R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) k.c[i] = C->c[i];
  k.cdist = C->cdist;
  return k;
}

// XC = true: write-through stores (sc1: nothing stays dirty in this XCD's L2) and L1-bypassing loads -- the fence-free hand-off form of
// MI355X_MICROARCH.md ("sc1 stores AND sc1 loads"). Measured for the per-ray march state inside the persistent tail launch (k_tail) and NOT
// used there: every step got slower than with plain accesses + one release / acquire fence per tile (137 x 137 / 100 steps: 7.94 against
// 7.46 ms) -- a write-through store is acknowledged by memory, not by the L2, and the cluster tile's counted vmcnt waits of the next
// layers queue up behind it. Kept for the one place that needs it: the mask words a HELPER member of a cluster stores (nobody releases
// those, store_own_mask_words). XC = false: plain accesses.
template <bool XC, class T>
__device__ __forceinline__ T ld_x(const T* p) {
  if constexpr (XC) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}
template <bool XC, class T>
__device__ __forceinline__ void st_x(T* p, T v) {
  if constexpr (XC) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}

// number of set bits of a ballot below this lane (v_mbcnt: no per-lane 64-bit mask to build and keep)
__device__ __forceinline__ int ballot_rank(unsigned long long ball) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ball >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ball, 0u));
}

// wave-level stream compaction: appends `id` of flagged lanes to list, one atomic per wavefront (counter2: a second counter that gets
// the same increment -- k_tail keeps the next step's count next to the step's barrier word)
template <bool XC = false>
__device__ __forceinline__ void wave_append(bool flag, int32_t id, int32_t* list, int32_t* counter, int32_t* counter2 = nullptr) {
  const unsigned long long ball = __ballot(flag);
  if (ball == 0ull) return;
  const int lane = threadIdx.x & 63;
  const int n = __popcll(ball);
  int base = 0;
  if (lane == 0) { base = atomicAdd(counter, n); if (counter2) atomicAdd(counter2, n); }
  base = __shfl(base, 0);
  if (flag) st_x<XC>(list + base + ballot_rank(ball), id);
}

// block-level stream compaction for the full-image setup kernels (256 threads): one atomic per BLOCK -- with one per
// wavefront the 4096 atomics of a 512x512 level on a single counter cost more than the rest of the kernel
__device__ __forceinline__ void block_append(bool flag, int32_t id, int32_t* list, int32_t* counter) {
  __shared__ int32_t s_n[4], s_base;
  const unsigned long long ball = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_n[wave] = __popcll(ball);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = s_n[0] + s_n[1] + s_n[2] + s_n[3];
    s_base = tot ? atomicAdd(counter, tot) : 0;
  }
  __syncthreads();
  int base = s_base;
  for (int w = 0; w < wave; ++w) base += s_n[w];
  if (flag) list[base + ballot_rank(ball)] = id;
}

// wave-level max, then one atomicMax per wavefront (skipped when the wave has nothing to contribute)
__device__ __forceinline__ void wave_atomic_max(uint32_t* dst, uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
  if ((threadIdx.x & 63) == 0 && v != 0u) atomicMax(dst, v);
}

// ------------------------------------------------------------------------------------------ k_prep
struct ViewFlags { uint8_t f[DISTR_MAX_VIEWS]; };   // VF_* of every view of the batch (host knowledge: the no_grad_* options)

// grid (4, nviews): camera constants, latent constants c0 / c4 and counter reset of every view of the batch; view b reads
// latent + b * lat_stride (lat_stride = 0: one shape code shared by all views), R + 9 b, T + 3 b
DISTR_GLOBAL void __launch_bounds__(256) k_prep(View V0, DecoderDev D, const float* __restrict__ latent0, int64_t lat_stride,
                                              const float* __restrict__ R0, const float* __restrict__ T0, ViewFlags vf) {
  const int vb = blockIdx.y;
  Consts* C = view_at(V0, vb).C;
  const float* latent = latent0 + (int64_t)vb * lat_stride;
  const float* R = R0 + 9 * vb;
  const float* T = T0 + 3 * vb;
  const int gid = blockIdx.x * 256 + threadIdx.x;  // 1024 threads
  {
    const int o = gid & 511;
    const float* Wt = (gid < 512) ? D.W0lat_t : D.W4lat_t;
    float acc = (gid < 512) ? D.b0[o] : D.b4[o];
#pragma unroll 8
    for (int k = 0; k < LAT; ++k) acc = __builtin_fmaf(Wt[k * HID + o], latent[k], acc);
    if (gid < 512) C->c0[o] = acc; else C->c4[o] = acc;
  }
  if (blockIdx.x == 0) {
    const int t = threadIdx.x;
    C->latent[t] = latent[t];
    for (int i = t; i < MAX_STEPS + 2; i += 256) { C->cnt_live[i] = 0; C->cnt_sticky[i] = 0; C->tail_sync[2 * i] = 0; C->tail_sync[2 * i + 1] = 0; }
    for (int i = t; i < PSTRIDE; i += 256) C->red[i] = 0.f;
    if (t < 12) C->cam_acc[t] = 0.f;
    if (t < MAX_LEVELS) { C->maxinit_bits[t] = 0u; C->cnt_level[t] = 0; }
    if (t == 0) {
      float c[3];
#pragma unroll
      for (int i = 0; i < 9; ++i) C->R[i] = R[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) C->T[i] = T[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) { c[i] = -(R[0 * 3 + i] * T[0] + R[1 * 3 + i] * T[1] + R[2 * 3 + i] * T[2]); C->c[i] = c[i]; }
      const float cd = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      C->cdist = cd;
      C->vflags = vf.f[vb];
      C->f_origin = 0.f;
      C->origin_done = 0;
      C->xchg_err = 0;
      C->f16_overflow = 0;
      C->tail_from = V0.tail_from;
      C->tail_steals = 0;
      C->cnt_valid = 0; C->cnt_normal = 0; C->cnt_samples = 0; C->pad_coef = 0.f;
    }
  }
}

// latent constants only (decode_sdf / decode_sdf_gradient entry points)
DISTR_GLOBAL void __launch_bounds__(256) k_latent_consts(float* c0c4 /*[1024]*/, DecoderDev D, const float* __restrict__ latent) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  const int o = gid & 511;
  const float* Wt = (gid < 512) ? D.W0lat_t : D.W4lat_t;
  float acc = (gid < 512) ? D.b0[o] : D.b4[o];
#pragma unroll 8
  for (int k = 0; k < D.nlat; ++k) acc = __builtin_fmaf(Wt[k * HID + o], latent[k], acc);
  c0c4[gid] = acc;
}

// ------------------------------------------------------------------------------------------ ray setup
// get_intersections_with_unit_spheres (renderer.py:254-273) for one pyramid level; coarse masks are the OR of the
// rdiv x rdiv children (maxpool_valid_mask_with_index / torch_scatter.scatter_max, renderer.py:668-680).
DISTR_GLOBAL void __launch_bounds__(256) k_setup_level(View V0, int lvl) {
  const View V = view_at(V0, blockIdx.y);
  const LevelView L = level_sel(V, lvl);
  Consts* C = V.C;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const CamRegs cam = load_cam(C);
  const bool inside = cam.cdist < V.cfg.radius;
  bool valid = false;
  uint32_t mx = 0u;
  if (i < L.n) {
    float px, py;
    level_center(L, i, px, py);
    const RayGeo g = make_ray(V.cfg.K_inv, cam.R, px, py);
    const Sph s = intersect(V.cfg.radius, cam.c, cam.cdist, g.d);
    if (lvl == 0) {
      valid = s.in;
    } else {
      const LevelView F = level_sel(V, lvl - 1);
      const int y = i / L.w, x = i % L.w, r = L.rdiv;
      for (int dy = 0; dy < r; ++dy)
        for (int dx = 0; dx < r; ++dx) {
          const int fy = r * y + dy, fx = r * x + dx;
          if (fy < F.h && fx < F.w) valid = valid || (F.valid[fy * F.w + fx] != 0);
        }
    }
    L.valid[i] = valid ? 1 : 0;
    if (!V.band && s.in && !inside) mx = __float_as_uint(s.init_raw);
  }
  wave_atomic_max(&C->maxinit_bits[lvl], mx);      // positive floats order like their bit patterns; one atomic per wavefront
  block_append(valid, i, L.list, &C->cnt_level[lvl]);
}

// row band: the fill depth of rays that miss the sphere is the maximum over the FULL image's level grid
// (renderer.py:268-270), not over the band -> one cheap pass over all of the level's pixel centres
DISTR_GLOBAL void __launch_bounds__(256) k_maxinit_full(View V0, int lvl) {
  const View V = view_at(V0, blockIdx.y);
  const LevelView L = level_sel(V, lvl);
  const int i = blockIdx.x * 256 + threadIdx.x;
  const CamRegs cam = load_cam(V.C);
  uint32_t mx = 0u;
  if (i < L.full_h * L.w && !(cam.cdist < V.cfg.radius)) {
    const float px = L.scale * (float)(i % L.w) + L.off, py = L.scale * (float)(i / L.w) + L.off;
    const RayGeo g = make_ray(V.cfg.K_inv, cam.R, px, py);
    const Sph s = intersect(V.cfg.radius, cam.c, cam.cdist, g.d);
    if (s.in) mx = __float_as_uint(s.init_raw);
  }
  wave_atomic_max(&V.C->maxinit_bits[lvl], mx);
}

// start depth of a coarse level: unit-sphere entry (coarsest) or the parent's last marched depth (renderer.py:766-769)
DISTR_GLOBAL void __launch_bounds__(256) k_coarse_init(View V0, int lvl) {
  const View V = view_at(V0, blockIdx.y);
  const LevelView L = level_sel(V, lvl);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= L.n) return;
  float init;
  if (lvl == V.nlev - 1) {
    const CamRegs cam = load_cam(V.C);
    float px, py;
    level_center(L, i, px, py);
    const RayGeo g = make_ray(V.cfg.K_inv, cam.R, px, py);
    const Sph s = intersect(V.cfg.radius, cam.c, cam.cdist, g.d);
    const bool inside = cam.cdist < V.cfg.radius;
    init = inside ? 0.f : (s.in ? s.init_raw : __uint_as_float(V.C->maxinit_bits[lvl]));
  } else {
    const LevelView Pp = level_sel(V, lvl + 1);
    const int par = ((i / L.w) / Pp.rdiv) * Pp.w + ((i % L.w) / Pp.rdiv);
    init = Pp.rza[(size_t)(Pp.steps - 1) * Pp.n + par];
  }
  L.cinit[i] = init;
  L.cm[i] = 0.f;
}

// selected-row buffer: bs entries sorted by |sdf| ascending, earlier row wins ties (renderer.py:314-318 topk)
__device__ __forceinline__ int topk_insert(const View& V, int px, float s, float zb, float za, int32_t src) {
  const int bs = V.cfg.buffer_size;
  const size_t P = (size_t)V.P;
  const float key = fabsf(s);
  int pos = bs;
  for (int k = bs - 1; k >= 0; --k) {
    const float sk = V.tk_s[k * P + px];
    if (key < fabsf(sk)) pos = k; else break;
  }
  if (pos >= bs) return -1;
  // the new row takes the free mask slot (the one of {0..bs} no entry uses); the evicted entry's slot becomes free
  int used = 0;
  for (int k = 0; k < bs; ++k) used += V.tk_slot[k * P + px];
  const int free_slot = bs * (bs + 1) / 2 - used;
  for (int k = bs - 1; k > pos; --k) {
    V.tk_s[k * P + px] = V.tk_s[(k - 1) * P + px];
    V.tk_zb[k * P + px] = V.tk_zb[(k - 1) * P + px];
    V.tk_za[k * P + px] = V.tk_za[(k - 1) * P + px];
    V.tk_src[k * P + px] = V.tk_src[(k - 1) * P + px];
    V.tk_slot[k * P + px] = V.tk_slot[(k - 1) * P + px];
  }
  V.tk_s[pos * P + px] = s;
  V.tk_zb[pos * P + px] = zb;
  V.tk_za[pos * P + px] = za;
  V.tk_src[pos * P + px] = src;
  V.tk_slot[pos * P + px] = free_slot;
  return free_slot;
}

// Per-ray march state requested in a tile's PROLOGUE (before the decoder runs) so that the epilogue needs no dependent
// global loads: with one workgroup per CU nothing else hides that latency (~5 us of a 380 us tile, more of a tail tile).
struct RayPre {
  float m, init_now, maxbound, minabs;
  float ks[MAX_BS];      // selected rows' sdf, |.| ascending
  int32_t sl[MAX_BS];    // their mask slots
};

// Per-ray arrays are addressed as (uniform base + k * P)[u] with an UNSIGNED 32-bit pixel index: the row base is scalar arithmetic and the
// access takes the "SGPR base + 32-bit VGPR offset" form. Written as base[k * P + px] with a 64-bit P the whole sum is 64-bit VECTOR
// arithmetic on a VGPR copy of the base -- inside the step loop of a sticky tile those copies were hoisted, kept across the decoder
// evaluation and spilled: scratch reloads with a vmcnt(0) each in the lead member's epilogue, the critical path of every sticky step.
template <class T>
__device__ __forceinline__ T* ray_row(T* base, int k, int32_t P) { return base + (size_t)k * (size_t)P; }

template <bool XC = false>
__device__ __forceinline__ void raypre_load(const View& V, int px, RayPre& st) {
  const uint32_t u = (uint32_t)px;
  const int bs = V.cfg.buffer_size;
  st.m = ld_x<XC>(V.m + u); st.init_now = V.init_now[u]; st.maxbound = V.maxbound[u]; st.minabs = ld_x<XC>(V.minabs + u);   // (init_now / maxbound: written once, by k_fine_init)
#pragma unroll
  for (int k = 0; k < MAX_BS; ++k) {
    st.ks[k] = (k < bs) ? ld_x<XC>(ray_row(V.tk_s, k, V.P) + u) : 0.f;
    st.sl[k] = (k < bs) ? ld_x<XC>(ray_row(V.tk_slot, k, V.P) + u) : 0;
  }
}

// topk_insert with the keys / slots already in registers (same result, same memory image)
template <bool XC = false>
__device__ __forceinline__ int topk_insert_pre(const View& V, const RayPre& st, int px, float s, float zb, float za, int32_t src) {
  const int bs = V.cfg.buffer_size;
  const uint32_t u = (uint32_t)px;
  const float key = fabsf(s);
  int pos = bs;
  bool open = true;
#pragma unroll
  for (int k = MAX_BS - 1; k >= 0; --k) {
    if (k < bs && open) { if (key < fabsf(st.ks[k])) pos = k; else open = false; }
  }
  if (pos >= bs) return -1;
  int used = 0;
#pragma unroll
  for (int k = 0; k < MAX_BS; ++k) used += (k < bs) ? st.sl[k] : 0;
  const int free_slot = bs * (bs + 1) / 2 - used;
#pragma unroll
  for (int k = MAX_BS - 1; k >= 1; --k) {
    if (k < bs && k > pos) {
      st_x<XC>(ray_row(V.tk_s, k, V.P) + u, st.ks[k - 1]);
      st_x<XC>(ray_row(V.tk_slot, k, V.P) + u, st.sl[k - 1]);
      st_x<XC>(ray_row(V.tk_zb, k, V.P) + u, ld_x<XC>(ray_row(V.tk_zb, k - 1, V.P) + u));
      st_x<XC>(ray_row(V.tk_za, k, V.P) + u, ld_x<XC>(ray_row(V.tk_za, k - 1, V.P) + u));
      st_x<XC>(ray_row(V.tk_src, k, V.P) + u, ld_x<XC>(ray_row(V.tk_src, k - 1, V.P) + u));
    }
  }
  // (the new row's place is data: one of bs unrolled, uniformly based stores)
#pragma unroll
  for (int k = 0; k < MAX_BS; ++k) {
    if (k < bs && k == pos) {
      st_x<XC>(ray_row(V.tk_s, k, V.P) + u, s);
      st_x<XC>(ray_row(V.tk_zb, k, V.P) + u, zb);
      st_x<XC>(ray_row(V.tk_za, k, V.P) + u, za);
      st_x<XC>(ray_row(V.tk_src, k, V.P) + u, src);
      st_x<XC>(ray_row(V.tk_slot, k, V.P) + u, (int32_t)free_slot);
    }
  }
  return free_slot;
}

// ... the slot only (what topk_insert_pre returns), nothing written: the members of a sticky cluster tile that do not lead mirror the
// lead's insertion on their copy of the keys to know where this step's mask block goes
__device__ __forceinline__ int topk_slot_pre(const View& V, const RayPre& st, float s) {
  const int bs = V.cfg.buffer_size;
  const float key = fabsf(s);
  int pos = bs;
  bool open = true;
#pragma unroll
  for (int k = MAX_BS - 1; k >= 0; --k) {
    if (k < bs && open) { if (key < fabsf(st.ks[k])) pos = k; else open = false; }
  }
  if (pos >= bs) return -1;
  int used = 0;
#pragma unroll
  for (int k = 0; k < MAX_BS; ++k) used += (k < bs) ? st.sl[k] : 0;
  return bs * (bs + 1) / 2 - used;
}

// The early break of the recursive march (renderer.py:562-567): when no ray is unfinished after L < buffer_size full-resolution steps, the
// reference pads its lists to buffer_size rows by REPEATING step L-1's rows -- sdf, point and depth of every ray, real rows of the rays that
// were evaluated at that step included -- and the selection (bs smallest |sdf|, earlier row wins ties) then takes copies of a ray's last row
// where this library's selected-row buffer holds the rows behind it (usually pad rows). Values do not change (top-1 is the original), the
// GRADIENT does: the copies are evaluated again, each carries the row's coefficient. Emulated on the buffer: with the last row at sorted
// position p, n = min(bs - L, bs - 1 - p) copies follow it and the last n rows of the buffer drop out. Returns bs - L (0: no early break
// below buffer_size steps, or not applicable) and L. Not applied to row bands: the break is a property of the WHOLE image's march, which a
// band cannot see. Found by the random options soak of round 6 (cameras inside the sphere, buffer_size 5..8).
__device__ __forceinline__ int early_dup(const View& V, const Consts* C, int& L) {
  L = 0;
  if (V.cfg.marcher == DISTR_MARCH_TRIVIAL || V.band) return 0;
  const int bs = V.cfg.buffer_size;
  for (int t = 0; t < bs && t < V.fine_steps; ++t)
    if (C->cnt_live[t] + C->cnt_sticky[t] == 0) { L = t; return t >= 1 ? bs - t : 0; }
  return 0;
}
// ... for one pixel: position of the duplicated row in its selected-row buffer (-1: none) and the number of copies selected
__device__ __forceinline__ void early_dup_px(const View& V, const Consts* C, int px, int& p, int& n) {
  p = -1; n = 0;
  int L;
  const int dup = early_dup(V, C, L);
  if (dup <= 0) return;
  const int bs = V.cfg.buffer_size;
  const size_t P = (size_t)V.P;
  for (int k = 0; k < bs; ++k) {
    const int32_t src = V.tk_src[k * P + px];
    if (src >= 0 && src_level(src) == 0 && src == L - 1) { p = k; break; }
  }
  if (p >= 0) n = (dup < bs - 1 - p) ? dup : (bs - 1 - p);
}

// per-ray state at the start of the full-resolution march (renderer.py:521-527, 795-804)
DISTR_GLOBAL void __launch_bounds__(256) k_fine_init(View V0) {
  const View V = view_at(V0, blockIdx.y);
  const LevelView& L0 = V.lv[0];
  Consts* C = V.C;
  const int px = blockIdx.x * 256 + threadIdx.x;
  bool live = false;
  if (px < V.P / 16 + 2) V.tclaim[px] = 0;
  if (px < V.P && L0.valid[px]) {
    const CamRegs cam = load_cam(C);
    float cx, cy;
    level_center(L0, px, cx, cy);
    const RayGeo g = make_ray(V.cfg.K_inv, cam.R, cx, cy);
    const Sph s = intersect(V.cfg.radius, cam.c, cam.cdist, g.d);
    const bool inside = cam.cdist < V.cfg.radius;
    const float init_orig = inside ? 0.f : s.init_raw;
    const float maxbound = init_orig + s.chord;
    float init_now = init_orig;
    const size_t P = (size_t)V.P;
    const int y = px / L0.w, x = px % L0.w;
    if (V.pyramid) {
      const LevelView& L1 = V.lv[1];
      const int par1 = (y / L1.rdiv) * L1.w + (x / L1.rdiv);
      init_now = L1.rza[(size_t)(L1.steps - 1) * L1.n + par1];
    }
    for (int k = 0; k < V.cfg.buffer_size; ++k) {
      V.tk_s[k * P + px] = 1.0f;
      V.tk_zb[k * P + px] = 0.f;
      V.tk_za[k * P + px] = V.pyramid ? init_now : 0.f;
      V.tk_src[k * P + px] = -1;
      V.tk_slot[k * P + px] = k;
    }
    if (V.pyramid) {
      for (int lvl = V.nlev - 1; lvl >= 1; --lvl) {
        const LevelView Lc = level_sel(V, lvl);
        const int par = (y / Lc.sdiv) * Lc.w + (x / Lc.sdiv);
        for (int st = 0; st < Lc.steps; ++st) {
          const size_t o = (size_t)st * Lc.n + par;
          topk_insert(V, px, Lc.rs[o], Lc.rzb[o], Lc.rza[o], src_coarse(lvl, st, par));
        }
      }
    }
    V.m[px] = 0.f;
    V.init_now[px] = init_now;
    V.maxbound[px] = maxbound;
    V.minabs[px] = INFINITY;
    V.first_sdf[px] = 1.0f;
    live = (V.cfg.marcher == DISTR_MARCH_TRIVIAL) ? true : ((0.f + init_now) < maxbound);
  }
  if (V.cfg.marcher != DISTR_MARCH_TRIVIAL) block_append(live, px, V.live[0], &C->cnt_live[0]);
}

// ------------------------------------------------------------------------------------------ the march kernel
// One launch = one marching step over a compacted list of rays: gather ray state -> sample point -> fused 9-layer
// decoder (distr_mlp.hpp) -> aggressive step update, online min-|sdf| selection, convergence/escape masking and
// ballot compaction of the rays that stay live (renderer.py:481-494, 528-567). Launched with the worst-case grid;
// workgroups beyond the device-side count exit immediately, so the host never synchronises inside the loop.
enum { MODE_EVAL = 0, MODE_COARSE = 1, MODE_FINE = 2 };

// A march step over the live rays (of all views of the batch) is split by tile size so that no launch pays a full 64-ray tile
// latency for a small remainder. n16 / n64 = the step's live rays in the virtual concatenation of the views, every view's count
// rounded up to 16 / 64 (vprefix). If n16 <= t16 (4096) the whole step runs on 16-ray tiles (one wave of 107 us tiles; cluster
// tiles below 2048 rays), granularity g = 16. Otherwise g = 64: rays [0, full), full = floor(n64 / 16384) * 16384 (whole rounds
// of 256 CUs x 64 rays), go to the 64-ray role and the remainder `rem`, by size, to
//     rem <= t16 (4096)          16-ray tiles
//     rem <= t32 (8192)          32-ray tiles          (203 us)
//     rem <= t32 + t16 (12288)   32-ray tiles for the first t32 rays + 16-ray tiles for the rest (203 + 111 us on the same CUs:
//                                the work is MFMA-bound, so 48 rays per CU cost 48/64 of a round whether the two tiles share
//                                the CU or follow each other)
//     else                       one more round of 64-ray tiles (362 us).
// Every role of the step evaluates this on the device-side counts; the host never needs to know them. (Host invariants,
// distr_create: 16 <= t16 <= t32, t16 + t32 < 16384, both multiples of 64.)
__device__ __forceinline__ void fine_split(int64_t n16, int64_t n64, int t16, int t32, int which, int64_t& lo, int64_t& hi, int& g) {
  if (n16 <= t16) { g = 16; lo = 0; hi = (which == 16) ? n16 : 0; return; }
  g = 64;
  const int64_t count = n64;
  const int64_t full = (count / 16384) * 16384, rem = count - full;
  // remainder split point: rays [full, cut) on 32-ray tiles, [cut, count) on 16-ray tiles (either part may be empty)
  int64_t cut;
  bool big = false;
  if (rem <= t16) cut = full;
  else if (rem <= t32) cut = count;
  else if (t16 < t32 && rem <= (int64_t)t32 + t16) cut = full + t32;
  else { cut = full; big = true; }                                                 // whole remainder on 64-ray tiles
  if (which == 64) { lo = 0; hi = big ? count : full; }
  else if (which == 32) { lo = full; hi = big ? full : cut; }
  else { lo = big ? count : cut; hi = count; }
}

struct MarchArgs {
  View V;
  int32_t lvl, step;
  int32_t origin_tile;       // the last nviews workgroups evaluate f(origin) of their view instead of a tile
  const float* xyz;          // MODE_EVAL
  float* sdf_out;
  const float* c0c4;         // MODE_EVAL: latent constants
  int64_t n;
  float clamp;
  // MODE_FINE tile-size split of the recursive marchers (see fine_split): t16 / t32 = largest remainder handled by 16- /
  // 32-ray tiles; which = tile size of THIS launch (a plain k_march launch; the roles of k_step pass their own)
  int32_t t16, t32, which;
  Xchg xc;                   // 16-ray launches: exchange region of the cluster tiles (buf == null: single-workgroup tiles only)
  DecoderB6 B6;              // split-bf16 weight planes (kernels instantiated with ARITH = 1, distr_render_cfg.arith)
  DecoderH3 H3;              // split-f16 weight planes (ARITH = 2)
  int32_t tail_absent;       // tests (DISTR_TAIL_TEST_ABSENT=n): workgroups 0 .. n-1 of k_tail leave at once, as if they never became resident
};

// One pyramid level of view b, read from the kernel-argument segment (MarchArgs is the first argument of every march kernel): a scalar
// load of the SELECTED level. level_sel on a register-resident View keeps all MAX_LEVELS x 17 members alive and selects among them --
// with the fourth level that was 150-240 spilled SGPRs (v_writelane / v_readlane next to the MFMAs) in the coarse launches.
__device__ __forceinline__ LevelView level_at(int l, int b) {
  const View& V0 = kernarg_ref<MarchArgs>(0).V;
  LevelView L = (&V0.lv[0])[l];
  if (V0.nviews > 1) {
    const int64_t d = (int64_t)b * V0.vstride;
    adv(L.valid, d); adv(L.list, d); adv(L.cinit, d); adv(L.cm, d); adv(L.rs, d); adv(L.rzb, d); adv(L.rza, d);
  }
  return L;
}
__device__ __forceinline__ int64_t moff_at(int l) { return (&kernarg_ref<MarchArgs>(0).V.moff[0])[l]; }

// KEEP: also save the ReLU masks of every row that enters a ray's selected-row buffer (and of every coarse row), so
// that the backward pass does not have to recompute the decoder forward (View::mstore).
// Body of one 32*RB-ray tile; `tile` / `ntile_grid` = index and count of the tiles this launch (or this role of a merged
// launch, k_step) provides, `which` = the tile size the split rule (fine_range) knows this role by.
// Returns false when the tile lies beyond this role's range (nothing done).
// ARITH = 0: exact f32 MFMA tile (distr_mlp.hpp); 1: six-product split-bf16 tile (distr_mlp_b6.hpp); 2: three-product split-f16
// tile (distr_mlp_h3.hpp)   (distr_render_cfg.arith)
template <int RB, int ARITH> struct TileSmem { using type = Smem<RB>; };
template <int RB> struct TileSmem<RB, 1> { using type = SmemB6<RB>; };
template <int RB> struct TileSmem<RB, 2> { using type = SmemH3<RB>; };

template <int MODE, int RB, bool KEEP, int ARITH = 0>
__device__ __forceinline__ bool march_tile(const MarchArgs& A, const DecoderDev& D, typename TileSmem<RB, ARITH>::type& S, int tile, int ntile_grid,
                                           int which, int origin_tile) {
  constexpr int TILE = 32 * RB;
  const View& V0 = A.V;
  const int tid = threadIdx.x;
  const int B = (MODE == MODE_EVAL) ? 1 : V0.nviews;
  const bool split = MODE == MODE_FINE && V0.cfg.marcher != DISTR_MARCH_TRIVIAL;
  bool origin = false;
  int vb = 0;
  int64_t base = (int64_t)tile * TILE, count = 0;
  if (MODE == MODE_EVAL) {
    count = A.n;
    if (base >= count) return false;
  } else if (origin_tile && tile >= ntile_grid - B) {
    origin = true;
    vb = tile - (ntile_grid - B);
  } else {
    // this tile's view and ray range from the device-side counts of all views (virtual concatenation, see vprefix)
    const int32_t* p0 = (MODE == MODE_COARSE) ? &V0.C->cnt_level[A.lvl] : split ? &V0.C->cnt_live[A.step] : &V0.C->cnt_level[0];
    const int32_t c = vload(p0, V0.vstride, B);
    int64_t lo = 0, hi;
    int g = TILE;
    int32_t incl;
    if (split) {
      const int32_t i16 = vprefix(c, B, 16), i64 = vprefix(c, B, 64);
      fine_split(vtotal(i16, B), vtotal(i64, B), A.t16, A.t32, which, lo, hi, g);
      incl = (g == 16) ? i16 : i64;
    } else {
      incl = vprefix(c, B, TILE);
      hi = vtotal(incl, B);
    }
    const int64_t vbase = lo + (int64_t)tile * TILE;
    if (vbase >= hi) return false;
    int64_t start;
    int32_t cnt;
    vfind(c, incl, B, g, vbase, vb, start, cnt);
    base = vbase - start;
    count = cnt;
    if (base >= count) return true;    // padding behind the view's last ray (tiles smaller than the granularity): nothing to do
  }
  const View V = view_at(V0, vb);
  const int32_t* list = nullptr;
  if (MODE == MODE_COARSE) list = level_at(A.lvl, vb).list;
  else if (MODE == MODE_FINE) list = split ? live_sel(V, A.step) : V.lv[0].list;
  const float* c0 = (MODE == MODE_EVAL) ? A.c0c4 : V.C->c0;
  const float* c4 = (MODE == MODE_EVAL) ? A.c0c4 + HID : V.C->c4;
  if constexpr (ARITH == 0) stage_bias<RB>(D, c0, c4, S);      // first thing: these loads travel under the prologue's dependent state loads

  int32_t id = -1;
  float zd = 0.f;
  bool valid = false;
  RayPre st;
  if (tid < TILE) {
    float p[3] = {0.f, 0.f, 0.f};
    if (!origin) {
      const int64_t r = base + tid;
      valid = r < count;
      if (valid) {
        if (MODE == MODE_EVAL) {
          id = (int32_t)r;
          p[0] = A.xyz[r * 3]; p[1] = A.xyz[r * 3 + 1]; p[2] = A.xyz[r * 3 + 2];
        } else {
          id = list[r];
          const LevelView L = (MODE == MODE_COARSE) ? level_at(A.lvl, vb) : V.lv[0];
          const CamRegs cam = load_cam(V.C);
          float cx, cy;
          level_center(L, id, cx, cy);
          const RayGeo g = make_ray(V.cfg.K_inv, cam.R, cx, cy);
          if (MODE == MODE_FINE) { raypre_load(V, id, st); zd = st.init_now + st.m; }
          else zd = L.cinit[id] + L.cm[id];
          make_point(V.cfg.M, cam.c, g.d, zd, p);
        }
      }
    }
    S.xyz[tid] = p[0]; S.xyz[TILE + tid] = p[1]; S.xyz[2 * TILE + tid] = p[2];
  }
  __syncthreads();

  uint32_t masks[8][4];
  float pre;
  if constexpr (ARITH == 0) pre = mlp_forward<RB, KEEP, false, true>(D, c0, c4, S, masks);
  else if constexpr (ARITH == 1) pre = mlp_forward_b6<RB, KEEP>(D, A.B6, c0, c4, S, masks);
  else pre = mlp_forward_h3<RB, KEEP>(D, A.H3, c0, c4, S, masks);

  // epilogue: wave 0 (kept whole for the ballot), lane = ray of the tile; lanes >= TILE are invalid
  int64_t mblock = -1;   // mask block this ray's row goes to (KEEP), -1: row not kept
  if (tid < 64) {
    const float s = tanh_spec(pre);
    if constexpr (ARITH == 2 && MODE != MODE_EVAL) {
      // an activation left the f16 range (non-finite result): counted, not hidden by the clamps below (one wave-uniform test per tile)
      const unsigned long long bad = __ballot((valid || origin) && !(fabsf(pre) <= 3.0e38f));
      if (bad != 0ull && tid == 0) atomicAdd(&V.C->f16_overflow, (int)__popcll(bad));
    }
    if (origin) {
      if (tid == 0) V.C->f_origin = s;
    } else if (MODE == MODE_EVAL) {
      if (valid) A.sdf_out[id] = (A.clamp >= 0.f) ? clampf(s, -A.clamp, A.clamp) : s;
    } else {
      const float cd = V.cfg.clamp_dist, ratio = V.cfg.ratio;
      if (MODE == MODE_COARSE) {
        if (valid) {
          const LevelView L = level_at(A.lvl, vb);
          const float mn = L.cm[id] + clampf(s, -cd, cd) * ratio;
          L.cm[id] = mn;
          const size_t o = (size_t)A.step * L.n + id;
          L.rs[o] = s;
          L.rzb[o] = zd;
          L.rza[o] = mn + L.cinit[id];
          mblock = V.mfine + moff_at(A.lvl) + (int64_t)o;
        }
      } else {  // MODE_FINE
        bool stay = false;
        if (valid) {
          const float mn = st.m + clampf(s, -cd, cd) * ratio;
          V.m[(uint32_t)id] = mn;
          const float za = mn + st.init_now;
          const int slot = topk_insert_pre(V, st, id, s, zd, V.pyramid ? za : mn, A.step);  // src: level 0 | march step (early_dup)
          if (slot >= 0) mblock = (int64_t)id * (V.cfg.buffer_size + 1) + slot;
          const float a = fabsf(s);
          if (a < st.minabs) V.minabs[(uint32_t)id] = a;
          if (A.step == 0) V.first_sdf[(uint32_t)id] = s;
          stay = (za < st.maxbound) && (a >= V.cfg.threshold);
        }
        if (V.cfg.marcher != DISTR_MARCH_TRIVIAL)
          wave_append(stay, id, live_sel(V, A.step + 1), &V.C->cnt_live[A.step + 1]);
      }
    }
  }
  if (KEEP && MODE != MODE_EVAL) {
    // hand every wave the mask-block index of each ray, then each lane stores the 64-byte chunks it owns
    long long* mb = reinterpret_cast<long long*>(S.aux);    // 8*TILE bytes; aux is not used by the forward tile
    if (tid < TILE) mb[tid] = origin ? (tid == 0 ? (long long)V.morigin : -1ll) : mblock;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const long long b = mb[32 * rb + (lane & 31)];
      if (b >= 0) store_mask_chunk<RB>(V.mstore + (size_t)b * 32, masks, rb, wave, lane >> 5);
    }
  }
  return true;
}

template <int MODE, int RB, bool KEEP, int ARITH = 0>
__global__ void __launch_bounds__(256, (RB == 1 && ARITH == 0) ? 2 : 1) k_march(MarchArgs A, DecoderDev D) {
  __shared__ typename TileSmem<RB, ARITH>::type S;
  (void)march_tile<MODE, RB, KEEP, ARITH>(A, D, S, (int)blockIdx.x, (int)gridDim.x, A.which, A.origin_tile);
}

// KEEP, cluster tile of CL members: this member's share of the rays' mask blocks. Every member recorded the ReLU bits of the rows IT
// computed in its S.mk (mlp_forward16_cl, MASK_OWN; zero elsewhere). In store_mask_chunk's format the bits of 64 consecutive rows of a
// 512-row layer are one 32-bit word per half h -- word 32 (g / 2) + 2 layer + (g & 1) + 16 h for row group g = 0..7 -- and a member owns
// 8 / CL whole groups; for lin3 (256 rows) four row blocks are word 32 w + 6 + 16 h (w = 0..3, the words 32 w + 7 + 16 h stay zero): a
// member of 4 or 2 owns whole words, two members of 8 share one (16 bits each). Thread (ray, layer, h) stores the member's words: the
// members together write every byte of the 512-byte block, nobody gathers. mb[ray] = index of the ray's block, < 0: none.
// WT (k_tail): write-through stores. Inside the persistent tail launch the same block can be rewritten a few steps later by a workgroup on
// ANOTHER XCD (the row was evicted, its slot reused); a helper member's plain stores are released by nobody before that (only a tile's lead
// member releases, k_tail), and a line left dirty in this XCD's L2 could be written back over the newer one. Write-through leaves none.
template <bool WT, class T>
__device__ __forceinline__ void st_mask(T* p, T v) {
  if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <int CL, bool WT = false>
__device__ __forceinline__ void store_own_mask_words(uint4* mstore, const long long* mb, const Smem16CL& S, int member, int tid) {
  const int j = tid >> 4, q = tid & 15;
  const long long b = mb[j];
  if (b < 0) return;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(&S.mk[j][0]);
  uint32_t* dst = reinterpret_cast<uint32_t*>(mstore + (size_t)b * 32);
  const int layer = q >> 1, h = q & 1;
  if (layer != 3) {
    constexpr int GPM = 8 / CL;
#pragma unroll
    for (int gi = 0; gi < GPM; ++gi) {
      const int g = member * GPM + gi, wi = 32 * (g >> 1) + 2 * layer + (g & 1) + 16 * h;
      st_mask<WT>(dst + wi, src[wi]);
    }
  } else if constexpr (CL == 8) {
    const int wi = 32 * (member >> 1) + 6 + 16 * h;
    st_mask<WT>(reinterpret_cast<uint16_t*>(dst + wi) + (member & 1), reinterpret_cast<const uint16_t*>(src + wi)[member & 1]);
    if (member & 1) st_mask<WT>(dst + wi + 1, 0u);
  } else {
    constexpr int WPM = 4 / CL;
#pragma unroll
    for (int wq = 0; wq < WPM; ++wq) {
      const int wi = 32 * (member * WPM + wq) + 6 + 16 * h;
      st_mask<WT>(dst + wi, src[wi]);
      st_mask<WT>(dst + wi + 1, 0u);
    }
  }
}

// kernel-argument offsets of the march kernels that take (MarchArgs, DecoderDev, DecoderDev16, ...): k_march16, k_step, k_tail
constexpr size_t KERNARG_OFF_D = (sizeof(MarchArgs) + alignof(DecoderDev) - 1) / alignof(DecoderDev) * alignof(DecoderDev);
constexpr size_t KERNARG_OFF_D16 = (KERNARG_OFF_D + sizeof(DecoderDev) + alignof(DecoderDev16) - 1) / alignof(DecoderDev16) * alignof(DecoderDev16);

// Sticky tail tile (renderer.py:528-567 from the point where few rays are left): once ALL live rays of a step fit the cluster
// tiles of one launch (<= 32 tiles of 16 rays, 8 compute units each), every tile marches ITS 16 rays through all remaining steps
// inside that launch -- no compaction, no launch boundary, no device-wide barrier: tiles never exchange anything. The ray state
// (march depth, bounds, selected-row keys) lives in registers of wave 0; after the layer-7 exchange every member holds h7, so
// every member computes lin8 and mirrors the depth update and the live test itself (identical arithmetic, nothing to
// broadcast); only the lead member writes the step's results (depths, selected rows, counters) -- with saved masks every member stores
// its own words of the rays' mask blocks (store_own_mask_words). Finished rays stay in the tile as dead lanes; the tile ends when none is
// live. A ray's arithmetic is exactly the per-step one (same points, same k-ordered chains) -> bit-identical outputs. The step
// launches the host still issues find empty live lists and exit in a few microseconds. A tile whose cluster does not assemble
// (compute units held by another stream) is evaluated by its lead member alone and hands its rays back to the next step's live
// list; if a barrier times out later, the lead member finishes the tile alone on the single-workgroup path.
// TAIL (k_tail: no launch follows): a tile whose cluster does not assemble is marched to the END by its lead member alone (no list to hand
// the rays back to); solo0: evaluated alone from the first step (a workgroup that took the tile over from a lead that is not resident).
template <bool KEEP, bool TAIL = false>
__device__ __forceinline__ void sticky_tile16(const MarchArgs& A, const DecoderDev& D, const DecoderDev16& D16, Smem16CLX& S, int vb,
                                              int tile, int member, int64_t base, int64_t count, int step0, const Xchg& xc, bool solo0 = false, int32_t lost = 0) {
  constexpr int TILE = 16;
  const int tid = threadIdx.x;
  const bool lead = member == cl_lead(8);
  const View V = view_at(A.V, vb);         // (setup and the per-step sample points; the epilogue of a step takes its own, see there)
  const int32_t* list = live_sel(V, step0);
  const float* c0 = V.C->c0;
  const float* c4 = V.C->c4;
  int32_t id = -1;
  bool live = false;
  float m = 0.f, init_now = 0.f, maxbound = 0.f, minabs = 0.f;      // march state of this lane's ray, in registers across steps
  if (tid < TILE && base + tid < count) {
    id = ld_x<false>(list + base + tid);
    live = true;
    RayPre st;
    raypre_load<false>(V, id, st);
    m = st.m; init_now = st.init_now; maxbound = st.maxbound; minabs = st.minabs;
#pragma unroll
    for (int k = 0; k < MAX_BS; ++k) { S.sk[k][tid] = st.ks[k]; S.ssl[k][tid] = st.sl[k]; }   // selected-row keys / slots: LDS between steps
  }
  if constexpr (TAIL) {   // lin0's operands into LDS, once for all of the tile's steps (Smem16CLX; visible behind the first step's barrier). Uniform bases
      // re-read from the kernel arguments + 32-bit lane offsets: no per-lane 64-bit address lives across the caller's job loop
    const f32x4* wb = reinterpret_cast<const f32x4*>(kernarg_ref<DecoderDev16>(KERNARG_OFF_D16).Wf[0]);
    const float* cb = view_at(kernarg_ref<MarchArgs>(0).V, vb).C->c0;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, t32 = tid;
#pragma unroll
    for (uint32_t ob = 0; ob < 8; ++ob) S.w0s[w][ob][ln] = wb[w * 512u + ob * 64u + ln];
    S.c0s[t32] = cb[t32];
    S.c0s[t32 + NTHREADS] = cb[t32 + (uint32_t)NTHREADS];
  }
  const float cd = V.cfg.clamp_dist, ratio = V.cfg.ratio;
  bool solo = solo0;                  // the cluster broke up: the lead member finishes the tile alone
  bool leaving = false;               // a member other than the lead that gave up after `go`: one evaluation of its own for the mask blocks, then out
  for (int step = step0, k = 0;; ++step, ++k) {
    if (tid < TILE) {
      float p[3] = {0.f, 0.f, 0.f};
      if (live) {        // (the ray direction is recomputed from the pixel id every step, like the per-step kernels do: ~40 flops
        const View& Vp = kernarg_ref<MarchArgs>(0).V;      // against three registers held across the decoder evaluation); camera constants re-read per step
        const CamRegs cam = load_cam(view_at(Vp, vb).C);
        float cx, cy;
        level_center(Vp.lv[0], id, cx, cy);
        const RayGeo g = make_ray(Vp.cfg.K_inv, cam.R, cx, cy);
        make_point(Vp.cfg.M, cam.c, g.d, init_now + m, p);
      }
      S.xyz[tid] = p[0]; S.xyz[TILE + tid] = p[1]; S.xyz[2 * TILE + tid] = p[2];
    }
    __syncthreads();
    uint32_t nib[8];
    float pre = 0.f;
    bool clustered = false;
    // a zero the optimiser cannot see through, added to the tile / member index: the weight-stream addresses of the decoder
    // evaluation then depend on the loop iteration and are NOT hoisted out of the step loop (hoisted, they stay live across
    // the whole loop body next to the 128-register weight ring and push the kernel into scratch)
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    if (!solo) {
      Xchg xs = xc;
      xs.epoch = xc.epoch + (uint32_t)k;          // one epoch per march step (the host reserved them: Xchg::epochs)
      xs.par = (xc.par + k) & 1;                   // alternate the exchange slots: layer 1 of step k+1 must not reuse layer 7's
      pre = mlp_forward16_cl<8, KEEP, true, true>(D, D16, c0 + zero, c4 + zero, S, xs, tile + zero, member + zero, k == 0, (TAIL && k == 0) ? lost : 0, TAIL);
      clustered = S.fail == 0;
      if (!clustered) {
        if (!lead) {
          // (see tile16_run: a member that gives up after `go` stores this step's whole mask blocks from an evaluation of its own, then leaves)
          if (!KEEP || !S.went) return;
          leaving = true;
        }
        if (lead && TAIL && k == 0) {   // the lead's claim had been lost (it told its members to leave): the tile belongs to somebody else
          if (tid == 0) S.cont = lost;
          __syncthreads();
          if (S.cont) return;
        }
        solo = true;
        if (tid == 0) atomicAdd(&V.C->xchg_err, 1);
        __syncthreads();
      }
    }
    if (solo) pre = mlp_forward16<KEEP>(D, D16, c0 + zero, c4 + zero, S, nib);
    if (TAIL && solo0 && k == 0) {     // taken over: only if the claim was won (known by now: the atomic was issued before the evaluation)
      if (tid == 0) S.cont = lost;
      __syncthreads();
      if (S.cont) return;
    }

    // the epilogue's view of the workspace, from kernel arguments re-read HERE (kernarg_ref; MarchArgs is the first argument of every kernel that holds 16-ray tiles): the ray-state pointers are then not alive
    // across the decoder evaluation (hoisted out of the step loop they were: AGPR / scratch copies reloaded in the lead member's epilogue)
    const View Ve = view_at(kernarg_ref<MarchArgs>(0).V, vb);
    long long mblock = -1;
    if (tid < 64) {
      const float s = tanh_spec(pre);            // (lane l holds ray l & 15)
      const float zd = init_now + m;              // depth of this step's sample (as computed before the evaluation)
      const unsigned long long was = __ballot(tid < TILE && live);
      bool stay = false;
      if (tid < TILE && live) {
        const float mn = m + clampf(s, -cd, cd) * ratio;
        const float za = mn + init_now;
        const float a = fabsf(s);
        {
          // (the ray's row addresses are recomputed from id + an opaque zero every step: hoisted out of the step loop they would be ~40
          // 64-bit values alive across the decoder evaluation)
          const int32_t id_ = id + zero;
          if (lead) Ve.m[(uint32_t)id_] = mn;
          RayPre st;                   // this step's view of the selected rows (only the keys and slots are read)
          st.m = m; st.init_now = init_now; st.maxbound = maxbound; st.minabs = minabs;
#pragma unroll
          for (int k = 0; k < MAX_BS; ++k) { st.ks[k] = S.sk[k][tid]; st.sl[k] = S.ssl[k][tid]; }
          // only the lead writes the selected rows; every member mirrors the insertion on its LDS copy of the keys / slots: the slot says
          // where this step's mask block goes, and every member stores its own words of it (mlp_forward16_cl, MASK_OWN)
          const int slot = lead ? topk_insert_pre<false>(Ve, st, id_, s, zd, Ve.pyramid ? za : mn, step) : topk_slot_pre(Ve, st, s);
          if (slot >= 0) {
            mblock = (long long)id_ * (Ve.cfg.buffer_size + 1) + slot;
            // the same insertion on the LDS copy: rows behind the new one move down, the new row takes its place
            const int bs = Ve.cfg.buffer_size;
            int pos = bs;
            bool open = true;
#pragma unroll
            for (int k = MAX_BS - 1; k >= 0; --k) {
              if (k < bs && open) { if (a < fabsf(st.ks[k])) pos = k; else open = false; }
            }
#pragma unroll
            for (int k = MAX_BS - 1; k >= 1; --k) {
              if (k < bs && k > pos) { S.sk[k][tid] = st.ks[k - 1]; S.ssl[k][tid] = st.sl[k - 1]; }
            }
#pragma unroll
            for (int k = 0; k < MAX_BS; ++k) {
              if (k == pos) { S.sk[k][tid] = s; S.ssl[k][tid] = slot; }
            }
          }
          if (lead) {
            if (a < minabs) Ve.minabs[(uint32_t)id_] = a;
            if (step == 0) Ve.first_sdf[(uint32_t)id_] = s;
          }
        }
        if (a < minabs) minabs = a;
        m = mn;
        stay = (za < maxbound) && (a >= Ve.cfg.threshold);
      }
      if (!TAIL && solo && k == 0 && lead) {
        // the cluster never assembled (its compute units are held by another stream / rank): this tile does NOT turn sticky -- the
        // lead member evaluated the step alone and hands the surviving rays to the next step's live list like a per-step tile
        // (marching 16 rays to the end on ONE compute unit would cost twice a cluster step, every step)
        wave_append(tid < TILE && stay, id, live_sel(Ve, step + 1), &Ve.C->cnt_live[step + 1]);
        stay = false;
      }
      live = stay;
      const unsigned long long now = __ballot(tid < TILE && live);
      if (tid == 0) {
        S.cont = now != 0ull;
        if (lead && k > 0) atomicAdd(&Ve.C->cnt_sticky[step], __popcll(was));   // (the tile's first step is counted in cnt_live)
      }
    }
    if (KEEP && (lead || clustered || leaving)) {
      if (tid < TILE) S.mb[tid] = mblock;
      __syncthreads();
      if (clustered) {
        store_own_mask_words<8, false>(Ve.mstore, S.mb, S, member, tid);    // every member: its own words of the rays' blocks
      } else {
        store_masks16<false>(Ve.mstore, S.mb, nib, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63);
      }
    }
    __syncthreads();                   // S.cont visible; this step's LDS reads are done before the next step's points land
    if (leaving) return;
    if (!S.cont || step + 1 >= Ve.fine_steps) break;
  }
}

// One 16-ray tile of a launch (or of a step of the persistent tail launch), everything about it decided by the caller:
struct Tile16 {
  int tile, member, cl, vb;     // virtual tile index, member index within its cluster of cl workgroups (cl = 1: single workgroup), view
  bool origin;                  // the tile evaluates f(0,0,0) of view vb
  bool sticky;                  // MODE_FINE: the tile keeps its rays to the end of the march (sticky_tile16)
  bool solo;                    // sticky tile evaluated by one workgroup from its first step (k_tail: taken over from an absent lead)
  int64_t base, count;          // the tile's rays: list[base .. min(base + 16, count))
  int32_t lost;                 // k_tail, lane 0 of the owner: != 0 when its claim on the tile failed (the claiming atomic is issued before the
                                // evaluation and looked at before anything is written or a cluster gets its `go`)
  int32_t* count2;              // k_tail: second counter of the rays that stay live (Consts::tail_sync), else null
  int32_t step;                 // MODE_FINE / MODE_COARSE: the march step (of the level) this tile evaluates (a launch: MarchArgs::step)
};

// TAIL (k_tail): write-through mask stores of the cluster members (store_own_mask_words), sticky tiles never hand rays back.
template <int MODE, bool KEEP, class SM, bool TAIL = false>
__device__ __forceinline__ void tile16_run(const MarchArgs& A, const DecoderDev& D, const DecoderDev16& D16, SM& S, const Tile16& t, const Xchg& xc) {
  constexpr int TILE = 16;
  const View& V0 = A.V;
  const int tid = threadIdx.x;
  const int tile = t.tile, member = t.member, cl = t.cl;
  const bool origin = t.origin;
  const int64_t base = t.base, count = t.count;
  const View V = view_at(V0, t.vb);
  if constexpr (MODE == MODE_FINE) {
    if (t.sticky && !origin) {
      sticky_tile16<KEEP, TAIL>(A, D, D16, S, t.vb, tile, member, base, count, t.step, xc, t.solo, t.lost);
      return;
    }
  }
  if (origin && V.C->origin_done) return;   // f(origin) was evaluated by an earlier launch (the one that turned sticky)
  const int32_t* list = (MODE == MODE_COARSE) ? level_at(A.lvl, t.vb).list : (MODE == MODE_FINE) ? live_sel(V, t.step) : nullptr;

  int32_t id = -1;
  float zd = 0.f;
  bool valid = false;
  RayPre st;
  if (tid < TILE) {
    float p[3] = {0.f, 0.f, 0.f};
    const int64_t r = base + tid;
    valid = !origin && r < count;
    if (valid) {
      if (MODE == MODE_EVAL) {
        id = (int32_t)r;
        p[0] = A.xyz[r * 3]; p[1] = A.xyz[r * 3 + 1]; p[2] = A.xyz[r * 3 + 2];
      } else {
        id = ld_x<false>(list + r);
        const LevelView L = (MODE == MODE_COARSE) ? level_at(A.lvl, t.vb) : V.lv[0];
        const CamRegs cam = load_cam(V.C);
        float cx, cy;
        level_center(L, id, cx, cy);
        const RayGeo g = make_ray(V.cfg.K_inv, cam.R, cx, cy);
        if (MODE == MODE_FINE) {
          raypre_load<false>(V, id, st);
          zd = st.init_now + st.m;
          if constexpr (TAIL) {
            // the persistent tail launch parks the ray's state in LDS across the evaluation instead of twenty registers per lane (inside
            // its step loop they end up in scratch). NOT re-read from memory in the epilogue: a cluster's members other than the lead
            // need the selected rows as they were BEFORE the lead member's epilogue rewrites them (topk_slot_pre: where the mask block goes)
            S.sst[0][tid] = st.m; S.sst[1][tid] = st.init_now; S.sst[2][tid] = st.maxbound; S.sst[3][tid] = st.minabs;
#pragma unroll
            for (int k = 0; k < MAX_BS; ++k) { S.sk[k][tid] = st.ks[k]; S.ssl[k][tid] = st.sl[k]; }
          }
        }
        else zd = L.cinit[id] + L.cm[id];
        make_point(V.cfg.M, cam.c, g.d, zd, p);
      }
    }
    S.xyz[tid] = p[0]; S.xyz[TILE + tid] = p[1]; S.xyz[2 * TILE + tid] = p[2];
  }
  __syncthreads();

  uint32_t nib[8];
  const float* c0 = (MODE == MODE_EVAL) ? A.c0c4 : V.C->c0;
  const float* c4 = (MODE == MODE_EVAL) ? A.c0c4 + HID : V.C->c4;
  float pre;
  bool clustered = false;               // the tile's value (and, KEEP, its mask blocks in S.mk) came from the cluster path
  bool helper = false;                  // a member of a cluster tile other than the lead (KEEP: it stores its words of the mask blocks, nothing else)
  if (MODE != MODE_EVAL && cl > 1) {
    if constexpr (MODE != MODE_EVAL) {
      // KEEP: every member evaluates lin8 too (ALL_LIN8) and records the ReLU bits of its own rows (MASK_OWN): it then knows where the rays'
      // mask blocks go and stores its words of them itself -- the lead member does not extract bits from staged rows (that made it the
      // last to publish in every layer). Without KEEP the other members leave after their last slice.
      const int32_t lost = TAIL ? t.lost : 0;     // (lead member, lane 0: its claim failed -> the cluster is told to leave)
      if (cl == 8) pre = mlp_forward16_cl<8, KEEP, KEEP, KEEP>(D, D16, c0, c4, S, xc, tile, member, true, lost);
      else if (cl == 4) pre = mlp_forward16_cl<4, KEEP, KEEP, KEEP>(D, D16, c0, c4, S, xc, tile, member, true, lost);
      else pre = mlp_forward16_cl<2, KEEP, KEEP, KEEP>(D, D16, c0, c4, S, xc, tile, member, true, lost);
    } else pre = 0.f;
    helper = member != cl_lead(cl);      // only the lead member runs the epilogue; with KEEP the others store their mask words
    clustered = S.fail == 0;
    if (helper && !clustered) {
      // Gave up before `go` (or no masks to save): the lead member evaluates the tile alone and stores whole mask blocks. Gave up AFTER
      // `go` with masks to save: it may have published every slice already (a timeout while staging h7) -- then the others complete the
      // tile and store THEIR words of the rays' mask blocks, and this member's words would keep what an earlier render left there (ADVICE
      // r5). It evaluates the tile on its own and stores the whole blocks (the same bits the others store: harmless where they overlap).
      if (!KEEP || !S.went) return;
      if (tid == 0) atomicAdd(&V.C->xchg_err, 1);
      __syncthreads();
      pre = mlp_forward16<KEEP>(D, D16, c0, c4, S, nib);
    } else
    if (!clustered) {
      if constexpr (TAIL) {      // ... or its lead's claim had been lost: the tile is somebody else's
        if (tid == 0) S.cont = t.lost;
        __syncthreads();
        if (S.cont) return;
      }
      // the cluster did not assemble (compute units held by other streams / ranks) or a barrier timed out: the lead member
      // evaluates the tile on its own -- identical values; counted in the render stats (cluster_fallbacks)
      if (tid == 0) atomicAdd(&V.C->xchg_err, 1);
      __syncthreads();
      pre = mlp_forward16<KEEP>(D, D16, c0, c4, S, nib);
    }
  } else {
    pre = mlp_forward16<KEEP>(D, D16, c0, c4, S, nib);
    if constexpr (TAIL) {        // the claim (issued before the evaluation) must have been won before anything is written
      if (tid == 0) S.cont = t.lost;
      __syncthreads();
      if (S.cont) return;
    }
  }

  // (the persistent tail launch: the epilogue takes its view of the workspace from kernel arguments re-read HERE, like a sticky tile's --
  // inside the step loop the prologue's pointers would otherwise stay alive across the evaluation, in scratch)
  const View Ve = TAIL ? view_at(kernarg_ref<MarchArgs>(0).V, t.vb) : V;
  long long mblock = -1;
  if (tid < 64) {
    const float s = tanh_spec(pre);
    RayPre st_again;         // (the persistent tail launch: the state parked in LDS by the prologue)
    if constexpr (TAIL && MODE == MODE_FINE) {
      st_again.m = 0.f; st_again.init_now = 0.f; st_again.maxbound = 0.f; st_again.minabs = 0.f;
#pragma unroll
      for (int k = 0; k < MAX_BS; ++k) { st_again.ks[k] = 0.f; st_again.sl[k] = 0; }
      if (valid) {
        st_again.m = S.sst[0][tid]; st_again.init_now = S.sst[1][tid]; st_again.maxbound = S.sst[2][tid]; st_again.minabs = S.sst[3][tid];
#pragma unroll
        for (int k = 0; k < MAX_BS; ++k) { st_again.ks[k] = S.sk[k][tid]; st_again.sl[k] = S.ssl[k][tid]; }
      }
    }
    const RayPre& sr = (TAIL && MODE == MODE_FINE) ? st_again : st;
    if (helper) {      // where the mask blocks go, nothing else (the lead member writes the step's results)
      if (origin) {
        if (tid == 0) mblock = Ve.morigin;
      } else if (MODE == MODE_COARSE) {
        if (valid) mblock = Ve.mfine + moff_at(A.lvl) + (long long)((size_t)t.step * level_at(A.lvl, t.vb).n + id);
      } else if (MODE == MODE_FINE) {
        if (valid) {
          const int slot = topk_slot_pre(Ve, sr, s);
          if (slot >= 0) mblock = (long long)id * (Ve.cfg.buffer_size + 1) + slot;
        }
      }
    } else if (origin) {
      if (tid == 0) { Ve.C->f_origin = s; Ve.C->origin_done = 1; mblock = Ve.morigin; }
    } else if (MODE == MODE_EVAL) {
      if (valid) A.sdf_out[id] = (A.clamp >= 0.f) ? clampf(s, -A.clamp, A.clamp) : s;
    } else if (MODE == MODE_COARSE) {
      if (valid) {
        const LevelView L = level_at(A.lvl, t.vb);
        const float mn = L.cm[id] + clampf(s, -Ve.cfg.clamp_dist, Ve.cfg.clamp_dist) * Ve.cfg.ratio;
        L.cm[id] = mn;
        const size_t o = (size_t)t.step * L.n + id;
        L.rs[o] = s;
        L.rzb[o] = zd;
        L.rza[o] = mn + L.cinit[id];
        mblock = Ve.mfine + moff_at(A.lvl) + (long long)o;
      }
    } else {
      const float cd = Ve.cfg.clamp_dist, ratio = Ve.cfg.ratio;
      bool stay = false;
      if (valid) {
        const float mn = sr.m + clampf(s, -cd, cd) * ratio;
        Ve.m[(uint32_t)id] = mn;
        const float za = mn + sr.init_now;
        const int slot = topk_insert_pre<false>(Ve, sr, id, s, zd, Ve.pyramid ? za : mn, t.step);
        if (slot >= 0) mblock = (long long)id * (Ve.cfg.buffer_size + 1) + slot;
        const float a = fabsf(s);
        if (a < sr.minabs) Ve.minabs[(uint32_t)id] = a;
        if (t.step == 0) Ve.first_sdf[(uint32_t)id] = s;
        stay = (za < sr.maxbound) && (a >= Ve.cfg.threshold);
      }
      wave_append<false>(stay, id, live_sel(Ve, t.step + 1), &Ve.C->cnt_live[t.step + 1], TAIL ? t.count2 : nullptr);
    }
  }
  if (KEEP && MODE != MODE_EVAL) {
    if (tid < TILE) S.mb[tid] = mblock;
    __syncthreads();
    if (clustered) {   // every member: its own words of the rays' blocks
      int tz = 0;        // (tail launch: the word indices derived from the thread index are computed HERE, not once in front of the step loop)
      if constexpr (TAIL) asm volatile("s_mov_b32 %0, 0" : "=s"(tz));
      if (cl == 8) store_own_mask_words<8, TAIL>(Ve.mstore, S.mb, S, member, tid + tz);
      else if (cl == 4) store_own_mask_words<4, TAIL>(Ve.mstore, S.mb, S, member, tid + tz);
      else store_own_mask_words<2, TAIL>(Ve.mstore, S.mb, S, member, tid + tz);
    } else {
      store_masks16<false>(Ve.mstore, S.mb, nib, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63);
    }
  }
}


// The same march step on 16-ray tiles (v_mfma_f32_16x16x4_f32), for the live-ray tail: see distr_mlp.hpp::Smem16.
// MODE_FINE (recursive marchers), MODE_COARSE (pyramid levels of small images) and MODE_EVAL.
// SM: Smem16CLX where cluster tiles can occur (MODE_FINE / MODE_COARSE), Smem16CL for MODE_EVAL (no landing zone: two workgroups per CU)
template <int MODE, bool KEEP, class SM>
__device__ __forceinline__ void march_tile16(const MarchArgs& A, const DecoderDev& D, const DecoderDev16& D16, SM& S, int bidx,
                                             int origin_tile) {
  constexpr int TILE = 16;
  const View& V0 = A.V;
  const int tid = threadIdx.x;
  const int B = (MODE == MODE_EVAL) ? 1 : V0.nviews;
  // virtual ray range [lo, hi) of this launch's 16-ray tiles over all views (16-ray tiles exist only for the recursive marchers'
  // steps, the coarse pyramid levels and explicit point lists)
  int64_t lo = 0, hi;
  int gran = TILE;
  int32_t c = 0, incl = 0;
  if (MODE == MODE_EVAL) {
    hi = A.n;
  } else {
    const int32_t* p0 = (MODE == MODE_COARSE) ? &V0.C->cnt_level[A.lvl] : &V0.C->cnt_live[A.step];
    c = vload(p0, V0.vstride, B);
    if (MODE == MODE_FINE) {
      const int32_t i16 = vprefix(c, B, 16), i64 = vprefix(c, B, 64);
      fine_split(vtotal(i16, B), vtotal(i64, B), A.t16, A.t32, 16, lo, hi, gran);
      incl = (gran == 16) ? i16 : i64;
    } else {
      incl = vprefix(c, B, TILE);
      hi = vtotal(incl, B);
    }
  }
  const int64_t n = hi - lo;
  const int64_t ntiles = (n + TILE - 1) / TILE;
  // the tiles after the last real one evaluate f(0,0,0) of each view (sample point of padded rows) in the launch that carries
  // `origin_tile` (the last step): on a tail step they ride along for free instead of adding tiles to a full round elsewhere
  int norigin = (MODE == MODE_FINE && origin_tile) ? B : 0;
  // Sticky launch: ALL live rays of the step (granularity 16: nothing went to the 32- / 64-ray roles) and the views' origin tiles
  // fit one launch of 8-CU cluster tiles -- every workgroup of the launch takes the same decision from the same counts. The
  // tiles then keep their rays to the end of the march (sticky_tile16) and the origin tiles ride on THIS launch (the last
  // step's launch would otherwise be a whole cluster-tile latency for them alone). A small remainder next to whole 64-ray
  // rounds must NOT go sticky: its rays would finish their march inside a launch whose other rays move on step by step.
  bool sticky_launch = false;
  if (MODE == MODE_FINE && A.xc.buf && A.xc.sticky && A.xc.max_cl >= 8 && gran == 16 && A.step + 1 < V0.fine_steps && ntiles + B <= 32) {
    sticky_launch = true;
    norigin = B;
  }
  // Cluster size from the (device-side) number of tiles of this launch: with at most 32 / 64 / 128 tiles -- counting the extra
  // tiles for f(origin) on the one launch that carries them, so that the grid stays within 256 workgroups -- 8 / 4 / 2 compute units
  // share each tile. Measured step time (C3 tail, profiles/r02_steps_c3.md): 52 us (8), 64 us (4), 94 us (2), 107 us (single
  // workgroup).
  int cl = 1;
  if (MODE != MODE_EVAL && A.xc.buf) {
    const int64_t need = ntiles + norigin;
    cl = (need <= 32 && A.xc.max_cl >= 8) ? 8 : (need <= 64 && A.xc.max_cl >= 4) ? 4 : (need <= 128 && A.xc.min_cl <= 2) ? 2 : 1;
  }
  int tile = bidx, member = 0;
  if (cl > 1) {   // members of a cluster = workgroups with equal index mod 8 (same XCD: every role of a launch starts at a multiple of 8).
    // (Measured alternative: member m of every cluster on XCD m, so that an XCD streams only its 1/cl slice of the weights from a
    // warm L2 -- the compute phase did not change, it is not bound by weight latency, and the exchange across XCDs cost 1 us more.)
    const int gq = bidx / (8 * cl), r = bidx % (8 * cl);
    tile = gq * 8 + (r & 7);
    member = r >> 3;
    if (A.xc.spread) { tile = gq * 8 + r / cl; member = r % cl; }     // tests: members on consecutive workgroups = different XCDs
  }
  const bool origin = norigin > 0 && tile >= ntiles && tile < ntiles + norigin;
  int vb = 0;
  int64_t base = (int64_t)tile * TILE, count = hi;
  if (origin) {
    vb = tile - (int)ntiles;
  } else {
    const int64_t vbase = lo + (int64_t)tile * TILE;
    if (vbase >= hi) return;
    if (MODE != MODE_EVAL) {
      int64_t start;
      int32_t cnt;
      vfind(c, incl, B, gran, vbase, vb, start, cnt);
      base = vbase - start;
      count = cnt;
      if (base >= count) return;       // padding behind the view's last ray (all members of a cluster agree)
    }
  }
  Tile16 t;
  t.tile = tile; t.member = member; t.cl = cl; t.vb = vb; t.origin = origin; t.sticky = sticky_launch; t.solo = false; t.base = base; t.count = count;
  t.lost = 0; t.count2 = nullptr; t.step = A.step;
  tile16_run<MODE, KEEP, SM>(A, D, D16, S, t, A.xc);
}

// (MODE_EVAL: point lists of any length on single-workgroup tiles, two per CU; MODE_COARSE: at most 256 tiles, cluster tiles among
// them -- one workgroup per CU and the whole register file for the weight ring + the granule requests in flight)
template <int MODE, bool KEEP>
__global__ void __launch_bounds__(256, (MODE == MODE_EVAL) ? 2 : 1) k_march16(MarchArgs A, DecoderDev D, DecoderDev16 D16) {
  __shared__ typename std::conditional<MODE == MODE_EVAL, Smem16CL, Smem16CLX>::type S;
  march_tile16<MODE, KEEP>(A, D, D16, S, (int)blockIdx.x, A.origin_tile);
}


// ------------------------------------------------------------------------------------------ persistent tail launch
// k_tail: the full-resolution steps [tail_from, fine_steps) of the recursive marchers inside ONE launch of 256 workgroups (one per
// compute unit). Every launch the host issues after the last ray has finished is pure waste, and the host cannot know when that is
// without synchronising (renderer.py:528-567 breaks out of its loop there: one host sync per step) -- 72 of the 94 full-resolution
// launches of a 137 x 137 / 100-step render found nothing to do. The host starts this launch where the previous render of the same
// configuration entered the sticky regime (distr_api.hip: tail_hint_slot; a hint, never a correctness input: the kernel is correct for
// any live count at any start step) and enqueues nothing behind it.
// Per step every workgroup derives the SAME plan from the device-side live counts (tail_plan): the step's rays on 16-ray tiles --
// cluster tiles of 8 / 4 / 2 workgroups while at most 32 / 64 / 128 tiles are left, single-workgroup tiles in rounds of 256 above
// that -- and, once everything fits 32 cluster tiles, sticky tiles that march their rays to the end (sticky_tile16); then the launch
// is over. A ray's arithmetic is exactly the per-step launches' one: bit-identical renders (tests/test_gpu_tail.py).
// Step barrier WITHOUT a co-residency assumption (the hardware promises none: another stream or process may hold compute units, and
// two spinning launches could starve each other for ever): a tile is OWNED through a claim word (atomicMax with the step's tag), its
// owner evaluates it, releases its stores (one agent-scope release fence per tile; every workgroup takes one acquire fence per step)
// and counts it in tail_sync[2 step]; a step is complete when the count equals the number of tiles. Tiles are handed out statically
// (workgroup b: tiles b, b + 256, ...; clusters as in march_tile16, the lead member claims), but a workgroup that has waited
// TAIL_T_STEAL for a step takes over every tile nobody has claimed and evaluates it alone -- so the launch finishes with ANY subset of
// its workgroups resident, and a lead that arrives late finds its tile taken and moves on.
constexpr long long TAIL_T_STEAL = 400 * 100;     // 400 us (100 MHz ticks): several tile latencies
constexpr int32_t TAIL_T_GO = 150 * 100;          // a cluster member waits this long for its lead's verdict (workgroups start a step together)

__device__ __forceinline__ int32_t vload_fresh(const int32_t* p0, int64_t stride, int B) {   // vload past the L1 (counters other workgroups of THIS launch wrote)
  if (B <= 1) return __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int lane = threadIdx.x & 63;
  asm volatile("" : "+v"(lane));      // (the 64-bit lane x stride product is formed here, not once at kernel entry and kept across the job loop)
  return (lane < B) ? __hip_atomic_load(reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(p0) + (int64_t)lane * stride), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT) : 0;
}

struct TailPlan {
  int32_t c, incl;     // this lane's view: live rays, inclusive prefix of the counts padded to 16 (vprefix)
  int64_t ntiles;      // 16-ray tiles of the step over all views
  int norigin;         // + one tile per view for f(0,0,0) on the launch's first step
  int cl;              // workgroups per tile
  bool sticky;         // the tiles keep their rays to the end of the march
};

// plan of a step from the views' live counts (c: this lane's view, vload layout)
__device__ __forceinline__ TailPlan tail_plan(const MarchArgs& A, const Xchg& xc, int32_t c, int step, bool first) {
  const View& V0 = A.V;
  const int B = V0.nviews;
  TailPlan P;
  P.c = c;
  P.incl = vprefix(P.c, B, 16);
  P.ntiles = vtotal(P.incl, B) / 16;
  P.norigin = first ? B : 0;
  const int64_t need = P.ntiles + P.norigin;
  P.sticky = xc.buf && xc.sticky && xc.max_cl >= 8 && step + 1 < V0.fine_steps && P.ntiles > 0 && need <= 32;
  P.cl = 1;
  if (xc.buf) P.cl = (need <= 32 && xc.max_cl >= 8) ? 8 : (need <= 64 && xc.max_cl >= 4) ? 4 : (need <= 128 && xc.min_cl <= 2) ? 2 : 1;
  return P;
}

// { tiles counted, rays entering the next step } of tail step k as ONE 8-byte word (Consts::tail_sync is 8-byte aligned)
__device__ __forceinline__ unsigned long long tail_sync_load(const Consts* C, int k) {
  return __hip_atomic_load(static_cast<const unsigned long long*>(__builtin_assume_aligned(&C->tail_sync[2 * k], 8)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// claim word of virtual tile vt: the views' tclaim arrays (P / 16 + 2 words each) taken as one array over the batch
__device__ __forceinline__ int32_t* tail_claim_word(const View& V0, int64_t vt) {
  const int64_t ch = V0.P / 16 + 2;
  return reinterpret_cast<int32_t*>(reinterpret_cast<char*>(V0.tclaim) + (vt / ch) * V0.vstride) + (vt % ch);
}

// Evaluates virtual tile vt of step T0 + k as member `member` of a cluster of cl workgroups (cl = 1: alone). The lead member (or the
// single workgroup) claims the tile -- the atomic is ISSUED here and its answer looked at inside the tile, before a cluster gets its `go`
// and before anything is written (tile16_run, Tile16::lost): a round trip to the claim word is not paid in front of every tile -- and
// counts the tile in tail_sync when its stores are complete.
template <bool KEEP>
__device__ __forceinline__ void tail_slot(const MarchArgs& A, const DecoderDev& D, const DecoderDev16& D16, Smem16CLX& S, int32_t* ctl, const TailPlan& P,
                                          const Xchg& xc, int64_t vt, int k, int cl, int member, bool sticky, bool steal) {
  const View& V0 = A.V;
  const int tid = threadIdx.x;
  const bool lead = member == cl_lead(cl);
  Tile16 t;
  t.tile = (int)vt; t.member = member; t.cl = cl; t.vb = 0; t.base = 0; t.count = 0; t.lost = 0; t.step = A.step + k;
  t.origin = vt >= P.ntiles;
  t.sticky = sticky; t.solo = sticky && cl == 1;
  if (t.solo) t.member = cl_lead(8);          // (sticky_tile16 is written for 8 members: alone = as their lead)
  if (steal) {
    // a scan for abandoned tiles looks before it claims (hundreds of failing atomics per scan would be the slow part of it), and only
    // touches a tile it can have: most are taken
    if (tid == 0) {
      int32_t* w = tail_claim_word(V0, vt);
      const bool got = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1 && atomicMax(w, k + 1) < k + 1;
      if (got) atomicAdd(&V0.C->tail_steals, 1);
      ctl[0] = got ? 1 : 0;
    }
    __syncthreads();
    const bool got = ctl[0] != 0;
    __syncthreads();
    if (!got) return;
  } else if (lead && tid == 0) {
    t.lost = (atomicMax(tail_claim_word(V0, vt), k + 1) < k + 1) ? 0 : 1;
  }
  if (t.origin) {
    t.vb = (int)(vt - P.ntiles);
  } else {
    int64_t start;
    int32_t cnt;
    vfind(P.c, P.incl, V0.nviews, 16, vt * 16, t.vb, start, cnt);
    t.base = vt * 16 - start;
    t.count = cnt;
  }
  // second counter of the rays that stay live: the count word next to the step's barrier word, in the tile's view
  t.count2 = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(&V0.C->tail_sync[2 * k + 1]) + (int64_t)t.vb * V0.vstride);
  tile16_run<MODE_FINE, KEEP, Smem16CLX, true>(A, D, D16, S, t, xc);
  // EVERY wave's stores have reached the L2 (vmcnt(0)) before ONE lane writes the L2's dirty lines back: a mask-block store of waves 1..3
  // that lands behind the write-back would stay dirty in this XCD's L2 until some later release -- and then overwrite what a cluster on
  // another XCD has meanwhile written THROUGH to the same block (the slot was reused two steps later): wrong ReLU masks in the backward,
  // found by the absent-workgroup test (plain whole-block stores of taken-over tiles next to write-through words of cluster members).
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                // ... and the tile's LDS is free for the next tile
  if (!sticky && tid == 0) {
    // release the tile's results (ray state, next step's list entries, whole mask blocks: plain stores) to the other XCDs -- every member
    // that stored anything (a member that gave up after `go` stores whole blocks too) --, THEN the lead member counts the tile
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lead && !t.lost) atomicAdd(&V0.C->tail_sync[2 * k], 1);
  }
}

template <bool KEEP>
__global__ void __launch_bounds__(256, 1) k_tail(MarchArgs A, DecoderDev D, DecoderDev16 D16) {
  __shared__ Smem16CLX S;
  __shared__ int32_t ctl[4];
  const View& V0 = A.V;
  const int tid = threadIdx.x, bidx = (int)blockIdx.x, nwg = (int)gridDim.x;
  const int T0 = A.step, B = V0.nviews;
  if (bidx < A.tail_absent) return;      // (tests: the others must take over this workgroup's tiles)
  int32_t c_next = vload(&V0.C->cnt_live[T0], V0.vstride, B);        // (written by earlier launches)
  bool behind = true;
  for (int k = 0; T0 + k < V0.fine_steps; ++k) {
    // a zero the optimiser cannot see through, added to the tile / member index of every tile: the weight-stream addresses of the decoder
    // evaluation then depend on the iteration and are NOT hoisted out of the step loop (hoisted they stay live next to the weight ring:
    // scratch; the same device as in sticky_tile16)
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    // (the kernel's arguments stay where they are: a modified COPY of MarchArgs would be hundreds of scalars loaded at the top of every
    // iteration and kept alive across the tile)
    Xchg xc = A.xc;
    xc.epoch = A.xc.epoch + (uint32_t)k;
    xc.epochs = A.xc.epochs - (uint32_t)k;
    xc.par = k & 1;            // alternate the exchange slots between steps (as sticky tiles do): layer 1 of step k + 1 never lands on step k's h7
    const TailPlan P = tail_plan(A, xc, c_next, T0 + k, k == 0);
    const int64_t slots = P.ntiles + P.norigin;
    if (slots == 0) break;     // no live ray left (every workgroup reads the same count)
    if (behind && !P.sticky) {
      // A workgroup that starts late (its compute unit was held by somebody else) must not evaluate tiles of steps that are long over
      // (it would find out only afterwards: the claim is looked at behind the evaluation): until it has seen ONE incomplete step it peeks.
      if (tid == 0) {
        const unsigned long long v = tail_sync_load(V0.C, k);
        ctl[1] = ((int32_t)(uint32_t)v >= (int32_t)slots) ? 1 : 0;
        ctl[2] = (int32_t)(uint32_t)(v >> 32);
      }
      __syncthreads();
      const bool over = ctl[1] != 0;
      const int32_t cn = ctl[2];
      __syncthreads();
      if (over) {
        c_next = (B > 1) ? vload_fresh(&V0.C->tail_sync[2 * k + 1], V0.vstride, B) : cn;
        continue;
      }
      behind = false;
      if (k > 0) {               // (it skipped steps without taking their acquire fences)
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __syncthreads();
      }
    }
    // This workgroup's jobs of the step, ONE call site for all of them (the tile code is large): first its own tile(s) -- cluster tiles as
    // in march_tile16 (members = workgroups with equal index mod 8: one XCD), single-workgroup tiles b, b + 256, ... --, then the wait for
    // the step (sticky step: for every tile to have an owner), and from a wait that took too long a scan over all tiles for abandoned ones.
    enum { OWN = 0, WAIT = 1, SCAN = 2 };
    int mode = OWN;
    int64_t cur = bidx, scan = 0;
    int member = 0;
    if (P.cl > 1) {
      const int gq = bidx / (8 * P.cl), r = bidx % (8 * P.cl);
      cur = gq * 8 + (r & 7);
      member = r >> 3;
      if (xc.spread) { cur = gq * 8 + r / P.cl; member = r % P.cl; }     // tests: members on consecutive workgroups = different XCDs
    }
    const long long t_step = (long long)wall_clock64();
    for (;;) {
      if (mode == OWN && cur >= slots) mode = WAIT;
      if (mode == WAIT) {
        int32_t state = 0;     // 1: the step is complete, 2: waited too long
        if (P.sticky) {
          // The tiles march to the end inside their workgroups: nothing to wait for -- but no tile may stay without an owner. Everybody (own tile
          // done, or none) watches the claim words until all are taken; after TAIL_T_STEAL an unclaimed tile's lead member is not coming.
          int32_t v = k + 1;
          if (tid < slots) v = __hip_atomic_load(tail_claim_word(V0, tid), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (ONE thread reads the clock: every wave must take the same branch -- the tile code behind it is full of workgroup barriers)
          if (tid == 0) ctl[1] = ((long long)wall_clock64() - t_step > TAIL_T_STEAL) ? 2 : 0;
          const bool open = __syncthreads_or(tid < 64 && v < k + 1) != 0;
          state = !open ? 1 : ctl[1];
          __syncthreads();
          if (state == 0) __builtin_amdgcn_s_sleep(32);
        } else {
          if (tid == 0) {
            // ONE 8-byte load per poll: { tiles counted, rays entering the next step } -- when the first says "all", the second is final (a
            // tile's count atomics are complete before it is counted)
            const long long t0 = (long long)wall_clock64();
            int32_t st;
            for (;;) {
              const unsigned long long v = tail_sync_load(V0.C, k);
              if ((int32_t)(uint32_t)v >= (int32_t)slots) { st = 1; ctl[2] = (int32_t)(uint32_t)(v >> 32); break; }
              if ((long long)wall_clock64() - t0 > TAIL_T_STEAL) { st = 2; break; }
              __builtin_amdgcn_s_sleep(2);
            }
            ctl[1] = st;
          }
          __syncthreads();
          state = ctl[1];
          c_next = ctl[2];
          __syncthreads();
        }
        if (state == 1) break;
        if (state == 0) continue;
        mode = SCAN; scan = 0;
      }
      if (mode == SCAN && scan >= slots) {      // every tile has been offered to this workgroup: all of them have an owner now
        if (P.sticky) break;
        mode = WAIT;
        continue;
      }
      const int64_t vt = (mode == OWN) ? cur : (bidx + scan) % slots;
      const int cl = (mode == OWN) ? P.cl : 1;
      // the tile reads the kernel's arguments through references taken HERE (kernarg_ref): none of them is loaded in front of the step
      // loop and carried across every tile
      constexpr size_t OFF_D = KERNARG_OFF_D, OFF_D16 = KERNARG_OFF_D16;
      tail_slot<KEEP>(kernarg_ref<MarchArgs>(0), kernarg_ref<DecoderDev>(OFF_D), kernarg_ref<DecoderDev16>(OFF_D16), S, ctl, P, xc, vt + zero, k, cl,
                      ((mode == OWN) ? member : 0) + zero, P.sticky && vt < P.ntiles, mode == SCAN);
      if (mode == OWN) cur = (P.cl > 1) ? slots : cur + nwg;
      else ++scan;
    }
    if (P.sticky) return;      // (the sticky tiles have marched their rays to the end)
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // ONE acquire per workgroup, then plain loads of what the step wrote
    __syncthreads();
    if (B > 1) c_next = vload_fresh(&V0.C->tail_sync[2 * k + 1], V0.vstride, B);      // a batch: every view's own count word
  }
}

// One full-resolution march step of the recursive marchers in ONE launch: the three tile sizes of the split (fine_range)
// are roles of the same grid -- n32 workgroups for 32-ray tiles, n16 for 16-ray / cluster tiles, n64 for 64-ray tiles
// (n32 a multiple of 8, so cluster members keep equal index mod 8 = the same XCD). Each role finds its
// range from the device-side live count and exits if it is empty. The 64-ray role is PERSISTENT: at most one workgroup
// per CU, each walking tiles b, b + n64, ... of the range (the tiles cost the same, so the static assignment loses
// nothing, a full round needs no re-dispatch, and a tail step has 512 idle workgroups to retire instead of 4 352).
// Against three launches per step this saves two empty launches (4-5 us each) on almost every step.
struct StepGrid { int32_t n64, n32, n16; };

template <bool KEEP, int ARITH = 0>
__global__ void __launch_bounds__(256, 1) k_step(MarchArgs A, DecoderDev D, DecoderDev16 D16, StepGrid G) {
  __shared__ __attribute__((aligned(16))) unsigned char raw[(sizeof(Smem<2>) > sizeof(SmemB6<2>)) ? sizeof(Smem<2>) : sizeof(SmemB6<2>)];
  static_assert(sizeof(Smem<2>) >= sizeof(Smem<1>) && sizeof(Smem<2>) >= sizeof(Smem16CLX) && sizeof(SmemB6<2>) >= sizeof(SmemB6<1>), "role shared memory");
  // role order in the grid: 32-ray tiles, 16-ray / cluster tiles, 64-ray tiles. On a tail step the cluster tiles start
  // after one wave of idle 32-ray workgroups and the idle 64-ray workgroups retire on the free CUs while the clusters
  // run; on a dense step the remainder tiles start first and the persistent 64-ray workgroups follow as CUs free up.
  // ARITH = 1 (split-bf16): 64- and 32-ray roles only (G.n16 = 0, t16 = 0); the views' origin tiles close the 32-ray role.
  const int b = blockIdx.x;
  if (b < G.n32) {
    (void)march_tile<MODE_FINE, 1, KEEP, ARITH>(A, D, *reinterpret_cast<typename TileSmem<1, ARITH>::type*>(raw), b, G.n32, 32, ARITH ? A.origin_tile : 0);
  } else if (b < G.n32 + G.n16) {
    if constexpr (ARITH == 0) march_tile16<MODE_FINE, KEEP>(A, D, D16, *reinterpret_cast<Smem16CLX*>(raw), b - G.n32, A.origin_tile);
  } else {
    for (int t = b - G.n32 - G.n16;; t += G.n64) {
      if (!march_tile<MODE_FINE, 2, KEEP, ARITH>(A, D, *reinterpret_cast<typename TileSmem<2, ARITH>::type*>(raw), t, 0x7fffffff, 64, 0)) break;
      __syncthreads();       // the tile's last LDS reads (mask store) are done before the next tile's points are written
    }
  }
}

// decode_color (core/utils/decoder_utils.py:94-112) for the surface points of SDFRenderer_color.render_color
// (core/sdfrenderer/renderer_rgb.py:20-38): the same fused tile on the colour decoder's weights (latent = shape code |
// colour code, folded into c0 / c4), lin8 with three rows, tanh on each. Forward only.
DISTR_GLOBAL void __launch_bounds__(256, 1) k_color(const float* __restrict__ xyz, int64_t n, const float* __restrict__ c0c4,
                                                  float* __restrict__ rgb, DecoderDev D) {
  constexpr int RB = 2, TILE = 64;
  __shared__ Smem<RB> S;
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * TILE;
  if (base >= n) return;
  if (tid < TILE) {
    const int64_t r = base + tid;
    const bool v = r < n;
    S.xyz[tid] = v ? xyz[r * 3] : 0.f; S.xyz[TILE + tid] = v ? xyz[r * 3 + 1] : 0.f; S.xyz[2 * TILE + tid] = v ? xyz[r * 3 + 2] : 0.f;
  }
  __syncthreads();
  uint32_t masks[8][4];
  float pre[3];
  pre[0] = mlp_forward<RB, false>(D, c0c4, c0c4 + HID, S, masks);
#pragma unroll
  for (int c = 1; c < 3; ++c) {
    __syncthreads();                       // S.part of the previous row has been read by everybody
    pre[c] = lin8_row<RB>(D.w8 + c * HID, D.b8x[c - 1], S);
  }
  if (tid < TILE && base + tid < n) {
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[(base + tid) * 3 + c] = tanh_spec(pre[c]);
  }
}

// Backward of decode_color (decoder_utils.py:94-112 differentiated by autograd in the reference): recompute the colour decoder's
// forward for a tile of 64 points keeping the ReLU masks, d8_c = g_rgb_c * (1 - rgb_c^2), dX chain with the 3-row lin8, per-tile
// delta sums for the [shape | colour] code gradient, d rgb / d xyz per point.
DISTR_GLOBAL void __launch_bounds__(256, 1) k_color_bwd(const float* __restrict__ xyz, int64_t n, const float* __restrict__ c0c4,
                                                      const float* __restrict__ g_rgb, float* __restrict__ g_xyz, float* __restrict__ partial,
                                                      DecoderDev D) {
  constexpr int RB = 2, TILE = 64;
  __shared__ Smem<RB> S;
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * TILE;
  if (base >= n) return;
  const bool valid = tid < TILE && base + tid < n;
  if (tid < TILE) {
    S.xyz[tid] = valid ? xyz[(base + tid) * 3] : 0.f; S.xyz[TILE + tid] = valid ? xyz[(base + tid) * 3 + 1] : 0.f;
    S.xyz[2 * TILE + tid] = valid ? xyz[(base + tid) * 3 + 2] : 0.f;
  }
  __syncthreads();
  uint32_t masks[8][4];
  float pre[3];
  pre[0] = mlp_forward<RB, true>(D, c0c4, c0c4 + HID, S, masks);
#pragma unroll
  for (int c = 1; c < 3; ++c) {
    __syncthreads();
    pre[c] = lin8_row<RB>(D.w8 + c * HID, D.b8x[c - 1], S);
  }
  __syncthreads();
  if (tid < TILE) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float y = tanh_spec(pre[c]);
      S.aux[c * TILE + tid] = valid ? g_rgb[(base + tid) * 3 + c] * __builtin_fmaf(-y, y, 1.0f) : 0.f;
    }
  }
  __syncthreads();
  float* part = partial + (size_t)blockIdx.x * PSTRIDE;
  mlp_backward<RB, 3>(D, S, masks, part, part + HID);
  if (g_xyz && valid) {
#pragma unroll
    for (int c = 0; c < 3; ++c) g_xyz[(base + tid) * 3 + c] = S.aux[(1 + c) * TILE + tid];
  }
}

// test/debug only: post-activation of layer `layer` for n points -> out[n][512] (see tests/test_gpu_parity.py)
template <int RB>
__global__ void __launch_bounds__(256, (RB == 1) ? 2 : 1) k_debug_layer(const float* xyz, int64_t n, const float* c0c4, int layer,
                                                                        float* out, DecoderDev D, long long* ts_out) {
  constexpr int TILE = 32 * RB;
  __shared__ Smem<RB> S;
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * TILE;
  if (tid < TILE) {
    const int64_t r = base + tid;
    const bool v = r < n;
    S.xyz[tid] = v ? xyz[r * 3] : 0.f; S.xyz[TILE + tid] = v ? xyz[r * 3 + 1] : 0.f; S.xyz[2 * TILE + tid] = v ? xyz[r * 3 + 2] : 0.f;
  }
  __syncthreads();
  uint32_t masks[8][4];
  long long* ts = ts_out ? ts_out + (size_t)blockIdx.x * 40 : nullptr;
  const float pre = mlp_forward<RB, false, true>(D, c0c4, c0c4 + HID, S, masks, layer, ts);
  if (ts_out) {   // timing mode: whole forward, no activation dump
    if (tid == 0) { ts[36] = (long long)__builtin_readcyclecounter(); ts[37] = (long long)wall_clock64(); }
    if (tid < TILE && base + tid < n) out[base + tid] = tanh_spec(pre);
    return;
  }
  __syncthreads();
  for (int i = tid; i < HID * TILE; i += 256) {
    const int f = i / TILE, ray = i % TILE;
    if (base + ray < n) out[(base + ray) * HID + f] = S.X[i];
  }
}

// ------------------------------------------------------------------------------------------ finalize
// render_depth's output assembly (renderer.py:859-878) + depth = Zdepth*calib (renderer.py:967-969)
// hint_out (host-mapped word, may be null): the first full-resolution step of THIS render that had at most tail_rays live rays over all
// views (fine_steps: none) -- where the next render of the same configuration lets the persistent tail launch take over (k_tail; the
// host reads the word without synchronising: a stale or missing value costs time, never rays).
DISTR_GLOBAL void __launch_bounds__(256) k_finalize(View V0, float* zdepth, uint8_t* mask, float* min_sdf, float* depth, int32_t* hint_out, int32_t tail_rays) {
  if (hint_out && blockIdx.x == 0 && blockIdx.y == 0 && V0.cfg.marcher != DISTR_MARCH_TRIVIAL) {
    __shared__ int32_t s_first;
    if (threadIdx.x == 0) s_first = V0.fine_steps;
    __syncthreads();
    for (int t = threadIdx.x; t < V0.fine_steps; t += 256) {
      int64_t tot = 0;
      for (int b = 0; b < V0.nviews; ++b) {
        const Consts* Cb = reinterpret_cast<const Consts*>(reinterpret_cast<const char*>(V0.C) + (int64_t)b * V0.vstride);
        tot += Cb->cnt_live[t] + Cb->cnt_sticky[t];
      }
      if (tot <= tail_rays) { atomicMin(&s_first, t); break; }     // (a thread's steps ascend: its first hit is its smallest)
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(hint_out, s_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const View V = view_at(V0, blockIdx.y);
  {
    const size_t o = (size_t)blockIdx.y * V.P;         // outputs of a batch: [nviews][P]
    if (zdepth) zdepth += o;
    if (mask) mask += o;
    if (min_sdf) min_sdf += o;
    if (depth) depth += o;
  }
  const int px = blockIdx.x * 256 + threadIdx.x;
  Consts* C = V.C;
  bool vout = false;
  if (px < V.P) {
    const LevelView& L0 = V.lv[0];
    const CamRegs cam = load_cam(C);
    float cx, cy;
    level_center(L0, px, cx, cy);
    const RayGeo g = make_ray(V.cfg.K_inv, cam.R, cx, cy);
    const Sph sp = intersect(V.cfg.radius, cam.c, cam.cdist, g.d);
    float Z = 1e11f, q;
    if (!L0.valid[px]) {
      q = sp.dist + V.cfg.threshold - V.cfg.radius;
    } else {
      const size_t P = (size_t)V.P;
      const float cd = V.cfg.clamp_dist, ratio = V.cfg.ratio;
      const bool inside = cam.cdist < V.cfg.radius;
      const float init_orig = inside ? 0.f : sp.init_raw;
      const bool pad0 = V.tk_src[px] < 0;
      const float s0 = V.tk_s[px];
      const float za0 = V.tk_za[px];
      const float m_row = V.pyramid ? (za0 - init_orig) : za0;
      float z = m_row + (1.0f - ratio) * clampf(s0, -cd, cd);
      if (C->vflags & VF_GRAD_DEPTH) {
        int dp, dn;
        early_dup_px(V, C, px, dp, dn);          // (the reference's op sequence runs over ITS selection: copies of the last row, early_dup)
        for (int j = 0; j < V.cfg.buffer_size; ++j) {
          const int k = (dn > 0 && j > dp) ? ((j - dn > dp) ? j - dn : dp) : j;
          const float sv = (V.tk_src[k * P + px] < 0) ? C->f_origin : V.tk_s[k * P + px];
          const float sc = clampf(sv, -cd, cd);
          z = z - sc * ratio;
          z = z + sc * ratio;
        }
      }
      Z = init_orig + z;
      q = pad0 ? C->f_origin : s0;
      bool v = (V.m[px] + V.init_now[px] < V.maxbound[px]) && (V.minabs[px] <= V.cfg.threshold);
      if (!V.pyramid) v = v && (V.first_sdf[px] > V.cfg.threshold);
      vout = v;
    }
    V.zdepth_s[px] = Z;
    V.mask_s[px] = vout ? 1 : 0;
    const float dp = vout ? Z * g.calib : 1e11f;
    V.depth_pre[px] = dp;
    if (zdepth) zdepth[px] = Z;
    if (mask) mask[px] = vout ? 1 : 0;
    if (min_sdf) min_sdf[px] = q;
    if (depth && !V.cfg.use_depth2normal) depth[px] = dp;
  }
  // valid-pixel list (autograd normals) + count
  const bool want_list = V.cfg.want_normal && !V.cfg.use_depth2normal;
  if (want_list) wave_append(vout, px, V.nlist, &C->cnt_normal);
  const unsigned long long ball = __ballot(vout);
  if ((threadIdx.x & 63) == 0 && ball) atomicAdd(&C->cnt_valid, __popcll(ball));
}

__device__ __forceinline__ float bg_depth(const float* dp, int i) {
  const float d = dp[i];
  return ((d > 1e5f) || (d == 0.f)) ? 0.f : d;
}

// rows whose vertical neighbours exist: not an image border row (render_utils.py:31-37 leaves those at zero) and, for a
// row band, not the band's first/last row (halo rows, cropped by the caller)
__device__ __forceinline__ bool d2n_row_inner(const View& V, int y) {
  const int gy = y + V.row0;
  return gy >= 1 && gy <= V.cfg.H - 2 && y >= 1 && y <= V.rows - 2;
}

// depth2normal (core/utils/render_utils.py:9-43) incl. its in-place zeroing of the background depth
DISTR_GLOBAL void __launch_bounds__(256) k_depth2normal(View V0, float* depth, float* normal) {
  const View V = view_at(V0, blockIdx.y);
  if (depth) depth += (size_t)blockIdx.y * V.P;
  if (normal) normal += (size_t)blockIdx.y * V.P * 3;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= V.P) return;
  const int Ww = V.cfg.W;
  const int y = i / Ww, x = i % Ww;
  const float* dp = V.depth_pre;
  const float d0 = bg_depth(dp, i);
  if (depth) depth[i] = d0;
  float n0 = 0.f, n1 = 0.f, n2 = 0.f;
  if (d0 != 0.f) {
    const bool ix = (x >= 1 && x <= Ww - 2), iy = d2n_row_inner(V, y);
    const float l = ix ? bg_depth(dp, i - 1) : 0.f, r = ix ? bg_depth(dp, i + 1) : 0.f;
    const float u = iy ? bg_depth(dp, i - Ww) : 0.f, d = iy ? bg_depth(dp, i + Ww) : 0.f;
    const float dzdx = (r - l) * V.cfg.fx / 2.0f, dzdy = (d - u) * V.cfg.fy / 2.0f;
    const float len = sqrtf(dzdx * dzdx + dzdy * dzdy + 1.0f);
    n0 = dzdx / (len + 1e-12f); n1 = dzdy / (len + 1e-12f); n2 = -1.0f / (len + 1e-12f);
  }
  if (normal) { normal[i * 3] = n0; normal[i * 3 + 1] = n1; normal[i * 3 + 2] = n2; }
}

// ------------------------------------------------------------------------------------------ backward MLP kernel
// Recomputes the decoder forward for a tile of 64 gradient-carrying samples, back-propagates coef * d f through it
// (dX chain with transposed weight fragments), and emits per tile: sum_n delta0, sum_n delta4 (for the shared-latent
// gradient, multiplied once by W_lat^T afterwards) and the camera-gradient partials; or, in POINTGRAD mode, the
// per-point sdf and d f/d xyz (decode_sdf_gradient, core/utils/decoder_utils.py:76-92).
enum { BWD_FULL = 0, BWD_POINTGRAD = 1, BWD_SAVED = 2 };   // SAVED: masks come from View::mstore, no forward recompute

struct BwdArgs {
  View V;                    // SAVED / FULL and POINTGRAD over the views' valid-pixel lists (V.nlist, counts V.C->cnt_normal)
  const Sample* samples;     // SAVED / FULL: view 0's sample list; view b's lies b * bstride bytes further (count: V.C->cnt_samples)
  float* partial;            // SAVED / FULL: view 0's [tiles][PSTRIDE] (same stride); POINTGRAD explicit with coef: [tiles][PSTRIDE]
  int64_t bstride;           // bytes between the backward workspaces of consecutive views
  int64_t n;                 // POINTGRAD explicit: number of points
  const float* xyz;          // POINTGRAD from explicit points (distr_mlp_grad); null: from the views' pixel lists + zdepth
  const float* zdepth;       // POINTGRAD from pixel lists: depth along the level-0 ray, view b at zdepth + b * zstride bytes
  int64_t zstride;
  const float* c0c4;         // POINTGRAD explicit: latent constants; null -> V.C
  const float* coef;         // POINTGRAD explicit: upstream gradient per point (decode_sdf backward); null -> 1
  float clamp;               // POINTGRAD explicit with coef: >= 0 -> zero the gradient where |f| > clamp (decode_sdf's clamp)
  int32_t split;             // SAVED / FULL: 1 = the sample list is split by tile size like a march step (bwd_range)
  float* out_sdf;            // POINTGRAD explicit: [n]   (pixel lists: V.n_sdf)
  float* out_g;              // POINTGRAD explicit: [n][3] (pixel lists: V.n_g)
  DecoderB6 B6;              // split-bf16 weight planes (k_bwd<BWD_SAVED, RB, 1>: distr_render_cfg.arith)
  DecoderH3 H3;              // split-f16 weight planes (k_bwd<BWD_SAVED, RB, 2>)
};

__device__ __forceinline__ int32_t vget(int32_t x, int b, int B) { return (B <= 1) ? x : __shfl(x, b); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Tile-size split of the gradient-sample list: whole rounds of 256 x 64 samples go to the 64-sample kernel, a remainder of
// at most 8192 samples to 32-sample tiles (two workgroups per CU: 212 us instead of a 380 us round for a handful of tiles).
// first_tile = index of this kernel's first tile in the common partial array, ntiles = total number of partial rows.
__device__ __forceinline__ void bwd_range(int64_t count, int split, int tile_size, int64_t& lo, int64_t& hi, int& first_tile, int& ntiles) {
  const int64_t full = (count / 16384) * 16384, rem = count - full;
  if (!split || rem > 8192) {
    lo = 0; hi = (tile_size == 64 || !split) ? count : 0; first_tile = 0;
    ntiles = (int)((count + tile_size - 1) / tile_size);
    if (split) ntiles = (int)((count + 63) / 64);
    return;
  }
  const int t64 = (int)(full / 64);
  ntiles = t64 + (int)((rem + 31) / 32);
  if (tile_size == 64) { lo = 0; hi = full; first_tile = 0; }
  else { lo = full; hi = count; first_tile = t64; }
}

template <int MODE, int RB, int ARITH = 0>
__global__ void __launch_bounds__(256, (RB == 1 && ARITH == 0) ? 2 : 1) k_bwd(BwdArgs A, DecoderDev D) {
  static_assert(ARITH == 0 || MODE == BWD_SAVED, "the split-bf16 backward exists for saved masks only");
  constexpr int TILE = 32 * RB;
  __shared__ typename TileSmem<RB, ARITH>::type S;
  const View& V0 = A.V;
  const int tid = threadIdx.x;
  const bool explicit_points = MODE == BWD_POINTGRAD && A.xyz;
  const int B = explicit_points ? 1 : V0.nviews;
  int tile = blockIdx.x, vb = 0;
  int64_t count, base;
  if (explicit_points) {
    count = A.n;
    base = (int64_t)tile * TILE;
    if (base >= count) return;
  } else if (MODE == BWD_POINTGRAD) {
    // valid-pixel lists of all views, each padded to whole tiles
    const int32_t c = vload(&V0.C->cnt_normal, V0.vstride, B);
    const int32_t incl = vprefix(c, B, TILE);
    const int64_t vbase = (int64_t)tile * TILE;
    if (vbase >= vtotal(incl, B)) return;
    int64_t start;
    int32_t cnt;
    vfind(c, incl, B, TILE, vbase, vb, start, cnt);
    base = vbase - start;
    count = cnt;
    if (base >= count) return;
  } else {
    // gradient samples: every view keeps its OWN tile decomposition (bwd_range of its own count: same tiles, same tile sizes,
    // same partial rows as a stand-alone backward of that view -> bit-identical gradients); this launch walks the
    // concatenation of the views' TILE-sized tiles
    const int32_t c = vload(&V0.C->cnt_samples, V0.vstride, B);
    int64_t lo, hi;
    int first, nt;
    bwd_range(c, A.split, TILE, lo, hi, first, nt);
    const int32_t mine = (int32_t)((hi - lo + TILE - 1) / TILE);      // tiles of this size in this lane's view
    const int32_t incl = vprefix(mine, B, 1);
    if (tile >= vtotal(incl, B)) return;
    int64_t start;
    int32_t cnt;
    vfind(mine, incl, B, 1, tile, vb, start, cnt);
    const int local = tile - (int)start;
    base = (int64_t)vget((int32_t)lo, vb, B) + (int64_t)local * TILE;
    count = vget((int32_t)hi, vb, B);
    tile = vget(first, vb, B) + local;                                 // partial row inside the view's own array
  }
  const View V = view_at(V0, vb);
  const Sample* samples = reinterpret_cast<const Sample*>(reinterpret_cast<const char*>(A.samples) + (int64_t)vb * A.bstride);
  float* partial = A.partial ? reinterpret_cast<float*>(reinterpret_cast<char*>(A.partial) + (int64_t)vb * A.bstride) : nullptr;
  const float* zdepth = A.zdepth ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(A.zdepth) + (int64_t)vb * A.zstride) : nullptr;
  const int32_t* pix_list = V.nlist;
  float* out_sdf = explicit_points ? A.out_sdf : V.n_sdf;
  float* out_g = explicit_points ? A.out_g : V.n_g;

  Sample sm; sm.src = -1; sm.zb = 0.f; sm.coef = 0.f; sm.flags = 0; sm.sdf = 0.f; sm.mblock = -1;
  bool valid = false;
  int64_t r = 0;
  uint32_t masks[8][4];
  float y = 0.f;
  if (MODE == BWD_SAVED) {
    // the ReLU masks of every gradient sample were saved by the march kernel: no forward recompute
    long long* mb = reinterpret_cast<long long*>(S.part);
    if (tid < TILE) {
      r = base + tid;
      valid = r < count;
      if (valid) sm = samples[r];
      y = sm.sdf;
      S.aux[tid] = valid ? sm.coef * __builtin_fmaf(-y, y, 1.0f) : 0.f;
      mb[tid] = valid ? (long long)sm.mblock : -1ll;
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
#pragma unroll
    for (int l = 0; l < 8; ++l)
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) masks[l][ob] = 0u;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const long long b = mb[32 * rb + (lane & 31)];
      if (b >= 0) load_mask_chunk(V.mstore + (size_t)b * 32, masks, rb, wave, lane >> 5);
    }
    __syncthreads();   // mb (S.part) is reused by mlp_backward
  } else {
    if (tid < TILE) {
      float p[3] = {0.f, 0.f, 0.f};
      r = base + tid;
      valid = r < count;
      if (valid) {
        if (MODE == BWD_POINTGRAD && A.xyz) {
          p[0] = A.xyz[r * 3]; p[1] = A.xyz[r * 3 + 1]; p[2] = A.xyz[r * 3 + 2];
          sm.coef = A.coef ? A.coef[r] : 1.0f;
        } else {
          if (MODE == BWD_POINTGRAD) { sm.src = pix_list[r]; sm.zb = zdepth[sm.src]; sm.coef = 1.0f; }
          else sm = samples[r];
          if (sm.src >= 0) {
            const int lvl = src_level(sm.src), ray = src_ray(sm.src);
            const CamRegs cam = load_cam(V.C);
            float cx, cy;
            level_center(level_sel(V, lvl), ray, cx, cy);
            const RayGeo g = make_ray(V.cfg.K_inv, cam.R, cx, cy);
            make_point(V.cfg.M, cam.c, g.d, sm.zb, p);
          }
        }
      }
      S.xyz[tid] = p[0]; S.xyz[TILE + tid] = p[1]; S.xyz[2 * TILE + tid] = p[2];
    }
    __syncthreads();
    const float* c0 = A.c0c4 ? A.c0c4 : V.C->c0;
    const float* c4 = A.c0c4 ? A.c0c4 + HID : V.C->c4;
    float pre = 0.f;
    if constexpr (ARITH == 0) pre = mlp_forward<RB, true>(D, c0, c4, S, masks);
    if (tid < TILE) {
      y = tanh_spec(pre);
      if (MODE == BWD_POINTGRAD && A.coef && A.clamp >= 0.f && !(fabsf(y) <= A.clamp)) sm.coef = 0.f;
      S.aux[tid] = valid ? sm.coef * __builtin_fmaf(-y, y, 1.0f) : 0.f;
    }
    __syncthreads();
  }
  float* part = (MODE != BWD_POINTGRAD || partial) ? partial + (size_t)tile * PSTRIDE : nullptr;
  if constexpr (ARITH == 0) mlp_backward<RB>(D, S, masks, part, part ? part + HID : nullptr);
  else if constexpr (ARITH == 1) mlp_backward_b6<RB>(D, A.B6, S, masks, part, part ? part + HID : nullptr);
  else mlp_backward_h3<RB>(D, A.H3, S, masks, part, part ? part + HID : nullptr);

  if (tid >= 64) return;   // wave 0 stays whole for the shuffle reduction; lanes >= TILE carry zeros
  const int rl = tid & (TILE - 1);
  const float gp0 = S.aux[TILE + rl], gp1 = S.aux[2 * TILE + rl], gp2 = S.aux[3 * TILE + rl];
  if (MODE == BWD_POINTGRAD) {
    if (valid) {
      if (out_sdf) out_sdf[r] = y;
      if (out_g) { out_g[r * 3] = gp0; out_g[r * 3 + 1] = gp1; out_g[r * 3 + 2] = gp2; }
    }
    return;
  }
  // sample point -> camera: p = M^T (cam_pos + ray * zb), zb detached (renderer.py:202-223)
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  if (valid && sm.src >= 0 && (sm.flags & 1)) {
    const int lvl = src_level(sm.src), ray = src_ray(sm.src);
    const CamRegs cam = load_cam(V.C);
    float cx, cy;
    level_center(level_sel(V, lvl), ray, cx, cy);
    const RayGeo g = make_ray(V.cfg.K_inv, cam.R, cx, cy);
    const float* M = V.cfg.M;
    float gq[3], gd[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) gq[j] = M[j * 3] * gp0 + M[j * 3 + 1] * gp1 + M[j * 3 + 2] * gp2;
#pragma unroll
    for (int j = 0; j < 3; ++j) { acc[9 + j] = gq[j]; gd[j] = sm.zb * gq[j]; }
    const float e = g.rn + 1e-12f;
    const float dot = gd[0] * g.r[0] + gd[1] * g.r[1] + gd[2] * g.r[2];
    const float hh[3] = {g.hx, g.hy, g.hz};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float gr = gd[i] / e - (g.rn > 0.f ? dot * g.r[i] / (g.rn * e * e) : 0.f);
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[j * 3 + i] = hh[j] * gr;
    }
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) part[2 * HID + i] = acc[i];
  }
}

// ------------------------------------------------------------------------------------------ normals (autograd path)
// render_normal (renderer.py:880-910) epilogue: n = normalize(3 * grad f) (torch-1.1 grad_outputs quirk,
// decoder_utils.py:84), t = M n; render(): out = flipx(R t) (renderer.py:977-980)
DISTR_GLOBAL void __launch_bounds__(256) k_normal_finish(View V0, float* normal_hw3, float* normal_3xP, int write_nrm_t) {
  const View V = view_at(V0, blockIdx.y);
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= V.C->cnt_normal) return;
  const size_t P = (size_t)V.P;
  if (normal_hw3) normal_hw3 += (size_t)blockIdx.y * P * 3;      // outputs of a batch: [nviews][...]
  if (normal_3xP) normal_3xP += (size_t)blockIdx.y * P * 3;
  const int32_t* pix_list = V.nlist;
  const float* n_sdf = V.n_sdf;
  const float* n_g = V.n_g;
  float* nrm_t = write_nrm_t ? V.nrm_t : nullptr;
  const int px = pix_list[r];
  const bool inclamp = fabsf(n_sdf[r]) <= V.cfg.clamp_dist;
  float g[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) g[j] = inclamp ? 3.0f * n_g[r * 3 + j] : 0.f;
  if (V.cfg.normalize_normal) {
    const float len = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
#pragma unroll
    for (int j = 0; j < 3; ++j) g[j] = g[j] / (len + 1e-12f);
  }
  const float* M = V.cfg.M_normal;     // renderer.py:899: the normals take the constructor's matrix whatever use_transform says
  float t[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = M[i * 3] * g[0] + M[i * 3 + 1] * g[1] + M[i * 3 + 2] * g[2];
  if (nrm_t) { nrm_t[px * 3] = t[0]; nrm_t[px * 3 + 1] = t[1]; nrm_t[px * 3 + 2] = t[2]; }
  if (normal_3xP) { normal_3xP[px] = t[0]; normal_3xP[P + px] = t[1]; normal_3xP[2 * P + px] = t[2]; }
  if (normal_hw3) {
    const float* R = V.C->R;
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = R[i * 3] * t[0] + R[i * 3 + 1] * t[1] + R[i * 3 + 2] * t[2];
    normal_hw3[px * 3] = -o[0]; normal_hw3[px * 3 + 1] = o[1]; normal_hw3[px * 3 + 2] = o[2];
  }
}

// valid-pixel list of every view from a caller-provided mask [nviews][P] (render_normal)
DISTR_GLOBAL void __launch_bounds__(256) k_mask_list(View V0, const uint8_t* mask) {
  const View V = view_at(V0, blockIdx.y);
  const int px = blockIdx.x * 256 + threadIdx.x;
  wave_append(px < V.P && mask[(size_t)blockIdx.y * V.P + px] != 0, px, V.nlist, &V.C->cnt_normal);
}

// ------------------------------------------------------------------------------------------ backward: image side
__device__ __forceinline__ void d2n_gv(const View& V, const float* g_normal, int i, float& gv0, float& gv1) {
  // gradient wrt (dzdx, dzdy) of pixel i of depth2normal's normalised vector
  const int Ww = V.cfg.W;
  const int y = i / Ww, x = i % Ww;
  const float* dp = V.depth_pre;
  const bool ix = (x >= 1 && x <= Ww - 2), iy = d2n_row_inner(V, y);
  const float l = ix ? bg_depth(dp, i - 1) : 0.f, r = ix ? bg_depth(dp, i + 1) : 0.f;
  const float u = iy ? bg_depth(dp, i - Ww) : 0.f, d = iy ? bg_depth(dp, i + Ww) : 0.f;
  const float v0 = (r - l) * V.cfg.fx / 2.0f, v1 = (d - u) * V.cfg.fy / 2.0f;
  const float len = sqrtf(v0 * v0 + v1 * v1 + 1.0f), e = len + 1e-12f;
  const float g0 = g_normal[i * 3], g1 = g_normal[i * 3 + 1], g2 = g_normal[i * 3 + 2];
  const float dot = g0 * v0 + g1 * v1 - g2;
  gv0 = g0 / e - dot * v0 / (len * e * e);
  gv1 = g1 / e - dot * v1 / (len * e * e);
}

__device__ __forceinline__ void ray_backward_acc(const RayGeo& g, const float* gd, float* acc /*gR[9]*/) {
  const float e = g.rn + 1e-12f;
  const float dot = gd[0] * g.r[0] + gd[1] * g.r[1] + gd[2] * g.r[2];
  const float hh[3] = {g.hx, g.hy, g.hz};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float gr = gd[i] / e - (g.rn > 0.f ? dot * g.r[i] / (g.rn * e * e) : 0.f);
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[j * 3 + i] += hh[j] * gr;
  }
}

// Turns the upstream image gradients into (i) per-sample coefficients on the selected rows (ii) camera-gradient
// terms that do not pass through the decoder (rays missing the sphere, renderer.py:863; the explicit R*n product,
// renderer.py:978). SURVEY.md Appendix A.6 steps 1-3, 5.
// Deterministic two-pass form: pass 0 (EMIT=false) counts the samples of every 256-pixel block and stores the block's
// camera / pad partial sums; k_bwd_scan turns the counts into offsets and adds the partials in block order; pass 1
// (EMIT=true) writes the samples at their fixed positions (block, row index k, thread). No atomics: the sample order
// and every reduction order are reproducible, so gradients are bit-reproducible run to run.
struct BwdBlocks { int32_t* cnt; int32_t* off; float* acc; };   // [nblk], [nblk], [nblk][16]

// Backward workspace of view 0 of a batch; view b's arrays lie b * bstride bytes further (bws_at).
struct BwdWs { Sample* samples; float* partial; float* chunk_part; BwdBlocks BB; int64_t bstride; };
__device__ __forceinline__ BwdWs bws_at(const BwdWs& W0, int b) {
  BwdWs W = W0;
  const int64_t d = (int64_t)b * W0.bstride;
  adv(W.samples, d); adv(W.partial, d); adv(W.chunk_part, d); adv(W.BB.cnt, d); adv(W.BB.off, d); adv(W.BB.acc, d);
  return W;
}

// grid (blocks of 256 pixels, nviews); upstream gradient images of a batch are [nviews][P] ([nviews][P][3] for normals)
template <bool EMIT>
__global__ void __launch_bounds__(256) k_bwd_prep(View V0, const float* g_zdepth, const float* g_min_sdf,
                                                  const float* g_depth, const float* g_normal, BwdWs W0) {
  __shared__ int32_t s_cnt[MAX_BS][4];
  __shared__ float s_acc[4][16];
  const View V = view_at(V0, blockIdx.y);
  const BwdWs Wv = bws_at(W0, blockIdx.y);
  Sample* samples = Wv.samples;
  const BwdBlocks B = Wv.BB;
  {
    const size_t o = (size_t)blockIdx.y * V.P;
    if (g_zdepth) g_zdepth += o;
    if (g_min_sdf) g_min_sdf += o;
    if (g_depth) g_depth += o;
    if (g_normal) g_normal += 3 * o;
  }
  const int px = blockIdx.x * 256 + threadIdx.x;
  Consts* C = V.C;
  const int vflags = C->vflags;
  const bool grad_depth = (vflags & VF_GRAD_DEPTH) != 0, grad_mask = (vflags & VF_GRAD_MASK) != 0, grad_camera = (vflags & VF_GRAD_CAMERA) != 0;
  const LevelView& L0 = V.lv[0];
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  float pad_acc = 0.f;
  const int bs = V.cfg.buffer_size;
  const bool act = px < V.P;
  const bool in = act && L0.valid[px];
  float gz = 0.f, gq = 0.f;
  if (act) {
    gq = g_min_sdf ? g_min_sdf[px] : 0.f;
    const CamRegs cam = load_cam(C);
    float cx, cy;
    level_center(L0, px, cx, cy);
    const RayGeo g = make_ray(V.cfg.K_inv, cam.R, cx, cy);
    if (!in) {
      if (gq != 0.f) {
        const Sph sp = intersect(V.cfg.radius, cam.c, cam.cdist, g.d);
        if (sp.dist > 0.f) {
          float gv[3], gd[3];
#pragma unroll
          for (int j = 0; j < 3; ++j) gv[j] = gq * sp.v[j] / sp.dist;
          const float gvd = gv[0] * g.d[0] + gv[1] * g.d[1] + gv[2] * g.d[2];
#pragma unroll
          for (int j = 0; j < 3; ++j) { acc[9 + j] += gv[j] - g.d[j] * gvd; gd[j] = -cam.c[j] * gvd - sp.ptq * gv[j]; }
          ray_backward_acc(g, gd, acc);
        }
      }
    } else {
      const bool m = V.mask_s[px] != 0;
      if (grad_depth) {
        if (g_zdepth) gz += g_zdepth[px];
        if (m) {
          float gd_eff = g_depth ? g_depth[px] : 0.f;
          if (V.cfg.want_normal && V.cfg.use_depth2normal) {
            if (bg_depth(V.depth_pre, px) == 0.f) {
              gd_eff = 0.f;
            } else if (g_normal) {
              const int Ww = V.cfg.W;
              const int y = px / Ww, x = px % Ww;
              float a0, a1, add = 0.f;
              if (x - 1 >= 1 && x - 1 <= Ww - 2 && bg_depth(V.depth_pre, px - 1) != 0.f) { d2n_gv(V, g_normal, px - 1, a0, a1); add += a0 * V.cfg.fx / 2.0f; }
              if (x + 1 >= 1 && x + 1 <= Ww - 2 && bg_depth(V.depth_pre, px + 1) != 0.f) { d2n_gv(V, g_normal, px + 1, a0, a1); add -= a0 * V.cfg.fx / 2.0f; }
              if (y >= 1 && d2n_row_inner(V, y - 1) && bg_depth(V.depth_pre, px - Ww) != 0.f) { d2n_gv(V, g_normal, px - Ww, a0, a1); add += a1 * V.cfg.fy / 2.0f; }
              if (y + 1 < V.rows && d2n_row_inner(V, y + 1) && bg_depth(V.depth_pre, px + Ww) != 0.f) { d2n_gv(V, g_normal, px + Ww, a0, a1); add -= a1 * V.cfg.fy / 2.0f; }
              gd_eff += add;
            }
          }
          gz += gd_eff * g.calib;
        }
      }
      if (m && V.cfg.want_normal && !V.cfg.use_depth2normal && g_normal) {
        const float go[3] = {-g_normal[px * 3], g_normal[px * 3 + 1], g_normal[px * 3 + 2]};
        const float* t = V.nrm_t + (size_t)px * 3;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int b = 0; b < 3; ++b) acc[a * 3 + b] += go[a] * t[b];
      }
    }
  }
  const size_t P = (size_t)V.P;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // per-row coefficient of this pixel (0: no sample); pad rows accumulate into pad_acc
  int dup_p = -1, dup_n = 0;       // early break below buffer_size steps: the ray's last row is selected 1 + dup_n times, the last dup_n rows drop out (early_dup)
  if (in) early_dup_px(V, C, px, dup_p, dup_n);
  auto row_coef = [&](int k, int32_t& src, float& sv) -> float {
    if (!in) return 0.f;
    src = V.tk_src[k * P + px];
    sv = src < 0 ? C->f_origin : V.tk_s[k * P + px];
    float c = 0.f;
    const bool dropped = dup_n > 0 && k > dup_p && k + dup_n > bs - 1;
    if (grad_depth && gz != 0.f && !dropped)
      c += V.cfg.ratio * gz * (fabsf(sv) <= V.cfg.clamp_dist ? 1.f : 0.f) * ((k == dup_p) ? (float)(1 + dup_n) : 1.f);
    if (k == 0 && grad_mask) c += gq;
    return c;
  };
  for (int k = 0; k < bs; ++k) {
    int32_t src = -1; float sv = 0.f;
    const float c = row_coef(k, src, sv);
    const bool emit = (c != 0.f) && src >= 0;
    if (!EMIT && c != 0.f && src < 0) pad_acc += c;
    const unsigned long long ball = __ballot(emit);
    if (lane == 0) s_cnt[k][wave] = __popcll(ball);
  }
  __syncthreads();
  if (!EMIT) {
    if (threadIdx.x == 0) {
      int tot = 0;
      for (int k = 0; k < bs; ++k) tot += s_cnt[k][0] + s_cnt[k][1] + s_cnt[k][2] + s_cnt[k][3];
      B.cnt[blockIdx.x] = tot;
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float v = wave_sum(acc[i]); if (lane == 0) s_acc[wave][i] = v; }
    { const float v = wave_sum(pad_acc); if (lane == 0) s_acc[wave][12] = v; }
    __syncthreads();
    if (threadIdx.x < 13) B.acc[blockIdx.x * 16 + threadIdx.x] = ((s_acc[0][threadIdx.x] + s_acc[1][threadIdx.x]) + s_acc[2][threadIdx.x]) + s_acc[3][threadIdx.x];
    return;
  }
  int base = B.off[blockIdx.x];
  for (int k = 0; k < bs; ++k) {
    int32_t src = -1; float sv = 0.f;
    const float c = row_coef(k, src, sv);
    const bool emit = (c != 0.f) && src >= 0;
    const unsigned long long ball = __ballot(emit);
    int wbase = base;
    for (int w = 0; w < wave; ++w) wbase += s_cnt[k][w];
    if (emit) {
      // (a fine row's src is its march step: the pixel is this one)
      Sample sm; sm.src = (src_level(src) == 0) ? px : src; sm.zb = V.tk_zb[k * P + px]; sm.coef = c; sm.sdf = sv; sm.mblock = -1; sm.pad0 = 0; sm.pad1 = 0;
      if (V.save_masks) {
        const int lv = src_level(src);
        sm.mblock = (lv == 0) ? (int32_t)((int64_t)px * (bs + 1) + V.tk_slot[k * P + px])
                              : (int32_t)(V.mfine + moff_sel(V, lv) + (int64_t)src_step(src) * level_sel(V, lv).n + src_ray(src));
      }
      const bool fine_row = src_level(src) == 0;
      sm.flags = (fine_row && !grad_camera && V.cfg.marcher != DISTR_MARCH_TRIVIAL) ? 0 : 1;
      samples[wbase + ballot_rank(ball)] = sm;
    }
    base += s_cnt[k][0] + s_cnt[k][1] + s_cnt[k][2] + s_cnt[k][3];
  }
}

// exclusive scan of the per-block sample counts, ordered sum of the per-block partials, pad sample (one block)
DISTR_GLOBAL void __launch_bounds__(256) k_bwd_scan(View V0, BwdWs W0, int nblk) {   // one block per view
  const View V = view_at(V0, blockIdx.x);
  const BwdWs Wv = bws_at(W0, blockIdx.x);
  const BwdBlocks B = Wv.BB;
  Sample* samples = Wv.samples;
  __shared__ int32_t s_part[256];
  __shared__ float s_accp[256][13];
  const int t = threadIdx.x;
  const int per = (nblk + 255) / 256;
  const int b0 = t * per, b1 = min(nblk, b0 + per);
  int sum = 0;
  float a[13];
#pragma unroll
  for (int i = 0; i < 13; ++i) a[i] = 0.f;
  for (int b = b0; b < b1; ++b) {
    sum += B.cnt[b];
#pragma unroll
    for (int i = 0; i < 13; ++i) a[i] += B.acc[b * 16 + i];
  }
  s_part[t] = sum;
#pragma unroll
  for (int i = 0; i < 13; ++i) s_accp[t][i] = a[i];
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int c = s_part[i]; s_part[i] = run; run += c; }
    V.C->cnt_samples = run;
  }
  __syncthreads();
  int off = s_part[t];
  for (int b = b0; b < b1; ++b) { B.off[b] = off; off += B.cnt[b]; }
  if (t < 13) {
    float v = 0.f;
    for (int i = 0; i < 256; ++i) v += s_accp[i][t];
    if (t < 12) V.C->cam_acc[t] = v; else V.C->pad_coef = v;
  }
  __syncthreads();
  if (t == 0 && V.C->pad_coef != 0.f) {
    // all padded rows sample the origin (points = 0, renderer.py:539): one combined sample, no camera dependence
    Sample sm; sm.src = -1; sm.zb = 0.f; sm.coef = V.C->pad_coef; sm.flags = 0; sm.sdf = V.C->f_origin; sm.mblock = (int32_t)V.morigin;
    sm.pad0 = 0; sm.pad1 = 0;
    samples[V.C->cnt_samples] = sm;
    V.C->cnt_samples = V.C->cnt_samples + 1;
  }
}

// column sums of the tile partials over one chunk of tiles -> chunk_part[chunk][col] (fixed order, no atomics)
DISTR_GLOBAL void __launch_bounds__(256) k_bwd_reduce(View V0, BwdWs W0, int chunk, int tile) {   // grid (columns, chunks, nviews)
  const View V = view_at(V0, blockIdx.z);
  const BwdWs Wv = bws_at(W0, blockIdx.z);
  const float* partial = Wv.partial;
  float* chunk_part = Wv.chunk_part;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= 2 * HID + 12) return;
  int ntiles = (V.C->cnt_samples + tile - 1) / tile;
  if (tile < 0) { int64_t lo, hi; int first; bwd_range(V.C->cnt_samples, 1, 64, lo, hi, first, ntiles); }   // tile < 0: split list
  const int t0 = blockIdx.y * chunk, t1 = min(ntiles, t0 + chunk);
  float s = 0.f;
  for (int t = t0; t < t1; ++t) s += partial[(size_t)t * PSTRIDE + col];
  chunk_part[(size_t)blockIdx.y * PSTRIDE + col] = s;
}

// decode_sdf backward (explicit points): ordered column sums of the tile partials, then g_latent as in k_bwd_final
DISTR_GLOBAL void __launch_bounds__(256) k_points_latent_grad(const float* partial, int ntiles, DecoderDev D, float* g_latent) {
  __shared__ float red[2 * HID];
  for (int col = threadIdx.x; col < 2 * HID; col += 256) {
    float s = 0.f;
    for (int t = 0; t < ntiles; ++t) s += partial[(size_t)t * PSTRIDE + col];
    red[col] = s;
  }
  __syncthreads();
  const int nlat = D.nlat;                           // 256 (SDF decoder) or 256 + color_size (colour decoder)
  for (int k = threadIdx.x; k < nlat; k += 256) {
    float a = 0.f;
#pragma unroll 16
    for (int o = 0; o < HID; ++o) a += D.W0lat[(size_t)o * nlat + k] * red[o] + D.W4lat[(size_t)o * nlat + k] * red[HID + o];
    g_latent[k] = a;
  }
}

// g_latent = W0lat^T sum(delta0) + W4lat^T sum(delta4); camera chain cam_pos = -R^T T (renderer.py:180-188)
DISTR_GLOBAL void __launch_bounds__(256) k_bwd_final(View V0, DecoderDev D, BwdWs W0, int nchunks_max, int chunk, int tile,
                                                   float* g_latent, float* g_R, float* g_T) {   // one block per view; outputs [nviews][256 | 9 | 3]
  const View V = view_at(V0, blockIdx.x);
  const float* chunk_part = bws_at(W0, blockIdx.x).chunk_part;
  if (g_latent) g_latent += (size_t)blockIdx.x * LAT;
  if (g_R) g_R += (size_t)blockIdx.x * 9;
  if (g_T) g_T += (size_t)blockIdx.x * 3;
  const int k = threadIdx.x;
  Consts* C = V.C;
  {
    int ntiles = (C->cnt_samples + tile - 1) / tile;
    if (tile < 0) { int64_t lo, hi; int first; bwd_range(C->cnt_samples, 1, 64, lo, hi, first, ntiles); }
    const int nchunks = min(nchunks_max, (ntiles + chunk - 1) / chunk);
    for (int col = k; col < 2 * HID + 12; col += 256) {
      float s = 0.f;
#pragma unroll 8
      for (int c = 0; c < nchunks; ++c) s += chunk_part[(size_t)c * PSTRIDE + col];   // ordered; the loads batch 8 deep
      C->red[col] = s;
    }
  }
  __syncthreads();
  const float* sd0 = C->red;
  const float* sd4 = C->red + HID;
  float a = 0.f;
#pragma unroll 16
  for (int o = 0; o < HID; ++o) a += D.W0lat[o * LAT + k] * sd0[o] + D.W4lat[o * LAT + k] * sd4[o];   // loads batch 16 deep
  if (g_latent) g_latent[k] = a;
  if (k < 12) {
    float gc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) gc[i] = C->red[2 * HID + 9 + i] + C->cam_acc[9 + i];
    if (k < 9) {
      const int j = k / 3, i = k % 3;
      if (g_R) g_R[k] = C->red[2 * HID + k] + C->cam_acc[k] - C->T[j] * gc[i];
    } else {
      const int j = k - 9;
      if (g_T) g_T[j] = -(C->R[j * 3] * gc[0] + C->R[j * 3 + 1] * gc[1] + C->R[j * 3 + 2] * gc[2]);
    }
  }
}

}  // namespace distr
