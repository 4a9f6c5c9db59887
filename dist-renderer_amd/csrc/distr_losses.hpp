// distr_losses.hpp -- the image-space consumers right after the hot path, fused:
//
//   f2  warp / photometric loss of SDFRenderer_warp.render_warp   core/sdfrenderer/renderer_warp.py:18-101
//       (back-project view-1 depth, project into view 2, bilinear depth + colour sampling of
//        grid_sample_on_img core/utils/loss_utils.py:9-25, depth-consistency test, L1 colour) forward + backward to
//       the view-1 depth and both cameras
//   f3  single-view losses of compute_all_loss                     core/utils/loss_utils.py:59-172
//       (silhouette hinges on the min-|sdf| sample, L1 depth, cosine normal) forward + backward to the render outputs
//
// These are HBM-bound element-wise passes (tens of bytes per pixel, one read each): one thread per pixel, coalesced
// row-major reads, and every reduction is two-level in a fixed order (wave shuffle -> LDS -> per-block partial ->
// one ordered pass over the partials), so the losses and gradients are bit-reproducible run to run. What they replace
// is ~40 (f3) / ~60 (f2) small ATen kernels per iteration plus the host synchronisations of boolean-mask indexing.
#pragma once
#include "distr_kernels.hpp"

namespace distr {

// ------------------------------------------------------------------------------------------ block reduction
// Sums NV values per thread over a 256-thread block; the result is valid on thread 0. Fixed order.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* lds /*[4*NV]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float x = v[k];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) x += __shfl_down(x, o);
    if (lane == 0) lds[wave * NV + k] = x;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = ((lds[k] + lds[NV + k]) + lds[2 * NV + k]) + lds[3 * NV + k];
  }
  __syncthreads();
}

// ordered sum of [nblk][nv] partials -> out[nv] (one block; thread k owns column k)
DISTR_GLOBAL void __launch_bounds__(64) k_sum_partials(const float* __restrict__ partial, int nblk, int nv, float* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= nv) return;
  float acc = 0.f;
  for (int b = 0; b < nblk; ++b) acc += partial[(size_t)b * nv + k];
  out[k] = acc;
}

// ========================================================================================== f3: single-view losses
struct SingleLossArgs {
  int32_t P;
  const float* depth;      // [P]    render(): depth
  const float* normal;     // [P][3] render(): normal
  const uint8_t* mask;     // [P]    render(): valid mask
  const float* min_sdf;    // [P]    render(): min_abs_query
  const float* gt_depth;   // [P] or null
  const float* gt_normal;  // [P][3] or null
  const uint8_t* gt_mask;  // [P]
  float threshold;
};

struct SinglePix { float v[4]; float c[4]; bool in_gt, in_out, in_d, in_n; float q, dd, nn, dot, bh[3]; };

// the four per-pixel terms (loss_utils.py:75-99, 118-131, 155-171): value v[k] and membership c[k] of
// k = 0 mask_gt (gt \ out), 1 mask_out (out \ gt), 2 depth, 3 normal
__device__ __forceinline__ SinglePix single_terms(const SingleLossArgs& A, int i) {
  SinglePix r;
#pragma unroll
  for (int k = 0; k < 4; ++k) { r.v[k] = 0.f; r.c[k] = 0.f; }
  const bool m = A.mask[i] != 0, g = A.gt_mask[i] != 0;
  r.q = A.min_sdf[i];
  r.in_gt = g && !m;
  r.in_out = m && !g;
  if (r.in_gt) { r.v[0] = fmaxf(r.q - A.threshold, 0.f); r.c[0] = 1.f; }
  if (r.in_out) { r.v[1] = fmaxf(-r.q + A.threshold, 0.f); r.c[1] = 1.f; }
  r.in_d = false; r.in_n = false; r.dd = 0.f; r.nn = 0.f; r.dot = 0.f;
  if (A.gt_depth) {
    const float gd = A.gt_depth[i];
    r.in_d = m && g && (gd > 0.f) && (gd < 1e5f);
    if (r.in_d) { r.dd = A.depth[i] - gd; r.v[2] = fabsf(r.dd); r.c[2] = 1.f; }
  }
  if (A.gt_normal) {
    const float n0 = A.normal[i * 3], n1 = A.normal[i * 3 + 1], n2 = A.normal[i * 3 + 2];
    r.nn = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
    r.in_n = m && g && (r.nn != 0.f);
    if (r.in_n) {
      const float b0 = A.gt_normal[i * 3], b1 = A.gt_normal[i * 3 + 1], b2 = A.gt_normal[i * 3 + 2];
      const float bn = sqrtf(b0 * b0 + b1 * b1 + b2 * b2);
      r.bh[0] = b0 / (bn + 1e-12f); r.bh[1] = b1 / (bn + 1e-12f); r.bh[2] = b2 / (bn + 1e-12f);
      const float e = r.nn + 1e-12f;
      r.dot = (n0 / e) * r.bh[0] + (n1 / e) * r.bh[1] + (n2 / e) * r.bh[2];
      r.v[3] = -r.dot; r.c[3] = 1.f;
    }
  }
  return r;
}

DISTR_GLOBAL void __launch_bounds__(256) k_single_loss_partial(SingleLossArgs A, float* __restrict__ partial /*[nblk][8]*/) {
  __shared__ float lds[32];
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.f;
  if (i < A.P) {
    const SinglePix r = single_terms(A, i);
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[k] = r.v[k]; v[4 + k] = r.c[k]; }
  }
  block_sum<8>(v, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) partial[(size_t)blockIdx.x * 8 + k] = v[k];
  }
}

// sums[0..3], counts[4..7] -> losses[k] = mean over the term's pixel set, 0 for an empty set (loss_utils.py:82-84 etc.)
DISTR_GLOBAL void __launch_bounds__(64) k_single_loss_final(const float* __restrict__ partial, int nblk, float* __restrict__ out /*[8]: losses[4], counts[4]*/) {
  __shared__ float s[8];
  const int k = threadIdx.x;
  if (k < 8) {
    float acc = 0.f;
    for (int b = 0; b < nblk; ++b) acc += partial[(size_t)b * 8 + k];
    s[k] = acc;
  }
  __syncthreads();
  if (k < 4) {
    const float cnt = s[4 + k];
    out[k] = cnt > 0.f ? s[k] / cnt : 0.f;
    out[4 + k] = cnt;
  }
}

// upstream gradients g[4] of the four means -> gradients of depth / normal / min_sdf images
DISTR_GLOBAL void __launch_bounds__(256) k_single_loss_bwd(SingleLossArgs A, const float* __restrict__ lc /*[8] losses, counts*/,
                                                         const float* __restrict__ g /*[4]*/, float* __restrict__ g_depth,
                                                         float* __restrict__ g_normal, float* __restrict__ g_min_sdf) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= A.P) return;
  const SinglePix r = single_terms(A, i);
  float gq = 0.f, gd = 0.f, gn[3] = {0.f, 0.f, 0.f};
  if (r.in_gt && (r.q - A.threshold) > 0.f) gq += g[0] / lc[4];
  if (r.in_out && (-r.q + A.threshold) > 0.f) gq -= g[1] / lc[5];
  if (r.in_d) gd = (r.dd > 0.f ? 1.f : (r.dd < 0.f ? -1.f : 0.f)) * (g[2] / lc[6]);
  if (r.in_n) {
    // loss_i = -(n / (|n| + eps)) . bh ;  d/dn = -(bh / e - (n . bh) n / (|n| e^2))
    const float e = r.nn + 1e-12f, w = g[3] / lc[7];
    const float n[3] = {A.normal[i * 3], A.normal[i * 3 + 1], A.normal[i * 3 + 2]};
    const float nb = n[0] * r.bh[0] + n[1] * r.bh[1] + n[2] * r.bh[2];
#pragma unroll
    for (int a = 0; a < 3; ++a) gn[a] = -w * (r.bh[a] / e - nb * n[a] / (r.nn * e * e));
  }
  if (g_min_sdf) g_min_sdf[i] = gq;
  if (g_depth) g_depth[i] = gd;
  if (g_normal) { g_normal[i * 3] = gn[0]; g_normal[i * 3 + 1] = gn[1]; g_normal[i * 3 + 2] = gn[2]; }
}

// ========================================================================================== f2: warp loss
struct WarpArgs {
  int32_t H, W;
  float K[9], K_inv[9];
  float thres_depth;
  const float* z1;         // [P] Zdepth of view 1 (carries the gradient)
  const uint8_t* m1;       // [P] valid mask of view 1
  const float* z2;         // [P] Zdepth of view 2
  const float* img1;       // [P][3]
  const float* img2;       // [P][3]
  const float* R1; const float* T1; const float* R2; const float* T2;   // device, row-major
};

// grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) at pixel coordinates (u, v), through the same
// normalise / un-normalise round trip as grid_sample_on_img (loss_utils.py:17-24)
struct Bilin { int x0, y0; float tx, ty; bool ok; };
__device__ __forceinline__ Bilin bilin_setup(float u, float v, int W, int H) {
  Bilin b;
  const float gx = 2.0f * u / (float)max(W - 1, 1) - 1.0f, gy = 2.0f * v / (float)max(H - 1, 1) - 1.0f;
  const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
  b.ok = (ix > -2.0f) && (ix < (float)W + 1.0f) && (iy > -2.0f) && (iy < (float)H + 1.0f);   // false also for NaN / inf
  const float fx = b.ok ? floorf(ix) : 0.f, fy = b.ok ? floorf(iy) : 0.f;
  b.x0 = (int)fx; b.y0 = (int)fy;
  b.tx = ix - fx; b.ty = iy - fy;
  return b;
}
__device__ __forceinline__ bool inb(int x, int y, int W, int H) { return x >= 0 && x < W && y >= 0 && y < H; }

struct WarpPix {
  bool valid, keep;
  RayGeo g;
  float z, pt[3], q[3], pr[3], u, v;
  Bilin b;
  float c2[3];
};

__device__ __forceinline__ WarpPix warp_pixel(const WarpArgs& A, int i, const float* R1, const float* c1, const float* R2, const float* T2) {
  WarpPix w;
  w.valid = A.m1[i] != 0;
  w.keep = false;
  if (!w.valid) return w;
  const int W = A.W, H = A.H;
  w.g = make_ray(A.K_inv, R1, (float)(i % W), (float)(i / W));
  w.z = A.z1[i];
#pragma unroll
  for (int a = 0; a < 3; ++a) w.pt[a] = w.g.d[a] * w.z + c1[a];
#pragma unroll
  for (int a = 0; a < 3; ++a) w.q[a] = R2[a * 3] * w.pt[0] + R2[a * 3 + 1] * w.pt[1] + R2[a * 3 + 2] * w.pt[2] + T2[a];
#pragma unroll
  for (int a = 0; a < 3; ++a) w.pr[a] = A.K[a * 3] * w.q[0] + A.K[a * 3 + 1] * w.q[1] + A.K[a * 3 + 2] * w.q[2];
  w.u = w.pr[0] / w.pr[2]; w.v = w.pr[1] / w.pr[2];
  w.b = bilin_setup(w.u, w.v, W, H);
  // depth of view 2 at the projection: Zdepth2 * calib_map, bilinear (renderer_warp.py:63-72)
  float d2 = 0.f;
  if (w.b.ok) {
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int x = w.b.x0 + dx, y = w.b.y0 + dy;
        if (inb(x, y, W, H)) {
          const float hx = A.K_inv[0] * x + A.K_inv[1] * y + A.K_inv[2], hy = A.K_inv[3] * x + A.K_inv[4] * y + A.K_inv[5],
                      hz = A.K_inv[6] * x + A.K_inv[7] * y + A.K_inv[8];
          const float calib = hz / (sqrtf(hx * hx + hy * hy + hz * hz) + 1e-12f);
          const float wgt = (dx ? w.b.tx : 1.0f - w.b.tx) * (dy ? w.b.ty : 1.0f - w.b.ty);
          d2 += A.z2[y * W + x] * calib * wgt;
        }
      }
  }
  const float err = w.pr[2] - d2;
  w.keep = (err * err) < A.thres_depth;
  if (w.keep) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) w.c2[ch] = 0.f;
    if (w.b.ok) {
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int x = w.b.x0 + dx, y = w.b.y0 + dy;
          if (inb(x, y, W, H)) {
            const float wgt = (dx ? w.b.tx : 1.0f - w.b.tx) * (dy ? w.b.ty : 1.0f - w.b.ty);
            const float* p = A.img2 + (size_t)(y * W + x) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) w.c2[ch] += p[ch] * wgt;
          }
        }
    }
  }
  return w;
}

__device__ __forceinline__ void load_cam_pair(const WarpArgs& A, float* R1, float* c1, float* R2, float* T2, float* T1) {
#pragma unroll
  for (int k = 0; k < 9; ++k) { R1[k] = A.R1[k]; R2[k] = A.R2[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) { T1[k] = A.T1[k]; T2[k] = A.T2[k]; }
#pragma unroll
  for (int j = 0; j < 3; ++j) c1[j] = -(R1[0 * 3 + j] * T1[0] + R1[1 * 3 + j] * T1[1] + R1[2 * 3 + j] * T1[2]);
}

// forward: per pixel keep flag, the two colour images (detached visualisation outputs), and per-block partials of
// { sum |c1 - c2|, #kept, #valid(view 1) }
DISTR_GLOBAL void __launch_bounds__(256) k_warp_fwd(WarpArgs A, uint8_t* __restrict__ keep, float* __restrict__ color1,
                                                  float* __restrict__ color2, float* __restrict__ partial /*[nblk][3]*/) {
  __shared__ float lds[12];
  const int P = A.H * A.W;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v[3] = {0.f, 0.f, 0.f};
  if (i < P) {
    float R1[9], c1[3], R2[9], T2[3], T1[3];
    load_cam_pair(A, R1, c1, R2, T2, T1);
    const WarpPix w = warp_pixel(A, i, R1, c1, R2, T2);
    float o1[3] = {0.f, 0.f, 0.f}, o2[3] = {0.f, 0.f, 0.f};
    if (w.valid) v[2] = 1.f;
    if (w.keep) {
      v[1] = 1.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) { o1[ch] = A.img1[(size_t)i * 3 + ch]; o2[ch] = w.c2[ch]; v[0] += fabsf(o1[ch] - o2[ch]); }
    }
    if (keep) keep[i] = w.keep ? 1 : 0;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      if (color1) color1[(size_t)i * 3 + ch] = o1[ch];
      if (color2) color2[(size_t)i * 3 + ch] = o2[ch];
    }
  }
  block_sum<3>(v, lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) partial[(size_t)blockIdx.x * 3 + k] = v[k];
  }
}

// out[0] = loss_color = mean |c1 - c2| over kept pixels x 3 channels (NaN for an empty kept set, as torch.mean of an
// empty tensor; 0 when view 1 has no valid pixel, renderer_warp.py:111-113), out[1] = #kept, out[2] = #valid
DISTR_GLOBAL void __launch_bounds__(64) k_warp_final(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
  __shared__ float s[3];
  const int k = threadIdx.x;
  if (k < 3) {
    float acc = 0.f;
    for (int b = 0; b < nblk; ++b) acc += partial[(size_t)b * 3 + k];
    s[k] = acc;
  }
  __syncthreads();
  if (k == 0) {
    out[0] = (s[2] == 0.f) ? 0.f : s[0] / (3.0f * s[1]);
    out[1] = s[1];
    out[2] = s[2];
  }
}

// backward: g_loss (device scalar) -> g_z1[P] and per-block partials of the camera gradients [R1 9 | T1 3 | R2 9 | T2 3]
DISTR_GLOBAL void __launch_bounds__(256) k_warp_bwd(WarpArgs A, const float* __restrict__ fwd /*[3] loss, kept, valid*/,
                                                  const float* __restrict__ g_loss, float* __restrict__ g_z1,
                                                  float* __restrict__ partial /*[nblk][24]*/) {
  __shared__ float lds[96];
  const int P = A.H * A.W, W = A.W, H = A.H;
  const int i = blockIdx.x * 256 + threadIdx.x;
  float acc[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) acc[k] = 0.f;
  float gz = 0.f;
  if (i < P) {
    float R1[9], c1[3], R2[9], T2[3], T1[3];
    load_cam_pair(A, R1, c1, R2, T2, T1);
    const WarpPix w = warp_pixel(A, i, R1, c1, R2, T2);
    if (w.keep && w.b.ok) {
      const float scale = g_loss[0] / (3.0f * fwd[1]);
      // d loss / d c2[ch] = -sign(c1 - c2) * scale
      float gc[3];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float df = A.img1[(size_t)i * 3 + ch] - w.c2[ch];
        gc[ch] = (df > 0.f ? -1.f : (df < 0.f ? 1.f : 0.f)) * scale;
      }
      // d c2 / d (ix, iy) of the bilinear sample (zero-padded corners)
      float val[2][2][3];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int x = w.b.x0 + dx, y = w.b.y0 + dy;
          const bool in = inb(x, y, W, H);
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) val[dy][dx][ch] = in ? A.img2[(size_t)(y * W + x) * 3 + ch] : 0.f;
        }
      float gix = 0.f, giy = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        gix += gc[ch] * ((val[0][1][ch] - val[0][0][ch]) * (1.0f - w.b.ty) + (val[1][1][ch] - val[1][0][ch]) * w.b.ty);
        giy += gc[ch] * ((val[1][0][ch] - val[0][0][ch]) * (1.0f - w.b.tx) + (val[1][1][ch] - val[0][1][ch]) * w.b.tx);
      }
      // (ix, iy) = (u, v) up to rounding; u = pr0 / pr2, v = pr1 / pr2
      const float gpr[3] = {gix / w.pr[2], giy / w.pr[2], -(gix * w.u + giy * w.v) / w.pr[2]};
      float gq[3], gpt[3];
#pragma unroll
      for (int b = 0; b < 3; ++b) gq[b] = A.K[0 * 3 + b] * gpr[0] + A.K[1 * 3 + b] * gpr[1] + A.K[2 * 3 + b] * gpr[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[12 + a * 3 + b] += gq[a] * w.pt[b];      // R2
        acc[21 + a] += gq[a];                                                    // T2
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) gpt[b] = R2[0 * 3 + b] * gq[0] + R2[1 * 3 + b] * gq[1] + R2[2 * 3 + b] * gq[2];
      gz = gpt[0] * w.g.d[0] + gpt[1] * w.g.d[1] + gpt[2] * w.g.d[2];
      // pt = d(R1) * z + c1(R1, T1)
      const float gd[3] = {gpt[0] * w.z, gpt[1] * w.z, gpt[2] * w.z};
      ray_backward_acc(w.g, gd, acc);                                            // R1 through the normalised ray
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[k * 3 + j] += -gpt[j] * T1[k];           // c1_j = -sum_k R1[k][j] T1[k]
        acc[9 + k] += -(R1[k * 3] * gpt[0] + R1[k * 3 + 1] * gpt[1] + R1[k * 3 + 2] * gpt[2]);
      }
    }
    if (g_z1) g_z1[i] = gz;
  }
  // 24 sums in 3 rounds of 8 (LDS budget)
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = acc[r * 8 + k];
    block_sum<8>(v, lds);
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) partial[(size_t)blockIdx.x * 24 + r * 8 + k] = v[k];
    }
  }
}

}  // namespace distr
