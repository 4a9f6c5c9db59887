// distr_mlp_b6.hpp -- the DeepSDF 8x512 decoder tile in SIX-PRODUCT SPLIT-bf16 arithmetic (opt-in), forward and backward.
//
// The default decoder tile (distr_mlp.hpp) computes every dense layer with f32-input MFMAs: exact f32, bit-identical to the
// oracle's k-ordered fmaf chains, at 1/16 of the bf16 MFMA rate. This tile evaluates the same network (Decoder.inference,
// core/graph/deep_sdf_decoder.py:80-111) with every f32 product W x replaced by six bf16 products
//     W = w0 + w1 + w2,  x = a0 + a1 + a2   (bf16 planes, round to nearest even of the running remainder)
//     W x ~ w0 a0 + w1 a0 + w0 a1 + w1 a1 + w2 a0 + w0 a2
// on v_mfma_f32_32x32x16_bf16 with f32 accumulation: both operands keep 24 significant bits, the dropped terms are below 2^-24
// relative, and one layer's error against a float64 product is no larger than the f32 chain's own (profiles/r03_split_bf16_layer.md:
// 1.4e-7 vs 2.0e-7). It is NOT bit-identical to the oracle (different summation order inside the MFMA), so it never replaces the
// exact path silently: callers ask for it (distr_mlp_eval_bf16x6; bulk SDF grids for meshing, core/evaluation/create_mesh.py, where
// marching cubes is indifferent to 1e-6). Measured: see profiles/ (dense decoder, TFLOP/s-equivalent and max |delta sdf|).
//
// Data flow of a 64-ray tile (one workgroup = 4 waves):
//   * activations stay f32 in LDS (128 KiB; three bf16 planes of 512 x 64 would need 192 KiB), k-MINOR: X[k >> 3][ray][k & 7], so
//     the 8 k-values a lane feeds to one bf16 MFMA are 32 contiguous bytes; they are split into the three planes in registers
//     while the next block's fragments load (VALU that hides in the gaps of bf16 MFMAs, unlike next to f32 MFMAs);
//   * weights: three pre-split bf16 planes per layer, pre-packed on the host as A fragments of v_mfma_f32_32x32x16_bf16
//     (fragment index (((kb * 4 + wave) * NOB + ob) * 3 + plane) * 64 + lane = 8 bf16 W_plane[o][16 kb + 8 h + 0..7]);
//   * wave w owns output rows [w O/4, (w+1) O/4) as NOB x 2 accumulator tiles of 32 x 32, started from the bias (exact f32);
//   * lin0 (K = 3) and lin8 (one row) are plain f32 VALU work, as in the exact tile;
//   * march tiles (round 3): the same forward inside k_march / k_step when distr_render_cfg.arith = DISTR_ARITH_BF16X6, on 64- and
//     32-ray tiles only (no 16-ray / cluster tiles: they are built on the f32 16x16x4 MFMA and would not be bit-identical to these);
//   * backward (mlp_backward_b6): the dX chain on the saved ReLU masks with the transposed weight planes, same loop.
#pragma once
#include "distr_mlp.hpp"

namespace distr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct DecoderB6 {
  const uint32_t* Wp[8];   // split-bf16 A-fragment planes of lin1..lin7 ([0] unused); lin3: O padded to 256; lin4: K = 256
  const uint32_t* Wb[8];   // the same for the TRANSPOSED matrices (backward dX chain), index = layer whose weights are used (1..7)
};

// A tile = RB blocks of 32 rays (RB = 2: 64 rays, RB = 1: 32 rays). Every output column of v_mfma_f32_32x32x16_bf16 depends only
// on its own B column, so a ray's value does not depend on RB or on which other rays share its tile: the two tile sizes are
// bit-identical (within this arithmetic), like the tile sizes of the exact path are among themselves.
template <int RB>
struct alignas(16) SmemB6 {
  static constexpr int TILE = 32 * RB;
  float X[HID * TILE];     // k-minor activations
  float xyz[4 * TILE];
  float part[12 * TILE];   // lin8 partial chains [4][TILE]; backward: xyz-gradient partials [3][4][TILE], mask-block indices
  float aux[4 * TILE];     // march tiles (KEEP): the rays' mask-block indices (8 bytes each); backward: row 0 = d8, rows 1..3 = d/dxyz
};

template <int TILE>
__device__ __forceinline__ int xk(int k, int ray) { return ((k >> 3) * TILE + ray) * 8 + (k & 7); }

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {   // {bf16(a) low half, bf16(b) high half}, round to nearest even
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf2));      // v_cvt_pk_bf16_f32
}
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = pk_bf16(a, b);
  const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
  p2 = pk_bf16(sa, sb);
}

// weight fragments of one 16-feature block: 3 planes x NOB row blocks, one 16-byte load each (buffers are sized for NOB = 4)
template <int NOB>
__device__ __forceinline__ void load_a_b6(const uint32_t* __restrict__ Wp, int kb, int wave, int lane, u32x4 (&dst)[4][3]) {
  const u32x4* wp = reinterpret_cast<const u32x4*>(Wp) + (size_t)wave * NOB * 3 * 64 + lane;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int p = 0; p < 3; ++p) dst[ob][p] = wp[(((size_t)kb * 4 * NOB + ob) * 3 + p) * 64];
}

// acc[ob][rb] (started by the caller) += W[rows of this wave][0..K) x X[0..K)[rays], six bf16 products per f32 product.
// Register double buffer: block kb + 1 (weights from L2, activations from LDS) is requested before the MFMAs of block kb, its
// activations are split into their bf16 planes between those MFMAs (one wave per SIMD: nothing else hides the L2 round trip).
// On entry `a` holds this layer's block 0 -- requested by the previous layer during ITS last block -- and the look-ahead slot of
// the last block requests block 0 of the NEXT layer (WpNext, NOBN row blocks per wave; null: nothing), returned in `a`: no layer
// begins with an exposed round trip (the same arrangement as dense_pf of the exact tile). (A three-deep ring, two blocks ahead,
// was tried: the compiler needs > 512 registers for it and spills inside the loop.)
template <int K, int NOB, int RB, int NOBN>
__device__ __forceinline__ void dense_b6(const uint32_t* __restrict__ Wp, const uint32_t* __restrict__ WpNext, const float* X, f32x16 (&acc)[NOB][RB],
                                         int wave, int lane, u32x4 (&a)[4][3]) {
  constexpr int NKB = K / 16, TILE = 32 * RB;
  const int j = lane & 31, h = lane >> 5;
  constexpr int PW[6] = {0, 1, 0, 1, 2, 0}, PA[6] = {0, 0, 1, 1, 0, 2};     // (weight plane, activation plane) of the six products
  u32x4 b[RB][3];
  f32x4 xr[RB][2];
  auto load_x = [&](f32x4 (&dst)[RB][2], int kb) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const f32x4* xp = reinterpret_cast<const f32x4*>(&X[xk<TILE>(16 * kb + 8 * h, 32 * rb + j)]);
      dst[rb][0] = xp[0]; dst[rb][1] = xp[1];
    }
  };
  auto split = [&](const f32x4 (&src)[RB][2], u32x4 (&dst)[RB][3]) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      uint32_t q0[4], q1[4], q2[4];
      split_pair(src[rb][0][0], src[rb][0][1], q0[0], q1[0], q2[0]);
      split_pair(src[rb][0][2], src[rb][0][3], q0[1], q1[1], q2[1]);
      split_pair(src[rb][1][0], src[rb][1][1], q0[2], q1[2], q2[2]);
      split_pair(src[rb][1][2], src[rb][1][3], q0[3], q1[3], q2[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) { dst[rb][0][i] = q0[i]; dst[rb][1][i] = q1[i]; dst[rb][2][i] = q2[i]; }
    }
  };
  load_x(xr, 0);
  split(xr, b);
#pragma unroll 2
  for (int kb = 0; kb < NKB; ++kb) {
    u32x4 an[4][3], bn[RB][3];
    f32x4 xn[RB][2];
    const bool last = (kb + 1 == NKB);
    if (!last) load_a_b6<NOB>(Wp, kb + 1, wave, lane, an);
    else if (WpNext) load_a_b6<NOBN>(WpNext, 0, wave, lane, an);
    load_x(xn, last ? kb : kb + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 6; ++q) {
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          acc[ob][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ob][PW[q]]), __builtin_bit_cast(bf16x8, b[rb][PA[q]]),
                                                                acc[ob][rb], 0, 0, 0);
      if (q == 0) split(xn, bn);
    }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[ob][p] = an[ob][p];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int p = 0; p < 3; ++p) b[rb][p] = bn[rb][p];
  }
}

// ReLU + write-back into the k-minor layout: D rows of register r on lane (j, h) are (r & 3) + 8 (r >> 2) + 4 h -> 4 consecutive
// features = one 16-byte store. KEEP: bit (rb * 16 + r) of mask[ob] = value > 0 -- the format of the exact tile's write-back
// (distr_mlp.hpp::writeback; the C/D layout of the bf16 MFMA is that of v_mfma_f32_32x32x2_f32), so the saved mask blocks feed the
// same backward kernel.
template <int NOB, int RB, bool KEEP>
__device__ __forceinline__ void writeback_b6(float* X, const f32x16 (&acc)[NOB][RB], int row0, int lane, uint32_t (&mask)[4]) {
  constexpr int TILE = 32 * RB;
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    uint32_t m = 0u;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int32_t rbits = max(__float_as_int(acc[ob][rb][4 * q + i]), 0);
          v[i] = __int_as_float(rbits);
          if (KEEP) m |= min((uint32_t)rbits, 1u) << (rb * 16 + 4 * q + i);
        }
        *reinterpret_cast<f32x4*>(&X[xk<TILE>(row0 + 32 * ob + 8 * q + 4 * h, 32 * rb + j)]) = v;
      }
    if (KEEP) {
      asm volatile("" : "+v"(m));      // (see distr_mlp.hpp::writeback: keeps LLVM from carrying the accumulators instead of the bits)
      mask[ob] = m;
    }
  }
}

// One wide layer; `a` = its first weight block on entry, the next layer's first block on return (see dense_b6)
template <int K, int NOB, int RB, bool KEEP, int NOBN>
__device__ __forceinline__ void layer_b6(const uint32_t* __restrict__ Wp, const uint32_t* __restrict__ WpNext, const float* __restrict__ bias, int nbias,
                                         float* X, int wave, int lane, uint32_t (&mask)[4], u32x4 (&a)[4][3]) {
  const int h = lane >> 5;
  f32x16 acc[NOB][RB];
  const int row0 = wave * 32 * NOB;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float bv = (row < nbias) ? bias[row] : 0.f;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[ob][rb][r] = bv;
    }
  dense_b6<K, NOB, RB, NOBN>(Wp, WpNext, X, acc, wave, lane, a);
  __syncthreads();                         // everybody is done reading the layer input
  writeback_b6<NOB, RB, KEEP>(X, acc, row0, lane, mask);
  __syncthreads();
}

// Preconditions as mlp_forward: S.xyz rows 0..2 hold the TILE points. Returns (every thread, ray = tid & (TILE - 1)) the pre-tanh
// output; masks[l] = ReLU bitmasks of layer l in the exact tile's format (KEEP).
template <int RB, bool KEEP>
__device__ __forceinline__ float mlp_forward_b6(const DecoderDev& D, const DecoderB6& B6, const float* __restrict__ c0, const float* __restrict__ c4,
                                                SmemB6<RB>& S, uint32_t (&masks)[8][4]) {
  constexpr int TILE = 32 * RB;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  float* X = S.X;
  // lin0 (K = 3): the f32 fmaf chain of the exact tile (bias, then x, y, z), computed in the accumulator layout of the wide layers
  // (lane (j, h), register r <-> row, column) so that its ReLU bits land in the common mask format
  {
    float px[RB], py[RB], pz[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { px[rb] = S.xyz[32 * rb + j]; py[rb] = S.xyz[TILE + 32 * rb + j]; pz[rb] = S.xyz[2 * TILE + 32 * rb + j]; }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      uint32_t m = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v[RB];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wave * 128 + 32 * ob + 8 * q + 4 * h + i;
          const float b0 = c0[row], wx = D.W0x[row], wy = D.W0x[HID + row], wz = D.W0x[2 * HID + row];
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) {
            float t = __builtin_fmaf(wx, px[rb], b0);
            t = __builtin_fmaf(wy, py[rb], t);
            t = __builtin_fmaf(wz, pz[rb], t);
            const int32_t rbits = max(__float_as_int(t), 0);
            v[rb][i] = __int_as_float(rbits);
            if (KEEP) m |= min((uint32_t)rbits, 1u) << (rb * 16 + 4 * q + i);
          }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) *reinterpret_cast<f32x4*>(&X[xk<TILE>(wave * 128 + 32 * ob + 8 * q + 4 * h, 32 * rb + j)]) = v[rb];
      }
      if (KEEP) {
        asm volatile("" : "+v"(m));      // (opaque, like writeback_b6: otherwise LLVM keeps lin0's 128 raw values alive instead of the 4 words)
        masks[0][ob] = m;
      }
    }
  }
  __syncthreads();
  u32x4 a[4][3];                           // first weight block of the next layer, travelling across the write-backs
  load_a_b6<4>(B6.Wp[1], 0, wave, lane, a);
  layer_b6<512, 4, RB, KEEP, 4>(B6.Wp[1], B6.Wp[2], D.bias[1], HID, X, wave, lane, masks[1], a);
  layer_b6<512, 4, RB, KEEP, 2>(B6.Wp[2], B6.Wp[3], D.bias[2], HID, X, wave, lane, masks[2], a);
  masks[3][2] = 0; masks[3][3] = 0;
  layer_b6<512, 2, RB, KEEP, 4>(B6.Wp[3], B6.Wp[4], D.bias[3], 253, X, wave, lane, masks[3], a);      // lin3: 512 -> 253 (+ 3 rows that carry xyz into lin4)
  if (tid < 3 * TILE) X[xk<TILE>(253 + tid / TILE, tid % TILE)] = S.xyz[tid];
  __syncthreads();
  layer_b6<256, 4, RB, KEEP, 4>(B6.Wp[4], B6.Wp[5], c4, HID, X, wave, lane, masks[4], a);             // lin4: [x3 (253) | xyz (3)] -> 512, latent part folded into c4
  layer_b6<512, 4, RB, KEEP, 4>(B6.Wp[5], B6.Wp[6], D.bias[5], HID, X, wave, lane, masks[5], a);
  layer_b6<512, 4, RB, KEEP, 4>(B6.Wp[6], B6.Wp[7], D.bias[6], HID, X, wave, lane, masks[6], a);
  layer_b6<512, 4, RB, KEEP, 4>(B6.Wp[7], nullptr, D.bias[7], HID, X, wave, lane, masks[7], a);
  // lin8: four 128-long f32 chains per ray (one per wave), combined in the exact tile's order
  const int ray = tid & (TILE - 1);
  {
    float p = 0.f;
    const float* w8 = D.w8 + wave * 128;
#pragma unroll 8
    for (int k = 0; k < 128; ++k) p = __builtin_fmaf(w8[k], X[xk<TILE>(wave * 128 + k, ray)], p);
    S.part[wave * TILE + ray] = p;
  }
  __syncthreads();
  return ((S.part[ray] + S.part[TILE + ray]) + (S.part[2 * TILE + ray] + S.part[3 * TILE + ray])) + D.b8;
}

// ---------------------------------------------------------------------------------------- backward tile (dX chain)
// Gate + write-back of a backward layer: delta = mask bit ? acc : 0 (bits in the forward's format), k-minor layout
template <int NOB, int RB>
__device__ __forceinline__ void writeback_gate_b6(float* X, const f32x16 (&acc)[NOB][RB], int row0, int lane, const uint32_t (&mask)[4]) {
  constexpr int TILE = 32 * RB;
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    const uint32_t m = mask[ob];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gate(acc[ob][rb][4 * q + i], (m >> (rb * 16 + 4 * q + i)) & 1u);
        *reinterpret_cast<f32x4*>(&X[xk<TILE>(row0 + 32 * ob + 8 * q + 4 * h, 32 * rb + j)]) = v;
      }
  }
}

template <int RB>
__device__ __forceinline__ void row_sums_b6(const float* X, float* __restrict__ dst, int tid) {
  constexpr int TILE = 32 * RB;
  const int lane = tid & 63;
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int row = tid + rr * 256;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < TILE; ++i) s += X[xk<TILE>(row, (i + lane) & (TILE - 1))];   // (same summation order as the exact tile's row_sums)
    dst[row] = s;
  }
}

// The dX chain of mlp_backward (distr_mlp.hpp) in split-bf16 arithmetic. Preconditions: masks = the ReLU bitmasks of the forward
// being differentiated (saved mask blocks); S.aux row 0 = d8[ray] = coef (1 - y^2). On return S.aux rows 1..3 hold d(coef f)/d xyz
// per ray; sd0 / sd4 receive the row sums over the tile's rays of delta0 / delta4 (latent gradient).
template <int RB>
__device__ __forceinline__ void mlp_backward_b6(const DecoderDev& D, const DecoderB6& B6, SmemB6<RB>& S, uint32_t (&masks)[8][4],
                                                float* __restrict__ sd0, float* __restrict__ sd4) {
  constexpr int TILE = 32 * RB;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int ray = tid & (TILE - 1);
  float* X = S.X;
  u32x4 a[4][3];
  load_a_b6<4>(B6.Wb[7], 0, wave, lane, a);
  // delta7[k][ray] = relu'(h7) * w8[k] * d8[ray]   (exact f32, like the exact tile)
  {
    float d8[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) d8[rb] = S.aux[32 * rb + j];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const uint32_t m = masks[7][ob];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v[RB];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wave * 128 + 32 * ob + 8 * q + 4 * h + i;
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) v[rb][i] = gate(D.w8[row] * d8[rb], (m >> (16 * rb + 4 * q + i)) & 1u);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) *reinterpret_cast<f32x4*>(&X[xk<TILE>(wave * 128 + 32 * ob + 8 * q + 4 * h, 32 * rb + j)]) = v[rb];
      }
    }
  }
  __syncthreads();
  auto zero4 = [&](f32x16 (&acc)[4][RB]) {
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ob][rb][r] = 0.f;
  };
#pragma unroll
  for (int l = 7; l >= 5; --l) {  // delta_l (512) -> delta_{l-1} (512)
    f32x16 acc[4][RB];
    zero4(acc);
    if (l > 5) dense_b6<512, 4, RB, 4>(B6.Wb[l], B6.Wb[l - 1], X, acc, wave, lane, a);
    else dense_b6<512, 4, RB, 2>(B6.Wb[5], B6.Wb[4], X, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_b6<4, RB>(X, acc, wave * 128, lane, masks[l - 1]);
    __syncthreads();
  }
  if (sd4) row_sums_b6<RB>(X, sd4, tid);  // X = delta4
  {  // lin4^T: delta4 (512) -> [delta3 (253) | d xyz (3)]
    f32x16 acc[2][RB];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ob][rb][r] = 0.f;
    dense_b6<512, 2, RB, 4>(B6.Wb[4], B6.Wb[3], X, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_b6<2, RB>(X, acc, wave * 64, lane, masks[3]);  // rows 253..255 have mask 0 -> written as 0
    if (wave == 3 && h == 1) {
#pragma unroll
      for (int r = 13; r < 16; ++r)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) S.aux[(1 + r - 13) * TILE + 32 * rb + j] = acc[1][rb][r];
    }
    __syncthreads();
  }
  {  // lin3^T: delta3 (256 rows, 253 real) -> delta2 (512)
    f32x16 acc[4][RB];
    zero4(acc);
    dense_b6<256, 4, RB, 4>(B6.Wb[3], B6.Wb[2], X, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_b6<4, RB>(X, acc, wave * 128, lane, masks[2]);
    __syncthreads();
  }
#pragma unroll
  for (int l = 2; l >= 1; --l) {
    f32x16 acc[4][RB];
    zero4(acc);
    if (l > 1) dense_b6<512, 4, RB, 4>(B6.Wb[2], B6.Wb[1], X, acc, wave, lane, a);
    else dense_b6<512, 4, RB, 4>(B6.Wb[1], nullptr, X, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_b6<4, RB>(X, acc, wave * 128, lane, masks[l - 1]);
    __syncthreads();
  }
  if (sd0) row_sums_b6<RB>(X, sd0, tid);  // X = delta0
  // d xyz through lin0's xyz columns: 3 x four 128-long f32 chains per ray
  {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    const float* wx = D.W0x + wave * 128;
#pragma unroll 4
    for (int k = 0; k < 128; ++k) {
      const float d = X[xk<TILE>(wave * 128 + k, ray)];
      p0 = __builtin_fmaf(wx[k], d, p0);
      p1 = __builtin_fmaf(wx[HID + k], d, p1);
      p2 = __builtin_fmaf(wx[2 * HID + k], d, p2);
    }
    S.part[(0 * 4 + wave) * TILE + ray] = p0;
    S.part[(1 * 4 + wave) * TILE + ray] = p1;
    S.part[(2 * 4 + wave) * TILE + ray] = p2;
  }
  __syncthreads();
  if (tid < 3 * TILE) {
    const int c = tid / TILE, r = tid % TILE;
    const float* p = S.part + c * 4 * TILE + r;
    S.aux[(1 + c) * TILE + r] = S.aux[(1 + c) * TILE + r] + ((p[0] + p[TILE]) + (p[2 * TILE] + p[3 * TILE]));
  }
  __syncthreads();
}

// decode_sdf (core/utils/decoder_utils.py:53-74) for n explicit points in split-bf16 arithmetic
DISTR_GLOBAL void __launch_bounds__(256, 1) k_eval_b6(const float* __restrict__ xyz, int64_t n, const float* __restrict__ c0c4, float clamp,
                                                    float* __restrict__ sdf, DecoderDev D, DecoderB6 B6) {
  __shared__ SmemB6<2> S;
  const int tid = threadIdx.x;
  const int64_t base = (int64_t)blockIdx.x * 64;
  if (base >= n) return;
  if (tid < 64) {
    const int64_t r = base + tid;
    const bool v = r < n;
    S.xyz[tid] = v ? xyz[r * 3] : 0.f; S.xyz[64 + tid] = v ? xyz[r * 3 + 1] : 0.f; S.xyz[128 + tid] = v ? xyz[r * 3 + 2] : 0.f;
  }
  __syncthreads();
  uint32_t masks[8][4];
  const float pre = mlp_forward_b6<2, false>(D, B6, c0c4, c0c4 + HID, S, masks);
  if (tid < 64 && base + tid < n) {
    const float s = tanh_spec(pre);
    sdf[base + tid] = (clamp >= 0.f) ? clampf(s, -clamp, clamp) : s;
  }
}

}  // namespace distr
