// distr_mlp_h3.hpp -- the DeepSDF 8x512 decoder tile in THREE-PRODUCT SPLIT-f16 arithmetic (opt-in), forward.
//
// Same network, same tile geometry and the same place in the march kernels as the six-product split-bf16 tile
// (distr_mlp_b6.hpp); what differs is the number format of the planes:
//     SW W = w0 + w1,  SX x = a0 + a1   (f16 planes of the SCALED operands, round to nearest even of the running remainder)
//     W x ~ (w0 a0 + w1 a0 + w0 a1) / (SW SX)             on v_mfma_f32_32x32x16_f16, f32 accumulation
// An f16 plane carries 11 significant bits, two planes 22, and the dropped product w1 a1 is 2^-22 of w0 a0 -- the size of the f32
// chain's own rounding. Measured on both fixture decoders (oracle/study_split_bf16.py, CPU restatement, 20 000 points against a
// float64 decoder): max |sdf - sdf_f64| 3.5e-7 for this form, 3.0e-7 for the exact f32 chain, 3.0e-7 for six bf16 products, 7.9e-6
// for three bf16 products. Three MFMAs per f32 product instead of six, and -- because two f16 planes of a 512 x 64 activation tile
// are 128 KiB, the size of the f32 tile -- the planes themselves live in LDS: the split is done ONCE by the wave that produces a
// value (write-back), not by each of the four waves that consume it, and the k-loop has no VALU work at all. The weight stream is
// two f16 planes = the bytes of f32 (the bf16 form streams 1.5 x). profiles/ubench/split_f16_layer.hip: 34.3 k cycles per
// 512 x 512 layer on a 64-ray tile (bf16x6 66.4 k, exact f32 133.7 k).
//
// What f16 costs is RANGE (5 exponent bits): a scaled operand beyond 65504 becomes inf. SX = SW = 64 keeps the second planes of
// this network's magnitudes (activations 1e-3 .. 1, weights ~ 0.05) above the f16 denormals and allows |x|, |W| < 1023;
// weights outside that range make the mode unavailable for the decoder (distr_set_decoder notes it, the calls that ask for the
// mode fail), and an activation that overflows turns the evaluation's result non-finite, which the tiles COUNT
// (Consts.f16_overflow -> distr_render_stats.f16_overflows; distr_mlp_eval_f16x3 writes NaN for the point) instead of hiding it
// behind the clamps of the march.
// The backward of a render in this mode (mlp_backward_h3) is the dX chain on the ReLU masks this forward saved, in the same
// arithmetic, on deltas NORMALISED per ray by the loss gradient (whose range has no bound), with transposed weight planes.
#pragma once
#include "distr_mlp_b6.hpp"

namespace distr {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

constexpr float H3_SX = 64.f, H3_SW = 64.f;      // powers of two: every rescale below is exact

constexpr float H3_SD = 1024.f;                  // scale of the backward's per-ray normalised deltas (|delta / d8| < 64)

struct DecoderH3 {
  const uint32_t* Wp[8];   // two f16 A-fragment planes of SW * lin1..lin7 ([0] unused); lin3: O padded to 256; lin4: K = 256
  const uint32_t* Wb[8];   // the same for the TRANSPOSED matrices (backward dX chain), index = layer whose weights are used (1..7)
};

template <int RB>
struct alignas(16) SmemH3 {
  static constexpr int TILE = 32 * RB;
  uint16_t P[2][HID * TILE];   // the two f16 planes of SX * activation, k-minor: P[p][k >> 3][ray][k & 7]
  float xyz[4 * TILE];
  float part[12 * TILE];       // lin8 partial chains [4][TILE]
  float aux[4 * TILE];         // march tiles (KEEP): the rays' mask-block indices (8 bytes each)
};
static_assert(sizeof(SmemH3<2>) == sizeof(SmemB6<2>) && sizeof(SmemH3<1>) == sizeof(SmemB6<1>), "the tiles share k_step's role buffer");

// {f16(a) low half, f16(b) high half} of the value and of its remainder (v_cvt_pk_f16_f32, v_cvt_f32_f16, v_sub_f32)
__device__ __forceinline__ void split2_f16(float a, float b, uint32_t& p0, uint32_t& p1) {
  const f32x2 v = {a, b};
  const f16x2 h0 = __builtin_convertvector(v, f16x2);
  const f32x2 r = v - __builtin_convertvector(h0, f32x2);
  const f16x2 h1 = __builtin_convertvector(r, f16x2);
  p0 = __builtin_bit_cast(uint32_t, h0);
  p1 = __builtin_bit_cast(uint32_t, h1);
}
// four consecutive features (>= 0: ReLU outputs) of one ray -> one 8-byte store per plane. `top` keeps the largest leading-plane
// bit pattern seen (v_pk_max_u16; non-negative f16 order like unsigned integers, inf = 0x7c00 and NaN above it): the range check
// of the tile costs one instruction per pair of values.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ bool top_overflowed(uint32_t top) { return (top & 0xffffu) >= 0x7c00u || (top >> 16) >= 0x7c00u; }

template <int TILE>
__device__ __forceinline__ void store4_h3(uint16_t (&P)[2][HID * TILE], int row, int ray, float v0, float v1, float v2, float v3, uint32_t& top) {
  u32x2 p0, p1;
  uint32_t t0, t1;
  split2_f16(v0, v1, t0, t1); p0[0] = t0; p1[0] = t1; top = pk_max_u16(top, t0);
  split2_f16(v2, v3, t0, t1); p0[1] = t0; p1[1] = t1; top = pk_max_u16(top, t0);
  *reinterpret_cast<u32x2*>(&P[0][xk<TILE>(row, ray)]) = p0;
  *reinterpret_cast<u32x2*>(&P[1][xk<TILE>(row, ray)]) = p1;
}

// weight fragments of one 16-feature block: 2 planes x NOB row blocks, one 16-byte load each (buffers are sized for NOB = 4)
template <int NOB>
__device__ __forceinline__ void load_a_h3(const uint32_t* __restrict__ Wp, int kb, int wave, int lane, u32x4 (&dst)[4][2]) {
  const u32x4* wp = reinterpret_cast<const u32x4*>(Wp) + (size_t)wave * NOB * 2 * 64 + lane;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int p = 0; p < 2; ++p) dst[ob][p] = wp[(((size_t)kb * 4 * NOB + ob) * 2 + p) * 64];
}

// acc[ob][rb] (started by the caller, in units of SW SX) += (SW W)[rows of this wave][0..K) x (SX X)[0..K)[rays], three f16 products
// per f32 product. Register double buffer as in dense_b6 (block kb + 1 requested before the MFMAs of block kb; `a` = this layer's
// block 0 on entry, the next layer's on return), with nothing but loads and MFMAs in the loop. (A ring of three buffers, two blocks
// ahead, is worth 13 % in the microbenchmark and 4.5 % in distr_mlp_eval_f16x3, but puts 380 bytes of k_step's state into scratch
// and leaves the march unchanged: not used.)
template <int K, int NOB, int RB, int NOBN>
__device__ __forceinline__ void dense_h3(const uint32_t* __restrict__ Wp, const uint32_t* __restrict__ WpNext, const uint16_t (&P)[2][HID * 32 * RB],
                                         f32x16 (&acc)[NOB][RB], int wave, int lane, u32x4 (&a)[4][2]) {
  constexpr int NKB = K / 16, TILE = 32 * RB;
  const int j = lane & 31, h = lane >> 5;
  constexpr int PW[3] = {0, 1, 0}, PA[3] = {0, 0, 1};     // (weight plane, activation plane) of the three products
  u32x4 b[RB][2];
  auto load_b = [&](u32x4 (&dst)[RB][2], int kb) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int p = 0; p < 2; ++p) dst[rb][p] = *reinterpret_cast<const u32x4*>(&P[p][xk<TILE>(16 * kb + 8 * h, 32 * rb + j)]);
  };
  load_b(b, 0);
#pragma unroll 2
  for (int kb = 0; kb < NKB; ++kb) {
    u32x4 an[4][2], bn[RB][2];
    const bool last = (kb + 1 == NKB);
    if (!last) load_a_h3<NOB>(Wp, kb + 1, wave, lane, an);
    else if (WpNext) load_a_h3<NOBN>(WpNext, 0, wave, lane, an);
    load_b(bn, last ? kb : kb + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          acc[ob][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ob][PW[q]]), __builtin_bit_cast(f16x8, b[rb][PA[q]]),
                                                               acc[ob][rb], 0, 0, 0);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int p = 0; p < 2; ++p) a[ob][p] = an[ob][p];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int p = 0; p < 2; ++p) b[rb][p] = bn[rb][p];
  }
}

// ReLU, rescale (acc = SW SX y -> SX relu(y)), split into the two planes, write back. KEEP: mask bits in the common format
// (distr_mlp.hpp::writeback: bit (rb * 16 + r) of mask[ob] = value > 0).
template <int NOB, int RB, bool KEEP>
__device__ __forceinline__ void writeback_h3(uint16_t (&P)[2][HID * 32 * RB], const f32x16 (&acc)[NOB][RB], int row0, int lane, uint32_t (&mask)[4],
                                             uint32_t (&top)[RB]) {
  constexpr int TILE = 32 * RB;
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    uint32_t m = 0u;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int32_t rbits = max(__float_as_int(acc[ob][rb][4 * q + i]), 0);
          v[i] = __int_as_float(rbits) * (1.0f / H3_SW);
          if (KEEP) m |= min((uint32_t)rbits, 1u) << (rb * 16 + 4 * q + i);
        }
        store4_h3<TILE>(P, row0 + 32 * ob + 8 * q + 4 * h, 32 * rb + j, v[0], v[1], v[2], v[3], top[rb]);
      }
    if (KEEP) {
      asm volatile("" : "+v"(m));
      mask[ob] = m;
    }
  }
}

template <int K, int NOB, int RB, bool KEEP, int NOBN>
__device__ __forceinline__ void layer_h3(const uint32_t* __restrict__ Wp, const uint32_t* __restrict__ WpNext, const float* __restrict__ bias, int nbias,
                                         uint16_t (&P)[2][HID * 32 * RB], int wave, int lane, uint32_t (&mask)[4], uint32_t (&top)[RB], u32x4 (&a)[4][2]) {
  const int h = lane >> 5;
  f32x16 acc[NOB][RB];
  const int row0 = wave * 32 * NOB;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float bv = (row < nbias) ? bias[row] * (H3_SW * H3_SX) : 0.f;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[ob][rb][r] = bv;
    }
  dense_h3<K, NOB, RB, NOBN>(Wp, WpNext, P, acc, wave, lane, a);
  __syncthreads();                         // everybody is done reading the layer input
  writeback_h3<NOB, RB, KEEP>(P, acc, row0, lane, mask, top);
  __syncthreads();
}

// Preconditions and results as mlp_forward_b6. Range check: every lane tracks the largest leading-plane pattern it wrote for each
// of its rays (store4_h3); a ray with an activation at or beyond the f16 range (inf / NaN pattern) gets NaN as its result -- which the
// callers test -- whatever the later layers made of the inf (a ReLU taken on the bit pattern turns a negative NaN into 0).
template <int RB, bool KEEP>
__device__ __forceinline__ float mlp_forward_h3(const DecoderDev& D, const DecoderH3& H3, const float* __restrict__ c0, const float* __restrict__ c4,
                                                SmemH3<RB>& S, uint32_t (&masks)[8][4]) {
  constexpr int TILE = 32 * RB;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int j = lane & 31, h = lane >> 5;
  uint32_t top[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) top[rb] = 0u;
  uint32_t* over = reinterpret_cast<uint32_t*>(S.part + 4 * TILE);      // [TILE] per-ray overflow flags
  if (tid < TILE) over[tid] = 0u;
  // lin0 (K = 3): the f32 fmaf chain of the exact tile in the accumulator layout of the wide layers, written as planes
  {
    float px[RB], py[RB], pz[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) { px[rb] = S.xyz[32 * rb + j]; py[rb] = S.xyz[TILE + 32 * rb + j]; pz[rb] = S.xyz[2 * TILE + 32 * rb + j]; }
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      uint32_t m = 0u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[RB][4];
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
            v[rb][i] = __int_as_float(rbits) * H3_SX;
            if (KEEP) m |= min((uint32_t)rbits, 1u) << (rb * 16 + 4 * q + i);
          }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) store4_h3<TILE>(S.P, wave * 128 + 32 * ob + 8 * q + 4 * h, 32 * rb + j, v[rb][0], v[rb][1], v[rb][2], v[rb][3], top[rb]);
      }
      if (KEEP) {
        asm volatile("" : "+v"(m));
        masks[0][ob] = m;
      }
    }
  }
  __syncthreads();
  u32x4 a[4][2];                           // first weight block of the next layer, travelling across the write-backs
  load_a_h3<4>(H3.Wp[1], 0, wave, lane, a);
  layer_h3<512, 4, RB, KEEP, 4>(H3.Wp[1], H3.Wp[2], D.bias[1], HID, S.P, wave, lane, masks[1], top, a);
  layer_h3<512, 4, RB, KEEP, 2>(H3.Wp[2], H3.Wp[3], D.bias[2], HID, S.P, wave, lane, masks[2], top, a);
  masks[3][2] = 0; masks[3][3] = 0;
  layer_h3<512, 2, RB, KEEP, 4>(H3.Wp[3], H3.Wp[4], D.bias[3], 253, S.P, wave, lane, masks[3], top, a);      // lin3: 512 -> 253 (+ 3 rows that carry xyz into lin4)
  if (tid < 3 * TILE) {                    // rows 253..255 <- SX * xyz (signed), as planes
    const _Float16 h0 = (_Float16)(S.xyz[tid] * H3_SX);
    const _Float16 h1 = (_Float16)(S.xyz[tid] * H3_SX - (float)h0);
    S.P[0][xk<TILE>(253 + tid / TILE, tid % TILE)] = __builtin_bit_cast(uint16_t, h0);
    S.P[1][xk<TILE>(253 + tid / TILE, tid % TILE)] = __builtin_bit_cast(uint16_t, h1);
  }
  __syncthreads();
  layer_h3<256, 4, RB, KEEP, 4>(H3.Wp[4], H3.Wp[5], c4, HID, S.P, wave, lane, masks[4], top, a);             // lin4: [x3 (253) | xyz (3)] -> 512, latent part folded into c4
  layer_h3<512, 4, RB, KEEP, 4>(H3.Wp[5], H3.Wp[6], D.bias[5], HID, S.P, wave, lane, masks[5], top, a);
  layer_h3<512, 4, RB, KEEP, 4>(H3.Wp[6], H3.Wp[7], D.bias[6], HID, S.P, wave, lane, masks[6], top, a);
  layer_h3<512, 4, RB, KEEP, 4>(H3.Wp[7], nullptr, D.bias[7], HID, S.P, wave, lane, masks[7], top, a);
  // lin8: four 128-long f32 chains per ray (one per wave) on x7 = (a0 + a1) / SX (the sum of the planes is exact in f32),
  // combined in the exact tile's order
  const int ray = tid & (TILE - 1);
  {
    float p = 0.f;
    const float* w8 = D.w8 + wave * 128;
#pragma unroll 8
    for (int k = 0; k < 128; ++k) {
      const int i = xk<TILE>(wave * 128 + k, ray);
      const float x = ((float)__builtin_bit_cast(_Float16, S.P[0][i]) + (float)__builtin_bit_cast(_Float16, S.P[1][i])) * (1.0f / H3_SX);
      p = __builtin_fmaf(w8[k], x, p);
    }
    S.part[wave * TILE + ray] = p;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
      if (top_overflowed(top[rb])) over[32 * rb + j] = 1u;
  }
  __syncthreads();
  const float y = ((S.part[ray] + S.part[TILE + ray]) + (S.part[2 * TILE + ray] + S.part[3 * TILE + ray])) + D.b8;
  return over[ray] ? __builtin_nanf("") : y;
}

// ---------------------------------------------------------------------------------------- backward tile (dX chain)
// Signed values (deltas): four consecutive features of one ray -> one 8-byte store per plane
template <int TILE>
__device__ __forceinline__ void store4s_h3(uint16_t (&P)[2][HID * TILE], int row, int ray, float v0, float v1, float v2, float v3) {
  u32x2 p0, p1;
  uint32_t t0, t1;
  split2_f16(v0, v1, t0, t1); p0[0] = t0; p1[0] = t1;
  split2_f16(v2, v3, t0, t1); p0[1] = t0; p1[1] = t1;
  *reinterpret_cast<u32x2*>(&P[0][xk<TILE>(row, ray)]) = p0;
  *reinterpret_cast<u32x2*>(&P[1][xk<TILE>(row, ray)]) = p1;
}
// Gate + write-back of a backward layer: e = mask bit ? acc / SW : 0 (bits in the forward's format), as planes
template <int NOB, int RB>
__device__ __forceinline__ void writeback_gate_h3(uint16_t (&P)[2][HID * 32 * RB], const f32x16 (&acc)[NOB][RB], int row0, int lane, const uint32_t (&mask)[4]) {
  constexpr int TILE = 32 * RB;
  const int j = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    const uint32_t m = mask[ob];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = gate(acc[ob][rb][4 * q + i] * (1.0f / H3_SW), (m >> (rb * 16 + 4 * q + i)) & 1u);
        store4s_h3<TILE>(P, row0 + 32 * ob + 8 * q + 4 * h, 32 * rb + j, v[0], v[1], v[2], v[3]);
      }
  }
}
// dst[row] = sum over the tile's rays of delta[row][ray] = sum of e[row][ray] * d8[ray]; d8s = d8 / SD (LDS), the exact tile's ray order
template <int RB>
__device__ __forceinline__ void row_sums_h3(const uint16_t (&P)[2][HID * 32 * RB], const float* d8s, float* __restrict__ dst, int tid) {
  constexpr int TILE = 32 * RB;
  const int lane = tid & 63;
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int row = tid + rr * 256;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < TILE; ++i) {
      const int ray = (i + lane) & (TILE - 1);
      const int x = xk<TILE>(row, ray);
      s += ((float)__builtin_bit_cast(_Float16, P[0][x]) + (float)__builtin_bit_cast(_Float16, P[1][x])) * d8s[ray];
    }
    dst[row] = s;
  }
}

// The dX chain of mlp_backward (distr_mlp.hpp) in split-f16 arithmetic. Interface of mlp_backward_b6: masks = the ReLU bitmasks of
// the forward being differentiated; S.aux row 0 = d8[ray] = coef (1 - y^2); on return S.aux rows 1..3 hold d(coef f)/d xyz per ray,
// sd0 / sd4 the row sums over the tile's rays of delta0 / delta4 (latent gradient).
// d8 is a loss gradient: its magnitude has no bound, f16 has 5 exponent bits. The chain is linear in d8, so it runs on the
// NORMALISED deltas e_l = delta_l / d8 (e7 = relu'(h7) w8: a property of the decoder alone, |e| ~ 1e-3 .. 1; planes hold SD e) and d8
// multiplies the results per ray: the row sums are weighted sums, the xyz gradient is scaled at the end. An e beyond the range (SD |e|
// > 65504) turns into inf / NaN planes, which the chain carries into non-finite gradients -- visible, not clamped.
template <int RB>
__device__ __forceinline__ void mlp_backward_h3(const DecoderDev& D, const DecoderH3& H3, SmemH3<RB>& S, uint32_t (&masks)[8][4],
                                                float* __restrict__ sd0, float* __restrict__ sd4) {
  constexpr int TILE = 32 * RB;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int ray = tid & (TILE - 1);
  float* d8s = S.xyz;                      // (the saved-mask backward does not stage points: S.xyz is free)
  if (tid < TILE) d8s[tid] = S.aux[tid] * (1.0f / H3_SD);
  u32x4 a[4][2];
  load_a_h3<4>(H3.Wb[7], 0, wave, lane, a);
  // SD e7[k][ray] = relu'(h7) * w8[k] * SD
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) {
    const uint32_t m = masks[7][ob];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[RB][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float w = D.w8[wave * 128 + 32 * ob + 8 * q + 4 * h + i] * H3_SD;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) v[rb][i] = gate(w, (m >> (16 * rb + 4 * q + i)) & 1u);
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) store4s_h3<TILE>(S.P, wave * 128 + 32 * ob + 8 * q + 4 * h, 32 * rb + j, v[rb][0], v[rb][1], v[rb][2], v[rb][3]);
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
  for (int l = 7; l >= 5; --l) {  // e_l (512) -> e_{l-1} (512)
    f32x16 acc[4][RB];
    zero4(acc);
    if (l > 5) dense_h3<512, 4, RB, 4>(H3.Wb[l], H3.Wb[l - 1], S.P, acc, wave, lane, a);
    else dense_h3<512, 4, RB, 2>(H3.Wb[5], H3.Wb[4], S.P, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_h3<4, RB>(S.P, acc, wave * 128, lane, masks[l - 1]);
    __syncthreads();
  }
  if (sd4) row_sums_h3<RB>(S.P, d8s, sd4, tid);  // planes = e4
  {  // lin4^T: e4 (512) -> [e3 (253) | d xyz (3)]
    f32x16 acc[2][RB];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ob][rb][r] = 0.f;
    dense_h3<512, 2, RB, 4>(H3.Wb[4], H3.Wb[3], S.P, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_h3<2, RB>(S.P, acc, wave * 64, lane, masks[3]);  // rows 253..255 have mask 0 -> written as 0
    if (wave == 3 && h == 1) {
#pragma unroll
      for (int r = 13; r < 16; ++r)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) S.aux[(1 + r - 13) * TILE + 32 * rb + j] = acc[1][rb][r] * (1.0f / (H3_SW * H3_SD));   // (per unit d8)
    }
    __syncthreads();
  }
  {  // lin3^T: e3 (256 rows, 253 real) -> e2 (512)
    f32x16 acc[4][RB];
    zero4(acc);
    dense_h3<256, 4, RB, 4>(H3.Wb[3], H3.Wb[2], S.P, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_h3<4, RB>(S.P, acc, wave * 128, lane, masks[2]);
    __syncthreads();
  }
#pragma unroll
  for (int l = 2; l >= 1; --l) {
    f32x16 acc[4][RB];
    zero4(acc);
    if (l > 1) dense_h3<512, 4, RB, 4>(H3.Wb[2], H3.Wb[1], S.P, acc, wave, lane, a);
    else dense_h3<512, 4, RB, 4>(H3.Wb[1], nullptr, S.P, acc, wave, lane, a);
    __syncthreads();
    writeback_gate_h3<4, RB>(S.P, acc, wave * 128, lane, masks[l - 1]);
    __syncthreads();
  }
  if (sd0) row_sums_h3<RB>(S.P, d8s, sd0, tid);  // planes = e0
  // d xyz through lin0's xyz columns: 3 x four 128-long f32 chains per ray on e0
  {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    const float* wx = D.W0x + wave * 128;
#pragma unroll 4
    for (int k = 0; k < 128; ++k) {
      const int x = xk<TILE>(wave * 128 + k, ray);
      const float d = (float)__builtin_bit_cast(_Float16, S.P[0][x]) + (float)__builtin_bit_cast(_Float16, S.P[1][x]);
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
    const float unit = S.aux[(1 + c) * TILE + r] + ((p[0] + p[TILE]) + (p[2 * TILE] + p[3 * TILE])) * (1.0f / H3_SD);
    S.aux[(1 + c) * TILE + r] = unit * S.aux[r];     // x d8 of the ray
  }
  __syncthreads();
}

// decode_sdf (core/utils/decoder_utils.py:53-74) for n explicit points in split-f16 arithmetic; a point whose evaluation left the
// f16 range gets NaN (not a clamped number)
DISTR_GLOBAL void __launch_bounds__(256, 1) k_eval_h3(const float* __restrict__ xyz, int64_t n, const float* __restrict__ c0c4, float clamp,
                                                    float* __restrict__ sdf, DecoderDev D, DecoderH3 H3) {
  __shared__ SmemH3<2> S;
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
  const float pre = mlp_forward_h3<2, false>(D, H3, c0c4, c0c4 + HID, S, masks);
  if (tid < 64 && base + tid < n) {
    const float s = tanh_spec(pre);
    const bool finite = fabsf(pre) <= 3.0e38f;
    sdf[base + tid] = !finite ? __builtin_nanf("") : (clamp >= 0.f) ? clampf(s, -clamp, clamp) : s;
  }
}

}  // namespace distr
