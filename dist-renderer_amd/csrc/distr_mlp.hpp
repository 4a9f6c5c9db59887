// distr_mlp.hpp -- the fused DeepSDF 8x512 decoder tile for gfx950 (CDNA4), forward and backward.
//
// Replaces Decoder.inference (core/graph/deep_sdf_decoder.py:80-111) + decode_sdf's latent expand/cat
// (core/utils/decoder_utils.py:53-74) + the autograd replay through them.
//
// One workgroup = 4 wavefronts (256 threads, one wave per SIMD) evaluates TILE=64 points through all nine
// layers without touching HBM in between:
//   * activations live in LDS, feature-major X[feature][ray] (512 x 64 f32 = 128 KiB), updated in place;
//   * each dense layer is computed transposed, Y^T = W * X^T, with v_mfma_f32_32x32x2_f32: the weight
//     matrix is the MFMA "A" operand (one f32 per lane, lane (i,h) holds W[o=i][k=h]) and the activations
//     are the "B" operand (lane (j,h) holds X[k=h][ray=j]) read from LDS with conflict-free ds_read_b32;
//   * wave w owns output rows [w*O/4, (w+1)*O/4) for all 64 rays: NOB x 2 accumulator tiles of 32x32
//     (128 accumulator registers for O=512), so every weight fragment is used for 2 MFMAs and every
//     activation fragment for NOB MFMAs; weights are streamed from L2 exactly once per workgroup per
//     layer as host-pre-packed, fully coalesced float4 fragments (1 KiB per wave-load), prefetched one
//     8-feature group (32 MFMAs = 2048 cycles) ahead;
//   * the k-loop visits features in natural order, and an f32 MFMA is bit-for-bit a k-ordered fmaf chain,
//     so the result equals the oracle's (and a scalar CPU) fmaf chain started from the bias;
//   * the latent columns of lin0 / lin4 are folded into per-render constant vectors c0 / c4 (k_prep).
// Backward re-uses the same dense loop with transposed weight fragments; ReLU masks of the recomputed
// forward are kept as bitmasks in registers (32 VGPRs) because forward outputs and backward deltas of a
// layer sit in the same lane/register positions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "distr_dense_asm.hpp"

namespace distr {

// Non-template kernels: defined once, by the translation unit that launches them (distr_api.hip). The translation units that only hold
// explicit instantiations of the big template kernels (distr_inst_*.hip, compiled in parallel) define DISTR_GLOBAL as `static __global__`:
// what they do not use is not emitted there.
#ifndef DISTR_GLOBAL
#define DISTR_GLOBAL __global__
#endif

// A reference to a kernel argument that the optimiser cannot see through (taken from the kernarg segment pointer, NOT from the address of
// the parameter: that would force a private copy of it): loads through it are still scalar loads from the constant
// address space, but they depend on THIS statement and are not hoisted above it. Inside a loop that runs a decoder evaluation per
// iteration (a sticky tile's march steps, the step loop of the persistent tail launch) loop-invariant kernel arguments are otherwise
// loaded once in front of the loop and kept alive across every evaluation -- dozens of 64-bit pointers next to the 512-register tile:
// parked in AGPRs, copied to VGPR pairs, spilled to scratch and reloaded (each reload a vmcnt(0)) on the critical path of every step.
template <class T>
__device__ __forceinline__ const T& kernarg_ref(size_t offset) {     // the kernel argument of type T at byte `offset` of the kernarg segment
  typedef const char __attribute__((address_space(4)))* cptr;
  cptr p = (cptr)__builtin_amdgcn_kernarg_segment_ptr() + offset;
  asm volatile("" : "+s"(p));
  return *(const T*)(uintptr_t)p;
}

constexpr int HID = 512;
constexpr int LAT = 256;
constexpr int NTHREADS = 256;

// (f32x16 / f32x4 / rsrc_t: distr_dense_asm.hpp)

// LDS byte offset of a __shared__ object (low half of its flat address) -- the address operand of the hand-written ds_* ops
__device__ __forceinline__ uint32_t lds_off(const void* p) { return (uint32_t)(uintptr_t)p; }
// raw buffer descriptor over a fragment stream (stride 0, 1 GiB range, gfx9 dword-3 flags): lets the dense loop walk the
// stream with an SGPR offset instead of 64-bit VGPR address arithmetic
__device__ __forceinline__ rsrc_t weight_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x40000000, 0x00020000);
}

struct DecoderDev {
  const float* Wf[8];   // forward A-fragments of lin0..lin7 (lin0: K padded to 8; lin3: O padded to 256; lin4: K=256)
  const float* Wb[8];   // backward (transposed) A-fragments, index = layer whose weights are used (1..7)
  const float* bias[8]; // biases of lin1,2,3(padded),5,6,7 ([0],[4]: unused -- c0/c4 are per render)
  const float* W0lat_t; // [256][512]  lin0 latent columns, k-major
  const float* W4lat_t; // [256][512]
  const float* W0lat;   // [512][256]
  const float* W4lat;   // [512][256]
  const float* b0;      // [512]
  const float* b4;      // [512]
  const float* w8;      // [nout][512]  lin8 (nout = 1: SDF; 3: the colour decoder of decode_color)
  const float* W0x;     // [3][512]   lin0 xyz columns
  float b8;             // lin8 bias of row 0
  float b8x[2];         // lin8 biases of rows 1, 2 (colour decoder)
  int32_t nlat;         // latent length folded into c0 / c4 (256; 256 + color_size for the colour decoder)
};

// A tile = RB blocks of 32 rays. RB=2: 64 rays, 133 KiB LDS, one workgroup per CU. RB=1: 32 rays, 67 KiB LDS, two
// workgroups per CU (each hides the other's prologue / epilogue / barrier stalls; half the tile latency).
// 64-ray tiles also stage the eight layers' accumulator start values (c0, b1, b2, b3, c4, b5, b6, b7: 16 KiB) in LDS once
// per tile, so that a layer does not begin with an exposed L2 round trip for its biases (the 32-ray tile has no LDS
// left for this: two of them share a CU).
template <int RB> struct SmemBias { };
template <> struct alignas(16) SmemBias<2> { float bias[8 * HID]; };

template <int RB>
struct alignas(16) Smem : SmemBias<RB> {
  static constexpr int TILE = 32 * RB;
  float X[HID * TILE];   // activations / deltas [feature][ray]
  float xyz[4 * TILE];   // rows 0..2: sample points of the tile
  float part[12 * TILE]; // lin8 partial chains [4][TILE]; backward: xyz-gradient partials [3][4][TILE]
  float aux[4 * TILE];   // backward: row 0 = d8, rows 1..3 = d/dxyz through lin4's xyz columns
};

template <int RB>
__device__ __forceinline__ void stage_bias(const DecoderDev& D, const float* __restrict__ c0, const float* __restrict__ c4, Smem<RB>& S) {
  if constexpr (RB == 2) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      const float* src = (l == 0) ? c0 : (l == 4) ? c4 : D.bias[l];
      const int n = (l == 3) ? 256 : HID;
#pragma unroll
      for (int i = tid; i < HID; i += NTHREADS) S.bias[l * HID + i] = (i < n) ? src[i] : 0.f;
    }
  }
}

// start values of layer l for this tile: LDS copy (RB == 2, after stage_bias + barrier) or the global arrays
template <int RB>
__device__ __forceinline__ const float* layer_init(const DecoderDev& D, const float* c0, const float* c4, const Smem<RB>& S, int l) {
  if constexpr (RB == 2) return S.bias + l * HID;
  else return (l == 0) ? c0 : (l == 4) ? c4 : D.bias[l];
}

// ---------------------------------------------------------------------------------------- scalar math
// tanh in explicit IEEE operations so that host (oracle) and device agree bit for bit.
__device__ __forceinline__ float exp_spec(float x) {
  const float LOG2E = 1.44269504088896341f;
  const float C1 = 0.693359375f;
  const float C2 = -2.12194440e-4f;
  float n = floorf(__builtin_fmaf(LOG2E, x, 0.5f));
  float r = __builtin_fmaf(n, -C1, x);
  r = __builtin_fmaf(n, -C2, r);
  float z = r * r;
  float p = 1.9875691500e-4f;
  p = __builtin_fmaf(p, r, 1.3981999507e-3f);
  p = __builtin_fmaf(p, r, 8.3334519073e-3f);
  p = __builtin_fmaf(p, r, 4.1665795894e-2f);
  p = __builtin_fmaf(p, r, 1.6666665459e-1f);
  p = __builtin_fmaf(p, r, 5.0000001201e-1f);
  p = __builtin_fmaf(p, z, r);
  p = p + 1.0f;
  int32_t bits = ((int32_t)n + 127) << 23;
  return p * __int_as_float(bits);
}

__device__ __forceinline__ float tanh_spec(float x) {
  float a = fabsf(x);
  if (a < 0.625f) {
    float z = x * x;
    float p = -5.70498872745e-3f;
    p = __builtin_fmaf(p, z, 2.06390887954e-2f);
    p = __builtin_fmaf(p, z, -5.37397155531e-2f);
    p = __builtin_fmaf(p, z, 1.33314422036e-1f);
    p = __builtin_fmaf(p, z, -3.33332819422e-1f);
    p = p * z;
    return __builtin_fmaf(p, x, x);
  }
  if (a > 10.0f) return x > 0 ? 1.0f : -1.0f;
  float e = exp_spec(a + a);
  float r = 1.0f - 2.0f / (e + 1.0f);
  return x > 0 ? r : -r;
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------------------------------- dense layer
// D rows of one 32x32 accumulator register r on lane (j,h): row = (r&3) + 8*(r>>2) + 4*h, col = j.
template <int NOB, int RB>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[NOB][RB], const float* __restrict__ init, int row0, int h) {
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(init + row0 + 32 * ob + 8 * q + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[ob][rb][4 * q + j] = b[j];
      }
    }
  }
}

template <int NOB, int RB>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[NOB][RB]) {
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[ob][rb][r] = 0.f;
    }
}

// acc[ob][rb] += W[rows of this wave][0..K) * X[0..K)[rays]   (K multiple of 8, natural k order)
// Wp: packed fragments, float4 index ((g*4 + wave)*NOB + ob)*64 + lane  holds
//     { W[o][8g+2s+h] : s=0..3 },  o = wave*32*NOB + 32*ob + (lane&31), h = lane>>5.
template <int K, int NOB, int RB>
__device__ __forceinline__ void dense(const float* __restrict__ Wp, const float* X, f32x16 (&acc)[NOB][RB], int wave,
                                      int lane) {
  constexpr int NG = K / 8;
  constexpr int TILE = 32 * RB;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)wave * NOB * 64 + lane;
  const float* xb = X + (lane >> 5) * TILE + (lane & 31);
  f32x4 a[NOB];
  float b[4][RB];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) a[ob] = wp[ob * 64];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) b[s][rb] = xb[2 * s * TILE + 32 * rb];
#pragma unroll 2
  for (int g = 0; g < NG; ++g) {
    // register double buffer: fetch group g+1 (weights from L2, activations from LDS) before the MFMAs of group g;
    // the scheduling barrier keeps the loads at the top so they get a full group of MFMA time as cover
    f32x4 an[NOB];
    float bn[4][RB];
    const int gn = (g + 1 < NG) ? g + 1 : g;
    const f32x4* wn = wp + (size_t)gn * (4 * NOB * 64);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) an[ob] = wn[ob * 64];
    const float* xg = xb + (size_t)gn * 8 * TILE;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) bn[s][rb] = xg[2 * s * TILE + 32 * rb];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          acc[ob][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ob][s], b[s][rb], acc[ob][rb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) a[ob] = an[ob];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) b[s][rb] = bn[s][rb];
  }
}

// dense() with (part of) the layer's first weight group already in registers -- requested during the previous layer --
// and the look-ahead slot of the last iteration, which would otherwise re-fetch the last group, pointed at the NEXT
// layer's first group: no layer starts with an exposed L2 round trip for its weights. pre[0..NIN) = this wave's first NIN
// fragments of group 0 (NIN <= NOB); on return pre[0..NOUT) = the first NOUT (<= NOB) fragments of the next layer's group 0
// (WpNext, whose waves own NOBN row blocks each); WpNext == nullptr: nothing is fetched.
template <int K, int NOB, int RB, int NIN, int NOUT, int NOBN>
__device__ __forceinline__ void dense_pf(const float* __restrict__ Wp, const float* X, f32x16 (&acc)[NOB][RB], int wave, int lane,
                                         f32x4 (&pre)[4], const float* __restrict__ WpNext) {
  constexpr int NG = K / 8;
  constexpr int TILE = 32 * RB;
  static_assert(NIN <= NOB && NOUT <= NOB && NOUT <= NOBN, "prefetch slots");
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)wave * NOB * 64 + lane;
  const f32x4* wnext = WpNext ? reinterpret_cast<const f32x4*>(WpNext) + (size_t)wave * NOBN * 64 + lane : wp + (size_t)(NG - 1) * (4 * NOB * 64);
  const float* xb = X + (lane >> 5) * TILE + (lane & 31);
  f32x4 a[NOB];
  float b[4][RB];
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) a[ob] = (ob < NIN) ? pre[ob] : wp[ob * 64];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) b[s][rb] = xb[2 * s * TILE + 32 * rb];
#pragma unroll 2
  for (int g = 0; g < NG; ++g) {
    f32x4 an[NOB];
    float bn[4][RB];
    const bool last = (g + 1 == NG);
    const int gn = last ? g : g + 1;
    const f32x4* wn = last ? wnext : wp + (size_t)gn * (4 * NOB * 64);
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) an[ob] = wn[((last && ob >= NOUT) ? 0 : ob) * 64];
    const float* xg = xb + (size_t)gn * 8 * TILE;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) bn[s][rb] = xg[2 * s * TILE + 32 * rb];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
          acc[ob][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ob][s], b[s][rb], acc[ob][rb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) a[ob] = an[ob];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) b[s][rb] = bn[s][rb];
  }
#pragma unroll
  for (int ob = 0; ob < NOUT; ++ob) pre[ob] = a[ob];
}

// Write a layer's accumulators back to X (in place). RELU: max(x,0); mask out: bit (rb*16+r) of mask[ob] = x>0.
// GATE: multiply by the given mask bits instead (backward: delta = mask ? acc : 0).
// (sign/zero tests are done with integer bit arithmetic: a float compare per element would make hipcc keep 128
// wave-wide lane masks in SGPR pairs and spill them)
__device__ __forceinline__ uint32_t pos_bit(float v) {  // 1 if v > 0 (v finite), else 0
  const uint32_t u = __float_as_uint(v);
  return ((~u) >> 31) & ((u | (0u - u)) >> 31);
}
__device__ __forceinline__ float gate(float v, uint32_t bit) {  // bit ? v : +0
  return __uint_as_float(__float_as_uint(v) & (0u - bit));
}

template <int NOB, int RB, bool RELU, bool GATE, bool KEEP = true>
__device__ __forceinline__ void writeback(float* X, const f32x16 (&acc)[NOB][RB], int row0, int lane, uint32_t (&mask)[4]) {
  constexpr int TILE = 32 * RB;
  const int h = lane >> 5, j = lane & 31;
  __builtin_amdgcn_sched_barrier(0);  // keep the mask packing here: do not let raw accumulators stay live (spill)
#pragma unroll
  for (int ob = 0; ob < NOB; ++ob) {
    uint32_t m = GATE ? mask[ob] : 0u;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[ob][rb][r];
        if (GATE) {
          v = gate(v, (m >> (rb * 16 + r)) & 1u);
        } else if (KEEP && RELU) {
          // relu on the bit pattern (v_max_i32), then bit = min(bits, 1) (v_min_u32): 3 VALU ops per element
          const int32_t rbits = max(__float_as_int(v), 0);
          v = __int_as_float(rbits);
          m |= min((uint32_t)rbits, 1u) << (rb * 16 + r);
        } else if (KEEP) {
          const uint32_t pb = pos_bit(v);
          m |= pb << (rb * 16 + r);
        } else if (RELU) {
          v = __int_as_float(max(__float_as_int(v), 0));  // relu on the bit pattern (v_max_i32)
        }
        X[row * TILE + 32 * rb + j] = v;
      }
    }
    if (!GATE && KEEP) {
      // opaque to the optimiser: otherwise LLVM sees through the bit packing ((m>>n)&1 == pos_bit(acc_n)) and keeps all
      // 128 x 8 forward accumulators alive (in scratch) until the backward pass instead of the 4 mask registers
      asm volatile("" : "+v"(m));
      mask[ob] = m;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------- saved ReLU masks
// A ray's 512-byte mask block = 8 chunks of 64 bytes, chunk (wave, h) = the 32 16-bit words masks[l][ob] (l = 0..7,
// ob = 0..3) that lane (j, h) of wave `wave` holds for that ray. The layout is per ray, so a block written by a
// 32- or 64-ray forward tile can be read by any backward tile that puts the ray on lane j / block rb.
template <int RB>
__device__ __forceinline__ void store_mask_chunk(uint4* blk /*block base*/, const uint32_t (&masks)[8][4], int rb, int wave, int h) {
  uint32_t q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e0 = 2 * i, e1 = 2 * i + 1;
    const uint32_t w0 = (masks[e0 >> 2][e0 & 3] >> (16 * rb)) & 0xffffu;
    const uint32_t w1 = (masks[e1 >> 2][e1 & 3] >> (16 * rb)) & 0xffffu;
    q[i] = w0 | (w1 << 16);
  }
  uint4* dst = blk + (wave * 2 + h) * 4;
#pragma unroll
  for (int v = 0; v < 4; ++v) dst[v] = make_uint4(q[4 * v], q[4 * v + 1], q[4 * v + 2], q[4 * v + 3]);
}

// masks[l][ob] |= chunk words << (16*rb)
__device__ __forceinline__ void load_mask_chunk(const uint4* blk, uint32_t (&masks)[8][4], int rb, int wave, int h) {
  const uint4* src = blk + (wave * 2 + h) * 4;
  uint32_t q[16];
#pragma unroll
  for (int v = 0; v < 4; ++v) { const uint4 t = src[v]; q[4 * v] = t.x; q[4 * v + 1] = t.y; q[4 * v + 2] = t.z; q[4 * v + 3] = t.w; }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e0 = 2 * i, e1 = 2 * i + 1;
    masks[e0 >> 2][e0 & 3] |= (q[i] & 0xffffu) << (16 * rb);
    masks[e1 >> 2][e1 & 3] |= (q[i] >> 16) << (16 * rb);
  }
}

// ---------------------------------------------------------------------------------------- forward tile
// Preconditions: S.xyz rows 0..2 hold the TILE points (x row, y row, z row), visible to all threads (barrier done).
// One row of lin8 on the h7 activations in S.X: four 128-long chains per ray (one per wave), combined in fixed order.
// Returns (every thread, for ray = tid & (TILE-1)) the pre-tanh value. Ends with every thread past a barrier on S.part
// reads only if the caller adds one before the next call (colour decoder: three rows).
template <int RB>
__device__ __forceinline__ float lin8_row(const float* __restrict__ w8row, float b8, Smem<RB>& S) {
  constexpr int TILE = 32 * RB;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ray = tid & (TILE - 1);
  {
    float p = 0.f;
    const float* w8 = w8row + wave * 128;
    const float* xr = S.X + (size_t)wave * 128 * TILE + ray;
#pragma unroll 8
    for (int k = 0; k < 128; ++k) p = __builtin_fmaf(w8[k], xr[k * TILE], p);
    S.part[wave * TILE + ray] = p;   // (RB=1: both half-waves hold the same value)
  }
  __syncthreads();
  return ((S.part[ray] + S.part[TILE + ray]) + (S.part[2 * TILE + ray] + S.part[3 * TILE + ray])) + b8;
}

// Returns (every thread, for ray = tid & (TILE-1)) the pre-tanh output. masks[l] = ReLU bitmasks of layer l
// (bit rb*16+r of masks[l][ob]). DEBUG_STOP: (test builds only) return right after layer `stop` is in X.
// STAGED: the caller already ran stage_bias (early, so that its loads overlap the tile's prologue).
template <int RB, bool KEEP, bool DEBUG_STOP = false, bool STAGED = false>
__device__ __forceinline__ float mlp_forward(const DecoderDev& D, const float* __restrict__ c0,
                                             const float* __restrict__ c4, Smem<RB>& S, uint32_t (&masks)[8][4],
                                             int stop = 8, long long* ts = nullptr) {
  constexpr int TILE = 32 * RB;
  // test builds: shader-clock / wall-clock stamps of wave 0 at phase boundaries (ts[2i], ts[2i+1])
#define DISTR_TS(i) do { if (DEBUG_STOP && ts && threadIdx.x == 0) { ts[2 * (i)] = (long long)__builtin_readcyclecounter(); ts[2 * (i) + 1] = (long long)wall_clock64(); } } while (0)
  DISTR_TS(0);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int h = lane >> 5;
  const int ray = tid & (TILE - 1);
  float* X = S.X;
  // first weight group of every layer travels while the previous layer finishes (dense_pf)
  f32x4 wpre[4];
  {
    const f32x4* w0 = reinterpret_cast<const f32x4*>(D.Wf[0]) + (size_t)wave * 4 * 64 + lane;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) wpre[ob] = w0[ob * 64];
  }
  if (!STAGED) stage_bias<RB>(D, c0, c4, S);
  // layer-0 input rows: xyz + zero padding to K=8
#pragma unroll
  for (int i = tid; i < 8 * TILE; i += NTHREADS) X[i] = (i < 3 * TILE) ? S.xyz[i] : 0.f;
  __syncthreads();
  {
    f32x16 acc[4][RB];
    acc_init<4, RB>(acc, layer_init<RB>(D, c0, c4, S, 0), wave * 128, h);
    dense_pf<8, 4, RB, 4, 4, 4>(D.Wf[0], X, acc, wave, lane, wpre, D.Wf[1]);
    DISTR_TS(1);
    __syncthreads();
    writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[0]);
    __syncthreads();
    DISTR_TS(2);
  }
  if (DEBUG_STOP && stop == 0) return 0.f;
  if constexpr (RB == 2) {
    // 64-ray tile: layers 1..7 on the hand-scheduled loop (distr_dense_asm.hpp). Every layer's accumulators start from the
    // LDS copy of its bias (srcC of the first MFMAs); the next layer's first weight group travels in `wpre` across the
    // write-back; S.part..S.aux (4 KiB, idle during the layers) is the scratch of the tuple -> register move.
    const uint32_t xaddr = lds_off(X) + (uint32_t)h * (TILE * 4) + (uint32_t)(lane & 31) * 4;
    const uint32_t voff = (uint32_t)lane * 16;
    const uint32_t scratch = lds_off(S.part) + (uint32_t)wave * 1024 + (uint32_t)lane * 16;
    const uint32_t bias4 = lds_off(S.bias) + (uint32_t)wave * 512 + (uint32_t)h * 16;   // + l * 2048: rows wave*128 .. of layer l
    const uint32_t bias2 = lds_off(S.bias) + 3 * 2048 + (uint32_t)wave * 256 + (uint32_t)h * 16;   // lin3: rows wave*64 ..
    const uint32_t soff4 = (uint32_t)wave * 4096, soff2 = (uint32_t)wave * 2048;
    rsrc_t rs[8];
#pragma unroll
    for (int l = 1; l < 8; ++l) rs[l] = weight_rsrc(D.Wf[l]);
    {
      f32x16 acc[4][2];
      dense_asm_k512_n4_o4_bias(acc, wpre, xaddr, voff, rs[1], rs[2], soff4, soff4, bias4 + 1 * 2048, scratch);
      DISTR_TS(3);
      __syncthreads();
      writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[1]);
      __syncthreads();
      DISTR_TS(4);
      if (DEBUG_STOP && stop == 1) return 0.f;
    }
    {
      f32x16 acc[4][2];
      dense_asm_k512_n4_o2_bias(acc, wpre, xaddr, voff, rs[2], rs[3], soff4, soff2, bias4 + 2 * 2048, scratch);
      DISTR_TS(5);
      __syncthreads();
      writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[2]);
      __syncthreads();
      DISTR_TS(6);
      if (DEBUG_STOP && stop == 2) return 0.f;
    }
    {  // lin3: 512 -> 253 (+3 rows that carry xyz into lin4)
      f32x16 acc[2][2];
      dense_asm_k512_n2_o4_bias(acc, wpre, xaddr, voff, rs[3], rs[4], soff2, soff4, bias2, scratch);
      DISTR_TS(7);
      __syncthreads();
      masks[3][2] = 0; masks[3][3] = 0;
      writeback<2, RB, true, false, KEEP>(X, acc, wave * 64, lane, masks[3]);
      __syncthreads();
      if (tid < 3 * TILE) X[253 * TILE + tid] = S.xyz[tid];
      __syncthreads();
      DISTR_TS(8);
      if (DEBUG_STOP && stop == 3) return 0.f;
    }
    {  // lin4: [x3(253) | xyz(3)] -> 512, latent part folded into c4
      f32x16 acc[4][2];
      dense_asm_k256_n4_o4_bias(acc, wpre, xaddr, voff, rs[4], rs[5], soff4, soff4, bias4 + 4 * 2048, scratch);
      DISTR_TS(9);
      __syncthreads();
      writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[4]);
      __syncthreads();
      DISTR_TS(10);
      if (DEBUG_STOP && stop == 4) return 0.f;
    }
#pragma unroll
    for (int l = 5; l <= 7; ++l) {
      f32x16 acc[4][2];
      if (l < 7) dense_asm_k512_n4_o4_bias(acc, wpre, xaddr, voff, rs[l], rs[l < 7 ? l + 1 : l], soff4, soff4, bias4 + l * 2048, scratch);
      else dense_asm_k512_n4_o0_bias(acc, wpre, xaddr, voff, rs[l], rs[l], soff4, soff4, bias4 + l * 2048, scratch);
      DISTR_TS(2 * l + 1);
      __syncthreads();
      writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[l]);
      __syncthreads();
      DISTR_TS(2 * l + 2);
      if (DEBUG_STOP && stop == l) return 0.f;
    }
    const float pre_asm = lin8_row<RB>(D.w8, D.b8, S);
    DISTR_TS(17);
    return pre_asm;
  }
  {
    f32x16 acc[4][RB];
    acc_init<4, RB>(acc, layer_init<RB>(D, c0, c4, S, 1), wave * 128, h);
    dense_pf<512, 4, RB, 4, 4, 4>(D.Wf[1], X, acc, wave, lane, wpre, D.Wf[2]);
    DISTR_TS(3);
    __syncthreads();
    writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[1]);
    __syncthreads();
    DISTR_TS(4);
    if (DEBUG_STOP && stop == 1) return 0.f;
  }
  {
    f32x16 acc[4][RB];
    acc_init<4, RB>(acc, layer_init<RB>(D, c0, c4, S, 2), wave * 128, h);
    dense_pf<512, 4, RB, 4, 2, 2>(D.Wf[2], X, acc, wave, lane, wpre, D.Wf[3]);
    DISTR_TS(5);
    __syncthreads();
    writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[2]);
    __syncthreads();
    DISTR_TS(6);
    if (DEBUG_STOP && stop == 2) return 0.f;
  }
  {  // lin3: 512 -> 253 (+3 rows that carry xyz into lin4)
    f32x16 acc[2][RB];
    acc_init<2, RB>(acc, layer_init<RB>(D, c0, c4, S, 3), wave * 64, h);
    dense_pf<512, 2, RB, 2, 2, 4>(D.Wf[3], X, acc, wave, lane, wpre, D.Wf[4]);
    DISTR_TS(7);
    __syncthreads();
    masks[3][2] = 0; masks[3][3] = 0;
    writeback<2, RB, true, false, KEEP>(X, acc, wave * 64, lane, masks[3]);
    __syncthreads();
    if (tid < 3 * TILE) X[253 * TILE + tid] = S.xyz[tid];
    __syncthreads();
    DISTR_TS(8);
  }
  if (DEBUG_STOP && stop == 3) return 0.f;
  {  // lin4: [x3(253) | xyz(3)] -> 512, latent part folded into c4
    f32x16 acc[4][RB];
    acc_init<4, RB>(acc, layer_init<RB>(D, c0, c4, S, 4), wave * 128, h);
    dense_pf<256, 4, RB, 2, 4, 4>(D.Wf[4], X, acc, wave, lane, wpre, D.Wf[5]);
    DISTR_TS(9);
    __syncthreads();
    writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[4]);
    __syncthreads();
    DISTR_TS(10);
  }
  if (DEBUG_STOP && stop == 4) return 0.f;
#pragma unroll
  for (int l = 5; l <= 7; ++l) {
    f32x16 acc[4][RB];
    acc_init<4, RB>(acc, layer_init<RB>(D, c0, c4, S, l), wave * 128, h);
    dense_pf<512, 4, RB, 4, 4, 4>(D.Wf[l], X, acc, wave, lane, wpre, (l < 7) ? D.Wf[l + 1] : nullptr);
    DISTR_TS(2 * l + 1);
    __syncthreads();
    writeback<4, RB, true, false, KEEP>(X, acc, wave * 128, lane, masks[l]);
    __syncthreads();
    DISTR_TS(2 * l + 2);
    if (DEBUG_STOP && stop == l) return 0.f;
  }
  const float pre = lin8_row<RB>(D.w8, D.b8, S);
  DISTR_TS(17);
#undef DISTR_TS
  return pre;
}

// ---------------------------------------------------------------------------------------- backward tile
// Preconditions: mlp_forward<RB,true> just ran on this tile (X = h7, masks filled); S.aux row 0 = d8[ray] =
// coef*(1-y^2), visible to all threads. On return: S.aux rows 1..3 hold d(coef*f)/d xyz per ray; sd0/sd4 (if non-null)
// receive the row sums over the TILE rays of delta0 / delta4.
template <int RB>
__device__ __forceinline__ void row_sums(const float* X, float* __restrict__ dst, int tid) {
  constexpr int TILE = 32 * RB;
  const int lane = tid & 63;
#pragma unroll 1
  for (int rr = 0; rr < 2; ++rr) {
    const int row = tid + rr * 256;
    float s = 0.f;
#pragma unroll 8
    for (int i = 0; i < TILE; ++i) s += X[row * TILE + ((i + lane) & (TILE - 1))];   // rotated: conflict-free
    dst[row] = s;
  }
}

// NOUT: rows of lin8 (1: SDF decoder; 3: colour decoder -- then S.aux rows 0..2 hold the three d8 rows and
// delta7 = relu'(h7) * sum_c w8[c][k] * d8_c, accumulated in channel order)
template <int RB, int NOUT = 1>
__device__ __forceinline__ void mlp_backward(const DecoderDev& D, Smem<RB>& S, uint32_t (&masks)[8][4],
                                             float* __restrict__ sd0, float* __restrict__ sd4) {
  constexpr int TILE = 32 * RB;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int h = lane >> 5, j = lane & 31;
  const int ray = tid & (TILE - 1);
  float* X = S.X;
  // delta7[k][ray] = relu'(h7) * w8[k] * d8[ray]
  {
    float d8[NOUT][RB];
#pragma unroll
    for (int c = 0; c < NOUT; ++c)
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) d8[c][rb] = S.aux[c * TILE + 32 * rb + j];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      const uint32_t m = masks[7][ob];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wave * 128 + 32 * ob + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          float v = D.w8[row] * d8[0][rb];
#pragma unroll
          for (int c = 1; c < NOUT; ++c) v = __builtin_fmaf(D.w8[c * HID + row], d8[c][rb], v);
          X[row * TILE + 32 * rb + j] = gate(v, (m >> (16 * rb + r)) & 1u);
        }
      }
    }
  }
  __syncthreads();
  if constexpr (RB == 2) {
    // 64-sample tile: the dX chain on the hand-scheduled loop (distr_dense_asm.hpp), accumulators starting from zero, the next
    // layer's first transposed weight group travelling in `t` across each write-back; scratch = S.xyz..S.part (4 KiB, idle here)
    const uint32_t xaddr = lds_off(X) + (uint32_t)h * (TILE * 4) + (uint32_t)j * 4;
    const uint32_t voff = (uint32_t)lane * 16;
    const uint32_t scratch = lds_off(S.xyz) + (uint32_t)wave * 1024 + (uint32_t)lane * 16;
    const uint32_t soff4 = (uint32_t)wave * 4096, soff2 = (uint32_t)wave * 2048;
    rsrc_t rs[8];
#pragma unroll
    for (int l = 1; l < 8; ++l) rs[l] = weight_rsrc(D.Wb[l]);
    f32x4 t[4];
    {
      const f32x4* w7 = reinterpret_cast<const f32x4*>(D.Wb[7]) + (size_t)wave * 4 * 64 + lane;
#pragma unroll
      for (int ob = 0; ob < 4; ++ob) t[ob] = w7[ob * 64];
    }
#pragma unroll
    for (int l = 7; l >= 5; --l) {  // delta_l (512) -> delta_{l-1} (512)
      f32x16 acc[4][2];
      if (l > 5) dense_asm_k512_n4_o4_zero(acc, t, xaddr, voff, rs[l], rs[l > 5 ? l - 1 : l], soff4, soff4, 0u, scratch);
      else dense_asm_k512_n4_o2_zero(acc, t, xaddr, voff, rs[5], rs[4], soff4, soff2, 0u, scratch);
      __syncthreads();
      writeback<4, RB, false, true>(X, acc, wave * 128, lane, masks[l - 1]);
      __syncthreads();
    }
    if (sd4) row_sums<RB>(X, sd4, tid);  // X = delta4
    {  // lin4^T: delta4 (512) -> [delta3 (253) | d xyz (3)]
      f32x16 acc[2][2];
      dense_asm_k512_n2_o4_zero(acc, t, xaddr, voff, rs[4], rs[3], soff2, soff4, 0u, scratch);
      __syncthreads();
      writeback<2, RB, false, true>(X, acc, wave * 64, lane, masks[3]);  // rows 253..255 have mask 0 -> written as 0
      if (wave == 3 && h == 1) {
#pragma unroll
        for (int r = 13; r < 16; ++r)
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) S.aux[(1 + r - 13) * TILE + 32 * rb + j] = acc[1][rb][r];
      }
      __syncthreads();
    }
    {  // lin3^T: delta3 (256 rows, 253 real) -> delta2 (512)
      f32x16 acc[4][2];
      dense_asm_k256_n4_o4_zero(acc, t, xaddr, voff, rs[3], rs[2], soff4, soff4, 0u, scratch);
      __syncthreads();
      writeback<4, RB, false, true>(X, acc, wave * 128, lane, masks[2]);
      __syncthreads();
    }
#pragma unroll
    for (int l = 2; l >= 1; --l) {
      f32x16 acc[4][2];
      if (l > 1) dense_asm_k512_n4_o4_zero(acc, t, xaddr, voff, rs[2], rs[1], soff4, soff4, 0u, scratch);
      else dense_asm_k512_n4_o0_zero(acc, t, xaddr, voff, rs[1], rs[1], soff4, soff4, 0u, scratch);
      __syncthreads();
      writeback<4, RB, false, true>(X, acc, wave * 128, lane, masks[l - 1]);
      __syncthreads();
    }
  } else {
#pragma unroll
  for (int l = 7; l >= 5; --l) {  // delta_l (512) -> delta_{l-1} (512)
    f32x16 acc[4][RB];
    acc_zero<4, RB>(acc);
    dense<512, 4, RB>(D.Wb[l], X, acc, wave, lane);
    __syncthreads();
    writeback<4, RB, false, true>(X, acc, wave * 128, lane, masks[l - 1]);
    __syncthreads();
  }
  if (sd4) row_sums<RB>(X, sd4, tid);  // X = delta4
  {  // lin4^T: delta4 (512) -> [delta3 (253) | d xyz (3)]
    f32x16 acc[2][RB];
    acc_zero<2, RB>(acc);
    dense<512, 2, RB>(D.Wb[4], X, acc, wave, lane);
    __syncthreads();
    writeback<2, RB, false, true>(X, acc, wave * 64, lane, masks[3]);  // rows 253..255 have mask 0 -> written as 0
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
    acc_zero<4, RB>(acc);
    dense<256, 4, RB>(D.Wb[3], X, acc, wave, lane);
    __syncthreads();
    writeback<4, RB, false, true>(X, acc, wave * 128, lane, masks[2]);
    __syncthreads();
  }
#pragma unroll
  for (int l = 2; l >= 1; --l) {
    f32x16 acc[4][RB];
    acc_zero<4, RB>(acc);
    dense<512, 4, RB>(D.Wb[l], X, acc, wave, lane);
    __syncthreads();
    writeback<4, RB, false, true>(X, acc, wave * 128, lane, masks[l - 1]);
    __syncthreads();
  }
  }
  if (sd0) row_sums<RB>(X, sd0, tid);  // X = delta0
  // d xyz through lin0's xyz columns: 3 x four 128-long chains per ray
  {
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    const float* wx = D.W0x + wave * 128;
    const float* xr = X + (size_t)wave * 128 * TILE + ray;
#pragma unroll 4
    for (int k = 0; k < 128; ++k) {
      const float d = xr[k * TILE];
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

// =====================================================================================================================
// 16-ray tile (forward only) on v_mfma_f32_16x16x4_f32 -- used for the live-ray tail of the march, where a launch has
// fewer tiles than CUs and its cost is ONE tile latency: 16 rays take half the MFMA time of 32.
// Fragment maps of the 16x16x4 form: A lane (i = l&15, kq = l>>4) = A[i][kq]; B lane (j = l&15, kq) = B[kq][j];
// D (4 registers) col = l&15, row = 4*(l>>4) + r. The chain order is again k-natural, so values are bit-identical to
// the 32x32x2 tiles. Wave w owns rows [w*O/4, (w+1)*O/4) as NB blocks of 16.
struct Smem16 {
  float X[HID * 16];
  float xyz[4 * 16];
  float part[4 * 16];
  long long mb[16];
};

struct Smem16CL : Smem16 {
  uint16_t mk[16][256];   // KEEP: the 16 rays' 512-byte mask blocks in store_mask_chunk's format (lead member: every row; sticky tiles: every member, its own rows)
  int32_t fail;           // != 0: this member gave up on the cluster (assembly or a staging unit timed out / aborted)
  int32_t sc1;            // the cluster's members sit on more than one XCD: slices are stored write-through (cl_assemble)
  int32_t cont;           // sticky tiles: any ray of the tile still live after this step
  int32_t went;           // the cluster assembled (`go` seen): a member that gives up AFTER that may have published every slice the others need
  float sk[8][16];        // sticky tiles: the rays' selected-row keys (sdf, |.| ascending) and mask slots between steps
  int32_t ssl[8][16];     // (DISTR_MAX_BUFFER_SIZE rows; kept here, not in registers, across the decoder evaluation)
  float sst[4][16];       // k_tail's per-step tiles: the rays' m, init_now, maxbound, minabs between a tile's prologue and its epilogue (with sk / ssl)
};

// ... where cluster tiles can occur (MODE_FINE / MODE_COARSE kernels: one workgroup per compute unit). The granule requests land in fixed
// registers, not here; what IS here since round 6: lin0's operands of a STICKY tile. A sticky tile evaluates the decoder ~70 times in a row
// with the same latent constants and the same lin0 weights; fetched from global memory each time, lin0's compiler-issued loads sit in the
// wave's in-order memory queue BEHIND the three weight chunks of lin1 requested just before (the compiler's vmcnt wait for them drains
// the whole queue): every step started with ~1 us of waiting for lin1's weights before lin0's first MFMA. From LDS lin0 needs no vector
// memory operation at all and runs while the ring fills.
struct Smem16CLX : Smem16CL {
  float c0s[HID];          // c0 = b0 + W0[:, :256] latent (this view's)
  f32x4 w0s[4][8][64];     // lin0's A-fragments: [wave][row block][lane] = what dense16<16, 8> loads from DecoderDev16::Wf[0]
};

struct DecoderDev16 {
  const float* Wf[8];   // 16x16x4 A-fragments: float4 ((g*4 + w)*NB + ob)*64 + lane = { W[w*16*NB + 16*ob + i][16g + 4s + kq] : s=0..3 }
};

template <int NB>
__device__ __forceinline__ void acc_init16(f32x4 (&acc)[NB], const float* __restrict__ init, int row0, int kq) {
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) acc[ob] = *reinterpret_cast<const f32x4*>(init + row0 + 16 * ob + 4 * kq);
}

template <int K, int NB>
__device__ __forceinline__ void dense16(const float* __restrict__ Wp, const float* X, f32x4 (&acc)[NB], int wave, int lane) {
  constexpr int NG = K / 16;
  const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)wave * NB * 64 + lane;
  const float* xb = X + lane;   // (16g + 4s + kq)*16 + j  with  kq*16 + j = lane
  f32x4 a[NB];
  float b[4];
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) a[ob] = wp[ob * 64];
#pragma unroll
  for (int s = 0; s < 4; ++s) b[s] = xb[4 * s * 16];
#pragma unroll 2
  for (int g = 0; g < NG; ++g) {
    f32x4 an[NB];
    float bn[4];
    const int gn = (g + 1 < NG) ? g + 1 : g;
    const f32x4* wn = wp + (size_t)gn * (4 * NB * 64);
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) an[ob] = wn[ob * 64];
    const float* xg = xb + (size_t)gn * 16 * 16;
#pragma unroll
    for (int s = 0; s < 4; ++s) bn[s] = xg[4 * s * 16];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][s], b[s], acc[ob], 0, 0, 0);
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) a[ob] = an[ob];
#pragma unroll
    for (int s = 0; s < 4; ++s) b[s] = bn[s];
  }
}

// lin0 (K = 16: one k-group) of a sticky tile with the A-fragments in LDS (Smem16CLX::w0s): the same MFMAs in the same order as
// dense16<16, 8> -- k-step s outer, row block inner; each accumulator's chain is k-ordered -- without a vector memory operation
__device__ __forceinline__ void dense16_lin0_lds(const f32x4 (&w0)[8][64], const float* X, f32x4 (&acc)[8], int lane) {
  f32x4 a[8];
  float b[4];
#pragma unroll
  for (int ob = 0; ob < 8; ++ob) a[ob] = w0[ob][lane];
#pragma unroll
  for (int s = 0; s < 4; ++s) b[s] = X[lane + 4 * s * 16];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) acc[ob] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob][s], b[s], acc[ob], 0, 0, 0);
}

// relu + write back; nib (KEEP): bit (4*ob + r) = output (ob, r) > 0
template <int NB, bool KEEP>
__device__ __forceinline__ uint32_t writeback16(float* X, const f32x4 (&acc)[NB], int row0, int lane) {
  const int kq = lane >> 4, j = lane & 15;
  uint32_t m = 0u;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int32_t rbits = max(__float_as_int(acc[ob][r]), 0);
      if (KEEP) m |= min((uint32_t)rbits, 1u) << (4 * ob + r);
      X[(row0 + 16 * ob + 4 * kq + r) * 16 + j] = __int_as_float(rbits);
    }
  }
  if (KEEP) asm volatile("" : "+v"(m));
  __builtin_amdgcn_sched_barrier(0);
  return m;
}

// nib[l]: per-lane nibble words of layer l (see writeback16). Returns pre-tanh for ray = tid & 15.
// Layers 1..7 run on the hand-scheduled loop (distr_dense_asm.hpp, dense16_asm_*): no VALU / 64-bit-address instruction beside the
// 32-cycle MFMAs, the next layer's first weight group and bias are requested before the current layer's loop (nothing is
// exposed at a layer start but one LDS round trip). S.mk (idle outside the cluster path) is the scratch of the tuple -> register move.
template <bool KEEP>
__device__ __forceinline__ float mlp_forward16(const DecoderDev& D, const DecoderDev16& D16, const float* __restrict__ c0,
                                               const float* __restrict__ c4, Smem16CL& S, uint32_t (&nib)[8]) {
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int kq = lane >> 4;
  const int ray = tid & 15;
  float* X = S.X;
  const uint32_t xaddr = lds_off(X) + (uint32_t)lane * 4;
  const uint32_t voff = (uint32_t)lane * 16;
  const uint32_t scratch = lds_off(S.mk) + (uint32_t)wave * 1024 + (uint32_t)lane * 16;
  const uint32_t soff8 = (uint32_t)wave * 8192, soff4 = (uint32_t)wave * 4096;
  rsrc_t rs[8];
#pragma unroll
  for (int l = 1; l < 8; ++l) rs[l] = weight_rsrc(D16.Wf[l]);
  f32x4 t[8];
  {  // lin1's first group travels while lin0 runs
    const f32x4* w1 = reinterpret_cast<const f32x4*>(D16.Wf[1]) + (size_t)wave * 8 * 64 + lane;
#pragma unroll
    for (int ob = 0; ob < 8; ++ob) t[ob] = w1[ob * 64];
  }
  f32x4 accA[8], accB[8];
  acc_init16<8>(accB, D.bias[1], wave * 128, kq);         // bias of the NEXT layer: requested one layer ahead throughout
  X[tid] = (tid < 48) ? S.xyz[tid] : 0.f;   // rows 0..15 of the layer-0 input: xyz + zero padding to K = 16
  __syncthreads();
  {
    acc_init16<8>(accA, c0, wave * 128, kq);
    dense16<16, 8>(D16.Wf[0], X, accA, wave, lane);
    __syncthreads();
    nib[0] = writeback16<8, KEEP>(X, accA, wave * 128, lane);
    __syncthreads();
  }
  {  // lin1 (bias in accB)
    acc_init16<8>(accA, D.bias[2], wave * 128, kq);
    dense16_asm_k512_n8_o8(accB, t, xaddr, voff, rs[1], rs[2], soff8, soff8, scratch);
    __syncthreads();
    nib[1] = writeback16<8, KEEP>(X, accB, wave * 128, lane);
    __syncthreads();
  }
  f32x4 acc3[4];
  {  // lin2 (bias in accA)
    acc_init16<4>(acc3, D.bias[3], wave * 64, kq);
    dense16_asm_k512_n8_o4(accA, t, xaddr, voff, rs[2], rs[3], soff8, soff4, scratch);
    __syncthreads();
    nib[2] = writeback16<8, KEEP>(X, accA, wave * 128, lane);
    __syncthreads();
  }
  {  // lin3: 512 -> 253 (+3 rows that carry xyz into lin4)
    acc_init16<8>(accA, c4, wave * 128, kq);
    dense16_asm_k512_n4_o8(acc3, t, xaddr, voff, rs[3], rs[4], soff4, soff8, scratch);
    __syncthreads();
    nib[3] = writeback16<4, KEEP>(X, acc3, wave * 64, lane);
    __syncthreads();
    if (tid < 48) X[253 * 16 + tid] = S.xyz[tid];
    __syncthreads();
  }
  {  // lin4 (c4 in accA)
    acc_init16<8>(accB, D.bias[5], wave * 128, kq);
    dense16_asm_k256_n8_o8(accA, t, xaddr, voff, rs[4], rs[5], soff8, soff8, scratch);
    __syncthreads();
    nib[4] = writeback16<8, KEEP>(X, accA, wave * 128, lane);
    __syncthreads();
  }
  {  // lin5 (accB)
    acc_init16<8>(accA, D.bias[6], wave * 128, kq);
    dense16_asm_k512_n8_o8(accB, t, xaddr, voff, rs[5], rs[6], soff8, soff8, scratch);
    __syncthreads();
    nib[5] = writeback16<8, KEEP>(X, accB, wave * 128, lane);
    __syncthreads();
  }
  {  // lin6 (accA)
    acc_init16<8>(accB, D.bias[7], wave * 128, kq);
    dense16_asm_k512_n8_o8(accA, t, xaddr, voff, rs[6], rs[7], soff8, soff8, scratch);
    __syncthreads();
    nib[6] = writeback16<8, KEEP>(X, accA, wave * 128, lane);
    __syncthreads();
  }
  {  // lin7 (accB)
    dense16_asm_k512_n8_o0(accB, t, xaddr, voff, rs[7], rs[7], soff8, soff8, scratch);
    __syncthreads();
    nib[7] = writeback16<8, KEEP>(X, accB, wave * 128, lane);
    __syncthreads();
  }
  {
    float p = 0.f;
    const float* w8 = D.w8 + wave * 128;
    const float* xr = X + (size_t)wave * 128 * 16 + ray;
#pragma unroll 8
    for (int k = 0; k < 128; ++k) p = __builtin_fmaf(w8[k], xr[k * 16], p);
    S.part[wave * 16 + ray] = p;
  }
  __syncthreads();
  return ((S.part[ray] + S.part[16 + ray]) + (S.part[32 + ray] + S.part[48 + ray])) + D.b8;
}

// Stores the ray's mask chunks in the common per-ray format (store_mask_chunk): the 16-bit word of 32-row block ob and
// half h is  nib(kq=h, 2ob) | nib(kq=2+h, 2ob)<<4 | nib(kq=h, 2ob+1)<<8 | nib(kq=2+h, 2ob+1)<<12.
// XC (k_tail): write-through stores (distr_kernels.hpp, st_x: the block may be rewritten a few steps later from another XCD)
template <bool XC = false>
__device__ __forceinline__ void store_masks16(uint4* mstore, const long long* mb /*[16] LDS*/, const uint32_t (&nib)[8], int wave,
                                              int lane) {
  uint32_t partner[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) partner[l] = __shfl(nib[l], (lane + 32) & 63);
  const int kq = lane >> 4, j = lane & 15;
  if (kq >= 2) return;
  const long long b = mb[j];
  if (b < 0) return;
  uint32_t q[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    uint32_t w2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = 2 * i + e, l = idx >> 2, ob = idx & 3;
      const uint32_t M = nib[l], Pn = partner[l];
      w2[e] = ((M >> (8 * ob)) & 0xfu) | (((Pn >> (8 * ob)) & 0xfu) << 4) | (((M >> (8 * ob + 4)) & 0xfu) << 8) |
              (((Pn >> (8 * ob + 4)) & 0xfu) << 12);
    }
    q[i] = w2[0] | (w2[1] << 16);
  }
  uint4* dst = mstore + (size_t)b * 32 + (wave * 2 + kq) * 4;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    const uint4 val = make_uint4(q[4 * v], q[4 * v + 1], q[4 * v + 2], q[4 * v + 3]);
    if constexpr (XC) {
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t vv = {val.x, val.y, val.z, val.w};
      asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst + v), "v"(vv) : "memory");     // (s_nop: the store reads its data registers after issue)
    } else dst[v] = val;
  }
}

// =====================================================================================================================
// Cluster tile: ONE 16-ray tile split over CL workgroups on CL compute units (the deep tail of the march, where a step
// has at most a few hundred live rays: most of the chip idles and a step costs exactly one tile latency, which on one CU
// cannot drop below ~100 us of MFMA issue). Member m computes rows [m*O/CL, (m+1)*O/CL) of every layer for all 16 rays
// and the members exchange their slices after each layer. Every output row is still one k-ordered fmaf chain computed by
// one wave, so the values are bit-identical to every other tile size.
//
// Exchange (round 5; rounds 2-4 used per-layer epoch-word barriers over uncached memory followed by a bulk read: store 0.6 +
// barrier 0.9 + read 1.5 us of every 5.5 us layer): the data IS the flag. A slice travels as 8-byte GRANULES { value, tag },
// tag = (launch epoch << 3) | layer -- unique per launch, march step and layer, so a granule validates itself and no word of
// the protocol needs resetting, ordering or a barrier (MI355X_MICROARCH.md, "inter-workgroup visibility", form R2; 8-byte
// halves of a 16-byte access, each self-validating). The consumer side is PIPELINED in k order: a layer's k-loop walks the
// previous layer's rows in natural order, i.e. member 0's slice first, so a member starts on the first 128 rows (one staging
// UNIT = 8 k-groups) as soon as THEY have arrived and stages unit u+1 into LDS while the MFMAs of unit u run; only the first
// unit's hand-off latency is exposed per layer (about 1 us) instead of a barrier plus a bulk read of all slices.
// All requests of a layer's input are issued at the end of the previous layer (their addresses do not depend on data); a
// granule whose tag is not the expected one yet is simply requested again (bounded: a timeout makes the lead member evaluate
// the tile alone, see below). The members of a cluster are workgroups with equal blockIdx mod 8, which the dispatcher places
// on one XCD: their stores then stay in that XCD's L2 and the polling loads (sc1: L1-bypassing) are served from it. The
// placement is CHECKED, not assumed: every member posts its XCC id with its arrival word and the lead member switches the
// cluster to write-through (sc1) stores when the ids differ -- plain stores are never seen by another XCD's L2.
struct Xchg {
  char* buf;           // [256 clusters][2 slots][32 row blocks][2 halves][64 lanes][16 B]  granule slots (cached device memory): row block rb,
                       // half h, lane l = rows 16rb + 4(l>>4) + 2h + {0, 1} of ray l&15 as { v, tag, v', tag } -- 1 KiB per (rb, h): one
                       // coalesced wave store on the producer, one LDS-DMA request on the consumer
  uint32_t* flags;     // [256 clusters][16][8] uncached words: slot 0 = arrival words (epoch << 4 | XCC id), slot 8 = { go, abort, go-mixed-XCD }
  uint32_t epoch;      // unique per launch (per region), < 2^28
  int32_t max_cl;      // largest cluster size to use (8 or 4)
  int32_t min_cl;      // smallest cluster size to use (2 = pair tiles up to 2032 rays; DISTR_CLUSTER_MIN=4 turns them off)
  int32_t test_abort;  // tests (DISTR_CLUSTER_TEST_ABORT=1): every lead member behaves as if its cluster had not assembled
  long long* ts;       // debug (DISTR_XCHG_TS=1): wall-clock stamps of cluster 0 / member 0 at phase boundaries, else null
  int32_t par;         // parity of the exchange slots (sticky tiles alternate it per march step, see sticky_tile16)
  int32_t sticky;      // 1: a launch whose clusters (8 CUs per tile) all fit may march its tiles to the end (DISTR_STICKY=0: off)
  uint32_t epochs;     // epochs this launch may use: epoch .. epoch + epochs - 1 (one per march step of a sticky tile)
  int32_t force_sc1;   // tests (DISTR_XCHG_SC1=1): write-through stores even when all members share an XCD (the mixed-XCD path)
  int32_t t_go;        // 100 MHz ticks a member waits for the lead's verdict; 0: CL_T_GO. The persistent tail kernel (k_tail) sets a short one:
                       // its workgroups start a step together, and a tile whose lead is not resident is evaluated by another workgroup
  int32_t spread;      // tests (DISTR_CLUSTER_SPREAD=1): the members of a cluster are CONSECUTIVE workgroups (one per XCD) instead of workgroups
                       // with equal index mod 8 (one XCD): every cluster then really spans XCDs -- the assembly sees different XCC ids and
                       // switches the cluster to write-through slice stores (the mixed-XCD path, otherwise only reached by dispatcher accident)
};
constexpr int XSLOT_BYTES = 2048 * 32;            // one granule slot: 512 rows x 16 rays x 8 B
constexpr int XCLUSTER_BYTES = 2 * XSLOT_BYTES;
// Wall-clock budgets (100 MHz ticks) of the cluster protocol. Co-residency of a cluster's workgroups is NOT guaranteed by the
// hardware (other streams / ranks may hold the compute units), so a cluster first ASSEMBLES: every member posts an arrival
// word, the lead member waits at most CL_T_ARRIVE (after its own lin0) for all of them and then publishes `go` or `abort`. After `go` all members
// are resident and the per-unit waits can only be delayed by compute skew; they are still bounded (CL_T_BARRIER). Whenever
// the lead member gives up -- at assembly or at any unit -- it evaluates the tile on its own (mlp_forward16: same values,
// bit for bit), so a scheduling surprise costs time, never correctness. Members that give up just leave.
constexpr long long CL_T_ARRIVE = 30 * 100;      // 30 us (members of a cluster are dispatched within ~1 us of each other when CUs are free)
constexpr long long CL_T_GO = 2000 * 100;        // 2 ms: a member waiting for the lead's verdict
constexpr long long CL_T_BARRIER = 1000 * 100;   // 1 ms per staging unit
// (tid: the cluster tile's own copy of the thread index, cl_tid)
#define DISTR_XTS(i) do { if (xc.ts && tid == 0 && member == 0 && (xbase == xc.buf)) { xc.ts[(i)] = (long long)wall_clock64(); \
    if ((i) == 0 || (i) == 32) xc.ts[40 + ((i) >> 5)] = (long long)__builtin_readcyclecounter();   /* shader clock of the phase: stamps 40 / 41 */ \
    __builtin_amdgcn_s_waitcnt(0); } } while (0)


// The cluster tile's own copy of the thread index / the wave index (opaque to the optimiser). k_step holds every role of a march step;
// with the thread index as ONE value across all of them the register allocator spills it (scratch) and re-derives the wave index from
// the reload in front of every use inside the cluster tile -- a scratch load followed by vmcnt(0), which drains the request pipeline.
__device__ __forceinline__ int cl_tid() {
  int t;
  asm("v_mov_b32 %0, %1" : "=v"(t) : "v"(threadIdx.x));
  return t;
}
__device__ __forceinline__ int cl_wave(int tid) {
  int w;
  asm("s_lshr_b32 %0, %1, 6" : "=s"(w) : "s"(__builtin_amdgcn_readfirstlane(tid)));
  return w;
}

// Geometry of one layer of the cluster tile: RBT 16-row blocks in total, PER per member, NBL per wave (ACT active waves).
// The A-fragments a wave needs are streamed in chunks of 8 float4 (G = 8/NBL feature groups of 16) through a RING of four
// register buffers, three chunks ahead of their use: a chunk is 32 MFMAs of work (about 0.45 us at the issue rate), a request
// needs about 1 us from L2, so one chunk of look-ahead (the first version) stalled every chunk. The ring runs across layers:
// the first chunks of the next layer(s) are requested during the last chunks of this one (weights do not depend on activations).
template <int K, int O, int CL>
struct ClGeom {
  static constexpr int RBT = O / 16, PER = RBT / CL, NBL = (PER >= 4) ? PER / 4 : 1, ACT = (PER >= 4) ? 4 : PER;
  static constexpr int NG = K / 16, G = 8 / NBL, NCH = (NG + G - 1) / G;
};
constexpr int CL_AHEAD = 3;   // chunks in flight (ring of CL_AHEAD + 1 buffers)

// The weight stream AND the granule stream are issued by inline asm and waited for with explicit vmcnt counts: the compiler's own
// wait-count insertion drains the whole queue (vmcnt(0)) in front of every use here, which serialises each request with its use.
// Memory operations of one wave complete in order, so "request R has landed" = "at most (number of requests issued after R) are
// outstanding". The counts are compile-time constants of the issue pattern, which is therefore the same in EVERY wave: every wave
// issues every weight chunk (waves without rows in a layer fetch another wave's fragments and drop them) and every granule request.
// Operations only some waves issue (slice stores, bias loads, debug stamps) are NOT counted: an uncounted operation in flight only
// makes a wait stricter, never weaker. A re-request of granules (tag not there yet) is followed by vmcnt(0), after which every
// statically counted wait is satisfied trivially.
//
// NO register the compiler allocates is ever the destination of such a request. A register the compiler sees defined by an asm load
// counts as written when the statement ends: under register pressure it is copied, parked in an accumulation register or reused
// BEFORE the data lands (garbage, or a memory fault when the reused register held an address) -- round 5's first version of the
// pipelined exchange did exactly that (profiles/tools/vm_hazard_scan.py finds such accesses in the generated code). So:
//   * the weight ring (4 chunks x 8 float4), the layers' accumulators and their start values live in FIXED accumulation registers
//     (ClRegs: a[120:255] for 8 members) that only the asm statements below name: the loads target them directly, the MFMAs read their A operand from them and
//     accumulate in them (gfx950: loads may write AGPRs, an MFMA takes A / B from either file), v_accvgpr_read moves the finished
//     rows out. Every statement lists the fixed range as clobbered, so the compiler keeps nothing of its own there across them; outside
//     the cluster tile the registers are ordinary (the other roles of k_step use them freely);
//   * the granule requests land in fixed registers too (v[224:255]: two slots of 16 per thread); after the counted wait their tags
//     are compared and their values stored to LDS by asm statements that name those registers. (LDS-DMA into a landing zone in LDS is register-safe as well and was tried first: it delivers 16 KiB per
//     0.6 us and compute unit -- the 64 KiB of granules of a layer then take longer than the layer's k-loop.)
#define CL_CLOB8 "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v202", "v203", "v204", "v205", "v206", "v207"
#define CL_CLOB4 "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
#define CL_CLOB2 "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"
// one statement, clobber list by cluster size (the fixed range is smaller for larger clusters: fewer row blocks per wave)
// (every statement opens with a comment naming its cluster size: profiles/tools/check_fixed_regs.py takes the fixed range that holds around a
// compiler instruction from the nearest statement's tag)
#define CL_ASM(CL, ...) do { if constexpr ((CL) == 8) asm volatile("; distr-cl 8\n\t" __VA_ARGS__, CL_CLOB8); \
                             else if constexpr ((CL) == 4) asm volatile("; distr-cl 4\n\t" __VA_ARGS__, CL_CLOB4); \
                             else asm volatile("; distr-cl 2\n\t" __VA_ARGS__, CL_CLOB2); } while (0)
// Fixed registers of a cluster of CL members (NBLM = row blocks per wave of a 512-row layer = 8 / CL):
//   accumulator set p (= layer & 1), row block ob, register r:   a[ACC0 + 4 NBLM p + 4 ob + r]        2 x 4 NBLM registers, top of the file
//   weight ring: chunk slot r, float4 i, element s:              a[RING0 + 32 r + 4 i + s]            128 registers below them
//   granule landing zone: slot u & 1, request q, dword d:        v[LAND0 + 16 (u & 1) + 4 q + d]      the top 32 ARCHITECTURAL registers
// CL = 8: a[120:255], CL = 4: a[112:255], CL = 2: a[96:255], and v[224:255] for all. The landing zone is NOT in accumulation registers:
// a v_accvgpr_read next to running MFMAs costs 30..95 cycles (profiles/ubench/mfma_fillers.log; the first version paid 16 of them per
// unit), an LDS store straight from a VGPR costs nothing and a tag compare ~12.
template <int CL>
struct ClRegs {
  static constexpr int NBLM = 8 / CL, ACC0 = 256 - 8 * NBLM, RING0 = ACC0 - 128, LAND0 = 224;
};

template <int N, class F, int... I>
__device__ __forceinline__ void static_for_impl(F& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(f, std::make_integer_sequence<int, N>{}); }

// Addresses: every request of the cluster tile is "wave-uniform base + 16 x lane" (or + 16 x (lane >> 4)): the base travels in an
// SGPR pair, the lane part in ONE 32-bit VGPR shared by all requests -- no per-request 64-bit address VGPRs (hoisted out of the
// step loop of a sticky tile they were dozens of live values next to the fixed registers).
// (an "s" operand must be PROVABLY wave-uniform, else the compiler substitutes a VGPR and the assembler rejects the statement: bases
// go through readfirstlane -- free when the value already sits in SGPRs. The compiler pads nothing INSIDE an asm string: a vector
// memory instruction that reads an SGPR a scalar instruction wrote fewer than 5 wait states ago gets the OLD value -- an address
// with a stale half, "memory access fault at (nil)" -- so every such statement opens with s_nop 4.)
template <class T>
__device__ __forceinline__ T* cl_uni(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}
template <int CL, int LO>
__device__ __forceinline__ void cl_ld_a(uint32_t voff, const void* sbase) {      // a[LO:LO+3] <- 16 bytes at sbase + voff
  CL_ASM(CL, "s_nop 4\n\tglobal_load_dwordx4 a[%2:%2+3], %0, %1" ::"v"(voff), "s"(cl_uni(sbase)), "n"(LO) : "memory");
}
// eight of them in one statement (a weight chunk): a[LO + 4 i : LO + 4 i + 3] <- p_i + voff
template <int CL, int LO>
__device__ __forceinline__ void cl_ld_a8(uint32_t voff, const void* p0, const void* p1, const void* p2, const void* p3, const void* p4, const void* p5,
                                         const void* p6, const void* p7) {
  CL_ASM(CL, "s_nop 4\n\t"
               "global_load_dwordx4 a[%9+0:%9+3], %0, %1\n\t"
               "global_load_dwordx4 a[%9+4:%9+7], %0, %2\n\t"
               "global_load_dwordx4 a[%9+8:%9+11], %0, %3\n\t"
               "global_load_dwordx4 a[%9+12:%9+15], %0, %4\n\t"
               "global_load_dwordx4 a[%9+16:%9+19], %0, %5\n\t"
               "global_load_dwordx4 a[%9+20:%9+23], %0, %6\n\t"
               "global_load_dwordx4 a[%9+24:%9+27], %0, %7\n\t"
               "global_load_dwordx4 a[%9+28:%9+31], %0, %8"
               ::"v"(voff), "s"(cl_uni(p0)), "s"(cl_uni(p1)), "s"(cl_uni(p2)), "s"(cl_uni(p3)), "s"(cl_uni(p4)), "s"(cl_uni(p5)), "s"(cl_uni(p6)),
                 "s"(cl_uni(p7)), "n"(LO) : "memory");
}
// One k-group (16 features = 4 MFMA k-steps) of a wave's NBL row blocks: accumulator block ob = a[AB + 4 ob : AB + 4 ob + 3], its
// A-fragment for k-step s = a[RB + 4 ob + s], B-fragment of k-step s = b_s. The NBL chains are interleaved, each stays k-ordered.
template <int CL, int NBL, int AB, int RB>
__device__ __forceinline__ void cl_mfma_group(float b0, float b1, float b2, float b3) {
  if constexpr (NBL == 1) {
    CL_ASM(CL, "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+0], %0, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+1], %1, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+2], %2, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+3], %3, a[%4+0:%4+3]"
                 ::"v"(b0), "v"(b1), "v"(b2), "v"(b3), "n"(AB), "n"(RB) : "memory");
  } else if constexpr (NBL == 2) {
    CL_ASM(CL, "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+0], %0, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+4], %0, a[%4+4:%4+7]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+1], %1, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+5], %1, a[%4+4:%4+7]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+2], %2, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+6], %2, a[%4+4:%4+7]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+3], %3, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+7], %3, a[%4+4:%4+7]"
                 ::"v"(b0), "v"(b1), "v"(b2), "v"(b3), "n"(AB), "n"(RB) : "memory");
  } else {
    static_assert(NBL == 4, "1, 2 or 4 row blocks per wave");
    CL_ASM(CL, "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+0], %0, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+4], %0, a[%4+4:%4+7]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+8:%4+11], a[%5+8], %0, a[%4+8:%4+11]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+12:%4+15], a[%5+12], %0, a[%4+12:%4+15]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+1], %1, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+5], %1, a[%4+4:%4+7]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+8:%4+11], a[%5+9], %1, a[%4+8:%4+11]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+12:%4+15], a[%5+13], %1, a[%4+12:%4+15]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+2], %2, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+6], %2, a[%4+4:%4+7]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+8:%4+11], a[%5+10], %2, a[%4+8:%4+11]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+12:%4+15], a[%5+14], %2, a[%4+12:%4+15]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+0:%4+3], a[%5+3], %3, a[%4+0:%4+3]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+4:%4+7], a[%5+7], %3, a[%4+4:%4+7]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+8:%4+11], a[%5+11], %3, a[%4+8:%4+11]\n\t"
                 "v_mfma_f32_16x16x4_f32 a[%4+12:%4+15], a[%5+15], %3, a[%4+12:%4+15]"
                 ::"v"(b0), "v"(b1), "v"(b2), "v"(b3), "n"(AB), "n"(RB) : "memory");
  }
}
// the four finished rows of accumulator block a[LO:LO+3] (after the MFMA's 8 passes: s_nop)
template <int CL, int LO>
__device__ __forceinline__ f32x4 cl_acc_read4() {
  float v0, v1, v2, v3;
  CL_ASM(CL, "s_nop 15\n\ts_nop 3\n\tv_accvgpr_read_b32 %0, a[%4+0]\n\tv_accvgpr_read_b32 %1, a[%4+1]\n\tv_accvgpr_read_b32 %2, a[%4+2]\n\t"
               "v_accvgpr_read_b32 %3, a[%4+3]" : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "n"(LO) : "memory");
  f32x4 v;
  v[0] = v0; v[1] = v1; v[2] = v2; v[3] = v3;
  return v;
}
// the four requests of staging unit U (L1-bypassing: polled data) -> landing registers a[LAND:LAND+15]
template <int CL, int LAND>
__device__ __forceinline__ void cl_req4(uint32_t voff, const void* p0, const void* p1, const void* p2, const void* p3) {
  CL_ASM(CL, "s_nop 4\n\t"
               "global_load_dwordx4 v[%5+0:%5+3], %0, %1 sc1\n\t"
               "global_load_dwordx4 v[%5+4:%5+7], %0, %2 sc1\n\t"
               "global_load_dwordx4 v[%5+8:%5+11], %0, %3 sc1\n\t"
               "global_load_dwordx4 v[%5+12:%5+15], %0, %4 sc1"
               ::"v"(voff), "s"(cl_uni(p0)), "s"(cl_uni(p1)), "s"(cl_uni(p2)), "s"(cl_uni(p3)), "n"(LAND) : "memory");
}
// ... the granules of the unit after the counted wait. Tag check of entry E (requests 2E, 2E+1 = four rows of one ray): the byte-wise
// sum of absolute differences of its four tags against `tag`, added to acc -- 0 iff every tag matched. One vector instruction per
// tag and no scalar round trip (a v_cmp / s_or pair per tag measured 0.16 us per unit beside the MFMAs: ~40 cycles a pair).
template <int CL, int LAND, int E>
__device__ __forceinline__ uint32_t cl_land_diff(uint32_t tag, uint32_t acc) {
  CL_ASM(CL, "v_sad_u8 %0, v[%2+1], %1, %0\n\tv_sad_u8 %0, v[%2+3], %1, %0\n\tv_sad_u8 %0, v[%2+5], %1, %0\n\tv_sad_u8 %0, v[%2+7], %1, %0"
             : "+v"(acc) : "v"(tag), "n"(LAND + 8 * E) : "memory");
  return acc;
}
// its four values -> X[row .. row+3][ray] (lds = byte address of X[row][ray]; rows are 64 bytes apart)
template <int CL, int LAND, int E>
__device__ __forceinline__ void cl_land_store(uint32_t lds) {
  CL_ASM(CL, "ds_write2_b32 %0, v[%1+0], v[%1+2] offset1:16\n\tds_write2_b32 %0, v[%1+4], v[%1+6] offset0:32 offset1:48" ::"v"(lds), "n"(LAND + 8 * E) : "memory");
}
// ... and their ReLU bits as a nibble (lead member, KEEP): bit r = value r > 0 (on the bit pattern)
template <int CL, int LAND, int E>
__device__ __forceinline__ uint32_t cl_land_nibble() {
  uint32_t nib, t;
  CL_ASM(CL, "v_med3_i32 %0, v[%2+6], 0, 1\n\tv_med3_i32 %1, v[%2+4], 0, 1\n\tv_lshl_or_b32 %0, %0, 1, %1\n\t"
             "v_med3_i32 %1, v[%2+2], 0, 1\n\tv_lshl_or_b32 %0, %0, 1, %1\n\t"
             "v_med3_i32 %1, v[%2+0], 0, 1\n\tv_lshl_or_b32 %0, %0, 1, %1"
             : "=&v"(nib), "=&v"(t) : "n"(LAND + 8 * E) : "memory");
  return nib;
}
__device__ __forceinline__ void cl_st(uint32_t voff, void* sbase, const f32x4& v, int sc1) {     // (s_nop 1: the store reads its data registers after issue)
  sbase = cl_uni(sbase);
  if (sc1) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");   // write-through: visible to every XCD
  else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");           // stays in this XCD's L2
}
template <int N>
__device__ __forceinline__ void cl_wait_vm() {   // (vmcnt has 6 bits: a smaller count only waits for more)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N > 63 ? 63 : N) : "memory");
}

// chunk C of a K x O layer -> ring slot SLOT: this wave's 8 float4 A-fragments (G = 8 / NBL k-groups of its NBL row blocks)
template <int K, int O, int CL, int SLOT, int C>
__device__ __forceinline__ void cl_load_chunk(const float* __restrict__ Wf, int member, int wave, int lane) {
  using Ge = ClGeom<K, O, CL>;
  static_assert(Ge::NG % Ge::G == 0, "whole chunks only");
  const int rb0 = member * Ge::PER + (wave & (Ge::ACT - 1)) * Ge::NBL;
  const char* wp = reinterpret_cast<const char*>(Wf) + (size_t)rb0 * 1024;          // float4 index (g*RBT + rb)*64 + lane
  auto at = [&](int i) { return wp + ((size_t)(C * Ge::G + i / Ge::NBL) * Ge::RBT + (i % Ge::NBL)) * 1024; };
  cl_ld_a8<CL, ClRegs<CL>::RING0 + 32 * SLOT>((uint32_t)lane * 16u, at(0), at(1), at(2), at(3), at(4), at(5), at(6), at(7));
}

// ---- 8 members: the k-loop of a unit (8 k-groups = 32 dependent MFMAs of a wave's ONE 16-row block) as two hand-scheduled statements.
// Between two MFMAs on the same accumulator every instruction that makes the wave leave the MFMA stream costs matrix time: a scalar
// instruction nothing, an LDS instruction nothing, a vector instruction 10..16 cycles, a v_accvgpr_read 30..95, a taken branch or a
// dozen scalar instructions in a row more than the 32 cycles the running MFMA hides (profiles/ubench/mfma_fillers.log; with one
// statement per k-group a 512-feature layer measured 5660 cycles for 4096 of MFMAs at 2.4 GHz). So everything else travels INSIDE
// the statements, in the shadow of the dependent MFMAs: the B fragments (ds_read2st64 two k-groups ahead into four fixed 4-register
// buffers v[208:223]), their lgkmcnt waits, the wait for the weight chunk, the eight requests of the chunk three further on and --
// statement AS -- the staging of the NEXT input unit (counted wait for its granules, tag check, LDS stores). Statement A / AS =
// groups 0..5, statement B = groups 6, 7 (reads the next unit's first two groups: its rows are in LDS by then, behind a barrier).
// What the compiler generates between two statements is a handful of scalar instructions: every per-lane value the statements need
// lives in FIXED registers only they name (the compiler parks long-lived per-lane values in accumulation registers and fetches
// them back with a v_accvgpr_read in front of every statement otherwise -- 50..95 cycles each, eight times a layer):
//   v207 = LDS address of X[0][lane]      (B fragments: unit u, k-step s at offset 256 B x (32 u + s), inside the 8-bit offset field)
//   v206 = LDS address of X[4 kq][ray]    (staging stores: + 1 KiB x row block)
//   v[202:205] = 16 lane + i x 32 KiB     (weight requests: request i of a chunk = base + i x 32 KiB for a 512-row layer -- two scalar
//                                          bases 128 KiB apart --, base + i x 16 KiB for the 256-row layer: bases 16 KiB apart)
// The staging block needs no register of its own: the tag differences are summed in place (v_sad_u8 over the landed tags of an
// entry), the store address then takes the place of the entry's first tag.
__device__ __forceinline__ void cl8_set_fixed(uint32_t xb0, uint32_t xs0, uint32_t lane16) {
  CL_ASM(8, "v_mov_b32 v207, %0\n\tv_mov_b32 v206, %1\n\tv_mov_b32 v202, %2\n\tv_add_u32 v203, 0x8000, %2\n\tv_add_u32 v204, 0x10000, %2\n\t"
            "v_add_u32 v205, 0x18000, %2" ::"v"(xb0), "v"(xs0), "v"(lane16) : "memory");
}
#define CL8_LDW(i, V, B) "global_load_dwordx4 a[%[nrb]+" #i "*4:%[nrb]+" #i "*4+3], " V ", " B "\n\t"
// 512-row layer: request i at base + i x 32 KiB (b1 = b0 + 128 KiB); 256-row layer: at base + i x 16 KiB (b1 = b0 + 16 KiB)
#define CL8_LOADS32 CL8_LDW(0, "v202", "%[b0]"), CL8_LDW(1, "v203", "%[b0]"), CL8_LDW(2, "v204", "%[b0]"), CL8_LDW(3, "v205", "%[b0]"), \
                    CL8_LDW(4, "v202", "%[b1]"), CL8_LDW(5, "v203", "%[b1]"), CL8_LDW(6, "v204", "%[b1]"), CL8_LDW(7, "v205", "%[b1]")
#define CL8_LOADS16 CL8_LDW(0, "v202", "%[b0]"), CL8_LDW(1, "v202", "%[b1]"), CL8_LDW(2, "v203", "%[b0]"), CL8_LDW(3, "v203", "%[b1]"), \
                    CL8_LDW(4, "v204", "%[b0]"), CL8_LDW(5, "v204", "%[b1]"), CL8_LDW(6, "v205", "%[b0]"), CL8_LDW(7, "v205", "%[b1]")
// (generated: gen_cl8_units.py prints this block, `--check distr_mlp.hpp` compares it (tests/test_host_logic.py). B buffers rotate 208 / 212 /
// 216 / 220, requests behind the first two MFMAs of groups 0..3, the staging block behind MFMAs 17..19, its verdict behind 20..22)
#define CL8_A_TEXT(CL8_L0, CL8_L1, CL8_L2, CL8_L3, CL8_L4, CL8_L5, CL8_L6, CL8_L7) \
  "s_nop 4\n\t" \
  "s_waitcnt vmcnt(%[nw])\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+8 offset1:%[uo]+9\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+10 offset1:%[uo]+11\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+0], v208, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L0 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+1], v209, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L1 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+2], v210, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+3], v211, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+12 offset1:%[uo]+13\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+14 offset1:%[uo]+15\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+4], v212, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L2 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+5], v213, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L3 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+6], v214, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+7], v215, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[208:209], v207 offset0:%[uo]+16 offset1:%[uo]+17\n\t" \
  "ds_read2st64_b32 v[210:211], v207 offset0:%[uo]+18 offset1:%[uo]+19\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+8], v216, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L4 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+9], v217, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L5 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+10], v218, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+11], v219, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[212:213], v207 offset0:%[uo]+20 offset1:%[uo]+21\n\t" \
  "ds_read2st64_b32 v[214:215], v207 offset0:%[uo]+22 offset1:%[uo]+23\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+12], v220, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L6 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+13], v221, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L7 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+14], v222, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+15], v223, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+24 offset1:%[uo]+25\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+26 offset1:%[uo]+27\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+16], v208, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+17], v209, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+18], v210, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+19], v211, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+28 offset1:%[uo]+29\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+30 offset1:%[uo]+31\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+20], v212, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+21], v213, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+22], v214, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+23], v215, a[%[ab]:%[ab]+3]\n\t"

#define CL8_AS_TEXT(CL8_L0, CL8_L1, CL8_L2, CL8_L3, CL8_L4, CL8_L5, CL8_L6, CL8_L7) \
  "s_nop 4\n\t" \
  "s_waitcnt vmcnt(%[nw])\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+8 offset1:%[uo]+9\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+10 offset1:%[uo]+11\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+0], v208, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L0 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+1], v209, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L1 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+2], v210, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+3], v211, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+12 offset1:%[uo]+13\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+14 offset1:%[uo]+15\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+4], v212, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L2 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+5], v213, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L3 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+6], v214, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+7], v215, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[208:209], v207 offset0:%[uo]+16 offset1:%[uo]+17\n\t" \
  "ds_read2st64_b32 v[210:211], v207 offset0:%[uo]+18 offset1:%[uo]+19\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+8], v216, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L4 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+9], v217, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L5 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+10], v218, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+11], v219, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[212:213], v207 offset0:%[uo]+20 offset1:%[uo]+21\n\t" \
  "ds_read2st64_b32 v[214:215], v207 offset0:%[uo]+22 offset1:%[uo]+23\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+12], v220, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L6 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+13], v221, a[%[ab]:%[ab]+3]\n\t" \
  CL8_L7 \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+14], v222, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+15], v223, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+24 offset1:%[uo]+25\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+26 offset1:%[uo]+27\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+16], v208, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+17], v209, a[%[ab]:%[ab]+3]\n\t" \
  "s_waitcnt vmcnt(%[nws])\n\t" \
  "s_bitcmp1_b32 %[own], 0\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_sad_u8 v[%[land]+1], v[%[land]+1], %[tag], 0\n\t" \
  "v_sad_u8 v[%[land]+3], v[%[land]+3], %[tag], v[%[land]+1]\n\t" \
  "v_sad_u8 v[%[land]+5], v[%[land]+5], %[tag], v[%[land]+3]\n\t" \
  "v_sad_u8 v[%[land]+7], v[%[land]+7], %[tag], v[%[land]+5]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+18], v210, a[%[ab]:%[ab]+3]\n\t" \
  "s_bitcmp1_b32 %[own], 1\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_sad_u8 v[%[land]+9], v[%[land]+9], %[tag], 0\n\t" \
  "v_sad_u8 v[%[land]+11], v[%[land]+11], %[tag], v[%[land]+9]\n\t" \
  "v_sad_u8 v[%[land]+13], v[%[land]+13], %[tag], v[%[land]+11]\n\t" \
  "v_sad_u8 v[%[land]+15], v[%[land]+15], %[tag], v[%[land]+13]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+19], v211, a[%[ab]:%[ab]+3]\n\t" \
  "v_add_u32 v[%[land]+1], %[so0], v206\n\t" \
  "s_bitcmp1_b32 %[own], 0\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "ds_write2_b32 v[%[land]+1], v[%[land]+0], v[%[land]+2] offset1:16\n\t" \
  "ds_write2_b32 v[%[land]+1], v[%[land]+4], v[%[land]+6] offset0:32 offset1:48\n\t" \
  "s_bitcmp1_b32 %[own], 1\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+8] offset:4096\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+10] offset:4160\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+12] offset:4224\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+14] offset:4288\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+28 offset1:%[uo]+29\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+30 offset1:%[uo]+31\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+20], v212, a[%[ab]:%[ab]+3]\n\t" \
  "s_bitcmp1_b32 %[own], 0\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_cmp_ne_u32 vcc, 0, v[%[land]+7]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+21], v213, a[%[ab]:%[ab]+3]\n\t" \
  "s_mov_b64 %[flag], vcc\n\t" \
  "s_bitcmp1_b32 %[own], 1\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_cmp_ne_u32 vcc, 0, v[%[land]+15]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+22], v214, a[%[ab]:%[ab]+3]\n\t" \
  "s_or_b64 %[flag], %[flag], vcc\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+23], v215, a[%[ab]:%[ab]+3]\n\t"

#define CL8_A0_TEXT \
  "s_waitcnt vmcnt(%[nw])\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+8 offset1:%[uo]+9\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+10 offset1:%[uo]+11\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+0], v208, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+1], v209, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+2], v210, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+3], v211, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+12 offset1:%[uo]+13\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+14 offset1:%[uo]+15\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+4], v212, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+5], v213, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+6], v214, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+7], v215, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[208:209], v207 offset0:%[uo]+16 offset1:%[uo]+17\n\t" \
  "ds_read2st64_b32 v[210:211], v207 offset0:%[uo]+18 offset1:%[uo]+19\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+8], v216, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+9], v217, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+10], v218, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+11], v219, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[212:213], v207 offset0:%[uo]+20 offset1:%[uo]+21\n\t" \
  "ds_read2st64_b32 v[214:215], v207 offset0:%[uo]+22 offset1:%[uo]+23\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+12], v220, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+13], v221, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+14], v222, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+15], v223, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+24 offset1:%[uo]+25\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+26 offset1:%[uo]+27\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+16], v208, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+17], v209, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+18], v210, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+19], v211, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+28 offset1:%[uo]+29\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+30 offset1:%[uo]+31\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+20], v212, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+21], v213, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+22], v214, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+23], v215, a[%[ab]:%[ab]+3]\n\t"

#define CL8_A0S_TEXT \
  "s_waitcnt vmcnt(%[nw])\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+8 offset1:%[uo]+9\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+10 offset1:%[uo]+11\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+0], v208, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+1], v209, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+2], v210, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+3], v211, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+12 offset1:%[uo]+13\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+14 offset1:%[uo]+15\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+4], v212, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+5], v213, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+6], v214, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+7], v215, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[208:209], v207 offset0:%[uo]+16 offset1:%[uo]+17\n\t" \
  "ds_read2st64_b32 v[210:211], v207 offset0:%[uo]+18 offset1:%[uo]+19\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+8], v216, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+9], v217, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+10], v218, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+11], v219, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[212:213], v207 offset0:%[uo]+20 offset1:%[uo]+21\n\t" \
  "ds_read2st64_b32 v[214:215], v207 offset0:%[uo]+22 offset1:%[uo]+23\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+12], v220, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+13], v221, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+14], v222, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+15], v223, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[216:217], v207 offset0:%[uo]+24 offset1:%[uo]+25\n\t" \
  "ds_read2st64_b32 v[218:219], v207 offset0:%[uo]+26 offset1:%[uo]+27\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+16], v208, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+17], v209, a[%[ab]:%[ab]+3]\n\t" \
  "s_waitcnt vmcnt(%[nws])\n\t" \
  "s_bitcmp1_b32 %[own], 0\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_sad_u8 v[%[land]+1], v[%[land]+1], %[tag], 0\n\t" \
  "v_sad_u8 v[%[land]+3], v[%[land]+3], %[tag], v[%[land]+1]\n\t" \
  "v_sad_u8 v[%[land]+5], v[%[land]+5], %[tag], v[%[land]+3]\n\t" \
  "v_sad_u8 v[%[land]+7], v[%[land]+7], %[tag], v[%[land]+5]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+18], v210, a[%[ab]:%[ab]+3]\n\t" \
  "s_bitcmp1_b32 %[own], 1\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_sad_u8 v[%[land]+9], v[%[land]+9], %[tag], 0\n\t" \
  "v_sad_u8 v[%[land]+11], v[%[land]+11], %[tag], v[%[land]+9]\n\t" \
  "v_sad_u8 v[%[land]+13], v[%[land]+13], %[tag], v[%[land]+11]\n\t" \
  "v_sad_u8 v[%[land]+15], v[%[land]+15], %[tag], v[%[land]+13]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+19], v211, a[%[ab]:%[ab]+3]\n\t" \
  "v_add_u32 v[%[land]+1], %[so0], v206\n\t" \
  "s_bitcmp1_b32 %[own], 0\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "ds_write2_b32 v[%[land]+1], v[%[land]+0], v[%[land]+2] offset1:16\n\t" \
  "ds_write2_b32 v[%[land]+1], v[%[land]+4], v[%[land]+6] offset0:32 offset1:48\n\t" \
  "s_bitcmp1_b32 %[own], 1\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+8] offset:4096\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+10] offset:4160\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+12] offset:4224\n\t" \
  "ds_write_b32 v[%[land]+1], v[%[land]+14] offset:4288\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "ds_read2st64_b32 v[220:221], v207 offset0:%[uo]+28 offset1:%[uo]+29\n\t" \
  "ds_read2st64_b32 v[222:223], v207 offset0:%[uo]+30 offset1:%[uo]+31\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+20], v212, a[%[ab]:%[ab]+3]\n\t" \
  "s_bitcmp1_b32 %[own], 0\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_cmp_ne_u32 vcc, 0, v[%[land]+7]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+21], v213, a[%[ab]:%[ab]+3]\n\t" \
  "s_mov_b64 %[flag], vcc\n\t" \
  "s_bitcmp1_b32 %[own], 1\n\ts_cselect_b64 exec, 0, -1\n\t" \
  "v_cmp_ne_u32 vcc, 0, v[%[land]+15]\n\t" \
  "s_mov_b64 exec, -1\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+22], v214, a[%[ab]:%[ab]+3]\n\t" \
  "s_or_b64 %[flag], %[flag], vcc\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+23], v215, a[%[ab]:%[ab]+3]\n\t"

#define CL8_B_NEXT_TEXT \
  "ds_read2st64_b32 v[208:209], v207 offset0:%[uo]+32 offset1:%[uo]+33\n\t" \
  "ds_read2st64_b32 v[210:211], v207 offset0:%[uo]+34 offset1:%[uo]+35\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+24], v216, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+25], v217, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+26], v218, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+27], v219, a[%[ab]:%[ab]+3]\n\t" \
  "ds_read2st64_b32 v[212:213], v207 offset0:%[uo]+36 offset1:%[uo]+37\n\t" \
  "ds_read2st64_b32 v[214:215], v207 offset0:%[uo]+38 offset1:%[uo]+39\n\t" \
  "s_waitcnt lgkmcnt(4)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+28], v220, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+29], v221, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+30], v222, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+31], v223, a[%[ab]:%[ab]+3]\n\t"

#define CL8_B_LAST_TEXT \
  "s_waitcnt lgkmcnt(2)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+24], v216, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+25], v217, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+26], v218, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+27], v219, a[%[ab]:%[ab]+3]\n\t" \
  "s_waitcnt lgkmcnt(0)\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+28], v220, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+29], v221, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+30], v222, a[%[ab]:%[ab]+3]\n\t" \
  "v_mfma_f32_16x16x4_f32 a[%[ab]:%[ab]+3], a[%[rb]+31], v223, a[%[ab]:%[ab]+3]\n\t"
#define CL8_LOADS_TEXT(L0, L1, L2, L3, L4, L5, L6, L7) L0 L1 L2 L3 L4 L5 L6 L7
#define CL8_EXPAND(M, ...) M(__VA_ARGS__)
__device__ __forceinline__ void cl8_b_prologue() {      // B fragments of groups 0, 1 of a layer's first unit
  CL_ASM(8, "ds_read2st64_b32 v[208:209], v207 offset0:0 offset1:1\n\t"
               "ds_read2st64_b32 v[210:211], v207 offset0:2 offset1:3\n\t"
               "ds_read2st64_b32 v[212:213], v207 offset0:4 offset1:5\n\t"
               "ds_read2st64_b32 v[214:215], v207 offset0:6 offset1:7"
         ::: "memory");
}
// AB: accumulator block, RB: ring slot of this unit's chunk, NWAIT: requests younger than it, NRB: ring slot of the chunk requested,
// UO = 32 x unit, S16: the requested chunk belongs to the 256-row layer (16 KiB between its requests)
template <int AB, int RB, int NWAIT, int NRB, int UO, bool S16>
__device__ __forceinline__ void cl8_unit_a(const char* b0) {
  const char* b1 = b0 + (S16 ? 16384 : 131072);
  if constexpr (S16) CL_ASM(8, CL8_EXPAND(CL8_A_TEXT, CL8_LOADS16) ::[b0] "s"(cl_uni(b0)), [b1] "s"(cl_uni(b1)), [ab] "n"(AB), [rb] "n"(RB), [nw] "n"(NWAIT > 63 ? 63 : NWAIT), [nrb] "n"(NRB), [uo] "n"(UO) : "memory");
  else CL_ASM(8, CL8_EXPAND(CL8_A_TEXT, CL8_LOADS32) ::[b0] "s"(cl_uni(b0)), [b1] "s"(cl_uni(b1)), [ab] "n"(AB), [rb] "n"(RB), [nw] "n"(NWAIT > 63 ? 63 : NWAIT), [nrb] "n"(NRB), [uo] "n"(UO) : "memory");
}
template <int AB, int RB, int NWAIT, int UO>
__device__ __forceinline__ void cl8_unit_a0() {          // ... when no further chunk exists
  CL_ASM(8, CL8_A0_TEXT ::[ab] "n"(AB), [rb] "n"(RB), [nw] "n"(NWAIT > 63 ? 63 : NWAIT), [uo] "n"(UO) : "memory");
}
// ... with the staging of the next input unit inside (landing slot LAND, NWS requests younger than its granules; own: bit e set = the
// wave's entry e holds the member's own rows -- neither checked nor stored: exec = 0 around its instructions; so0 = LDS offset of the
// wave's first row block of the unit, the second one is 4 KiB further). Returns nonzero when a tag did not match (the caller then
// re-requests the unit until it is there: cl_stage_unit). exec is all ones here (every branch around the tile is wave-uniform).
struct Cl8Stage { uint32_t tag, so0, own; };
template <int AB, int RB, int NWAIT, int NRB, int UO, bool S16, int NWS, int LAND>
__device__ __forceinline__ uint64_t cl8_unit_as(const char* b0, const Cl8Stage& st) {
  const char* b1 = b0 + (S16 ? 16384 : 131072);
  uint64_t flag;
  if constexpr (S16) CL_ASM(8, CL8_EXPAND(CL8_AS_TEXT, CL8_LOADS16) : [flag] "=&s"(flag) : [b0] "s"(cl_uni(b0)), [b1] "s"(cl_uni(b1)), [ab] "n"(AB), [rb] "n"(RB),
                               [nw] "n"(NWAIT > 63 ? 63 : NWAIT), [nrb] "n"(NRB), [uo] "n"(UO), [nws] "n"(NWS > 63 ? 63 : NWS), [land] "n"(LAND), [tag] "s"(st.tag),
                               [own] "s"(st.own), [so0] "s"(st.so0) : "memory", "vcc", "scc");
  else CL_ASM(8, CL8_EXPAND(CL8_AS_TEXT, CL8_LOADS32) : [flag] "=&s"(flag) : [b0] "s"(cl_uni(b0)), [b1] "s"(cl_uni(b1)), [ab] "n"(AB), [rb] "n"(RB),
                 [nw] "n"(NWAIT > 63 ? 63 : NWAIT), [nrb] "n"(NRB), [uo] "n"(UO), [nws] "n"(NWS > 63 ? 63 : NWS), [land] "n"(LAND), [tag] "s"(st.tag),
                 [own] "s"(st.own), [so0] "s"(st.so0) : "memory", "vcc", "scc");
  return flag;
}
template <int AB, int RB, int NWAIT, int UO, int NWS, int LAND>
__device__ __forceinline__ uint64_t cl8_unit_a0s(const Cl8Stage& st) {
  uint64_t flag;
  CL_ASM(8, CL8_A0S_TEXT : [flag] "=&s"(flag) : [ab] "n"(AB), [rb] "n"(RB), [nw] "n"(NWAIT > 63 ? 63 : NWAIT), [uo] "n"(UO), [nws] "n"(NWS > 63 ? 63 : NWS),
            [land] "n"(LAND), [tag] "s"(st.tag), [own] "s"(st.own), [so0] "s"(st.so0) : "memory", "vcc", "scc");
  return flag;
}
template <int NRB, bool S16>
__device__ __forceinline__ void cl8_unit_loads(const char* b0) {     // a wave without rows in this layer: the requests only
  const char* b1 = b0 + (S16 ? 16384 : 131072);
  if constexpr (S16) CL_ASM(8, "s_nop 4\n\t" CL8_EXPAND(CL8_LOADS_TEXT, CL8_LOADS16) ::[b0] "s"(cl_uni(b0)), [b1] "s"(cl_uni(b1)), [nrb] "n"(NRB) : "memory");
  else CL_ASM(8, "s_nop 4\n\t" CL8_EXPAND(CL8_LOADS_TEXT, CL8_LOADS32) ::[b0] "s"(cl_uni(b0)), [b1] "s"(cl_uni(b1)), [nrb] "n"(NRB) : "memory");
}
template <int AB, int RB, bool NEXT, int UO>
__device__ __forceinline__ void cl8_unit_b() {
  if constexpr (NEXT) CL_ASM(8, CL8_B_NEXT_TEXT ::[ab] "n"(AB), [rb] "n"(RB), [uo] "n"(UO) : "memory");
  else CL_ASM(8, CL8_B_LAST_TEXT ::[ab] "n"(AB), [rb] "n"(RB) : "memory");
}


// request base of chunk C of a K x O layer for this wave, 8 members (one row block per wave; see cl_load_chunk): request i of the chunk is
// base + i x RBT KiB
template <int K, int O, int C>
__device__ __forceinline__ const char* cl8_chunk_base(const float* __restrict__ Wf, int member, int wave) {
  using Ge = ClGeom<K, O, 8>;
  static_assert(Ge::NBL == 1 && (Ge::RBT == 32 || Ge::RBT == 16), "512- or 256-row layers");
  const int rb0 = member * Ge::PER + (wave & (Ge::ACT - 1));
  return reinterpret_cast<const char*>(Wf) + ((size_t)rb0 + (size_t)(C * 8) * Ge::RBT) * 1024;
}

// start values of a layer's accumulators (this wave's rows of the bias / latent-constant vector) straight into accumulator set P
// (uncounted requests: only the waves with rows issue them)
template <int K, int O, int CL, int P>
__device__ __forceinline__ void cl_load_start(const float* __restrict__ init, int member, int wave, int kq) {
  using Ge = ClGeom<K, O, CL>;
  static_assert(Ge::NBL <= ClRegs<CL>::NBLM, "an accumulator set holds NBLM row blocks");
  if (wave >= Ge::ACT) return;
  const int rb0 = member * Ge::PER + wave * Ge::NBL;
  static_for<Ge::NBL>([&](auto ob_) {
    constexpr int ob = decltype(ob_)::value;
    cl_ld_a<CL, ClRegs<CL>::ACC0 + 4 * ClRegs<CL>::NBLM * P + 4 * ob>((uint32_t)kq * 16u, init + 16 * (rb0 + ob));
  });
}

// Assembly of a cluster (see CL_T_*). Returns with *fail set (all threads see it after the barrier) when this member must not
// take part: the lead then evaluates the tile alone, the others leave. *sc1 (LDS, valid when the cluster assembled) = 1: the
// members sit on more than one XCD, slices must be stored write-through.
// The LEAD member of a cluster (it coordinates the assembly, runs the tile's epilogue -- march update, selected rows, mask blocks --
// and evaluates the tile alone if the cluster breaks up) is its LAST member, not its first: everything only the lead does (the
// epilogue of a step; before the own-row masks also the ReLU bits of every staged unit) delays ITS slice, and a layer's k-loop consumes the slices in member
// order -- the last member's rows are needed ~1.5 us after the first member's, so up to that much lead-only work is hidden, while
// the same work on member 0 stalls every member at the start of every layer.
__device__ __forceinline__ constexpr int cl_lead(int cl) { return cl - 1; }

template <int CL>
__device__ __forceinline__ void cl_assemble(uint32_t* flags, int member, uint32_t epoch, int tid, int32_t* fail, int32_t* sc1, int test_abort, int force_sc1, long long t_go) {
  if (tid < 64) {
    const long long t0 = (long long)wall_clock64();
    if (member == cl_lead(CL)) {
      bool ok = false, mixed = false;
      for (;;) {
        const uint32_t v = (tid < CL) ? __hip_atomic_load(flags + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (epoch << 4);
        if (__ballot((v >> 4) != epoch) == 0ull) {
          ok = __ballot(test_abort != 0) == 0ull;     // (any lane: k_tail hands the lead's lost-claim flag in on lane 0)
          const uint32_t x0 = __shfl(v & 15u, 0);
          mixed = __ballot(tid < CL && (v & 15u) != x0) != 0ull;
          break;
        }
        if ((long long)wall_clock64() - t0 > CL_T_ARRIVE) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (tid == 0) {
        const bool wt = mixed || force_sc1;
        __hip_atomic_store(flags + 8 * 8 + (ok ? (wt ? 2 : 0) : 1), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sc1 = wt ? 1 : 0;
        if (!ok) *fail = 1;
      }
    } else {
      for (;;) {
        const uint32_t v = (tid < 3) ? __hip_atomic_load(flags + 8 * 8 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const unsigned long long hit = __ballot(v == epoch);
        if (hit & 5ull) { if (tid == 0) *sc1 = (hit & 4ull) ? 1 : 0; break; }     // go (same XCD / mixed)
        if ((hit & 2ull) || (long long)wall_clock64() - t0 > t_go) {   // abort, or no verdict: withdraw the arrival word and leave
          if (tid == 0) { *fail = 1; __hip_atomic_store(flags + member, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
}

// KEEP: the ReLU bits of 4 consecutive rows (row0 = multiple of 4) of ray jj, taken from the REGISTERS that hold the values anyway
// (own slice at write-back; in the lead-gathers form, keep bit 0, also the other members' slices on their way into LDS), go into the ray's mask block as one
// nibble with an LDS atomic-or (S.mk is zeroed at the start of the tile). Format of store_mask_chunk: chunk (w, h), word
// layer*4 + ob, bit r <-> row w*WR + 32*ob + (r&3) + 8*(r>>2) + 4*h, WR = 1 << wr_log = rows per wave of the 32x32 tiles (128; 64
// for lin3, whose words ob = 2, 3 stay zero). (Re-reading the finished layer from LDS cost 2 us per layer on the lead member.)
// rb = row0 / 16 is wave-uniform (a wave's entry is a whole 16-row block, lane (kq, jj) holds its rows 4kq .. 4kq+3): with R = 16 rb,
//   w = R >> wr_log, ob = (R & (WR - 1)) >> 5, h = kq & 1, n = 2 (rb & 1) + (kq >> 1),  idx = [64 w + 4 layer + ob] + 32 h
// so everything but 64 bytes x (kq & 1) of the address and 4 x (kq >> 1) of the shift is scalar arithmetic; the nibble itself is four
// clamps to [0, 1] (v_med3_i32 on the bit pattern) and three shift-ors -- about 10 vector instructions per entry instead of 35 (they
// run beside the MFMAs of the lead member's k-loop, where every vector instruction costs ~13 cycles of matrix time).
__device__ __forceinline__ void mask_nibble_put(Smem16CL& S, int layer, int rb, int wr_log, int kq, int jj, uint32_t nib) {
  const int R = 16 * rb, w = R >> wr_log, ob = (R & ((1 << wr_log) - 1)) >> 5;
  const int idxu = 64 * w + 4 * layer + ob;
  uint32_t* p = reinterpret_cast<uint32_t*>(&S.mk[jj][0]) + (idxu >> 1) + 16 * (kq & 1);
  atomicOr(p, nib << (8 * (rb & 1) + 16 * (idxu & 1) + 4 * (kq >> 1)));
}
__device__ __forceinline__ uint32_t mask_nibble_of(const f32x4& v) {     // bit r = "the float's integer pattern is positive"
  const int b0 = min(max(__float_as_int(v[0]), 0), 1), b1 = min(max(__float_as_int(v[1]), 0), 1), b2 = min(max(__float_as_int(v[2]), 0), 1),
            b3 = min(max(__float_as_int(v[3]), 0), 1);
  return (uint32_t)(((((b3 << 1) | b2) << 1 | b1) << 1) | b0);
}
__device__ __forceinline__ void mask_nibble_or(Smem16CL& S, int layer, int row0, int wr_log, int jj, const f32x4& v) {
  mask_nibble_put(S, layer, row0 >> 4, wr_log, (row0 >> 2) & 3, jj, mask_nibble_of(v));
}

// Requests the NU staging units of a slot (a layer's whole output, all members' slices incl. the own one: the request pattern must
// not depend on the member). Unit u = row blocks 8u .. 8u+7; wave w takes row blocks 8u + w and 8u + w + 4 of it, both halves:
// request q = 2e + h -> row block 8u + w + 4e, half h, landing at stage[u][q][w] (1 KiB each).
// [U0, U1): unit 0 is requested right behind the own slice store (and re-requested until its granules are there: the exposed
// hand-off); the others only once unit 0 has been seen -- requested together with it they would all come back stale (every member
// publishes at about the same time) and each would cost its own re-request round trip in the middle of the k-loop.
// Unit u = row blocks 8u .. 8u+7; wave w takes row blocks 8u + w and 8u + w + 4 of it, both halves: request q = 2e + h -> row block
// 8u + w + 4e, half h (the whole layer output incl. the own slice: the request pattern must not depend on the member).
template <int CL, int U>
__device__ __forceinline__ void cl_request_unit(const char* slot, int wave, int lane) {      // -> landing slot U & 1
  const char* b = slot + (size_t)(8 * U + wave) * 2048;
  cl_req4<CL, ClRegs<CL>::LAND0 + 16 * (U & 1)>((uint32_t)lane * 16u, b, b + 1024, b + 4 * 2048, b + 4 * 2048 + 1024);
}

// Request schedule of a layer input of NU units over the two landing slots: unit 0 behind the producer's own slice store; units 1 and
// 2 when unit 0 has been staged (slot 0 is free again); unit 3 when unit 1 has been staged.
// Stages unit U of the slot into X: waits (statically counted, NWAIT younger requests) for its four requests, moves them out of
// the landing registers, validates the granule tags, re-requests until they are all there (bounded), writes the values of the OTHER
// members' rows to X (the own rows were written at write-back) and ORs their ReLU bits into the mask blocks (lead member, KEEP).
// The caller adds the barrier. own rows of the producing layer: row blocks [own_lo, own_lo + own_n).
template <int CL, int U, int NWAIT>
__device__ __forceinline__ void cl_stage_unit(const char* slot, uint32_t tag, int own_lo, int own_n, Smem16CL& S, int tid, int wave, bool keep,
                                              int layer_of_data, int wr_log) {
  constexpr int LAND = ClRegs<CL>::LAND0 + 16 * (U & 1);
  const int lw = tid & 63, kq = lw >> 4, j = lw & 15;
  cl_wait_vm<NWAIT>();
  // own rows of the producing layer (wave-uniform: a wave's two entries are whole row blocks): neither checked nor copied
  const bool own0 = (unsigned)(U * 8 + wave - own_lo) < (unsigned)own_n, own1 = (unsigned)(U * 8 + wave + 4 - own_lo) < (unsigned)own_n;
  auto bad = [&]() {
    uint32_t d = 0u;
    if (!own0) d = cl_land_diff<CL, LAND, 0>(tag, d);
    if (!own1) d = cl_land_diff<CL, LAND, 1>(tag, d);
    return __ballot(d != 0u) != 0ull;
  };
  if (bad()) {
    const long long t0 = (long long)wall_clock64();
    for (;;) {
      cl_request_unit<CL, U>(slot, wave, lw);
      cl_wait_vm<0>();
      if (!bad()) break;
      if (*reinterpret_cast<volatile int32_t*>(&S.fail) != 0) break;
      if ((long long)wall_clock64() - t0 > CL_T_BARRIER) { S.fail = 1; break; }
    }
  }
  const uint32_t x0 = lds_off(S.X) + (uint32_t)(((16 * (U * 8 + wave) + 4 * kq) * 16 + j) * 4);
  if (!own0) {
    cl_land_store<CL, LAND, 0>(x0);
    if (keep) mask_nibble_put(S, layer_of_data, U * 8 + wave, wr_log, kq, j, cl_land_nibble<CL, LAND, 0>());
  }
  if (!own1) {
    cl_land_store<CL, LAND, 1>(x0 + 4 * 16 * 16 * 4);
    if (keep) mask_nibble_put(S, layer_of_data, U * 8 + wave + 4, wr_log, kq, j, cl_land_nibble<CL, LAND, 1>());
  }
}

// number of weight-chunk requests issued in iterations 0..upto of a layer whose chunk 0 sits at position GB of the TOT-chunk sequence
constexpr int cl_issued(int GB, int TOT, int upto) {
  int n = 0;
  for (int i = 0; i <= upto; ++i) n += (GB + i + CL_AHEAD < TOT) ? 1 : 0;
  return n;
}

// One dense layer of the cluster tile: this member's row slice over the (staged) input, ReLU, slice -> own LDS + granule slot,
// requests for the whole layer output (the next layer's input). FIRST: the input (lin0's output) was computed whole by every member
// (nothing staged, the start values and first weight chunks were requested before lin0). Otherwise the input is the previous layer's
// slot: unit 0 is staged before the k-loop, unit u+1 two k-groups before the end of unit u (the B fragments run two groups ahead).
// The accumulators are set LAYER & 1 of the fixed registers; their start values were requested into them by the previous layer
// before its slice stores, the next layer's (initNext) are requested into the other set after the chunk loop.
// GB = index of this layer's chunk 0 in the network-wide chunk sequence (ring slot = index & 3). On entry the first
// min(CL_AHEAD, NCH) chunks of this layer are in flight / in their slots; while chunk c is used, chunk c + CL_AHEAD of the
// sequence is requested: of this layer, of the next one (KN x ON, WfNext) or of the one after it (KN2 x ON2, WfNext2).
// REQ_OUT: request the output slot afterwards (false for a member that leaves after its last slice store).
// Returns false when this member gave up (S.fail set; nothing of its requests is in flight any more).
// keep (ReLU bits into S.mk): bit 0 = of the rows staged from the other members, bit 1 = of the own rows at write-back.
template <int LAYER, int K, int O, int CL, int GB, int TOT, int KN, int ON, int KN2, int ON2, bool FIRST, bool REQ_OUT>
__device__ __forceinline__ bool layer_cl(const float* __restrict__ Wf, const float* __restrict__ WfNext, const float* __restrict__ WfNext2,
                                         const float* __restrict__ initNext, Smem16CLX& S, const Xchg& xc, char* xbase, int member, int keep) {
  using Ge = ClGeom<K, O, CL>;
  using GeN = ClGeom<(KN > 0 ? KN : 128), (KN > 0 ? ON : 64 * CL), CL>;
  using GeN2 = ClGeom<(KN2 > 0 ? KN2 : 128), (KN2 > 0 ? ON2 : 64 * CL), CL>;
  constexpr int RBT = Ge::RBT, PER = Ge::PER, NBL = Ge::NBL, ACT = Ge::ACT, G = Ge::G, NCH = Ge::NCH, NG = Ge::NG;
  constexpr int NCHN = (KN > 0) ? GeN::NCH : 0, NCHN2 = (KN2 > 0) ? GeN2::NCH : 0;
  constexpr int NUIN = FIRST ? 0 : K / 128, PERIN = (K / 16) / CL, NUOUT = O / 128;
  constexpr int ACC = ClRegs<CL>::ACC0 + 4 * ClRegs<CL>::NBLM * (LAYER & 1);
  constexpr int layer = LAYER;
  static_assert(FIRST || (NG % 8 == 0 && PERIN >= 1), "staging units of 8 k-groups");
  const int tid = cl_tid();
  const int wave = cl_wave(tid);
  const int lane = tid & 63, kq = lane >> 4, j = lane & 15;
  float* X = S.X;
  const char* slot_in = xbase + ((layer - 1 + xc.par) & 1) * XSLOT_BYTES;
  char* slot_out = xbase + ((layer + xc.par) & 1) * XSLOT_BYTES;
  const uint32_t tag_in = (xc.epoch << 3) | (uint32_t)(layer - 1), tag_out = (xc.epoch << 3) | (uint32_t)layer;
  constexpr int wr_in = (LAYER - 1 == 3) ? 6 : 7;
  const int rb0 = member * PER + wave * NBL;
  if constexpr (NUIN > 0) {   // the first 128 input rows (the only exposed hand-off of the layer); everything requested earlier has landed with them
    cl_stage_unit<CL, 0, 0>(slot_in, tag_in, member * PERIN, PERIN, S, tid, wave, (keep & 1) != 0, layer - 1, wr_in);
    // (units 1 and 2 before this layer's first weight request: the counts below rely on it)
    if constexpr (NUIN > 1) cl_request_unit<CL, 1>(slot_in, wave, lane);
    if constexpr (NUIN > 2) cl_request_unit<CL, 2>(slot_in, wave, lane);
    __syncthreads();
    if (S.fail) { cl_wait_vm<0>(); return false; }
    DISTR_XTS(4 * (layer - 1) + 3);
  }
  if constexpr (CL == 8) {
    // hand-scheduled units (cl8_unit_a / _as / _b): a weight chunk = a unit (G = 8), one row block per wave
    static_assert(G == 8 && NBL == 1 && NG % 8 == 0, "8 members: one 16-row block per wave, a chunk per unit");
    constexpr int NU = NG / 8;
    const bool act = (ACT == 4) || wave < ACT;       // (a scalar branch costs what the MFMA before it hides: none where every wave has rows)
    if (act) cl8_b_prologue();
    static_for<NU>([&](auto u_) {
      constexpr int u = decltype(u_)::value;
      constexpr int t = u + CL_AHEAD;
      constexpr int RB = ClRegs<8>::RING0 + 32 * ((GB + u) & 3), NRB = ClRegs<8>::RING0 + 32 * ((GB + t) & 3);
      // this unit's chunk: requested three units ago -- younger = the chunks of the two units in between, and unit 3 of the input if it
      // was requested (behind the staging of unit 1, inside unit 0) since; chunks 0..2 of a layer landed with its first input unit
      // (first layer: requested before lin0, like the start values, which are the youngest request there)
      constexpr int NWAIT = (FIRST && u == 0) ? 0 : (u >= CL_AHEAD) ? 8 * (((GB + u - 1 + CL_AHEAD < TOT) ? 1 : 0) + ((GB + u - 2 + CL_AHEAD < TOT) ? 1 : 0)) +
                                                                        ((NUIN == 4 && u - 3 <= 0 && 0 <= u - 1) ? 4 : 0) : 63;
      constexpr bool HAS = GB + t < TOT;                      // a chunk to request: of this layer, the next one or the one after it
      constexpr int KT = (t < NCH) ? K : (t - NCH < NCHN) ? (KN > 0 ? KN : 128) : (KN2 > 0 ? KN2 : 128);
      constexpr int OT = (t < NCH) ? O : (t - NCH < NCHN) ? (KN > 0 ? ON : 512) : (KN2 > 0 ? ON2 : 512);
      constexpr int CT = (t < NCH) ? t : (t - NCH < NCHN) ? t - NCH : t - NCH - NCHN;
      constexpr bool S16 = OT == 256;
      const char* b0 = nullptr;
      if constexpr (HAS) b0 = cl8_chunk_base<KT, OT, CT>((t < NCH) ? Wf : (t - NCH < NCHN) ? WfNext : WfNext2, member, wave);
      if constexpr (NUIN > 1 && u + 1 < NU) {     // the next unit: in LDS before statement B reads the B fragments of its first groups
        constexpr int un = u + 1;
        // younger than unit un's requests: unit 1: unit 2's + the weight chunks requested in units 0..u; unit 2: unit 3's (requested
        // behind the staging of unit 1) + the chunks of units 0..u; unit 3: the chunks requested in units 1..u
        constexpr int NW = (un == 1) ? ((NUIN > 2 ? 4 : 0) + 8 * cl_issued(GB, TOT, u)) : (un == 2) ? ((NUIN > 3 ? 4 : 0) + 8 * cl_issued(GB, TOT, u))
                                     : 8 * (cl_issued(GB, TOT, u) - cl_issued(GB, TOT, 0));
        constexpr int LAND = ClRegs<8>::LAND0 + 16 * (un & 1);
        if (act) {
          // staged inside the statement (this unit's chunk requests are older than the wait in there, like in the separate form)
          const int blk = un * 8 + wave;
          const bool own0 = (unsigned)(blk - member * PERIN) < (unsigned)PERIN, own1 = (unsigned)(blk + 4 - member * PERIN) < (unsigned)PERIN;
          Cl8Stage st;
          // ("s" operands must be provably wave-uniform: see cl_uni)
          st.tag = __builtin_amdgcn_readfirstlane(tag_in); st.so0 = __builtin_amdgcn_readfirstlane((uint32_t)blk * 1024u);
          st.own = __builtin_amdgcn_readfirstlane((own0 ? 1u : 0u) | (own1 ? 2u : 0u));
          uint64_t bad;
          if constexpr (HAS) bad = cl8_unit_as<ACC, RB, NWAIT, NRB, 32 * u, S16, NW, LAND>(b0, st);
          else bad = cl8_unit_a0s<ACC, RB, NWAIT, 32 * u, NW, LAND>(st);
          if (bad) {          // a granule was not there yet: re-request until it is (bounded), store again
            cl_stage_unit<CL, un, 0>(slot_in, tag_in, member * PERIN, PERIN, S, tid, wave, (keep & 1) != 0, layer - 1, wr_in);
          } else if (keep & 1) {  // lead member: the ReLU bits of the staged rows
            if (!own0) mask_nibble_put(S, layer - 1, blk, wr_in, kq, j, cl_land_nibble<CL, LAND, 0>());
            if (!own1) mask_nibble_put(S, layer - 1, blk + 4, wr_in, kq, j, cl_land_nibble<CL, LAND, 1>());
          }
        } else {
          if constexpr (HAS) cl8_unit_loads<NRB, S16>(b0);
          cl_stage_unit<CL, un, NW>(slot_in, tag_in, member * PERIN, PERIN, S, tid, wave, (keep & 1) != 0, layer - 1, wr_in);
        }
        if constexpr (un == 1 && NUIN > 3) cl_request_unit<CL, 3>(slot_in, wave, lane);
        // lin4's input: rows 253..255 carry xyz, over the (staged or own) zeros of lin3's padded rows -- written by the wave that staged
        // that row block (wave 3: row block 15), behind its own copy in program order (another wave would race with it)
        if (K == 256 && un == NUIN - 1 && wave == 3 && lane < 48) X[253 * 16 + lane] = S.xyz[lane];
        __syncthreads();
      } else {
        if (act) {
          if constexpr (HAS) cl8_unit_a<ACC, RB, NWAIT, NRB, 32 * u, S16>(b0);
          else cl8_unit_a0<ACC, RB, NWAIT, 32 * u>();
        } else {
          if constexpr (HAS) cl8_unit_loads<NRB, S16>(b0);
        }
      }
#ifdef DISTR_XTS_UNITS      // diagnostics build: stamps around the statements of layer 2's units (profiles/tools: gpu_diag_cluster.py prints them)
      if (LAYER == 2) DISTR_XTS(42 + 2 * u);
#endif
      if (act) cl8_unit_b<ACC, RB, (u + 1 < NU), 32 * u>();
#ifdef DISTR_XTS_UNITS
      if (LAYER == 2) DISTR_XTS(43 + 2 * u);
#endif
    });
  } else {
  const float* xb = X + lane;
  float b[3][4];                      // B fragments (LDS) run two feature groups ahead
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) { b[0][s4] = xb[(4 * s4) * 16]; b[1][s4] = xb[(16 * (NG > 1 ? 1 : 0) + 4 * s4) * 16]; }
  static_for<NCH>([&](auto c_) {
    constexpr int c = decltype(c_)::value;
    constexpr int t = c + CL_AHEAD;
    if constexpr (t < NCH) cl_load_chunk<K, O, CL, (GB + t) & 3, t>(Wf, member, wave, lane);
    else if constexpr (t - NCH < NCHN) cl_load_chunk<(KN > 0 ? KN : 128), (KN > 0 ? ON : 64 * CL), CL, (GB + t) & 3, t - NCH>(WfNext, member, wave, lane);
    else if constexpr (t - NCH - NCHN < NCHN2) cl_load_chunk<(KN2 > 0 ? KN2 : 128), (KN2 > 0 ? ON2 : 64 * CL), CL, (GB + t) & 3, t - NCH - NCHN>(WfNext2, member, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    if (wave < ACT) {
      if constexpr (FIRST && c == 0) {
        // first layer: chunks 0..2 and the start values were requested before lin0 (in that order); the only younger request is this
        // iteration's chunk
        if constexpr (GB + CL_AHEAD < TOT) cl_wait_vm<8>(); else cl_wait_vm<0>();
      } else if constexpr (c >= CL_AHEAD) {
        // chunk c was requested in iteration c - CL_AHEAD of this layer: younger = the weight requests of iterations c-2 .. c, and unit 3
        // of the input if it was requested (behind the staging of unit 1, iteration CS1) in iterations c-3 .. c-1. Chunks c < CL_AHEAD were
        // requested during the previous layer(s), before this layer's input units: they (and the start values) landed with unit 0.
        constexpr int younger = ((GB + c + CL_AHEAD < TOT) ? 1 : 0) + ((GB + c - 1 + CL_AHEAD < TOT) ? 1 : 0) + ((GB + c - 2 + CL_AHEAD < TOT) ? 1 : 0);
        constexpr int CS1 = 6 / G;
        cl_wait_vm<8 * younger + ((NUIN == 4 && c - 3 <= CS1 && CS1 <= c - 1) ? 4 : 0)>();
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<G>([&](auto gi_) {
      constexpr int gi = decltype(gi_)::value;
      constexpr int g = c * G + gi;
        if constexpr (g < NG) {
        if constexpr (NUIN > 1 && (g % 8) == 6 && g + 2 < NG) {   // next unit: in LDS before the B fragments of its first group are read (two groups ahead)
          constexpr int u = (g + 2) / 8;
          // younger than unit u's requests: unit 1: unit 2's + the weight chunks of iterations 0..c; unit 2: unit 3's (requested behind the
          // staging of unit 1) + the chunks of iterations 0..c; unit 3: the chunks of the iterations after unit 1's staging .. c
          constexpr int CS1 = 6 / G;
          constexpr int NW = (u == 1) ? ((NUIN > 2 ? 4 : 0) + 8 * cl_issued(GB, TOT, c)) : (u == 2) ? ((NUIN > 3 ? 4 : 0) + 8 * cl_issued(GB, TOT, c))
                                      : 8 * (cl_issued(GB, TOT, c) - cl_issued(GB, TOT, CS1));
          cl_stage_unit<CL, u, NW>(slot_in, tag_in, member * PERIN, PERIN, S, tid, wave, (keep & 1) != 0, layer - 1, wr_in);
          if constexpr (u == 1 && NUIN > 3) cl_request_unit<CL, 3>(slot_in, wave, lane);
          // lin4's input: rows 253..255 carry xyz, over the (staged or own) zeros of lin3's padded rows -- written by the wave that staged
          // that row block (wave 3: row block 15), behind its own copy in program order (another wave would race with it)
          if (K == 256 && u == NUIN - 1 && wave == 3 && lane < 48) X[253 * 16 + lane] = S.xyz[lane];
          __syncthreads();
        }
        if (wave < ACT) {
          constexpr int gn = (g + 2 < NG) ? g + 2 : NG - 1;
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) b[(g + 2) % 3][s4] = xb[(16 * gn + 4 * s4) * 16];
          __builtin_amdgcn_sched_barrier(0);
          cl_mfma_group<CL, NBL, ACC, ClRegs<CL>::RING0 + 32 * ((GB + c) & 3) + 4 * gi * NBL>(b[g % 3][0], b[g % 3][1], b[g % 3][2], b[g % 3][3]);
        }
      }
    });
    __builtin_amdgcn_sched_barrier(0);
  });
  }
  if constexpr (KN > 0) cl_load_start<KN, ON, CL, (LAYER + 1) & 1>(initNext, member, wave, kq);   // lands with the next layer's first unit
  DISTR_XTS(4 * layer);
  __syncthreads();                         // everybody is done reading the layer input
  // A member that gave up inside this layer (a unit never arrived) must not publish: its rows are garbage. (S.fail is looked at here
  // and after the first unit only: a later unit that times out ends its wait and the layer finishes on whatever was there.)
  if (NUIN > 1 && S.fail) { cl_wait_vm<0>(); return false; }
  if (wave < ACT) {
    f32x4 tg;
    tg[1] = __uint_as_float(tag_out); tg[3] = tg[1];
    const int sc1 = S.sc1;
    static_for<NBL>([&](auto ob_) {
      constexpr int ob = decltype(ob_)::value;
      f32x4 v = cl_acc_read4<CL, ACC + 4 * ob>();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = __int_as_float(max(__float_as_int(v[r]), 0));
        X[(16 * (rb0 + ob) + 4 * kq + r) * 16 + j] = v[r];
      }
      char* p = slot_out + (size_t)(rb0 + ob) * 2048;
      f32x4 g0 = tg, g1 = tg;
      g0[0] = v[0]; g0[2] = v[1]; g1[0] = v[2]; g1[2] = v[3];
      cl_st((uint32_t)lane * 16u, p, g0, sc1);
      cl_st((uint32_t)lane * 16u, p + 1024, g1, sc1);
      if (keep & 2) mask_nibble_put(S, layer, rb0 + ob, (LAYER == 3) ? 6 : 7, kq, j, mask_nibble_of(v));
    });
  }
  if (REQ_OUT) cl_request_unit<CL, 0>(slot_out, wave, lane);
  DISTR_XTS(4 * layer + 1);
  return true;
}

// Cluster forward. Every member returns after its last contribution; the lead member returns the pre-tanh value (ray = tid & 15)
// and, with KEEP and without MASK_OWN, has the rays' mask blocks in S.mk. The other members return 0 (unless ALL_LIN8). On return S.fail != 0 (uniform over the
// workgroup) means this member gave up (cluster not assembled in time / a unit timed out): the lead member's caller then
// evaluates the tile with mlp_forward16 (S.xyz is untouched), the other members simply leave.
// ALL_LIN8 (sticky tiles; every cluster tile with saved masks): every member computes lin8 from its own copy of h7 and returns the pre-tanh value (all members then
// mirror the march update in registers, no broadcast needed). MASK_OWN (with KEEP): EVERY member records the ReLU bits of the rows it
// computes itself (its accumulators at write-back, its quarter-of-a-wave share of lin0) in its own S.mk -- in the mask-block format the four
// row blocks of a member are exactly one 32-bit word per layer and half (two members share a word for the 256-row lin3) -- and nobody
// extracts bits from staged rows: the lead member's per-unit bit extraction made it the last to publish in every layer. assemble = false: the members are known to be resident (a later
// march step of the same launch), no arrival / go handshake (S.sc1 keeps the first step's verdict).
template <int CL, bool KEEP, bool ALL_LIN8 = false, bool MASK_OWN = false>
__device__ __forceinline__ float mlp_forward16_cl(const DecoderDev& D, const DecoderDev16& D16, const float* __restrict__ c0,
                                                  const float* __restrict__ c4, Smem16CLX& S, const Xchg& xc, int cluster, int member,
                                                  bool assemble = true, int32_t abort_lane = 0, bool lin0_lds = false) {
  const int tid = cl_tid();
  const int wave = cl_wave(tid);
  const int lane = tid & 63;
  const int kq = lane >> 4;
  const int ray = tid & 15;
  cluster = __builtin_amdgcn_readfirstlane(cluster);      // (uniform by construction; the request bases are SGPR operands)
  member = __builtin_amdgcn_readfirstlane(member);
  float* X = S.X;
  char* xbase = xc.buf + (size_t)cluster * XCLUSTER_BYTES;
  uint32_t* flags = xc.flags + (size_t)cluster * 128;
  const bool lead = (member == CL - 1);      // the LAST member leads (see cl_lead)
  DISTR_XTS(0);
  if (tid == 0) {   // arrival word first: the lead member counts them while everybody computes lin0
    S.fail = 0;
    if (assemble) {
      S.went = 0;
      uint32_t xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      S.sc1 = 1;    // (until the verdict)
      __hip_atomic_store(flags + member, (xc.epoch << 4) | (xcc & 15u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // lin1's first weights (ring slots 0..2) and start values (accumulator set 1) travel while lin0 runs
  cl_load_chunk<512, 512, CL, 0, 0>(D16.Wf[1], member, wave, lane);
  cl_load_chunk<512, 512, CL, 1, 1>(D16.Wf[1], member, wave, lane);
  cl_load_chunk<512, 512, CL, 2, 2>(D16.Wf[1], member, wave, lane);
  cl_load_start<512, 512, CL, 1>(D.bias[1], member, wave, kq);
  X[tid] = (tid < 48) ? S.xyz[tid] : 0.f;
  if (KEEP && (lead || MASK_OWN)) {   // the rays' mask blocks are OR-ed together nibble by nibble (mask_nibble_or)
    uint4* z = reinterpret_cast<uint4*>(&S.mk[0][0]);
    z[tid] = make_uint4(0u, 0u, 0u, 0u);
    z[tid + NTHREADS] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  {  // lin0 (K = 16 padded): cheaper to compute whole in every member than to exchange
    f32x4 acc[8];
    if (lin0_lds) {          // sticky tiles: latent constants and lin0 fragments from LDS (Smem16CLX, filled once per tile)
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) acc[ob] = *reinterpret_cast<const f32x4*>(&S.c0s[wave * 128 + 16 * ob + 4 * kq]);
    } else {
      acc_init16<8>(acc, c0, wave * 128, kq);
    }
    CL_ASM(CL, "" ::: "memory");        // (lin0's accumulators live across this point: not in the fixed registers the ring is landing in)
    if (lin0_lds) dense16_lin0_lds(S.w0s[wave], X, acc, lane);
    else dense16<16, 8>(D16.Wf[0], X, acc, wave, lane);
    CL_ASM(CL, "" ::: "memory");
    __syncthreads();
    (void)writeback16<8, false>(X, acc, wave * 128, lane);
    if (KEEP && (lead || MASK_OWN)) {
#pragma unroll
      for (int ob = 0; ob < 8; ++ob) {     // (MASK_OWN: the member's share of lin0 = the row blocks it owns in every 512-row layer)
        if (!MASK_OWN || (wave * 8 + ob) / (32 / CL) == member) mask_nibble_put(S, 0, wave * 8 + ob, 7, kq, ray, mask_nibble_of(acc[ob]));
      }
    }
    __syncthreads();
  }
  DISTR_XTS(1);
  if (assemble) {
    cl_assemble<CL>(flags, member, xc.epoch, tid, &S.fail, &S.sc1, (xc.test_abort & 1) | abort_lane, xc.force_sc1, xc.t_go > 0 ? (long long)xc.t_go : CL_T_GO);
    if (S.fail) { cl_wait_vm<0>(); return 0.f; }
    if (tid == 0) S.went = 1;        // (read behind later barriers only)
  }
  // position of every layer's chunk 0 in the network-wide chunk sequence (see layer_cl)
  constexpr int N1 = ClGeom<512, 512, CL>::NCH, N3 = ClGeom<512, 256, CL>::NCH, N4 = ClGeom<256, 512, CL>::NCH;
  static_assert(N1 >= CL_AHEAD, "the initial requests cover lin1's first chunks");
  constexpr int G1 = 0, G2 = G1 + N1, G3 = G2 + N1, G4 = G3 + N3, G5 = G4 + N4, G6 = G5 + N1, G7 = G6 + N1, GT = G7 + N1;
  const int kp = !KEEP ? 0 : MASK_OWN ? 2 : (lead ? 3 : 0);
  // (behind lin0, whose compiler-generated loop may use any register: from here to the end of lin7 only the statements name these)
  if constexpr (CL == 8) cl8_set_fixed(lds_off(X) + (uint32_t)lane * 4u, lds_off(X) + (uint32_t)((4 * kq * 16 + ray) * 4), (uint32_t)lane * 16u);
  if (!layer_cl<1, 512, 512, CL, G1, GT, 512, 512, 512, 256, true, true>(D16.Wf[1], D16.Wf[2], D16.Wf[3], D.bias[2], S, xc, xbase, member, kp)) return 0.f;
  if (!layer_cl<2, 512, 512, CL, G2, GT, 512, 256, 256, 512, false, true>(D16.Wf[2], D16.Wf[3], D16.Wf[4], D.bias[3], S, xc, xbase, member, kp)) return 0.f;
  if (!layer_cl<3, 512, 256, CL, G3, GT, 256, 512, 512, 512, false, true>(D16.Wf[3], D16.Wf[4], D16.Wf[5], c4, S, xc, xbase, member, kp)) return 0.f;
  if (!layer_cl<4, 256, 512, CL, G4, GT, 512, 512, 512, 512, false, true>(D16.Wf[4], D16.Wf[5], D16.Wf[6], D.bias[5], S, xc, xbase, member, kp)) return 0.f;
  if (!layer_cl<5, 512, 512, CL, G5, GT, 512, 512, 512, 512, false, true>(D16.Wf[5], D16.Wf[6], D16.Wf[7], D.bias[6], S, xc, xbase, member, kp)) return 0.f;
  if (!layer_cl<6, 512, 512, CL, G6, GT, 512, 512, 0, 0, false, true>(D16.Wf[6], D16.Wf[7], nullptr, D.bias[7], S, xc, xbase, member, kp)) return 0.f;
  if (!lead && !ALL_LIN8) {
    (void)layer_cl<7, 512, 512, CL, G7, GT, 0, 0, 0, 0, false, false>(D16.Wf[7], nullptr, nullptr, nullptr, S, xc, xbase, member, 0);
    return 0.f;                          // (the slice stores complete before the wave ends)
  }
  if (!layer_cl<7, 512, 512, CL, G7, GT, 0, 0, 0, 0, false, true>(D16.Wf[7], nullptr, nullptr, nullptr, S, xc, xbase, member, kp)) return 0.f;
  {  // h7: all four units (nothing is requested after them)
    const char* slot7 = xbase + ((7 + xc.par) & 1) * XSLOT_BYTES;
    const uint32_t tag7 = (xc.epoch << 3) | 7u;
    constexpr int PER7 = 32 / CL;
    cl_stage_unit<CL, 0, 0>(slot7, tag7, member * PER7, PER7, S, tid, wave, KEEP && lead && !MASK_OWN, 7, 7);
    cl_request_unit<CL, 1>(slot7, wave, lane);
    cl_request_unit<CL, 2>(slot7, wave, lane);
    cl_stage_unit<CL, 1, 4>(slot7, tag7, member * PER7, PER7, S, tid, wave, KEEP && lead && !MASK_OWN, 7, 7);
    cl_request_unit<CL, 3>(slot7, wave, lane);
    cl_stage_unit<CL, 2, 4>(slot7, tag7, member * PER7, PER7, S, tid, wave, KEEP && lead && !MASK_OWN, 7, 7);
    cl_stage_unit<CL, 3, 0>(slot7, tag7, member * PER7, PER7, S, tid, wave, KEEP && lead && !MASK_OWN, 7, 7);
    // tests (test_abort bit 1): member 0 gives up HERE, behind its last slice -- the rest of the cluster completes without noticing
    if ((xc.test_abort & 2) && member == 0 && !lead && tid == 0) S.fail = 1;
    __syncthreads();
    if (S.fail) { cl_wait_vm<0>(); return 0.f; }
  }
  DISTR_XTS(32);
  {
    float p = 0.f;
    const float* w8 = D.w8 + wave * 128;
    const float* xr = X + (size_t)wave * 128 * 16 + ray;
#pragma unroll 8
    for (int k = 0; k < 128; ++k) p = __builtin_fmaf(w8[k], xr[k * 16], p);
    S.part[wave * 16 + ray] = p;
  }
  __syncthreads();
  return ((S.part[ray] + S.part[16 + ray]) + (S.part[32 + ray] + S.part[48 + ray])) + D.b8;
}

}  // namespace distr
