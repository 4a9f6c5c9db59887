// Where do the 26 % between the six-product split-bf16 layer (66.4 k cycles per 512 x 512 layer on a 64-ray tile,
// split_bf16_layer.hip) and its MFMA-only bound (49.2 k) go? Same harness, same data flow; experiments switch ONE ingredient off:
//   EXP 0  the layer as measured in split_bf16_layer.hip
//   EXP 1  no in-loop split: the B fragments of the first block are reused for every block (no VALU between the MFMAs; LDS reads stay)
//   EXP 2  no weight stream: every block re-reads the layer's block 0 (same instructions, all hits in the CU's own L1)
//   EXP 3  neither
// Results are wrong on purpose in EXP 1..3 (timing only). Build: hipcc -O3 --offload-arch=gfx950 split_bf16_bounds.hip -o split_bf16_bounds
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int HID = 512, TILE = 64, LAYERS = 8;

// ---- host-side bf16 helpers (round to nearest even)
static inline uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- device: split two f32 into bf16 planes (RNE), packed pairs
__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {   // {bf16(a) low, bf16(b) high}
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  const bf2 r = __builtin_convertvector(v, bf2);      // v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = pk_bf16(a, b);
  const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
  p1 = pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);
  p2 = pk_bf16(sa, sb);
}

// LDS activation layout, k-minor: X[k >> 3][ray][k & 7] f32 -> the 8 k-values a lane feeds to one bf16 MFMA are 32 contiguous bytes
__device__ __forceinline__ int xidx(int k, int ray) { return ((k >> 3) * TILE + ray) * 8 + (k & 7); }

// MODE 0: f32 MFMA (A fragments: f32, layout of distr_mlp.hpp::dense but with the k-minor LDS layout)
// MODE 6 / 3: six / three bf16 products. Wp: packed fragment stream of all layers.
template <int MODE, int EXP>
__global__ void __launch_bounds__(256, 1) k_layers(const uint32_t* __restrict__ Wp, const float* __restrict__ x0, float* __restrict__ y_out,
                                                   long long* __restrict__ cyc, int tiles, int layers) {
  __shared__ float X[HID * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  long long tsum = 0;
  for (int t = 0; t < tiles; ++t) {
    for (int i = tid; i < HID * TILE; i += 256) X[i] = x0[i];     // (tile input; same for every tile: timing only)
    __syncthreads();
    const long long c0 = __builtin_readcyclecounter();
    for (int l = 0; l < layers; ++l) {
      f32x16 acc[4][2];
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ob][rb][r] = 0.f;
      if (MODE == 0) {
        // f32: per group of 8 features, lane (i, h) holds W[o][8g + 2s + h], s = 0..3 (one float4 per row block)
        const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)l * (HID / 8) * 4 * 4 * 64 + (size_t)wave * 4 * 64 + lane;
        f32x4 a[4];
        float b[4][2];
#pragma unroll
        for (int ob = 0; ob < 4; ++ob) a[ob] = wp[ob * 64];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) b[s][rb] = X[xidx(2 * s + h, 32 * rb + j)];
#pragma unroll 2
        for (int g = 0; g < HID / 8; ++g) {
          f32x4 an[4];
          float bn[4][2];
          const int gn = (g + 1 < HID / 8) ? g + 1 : g;
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) an[ob] = wp[((size_t)gn * 4 * 4 + ob) * 64];
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) bn[s][rb] = X[xidx(8 * gn + 2 * s + h, 32 * rb + j)];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
              for (int rb = 0; rb < 2; ++rb) acc[ob][rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ob][s], b[s][rb], acc[ob][rb], 0, 0, 0);
#pragma unroll
          for (int ob = 0; ob < 4; ++ob) a[ob] = an[ob];
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) b[s][rb] = bn[s][rb];
        }
      } else {
        // bf16: per block of 16 features, lane (i, h) holds 8 bf16 W_plane[o][16 kb + 8 h + 0..7] per (row block, plane).
        // Register double buffer like distr_mlp.hpp::dense: the weight fragments and the f32 activations of block kb + 1 are
        // requested before the MFMAs of block kb (one wave per SIMD: nothing else hides the L2 round trip); the split of the
        // next block's activations into bf16 planes sits between the MFMAs of the current one.
        const u32x4* wp = reinterpret_cast<const u32x4*>(Wp) + (size_t)l * (HID / 16) * 4 * 4 * 3 * 64 + (size_t)wave * 4 * 3 * 64 + lane;
        constexpr int NPL = (MODE == 6) ? 3 : 2;
        constexpr int NP = (MODE == 6) ? 6 : 3;
        constexpr int PW[6] = {0, 1, 0, 1, 2, 0}, PA[6] = {0, 0, 1, 1, 0, 2};     // w0a0 w1a0 w0a1 | w1a1 w2a0 w0a2
        u32x4 a[4][3], b[2][3];
        f32x4 xr[2][2];
        auto load_a = [&](u32x4 (&dst)[4][3], int kb) {
#pragma unroll
          for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int p = 0; p < NPL; ++p) dst[ob][p] = wp[(((size_t)kb * 4 * 4 + ob) * 3 + p) * 64];
        };
        auto load_x = [&](f32x4 (&dst)[2][2], int kb) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            const f32x4* xp = reinterpret_cast<const f32x4*>(&X[xidx(16 * kb + 8 * h, 32 * rb + j)]);
            dst[rb][0] = xp[0]; dst[rb][1] = xp[1];
          }
        };
        auto split = [&](const f32x4 (&src)[2][2], u32x4 (&dst)[2][3]) {
#pragma unroll
          for (int rb = 0; rb < 2; ++rb) {
            uint32_t q0[4], q1[4], q2[4];
            split_pair(src[rb][0][0], src[rb][0][1], q0[0], q1[0], q2[0]);
            split_pair(src[rb][0][2], src[rb][0][3], q0[1], q1[1], q2[1]);
            split_pair(src[rb][1][0], src[rb][1][1], q0[2], q1[2], q2[2]);
            split_pair(src[rb][1][2], src[rb][1][3], q0[3], q1[3], q2[3]);
#pragma unroll
            for (int i = 0; i < 4; ++i) { dst[rb][0][i] = q0[i]; dst[rb][1][i] = q1[i]; dst[rb][2][i] = q2[i]; }
          }
        };
        load_a(a, 0);
        load_x(xr, 0);
        split(xr, b);
#pragma unroll 2
        for (int kb = 0; kb < HID / 16; ++kb) {
          u32x4 an[4][3], bn[2][3];
          f32x4 xn[2][2];
          const int kn = (kb + 1 < HID / 16) ? kb + 1 : kb;
          load_a(an, (EXP & 2) ? 0 : kn);
          load_x(xn, kn);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob)
#pragma unroll
              for (int rb = 0; rb < 2; ++rb)
                acc[ob][rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ob][PW[q]]), __builtin_bit_cast(bf16x8, b[rb][PA[q]]),
                                                                      acc[ob][rb], 0, 0, 0);
            if (q == 0) {
              if (EXP & 1) {              // keep the LDS reads alive without the split
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
#pragma unroll
                  for (int p = 0; p < 3; ++p) bn[rb][p] = b[rb][p];
                  asm volatile("" :: "v"(xn[rb][0]), "v"(xn[rb][1]));
                }
              } else split(xn, bn);
            }
          }
#pragma unroll
          for (int ob = 0; ob < 4; ++ob)
#pragma unroll
            for (int p = 0; p < NPL; ++p) a[ob][p] = an[ob][p];
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[rb][p] = bn[rb][p];
        }
      }
      __syncthreads();
      // write-back: ReLU (f32) in place; D rows of register r on lane (j, h): (r & 3) + 8 (r >> 2) + 4 h -> 4 consecutive k = one 16-byte store
#pragma unroll
      for (int ob = 0; ob < 4; ++ob)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(acc[ob][rb][4 * q + i], 0.f) * (1.0f / 16.0f);   // (scaled: keeps 8 random layers bounded)
            const int row = wave * 128 + 32 * ob + 8 * q + 4 * h;
            *reinterpret_cast<f32x4*>(&X[xidx(row, 32 * rb + j)]) = v;
          }
      __syncthreads();
    }
    tsum += __builtin_readcyclecounter() - c0;
  }
  if (tid == 0) cyc[blockIdx.x] = tsum;
  if (blockIdx.x == 0)
    for (int i = tid; i < HID * TILE; i += 256) y_out[i] = X[i];
}

int main() {
  const int NWG = 256;
  // weights: 8 layers of N(0, sqrt(2/512)); tile input in [0, 1)
  std::vector<float> W((size_t)LAYERS * HID * HID), x((size_t)HID * TILE);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f; };
  for (auto& w : W) { float u1 = rnd() + 1e-6f, u2 = rnd(); w = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * sqrtf(2.0f / HID); }
  std::vector<float> xk((size_t)HID * TILE);                    // x[k][ray]
  for (auto& v : xk) v = rnd();
  for (int k = 0; k < HID; ++k) for (int r = 0; r < TILE; ++r) x[((k >> 3) * TILE + r) * 8 + (k & 7)] = xk[(size_t)k * TILE + r];

  // packed streams
  std::vector<float> Wf((size_t)LAYERS * HID * HID);            // f32 fragments
  std::vector<uint16_t> Wb((size_t)LAYERS * HID * HID * 3);     // three bf16 planes, fragment order
  for (int l = 0; l < LAYERS; ++l) {
    const float* Wl = &W[(size_t)l * HID * HID];
    for (int g = 0; g < HID / 8; ++g) for (int w = 0; w < 4; ++w) for (int ob = 0; ob < 4; ++ob) for (int lane = 0; lane < 64; ++lane) {
      const int o = w * 128 + 32 * ob + (lane & 31), h = lane >> 5;
      float* d = &Wf[(size_t)l * HID * HID + ((((size_t)g * 4 + w) * 4 + ob) * 64 + lane) * 4];
      for (int sidx = 0; sidx < 4; ++sidx) d[sidx] = Wl[(size_t)o * HID + 8 * g + 2 * sidx + h];
    }
    for (int kb = 0; kb < HID / 16; ++kb) for (int w = 0; w < 4; ++w) for (int ob = 0; ob < 4; ++ob) for (int lane = 0; lane < 64; ++lane) {
      const int o = w * 128 + 32 * ob + (lane & 31), h = lane >> 5;
      for (int i = 0; i < 8; ++i) {
        const float v = Wl[(size_t)o * HID + 16 * kb + 8 * h + i];
        const uint16_t p0 = f2bf(v); const float r1 = v - bf2f(p0);
        const uint16_t p1 = f2bf(r1); const float r2 = r1 - bf2f(p1);
        const uint16_t p2 = f2bf(r2);
        const uint16_t pl[3] = {p0, p1, p2};
        for (int p = 0; p < 3; ++p)
          Wb[(size_t)l * HID * HID * 3 + ((((((size_t)kb * 4 + w) * 4 + ob) * 3 + p) * 64 + lane) * 8) + i] = pl[p];
      }
    }
  }
  float *dWf, *dx, *dy; uint16_t* dWb; long long* dcyc;
  hipMalloc(&dWf, Wf.size() * 4); hipMalloc(&dWb, Wb.size() * 2); hipMalloc(&dx, x.size() * 4); hipMalloc(&dy, x.size() * 4);
  hipMalloc(&dcyc, NWG * sizeof(long long));
  hipMemcpy(dWf, Wf.data(), Wf.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dWb, Wb.data(), Wb.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);

  // float64 reference of ONE layer on the tile (layer 0; the kernels scale by 1/16 after the ReLU)
  std::vector<double> ref((size_t)HID * TILE);
  for (int o = 0; o < HID; ++o) for (int r = 0; r < TILE; ++r) {
    double a = 0; for (int k = 0; k < HID; ++k) a += (double)W[(size_t)o * HID + k] * (double)xk[(size_t)k * TILE + r];
    ref[(size_t)o * TILE + r] = (a > 0 ? a : 0) / 16.0;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](int exp, const char* name) {
    const uint32_t* wp = (const uint32_t*)dWb;
    auto launch = [&](int tiles, int layers) {
      if (exp == 0) hipLaunchKernelGGL((k_layers<6, 0>), dim3(NWG), dim3(256), 0, 0, wp, dx, dy, dcyc, tiles, layers);
      else if (exp == 1) hipLaunchKernelGGL((k_layers<6, 1>), dim3(NWG), dim3(256), 0, 0, wp, dx, dy, dcyc, tiles, layers);
      else if (exp == 2) hipLaunchKernelGGL((k_layers<6, 2>), dim3(NWG), dim3(256), 0, 0, wp, dx, dy, dcyc, tiles, layers);
      else hipLaunchKernelGGL((k_layers<6, 3>), dim3(NWG), dim3(256), 0, 0, wp, dx, dy, dcyc, tiles, layers);
    };
    const int tiles = 16;
    double best_ms = 1e9; double cyc_layer = 0;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0); launch(tiles, LAYERS); hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long hc[NWG]; hipMemcpy(hc, dcyc, sizeof(hc), hipMemcpyDeviceToHost);
      double c = 0; for (int i = 0; i < NWG; ++i) c += (double)hc[i];
      if (rep > 0 && ms < best_ms) { best_ms = ms; cyc_layer = c / NWG / tiles / LAYERS; }
    }
    printf("%-40s %8.0f cycles / 512x512 layer / tile (MFMA-only bound 49152)   %7.3f ms = %.2f GHz\n", name, cyc_layer, best_ms,
           cyc_layer * tiles * LAYERS / best_ms / 1e6);
  };
  (void)ref;
  run(0, "bf16x6 as shipped");
  run(1, "no in-loop split (VALU off)");
  run(2, "no weight stream (L1 hits)");
  run(3, "neither");
  return 0;
}
