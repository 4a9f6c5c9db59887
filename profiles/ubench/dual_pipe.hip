// Microbenchmark: can a CDNA4 SIMD sustain f32 MFMA and f32 VALU FMA streams from two co-resident waves at once?
// 512-thread workgroups: waves 0-3 (one per SIMD) issue v_mfma_f32_32x32x2_f32, waves 4-7 issue v_fma_f32 with an
// SGPR multiplicand (the shape a lane=ray dot-product chain would have). Modes: 1 = MFMA only, 2 = VALU only, 3 = both.
// Build: hipcc -O3 --offload-arch=gfx950 -o dual_pipe dual_pipe.hip ; run: ./dual_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512, 2) k(int mode, int iters, const float* __restrict__ w, float* out) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!(mode & 1)) return;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = lane * 0.001f, b = lane * 0.002f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  } else {
    if (!(mode & 6)) return;
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    float x = lane * 0.001f;
    // per "iteration" of the MFMA wave (8 MFMA = 512 cycles) a VALU wave can issue 256 v_fma (2 cycles each);
    // weights stream through the scalar cache (a different 1 KiB slice every iteration), x changes per rep
    if (mode & 2) {
      for (int it = 0; it < iters; ++it) {
        const float* wi = w + (it & 63) * 256;
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
          x += 1.0f;
#pragma unroll
          for (int i = 0; i < 32; ++i) acc[i] = __builtin_fmaf(wi[rep * 32 + i], x, acc[i]);   // compiler packs to v_pk_fma_f32
        }
      }
    } else {
      for (int it = 0; it < iters; ++it) {
        const float* wi = w + (it & 63) * 256;
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
          x += 1.0f;
#pragma unroll
          for (int i = 0; i < 32; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(wi[rep * 32 + i]), "v"(x));
        }
      }
    }
    float s = 0.f;
    for (int i = 0; i < 32; ++i) s += acc[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
}

int main() {
  float *w, *out;
  hipMalloc(&w, 64 * 1024);
  hipMalloc(&out, 4096 * 512 * 4);
  hipMemset(w, 0, 64 * 1024);
  const int iters = 4000, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode : {1, 2, 3, 4, 5}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, 100, w, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, iters, w, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma_flop = (mode & 1) ? (double)blocks * 4 * iters * 8 * 4096.0 : 0;          // 32*32*2*2 flop per MFMA
    double valu_flop = (mode & 6) ? (double)blocks * 4 * iters * 256 * 128.0 : 0;         // 64 lanes * 2 flop per v_fma
    printf("mode %d: %.3f ms  MFMA %.1f TF/s  VALU %.1f TF/s  total %.1f TF/s\n", mode, ms, mfma_flop / ms / 1e9, valu_flop / ms / 1e9,
           (mfma_flop + valu_flop) / ms / 1e9);
  }
  return 0;
}
