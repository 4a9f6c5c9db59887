// Microbenchmark: latency of DEPENDENT f32 MFMAs (one accumulator chain: D = A*B + D, the k-ordered chain of one output row)
// versus the number of independent chains a wave interleaves -- what bounds a cluster tile's compute phase (distr_mlp.hpp::layer_cl).
// One wave per SIMD, 256 workgroups. Cycles (s_memtime) per MFMA per wave. Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int SHAPE, bool VG = false>
__global__ void __launch_bounds__(256) k(int iters, float av, float bv, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63;
  float a = av * (1.0f + lane * 0.01f), b = bv * (1.0f - lane * 0.003f);
  float s = 0.f;
  long long t0, t1;
  if (SHAPE == 16) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
          if (VG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));   // accumulator in architectural VGPRs
          else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);                        // (the compiler picks AGPRs)
        }
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  } else if (SHAPE == 32) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    t1 = __builtin_readcyclecounter();
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int SHAPE, bool VG = false>
void run(const char* name, float* out, long long* cyc) {
  const int iters = 4000, blocks = 256;
  hipLaunchKernelGGL((k<NACC, SHAPE, VG>), dim3(blocks), dim3(256), 0, 0, 100, 0.37f, 0.91f, out, cyc);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, SHAPE, VG>), dim3(blocks), dim3(256), 0, 0, iters, 0.37f, 0.91f, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks;
  const double n = (double)iters * 4 * NACC;
  const int kper = SHAPE == 16 ? 4 : SHAPE == 32 ? 2 : 1;
  printf("%-52s %6.1f cycles per MFMA (wave-local), %6.1f per MFMA of ONE chain, %5.1f cycles per k-step of a chain; clock %.2f GHz\n",
         name, c / n, c / n * NACC, c / n * NACC / kper, c / (ms * 1e6));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<1, 16>("v_mfma_f32_16x16x4_f32, 1 chain", out, cyc);
  run<2, 16>("v_mfma_f32_16x16x4_f32, 2 chains", out, cyc);
  run<3, 16>("v_mfma_f32_16x16x4_f32, 3 chains", out, cyc);
  run<4, 16>("v_mfma_f32_16x16x4_f32, 4 chains", out, cyc);
  run<8, 16>("v_mfma_f32_16x16x4_f32, 8 chains", out, cyc);
  run<1, 16, true>("v_mfma_f32_16x16x4_f32, 1 chain, VGPR accumulator", out, cyc);
  run<2, 16, true>("v_mfma_f32_16x16x4_f32, 2 chains, VGPR accumulators", out, cyc);
  run<4, 16, true>("v_mfma_f32_16x16x4_f32, 4 chains, VGPR accumulators", out, cyc);
  run<1, 32>("v_mfma_f32_32x32x2_f32, 1 chain", out, cyc);
  run<2, 32>("v_mfma_f32_32x32x2_f32, 2 chains", out, cyc);
  run<4, 32>("v_mfma_f32_32x32x2_f32, 4 chains", out, cyc);
  run<1, 4>("v_mfma_f32_4x4x1_16B_f32, 1 chain", out, cyc);
  run<2, 4>("v_mfma_f32_4x4x1_16B_f32, 2 chains", out, cyc);
  run<4, 4>("v_mfma_f32_4x4x1_16B_f32, 4 chains", out, cyc);
  run<8, 4>("v_mfma_f32_4x4x1_16B_f32, 8 chains", out, cyc);
  return 0;
}
