// Microbenchmark: sustained issue interval of v_mfma_f32_32x32x2_f32 (cycles per MFMA per SIMD, s_memtime) as a function
// of waves per SIMD, operand data (zero / non-zero) and independent accumulators. Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k(int iters, float av, float bv, float* out, long long* cyc) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = av * (1.0f + lane * 0.01f), b = bv * (1.0f - lane * 0.003f);
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
void run(const char* name, int blocks_per_cu, float av, float bv, float* out, long long* cyc) {
  const int iters = 20000 / NACC * 8 / 8, blocks = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, 100, av, bv, out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, iters, av, bv, out, cyc);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[4096]; hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks;
  const double n_mfma_per_simd = (double)iters * NACC * blocks_per_cu;   // waves of blocks_per_cu blocks share a SIMD
  printf("%-44s %7.3f ms  %6.1f TF/s  %5.1f cycles per MFMA per SIMD (wave-local %5.1f)  clock %.2f GHz\n", name, ms,
         (double)blocks * 4 * iters * NACC * 4096.0 / ms / 1e9, c / n_mfma_per_simd, c / ((double)iters * NACC), c / (ms * 1e6));
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 8);
  run<8>("1 wave/SIMD, 8 acc, non-zero data", 1, 0.37f, 0.91f, out, cyc);
  run<8>("1 wave/SIMD, 8 acc, zero data", 1, 0.f, 0.f, out, cyc);
  run<4>("1 wave/SIMD, 4 acc, non-zero data", 1, 0.37f, 0.91f, out, cyc);
  run<2>("1 wave/SIMD, 2 acc, non-zero data", 1, 0.37f, 0.91f, out, cyc);
  run<8>("2 waves/SIMD, 8 acc, non-zero data", 2, 0.37f, 0.91f, out, cyc);
  run<8>("2 waves/SIMD, 8 acc, zero data", 2, 0.f, 0.f, out, cyc);
  run<8>("4 waves/SIMD, 8 acc, non-zero data", 4, 0.37f, 0.91f, out, cyc);
  return 0;
}
