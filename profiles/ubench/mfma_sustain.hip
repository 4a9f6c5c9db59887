// Microbenchmark: SUSTAINED rate and shader clock of a pure v_mfma_f32_32x32x2_f32 stream over several seconds, per operand
// pattern. One workgroup of 4 waves per CU (the decoder tile's shape), 8 independent accumulators per wave (no dependent-issue
// stalls). Each launch runs ~8 ms; the launches of one pattern are repeated for `secs` seconds and every launch reports
//   TF/s = MFMA flops / hipEvent time,   clock = s_memtime cycles / hipEvent time,   cycles per MFMA per SIMD.
// Patterns: zero operands; a constant non-zero operand pair per lane (accumulators grow -> exponent/mantissa toggling);
// "decoder-like": operands re-drawn every iteration from a small in-register LCG (sign and magnitude change every MFMA, like
// weights x post-ReLU activations; the random signs keep the accumulators a bounded random walk). Build: hipcc -O3 --offload-arch=gfx950 mfma_sustain.hip -o mfma_sustain
// Run beside `rocm-smi --showclocks --showpower` sampling (profiles/ubench/run_ubench.sh) to see the power-management state.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: zero, 1: constant non-zero, 2: decoder-like (changing operands)
__global__ void __launch_bounds__(256) k(int iters, float* out, long long* cyc) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a = (MODE == 0) ? 0.f : 0.037f * (1.0f + lane * 0.01f), b = (MODE == 0) ? 0.f : 0.91f * (1.0f - lane * 0.003f);
  uint32_t s = 12345u + threadIdx.x * 2654435761u;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 2) {
      // new operands every iteration: |a| ~ 0.03 weights with random sign, b >= 0 (post-ReLU) in [0, 1), 40 % zeros
      s = s * 1664525u + 1013904223u;
      a = __uint_as_float((s & 0x807fffffu) | 0x3c800000u);                                   // +-[1/64, 1/32)
      const uint32_t t = s * 2246822519u;
      b = ((t >> 28) < 6u) ? 0.f : __uint_as_float(((t >> 9) & 0x007fffffu) | 0x3f000000u) - 0.5f;   // [0, 0.5) or 0
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void sustain(const char* name, double secs, float* out, long long* cyc) {
  const int iters = 40000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const auto w0 = std::chrono::steady_clock::now();
  int n = 0;
  double tf_min = 1e9, tf_max = 0, tf_sum = 0, ck_sum = 0, ck_min = 1e9, cpm_sum = 0;
  printf("## %s\n", name);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < secs) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, iters, out, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks;
    const double tf = (double)blocks * 4 * iters * 8 * 4096.0 / ms / 1e9, ck = c / (ms * 1e6), cpm = c / ((double)iters * 8);
    if (n % 25 == 0) printf("  t=%6.2fs  launch %4d  %7.3f ms  %6.1f TF/s  clock %.3f GHz  %5.2f cycles/MFMA/SIMD\n",
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count(), n, ms, tf, ck, cpm);
    tf_min = tf < tf_min ? tf : tf_min; tf_max = tf > tf_max ? tf : tf_max; tf_sum += tf; ck_sum += ck; ck_min = ck < ck_min ? ck : ck_min; cpm_sum += cpm;
    ++n;
  }
  printf("  => %d launches: %.1f TF/s mean (min %.1f, max %.1f), clock %.3f GHz mean (min %.3f), %.2f cycles/MFMA/SIMD; peak 157.3 TF/s = 64 cycles at 2.4 GHz\n\n",
         n, tf_sum / n, tf_min, tf_max, ck_sum / n, ck_min, cpm_sum / n);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 4.0;
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("# %s, %d CUs, clockRate %.0f MHz; one 4-wave workgroup per CU, 8 accumulators per wave, %g s per pattern\n\n", p.name,
         p.multiProcessorCount, p.clockRate / 1e3, secs);
  sustain<0>("zero operands", secs, out, cyc);
  sustain<1>("constant non-zero operands", secs, out, cyc);
  sustain<2>("decoder-like operands (new random weights x post-ReLU activations every MFMA group)", secs, out, cyc);
  sustain<0>("zero operands again (after the chip is warm)", secs, out, cyc);
  return 0;
}
