// Microbenchmark (round 5): does a DEPENDENT chain of v_mfma_f32_16x16x4_f32 run at 32 cycles per MFMA when the A operand and / or the
// accumulator sit in accumulation registers, and with LDS reads / vector-memory requests interleaved (the cluster tile's k-loop,
// distr_mlp.hpp cl8_unit_a)? One wave per SIMD, 256 workgroups, cycles (s_memtime) per MFMA. Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(int iters, const float* src, float* out, long long* cyc) {
  __shared__ float X[4096];
  const int lane = threadIdx.x & 63;
  X[threadIdx.x] = src[threadIdx.x]; X[threadIdx.x + 256] = src[threadIdx.x + 256];
  __syncthreads();
  float a = src[lane] + 1.0f, b = src[64 + lane] * 0.5f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const unsigned xb = (unsigned)(size_t)X + lane * 4;
  const float* gp = src + lane * 4;
  asm volatile("v_accvgpr_write_b32 a100, %0\n\tv_accvgpr_write_b32 a101, %0\n\tv_accvgpr_write_b32 a102, %0\n\tv_accvgpr_write_b32 a103, %0\n\t"
               "v_accvgpr_write_b32 a240, %1\n\tv_accvgpr_write_b32 a241, %1\n\tv_accvgpr_write_b32 a242, %1\n\tv_accvgpr_write_b32 a243, %1" ::"v"(a), "v"(0.f) : "a100", "a101", "a102", "a103", "a240", "a241", "a242", "a243");
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {        // accumulator VGPR, A VGPR
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    } else if (MODE == 1) { // accumulator AGPR (fixed), A VGPR
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_16x16x4_f32 a[240:243], %0, %1, a[240:243]" ::"v"(a), "v"(b) : "a240", "a241", "a242", "a243");
    } else if (MODE == 2) { // accumulator AGPR, A AGPR (fixed)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_16x16x4_f32 a[240:243], a[100+%1], %0, a[240:243]" ::"v"(b), "n"(0) : "a240", "a241", "a242", "a243");
    } else if (MODE == 3) { // MODE 2 + two ds_read2st64 and an lgkmcnt wait per 4 MFMAs (inside ONE statement)
      asm volatile(
          "ds_read2st64_b32 v[208:209], %1 offset0:0 offset1:1\n\tds_read2st64_b32 v[210:211], %1 offset0:2 offset1:3\n\ts_waitcnt lgkmcnt(0)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v208, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v209, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v210, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v211, a[240:243]\n\t"
          "ds_read2st64_b32 v[212:213], %1 offset0:4 offset1:5\n\tds_read2st64_b32 v[214:215], %1 offset0:6 offset1:7\n\ts_waitcnt lgkmcnt(0)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v212, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v213, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v214, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v215, a[240:243]\n\t"
          "ds_read2st64_b32 v[208:209], %1 offset0:8 offset1:9\n\tds_read2st64_b32 v[210:211], %1 offset0:10 offset1:11\n\ts_waitcnt lgkmcnt(0)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v208, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v209, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v210, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v211, a[240:243]\n\t"
          "ds_read2st64_b32 v[212:213], %1 offset0:12 offset1:13\n\tds_read2st64_b32 v[214:215], %1 offset0:14 offset1:15\n\ts_waitcnt lgkmcnt(0)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v212, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v213, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v214, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v215, a[240:243]"
          ::"v"(b), "v"(xb) : "memory", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "a240", "a241", "a242", "a243");
    } else if (MODE == 4) { // MODE 3 with the reads TWO groups ahead (lgkmcnt(4)): what cl8_unit_a does
      asm volatile(
          "ds_read2st64_b32 v[208:209], %1 offset0:0 offset1:1\n\tds_read2st64_b32 v[210:211], %1 offset0:2 offset1:3\n\t"
          "ds_read2st64_b32 v[212:213], %1 offset0:4 offset1:5\n\tds_read2st64_b32 v[214:215], %1 offset0:6 offset1:7\n\t"
          "ds_read2st64_b32 v[216:217], %1 offset0:8 offset1:9\n\tds_read2st64_b32 v[218:219], %1 offset0:10 offset1:11\n\ts_waitcnt lgkmcnt(4)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v208, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v209, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v210, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v211, a[240:243]\n\t"
          "ds_read2st64_b32 v[220:221], %1 offset0:12 offset1:13\n\tds_read2st64_b32 v[222:223], %1 offset0:14 offset1:15\n\ts_waitcnt lgkmcnt(4)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v212, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v213, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v214, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v215, a[240:243]\n\t"
          "s_waitcnt lgkmcnt(2)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v216, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v217, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v218, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v219, a[240:243]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v220, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[101], v221, a[240:243]\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v222, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v223, a[240:243]"
          ::"v"(b), "v"(xb) : "memory", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223",
            "a240", "a241", "a242", "a243");
    } else {                // MODE 4 + two vector-memory requests into AGPRs per 4 MFMAs
      asm volatile(
          "ds_read2st64_b32 v[208:209], %1 offset0:0 offset1:1\n\tds_read2st64_b32 v[210:211], %1 offset0:2 offset1:3\n\t"
          "ds_read2st64_b32 v[212:213], %1 offset0:4 offset1:5\n\tds_read2st64_b32 v[214:215], %1 offset0:6 offset1:7\n\ts_waitcnt lgkmcnt(2)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v208, a[240:243]\n\tglobal_load_dwordx4 a[120:123], %2, off\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[101], v209, a[240:243]\n\tglobal_load_dwordx4 a[124:127], %2, off offset:16\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v210, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v211, a[240:243]\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[100], v212, a[240:243]\n\tglobal_load_dwordx4 a[128:131], %2, off offset:32\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[101], v213, a[240:243]\n\tglobal_load_dwordx4 a[132:135], %2, off offset:48\n\t"
          "v_mfma_f32_16x16x4_f32 a[240:243], a[102], v214, a[240:243]\n\tv_mfma_f32_16x16x4_f32 a[240:243], a[103], v215, a[240:243]\n\t"
          "s_waitcnt vmcnt(0)"
          ::"v"(b), "v"(xb), "v"(gp) : "memory", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127",
            "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a240", "a241", "a242", "a243");
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = acc[0] + acc[1] + acc[2] + acc[3];
  float r;
  asm volatile("s_nop 15\n\ts_nop 7\n\tv_accvgpr_read_b32 %0, a240" : "=v"(r));
  out[blockIdx.x * 256 + threadIdx.x] = s + r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, const float* src, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, iters, src, out, cyc);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double av = 0; for (int i = 0; i < 256; ++i) av += (double)h[i]; av /= 256;
  printf("%-86s %6.1f cycles per MFMA\n", name, av / iters / per_iter);
}

int main() {
  float *src, *out; long long* cyc;
  hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0>("dependent chain, accumulator VGPR, A VGPR", 16, src, out, cyc);
  run<1>("dependent chain, accumulator AGPR, A VGPR", 16, src, out, cyc);
  run<2>("dependent chain, accumulator AGPR, A AGPR", 16, src, out, cyc);
  run<3>("... + 2 ds_read2st64 per 4 MFMAs, waited for at once (lgkmcnt(0) before their group)", 16, src, out, cyc);
  run<4>("... + the reads two groups ahead (lgkmcnt(4)): the cluster tile's statement", 16, src, out, cyc);
  run<5>("... + reads one group ahead and 2 global_load_dwordx4 -> AGPR per 4 MFMAs", 8, src, out, cyc);
  return 0;
}
