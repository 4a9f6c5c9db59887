// Microbenchmark: what does ONE other instruction cost next to a back-to-back v_mfma_f32_32x32x2_f32 stream?
// One wave per SIMD (256-thread workgroup per CU, the decoder tile's shape), 8 independent accumulators; per loop iteration
// 8 MFMAs (512 cycles of MFMA pipe) with NF filler instructions of one type spread between them. Reports cycles per
// iteration (s_memtime) and the extra cycles per filler over the filler-free loop. An instruction that is "free" beside the
// 64-cycle f32 MFMA shows ~0; one that shares the f32 datapath (or stalls MFMA issue) shows its real price.
// Build: hipcc -O3 -w --offload-arch=gfx950 mfma_fillers.hip -o mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { F_NONE, F_VMOV, F_VADD, F_VADD64, F_VFMA, F_SADD, F_DSREAD2, F_GLOAD4, F_SNOP, F_ACCREAD, F_CNDMASK, F_VMULLO, F_DSWRITE, F_VMAXI, F_VLSHLOR,
       F_ACCMOV, F_BUFLOAD4, F_DSREAD1, F_COUNT };
static const char* NAMES[] = {"none", "v_mov_b32", "v_add_u32", "v_lshl_add_u64", "v_fma_f32", "s_add_u32", "ds_read2_b32", "global_load_dwordx4",
                              "s_nop 0", "v_accvgpr_read_b32", "v_cndmask_b32", "v_mul_lo_u32", "ds_write_b32", "v_max_i32", "v_lshl_or_b32",
                              "v_accvgpr_mov_b32", "buffer_load_dwordx4 soff", "ds_read_b32 offset:imm"};

template <int TYPE>
__device__ __forceinline__ void filler(uint32_t& x, uint32_t& y, float& f, unsigned long long& q, const f32x4* g, f32x4& gl, uint32_t lds_addr, f32x16& acc, const f32x4* gb, int soff) {
  if (TYPE == F_VMOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
  if (TYPE == F_VADD) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x) : "v"(y));
  if (TYPE == F_VADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"(q));
  if (TYPE == F_VFMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f) : "v"(f));
  if (TYPE == F_SADD) { uint32_t s; asm volatile("s_add_u32 %0, %1, 1" : "=s"(s) : "s"(0x1234)); }
  if (TYPE == F_DSREAD2) { unsigned long long r; asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(r) : "v"(lds_addr)); }
  if (TYPE == F_GLOAD4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gl) : "v"(g));
  if (TYPE == F_SNOP) asm volatile("s_nop 0");
  if (TYPE == F_ACCREAD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc[0]));
  if (TYPE == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(x) : "v"(y));
  if (TYPE == F_VMULLO) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x) : "v"(y));
  if (TYPE == F_DSWRITE) asm volatile("ds_write_b32 %0, %1" : : "v"(lds_addr), "v"(x));
  if (TYPE == F_VMAXI) asm volatile("v_max_i32 %0, %1, %0" : "+v"(x) : "v"(y));
  if (TYPE == F_VLSHLOR) asm volatile("v_lshl_or_b32 %0, %1, 3, %0" : "+v"(x) : "v"(y));
  if (TYPE == F_BUFLOAD4) {
    // SGPR buffer descriptor + SGPR offset: no VGPR address arithmetic needed to walk the weight stream
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, 0x7fffffff, 0x00020000);
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(gl) : "v"(lds_addr), "s"(r), "s"(soff));
  }
  if (TYPE == F_DSREAD1) { uint32_t r; asm volatile("ds_read_b32 %0, %1 offset:8192" : "=v"(r) : "v"(lds_addr)); }
  if (TYPE == F_ACCMOV) { float t; asm volatile("v_accvgpr_mov_b32 %0, %1" : "=a"(t) : "a"(acc[1])); }
}

template <int TYPE, int NF>
__global__ void __launch_bounds__(256) k(int iters, const f32x4* gbuf, float* out, long long* cyc) {
  __shared__ float lds[8192];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 0.f;
  __syncthreads();
  float a = 0.037f * (1.0f + lane * 0.01f), b = 0.91f * (1.0f - lane * 0.003f);
  uint32_t x = threadIdx.x, y = 3; float f = 0.5f; unsigned long long q = threadIdx.x;
  f32x4 gl = {0.f, 0.f, 0.f, 0.f};
  const f32x4* g = gbuf + threadIdx.x;
  const uint32_t lds_addr = (uint32_t)(threadIdx.x & 63) * 4;
  constexpr int PER = (NF + 7) / 8;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < PER; ++j)
        if (i * PER + j < NF) filler<TYPE>(x, y, f, q, g, gl, lds_addr, acc[(i + 4) & 7], gbuf, __builtin_amdgcn_readfirstlane(it & 1023));
      __builtin_amdgcn_sched_barrier(0);
    }
    if (TYPE == F_GLOAD4 || TYPE == F_BUFLOAD4) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NF * 3 > 60 ? 60 : NF * 3));
    if (TYPE == F_DSREAD2 || TYPE == F_DSWRITE || TYPE == F_DSREAD1) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(NF > 12 ? 12 : NF));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  long long t1 = __builtin_readcyclecounter();
  float sum = (float)x + f + (float)q + gl[0];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static double base_cycles = 0;

template <int TYPE, int NF>
void run(const f32x4* gbuf, float* out, long long* cyc) {
  const int iters = 4000, blocks = 256;
  hipLaunchKernelGGL((k<TYPE, NF>), dim3(blocks), dim3(256), 0, 0, iters, gbuf, out, cyc);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<TYPE, NF>), dim3(blocks), dim3(256), 0, 0, iters, gbuf, out, cyc);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks; c /= iters;
  if (TYPE == F_NONE) base_cycles = c;
  printf("%-22s x%2d per 8 MFMAs: %7.1f cycles per iteration (MFMA alone %.1f) -> %+6.2f cycles per filler\n", NAMES[TYPE], NF, c, base_cycles,
         NF ? (c - base_cycles) / NF : 0.0);
}

template <int TYPE>
void sweep(const f32x4* gbuf, float* out, long long* cyc) {
  run<TYPE, 8>(gbuf, out, cyc);
  run<TYPE, 16>(gbuf, out, cyc);
  run<TYPE, 32>(gbuf, out, cyc);
}

int main() {
  float* out; long long* cyc; f32x4* gbuf;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&gbuf, 4096 * 16); hipMemset(gbuf, 0, 4096 * 16);
  run<F_NONE, 0>(gbuf, out, cyc);
  sweep<F_SNOP>(gbuf, out, cyc);
  sweep<F_SADD>(gbuf, out, cyc);
  sweep<F_VMOV>(gbuf, out, cyc);
  sweep<F_VADD>(gbuf, out, cyc);
  sweep<F_VMAXI>(gbuf, out, cyc);
  sweep<F_VLSHLOR>(gbuf, out, cyc);
  sweep<F_CNDMASK>(gbuf, out, cyc);
  sweep<F_VADD64>(gbuf, out, cyc);
  sweep<F_VFMA>(gbuf, out, cyc);
  sweep<F_VMULLO>(gbuf, out, cyc);
  sweep<F_ACCREAD>(gbuf, out, cyc);
  sweep<F_ACCMOV>(gbuf, out, cyc);
  sweep<F_DSREAD2>(gbuf, out, cyc);
  sweep<F_DSWRITE>(gbuf, out, cyc);
  sweep<F_DSREAD1>(gbuf, out, cyc);
  run<F_GLOAD4, 1>(gbuf, out, cyc);
  run<F_GLOAD4, 2>(gbuf, out, cyc);
  run<F_GLOAD4, 4>(gbuf, out, cyc);
  sweep<F_GLOAD4>(gbuf, out, cyc);
  run<F_BUFLOAD4, 1>(gbuf, out, cyc);
  run<F_BUFLOAD4, 2>(gbuf, out, cyc);
  run<F_BUFLOAD4, 4>(gbuf, out, cyc);
  run<F_BUFLOAD4, 8>(gbuf, out, cyc);
  return 0;
}
