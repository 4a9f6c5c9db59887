// Microbenchmark: latency of one "layer exchange" between the CL workgroups of a cluster that would split one 16-ray
// decoder tile's output rows over CL compute units: every workgroup writes its slice (512/CL rows x 16 rays x 4 B) to a
// global buffer, signals an agent-scope counter (release), spins until all CL have arrived (acquire), then reads the
// whole 32 KB activation block back. Reports wall time per exchange for clusters inside one XCD (workgroup ids congruent
// mod 8) and spread over XCDs. Spins are bounded (no hang on a scheduling surprise). Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ long long wall() { return (long long)wall_clock64(); }

template <int CL, int UC>
__global__ void __launch_bounds__(256) k(int iters, int same_xcd, float* buf /*[clusters][2][512*16]*/, unsigned* flags /*[clusters]*/,
                                         float* out, long long* ticks, int* fail) {
  __shared__ float X[512 * 16];
  const int tid = threadIdx.x;
  // cluster / member from the block id: same_xcd -> members are blocks b, b+8, b+16, ... (same id mod 8 = same XCD)
  int cluster, member;
  const int nclusters = gridDim.x / CL;
  if (same_xcd) { const int g = blockIdx.x / (8 * CL), r = blockIdx.x % (8 * CL); cluster = g * 8 + (r % 8); member = r / 8; }
  else { cluster = blockIdx.x / CL; member = blockIdx.x % CL; }
  if (cluster >= nclusters) return;
  float* base = buf + (size_t)cluster * 2 * 8192;
  unsigned* flag = flags + cluster * 32;    // one 128-byte line per cluster
  constexpr int SL = 8192 / CL;             // floats per slice
  float v = (float)(member + 1);
  long long t0 = 0;
  for (int it = 0; it < iters; ++it) {
    if (it == 8) t0 = wall();
    float* dst = base + (it & 1) * 8192 + member * SL;
    for (int i = tid * 4; i < SL; i += 1024) *reinterpret_cast<float4*>(dst + i) = make_float4(v, v + 1, v + 2, v + 3);
    if (UC) __builtin_amdgcn_s_waitcnt(0);   // every wave: its stores have reached (uncached) memory
    __syncthreads();
    if (tid == 0) {
      if (UC) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // stores already drained (syncthreads waits vmcnt)
      else __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(CL * (it + 1));
      int spins = 0;
      while ((UC ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                 : __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < target) {
        if (++spins > (1 << 22)) { *fail = 1; break; }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (!UC) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const float* src = base + (it & 1) * 8192;
    for (int i = tid * 4; i < 8192; i += 1024) *reinterpret_cast<float4*>(X + i) = *reinterpret_cast<const float4*>(src + i);
    __syncthreads();
    v = X[(tid * 37 + it) & 8191] * 0.5f + (float)member;    // data dependence between iterations
  }
  const long long t1 = wall();
  out[blockIdx.x * 256 + tid] = v;
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int CL, int UC>
void run(int clusters, int same_xcd, float* buf, unsigned* flags, float* out, long long* ticks, int* fail) {
  const int iters = 2008, blocks = clusters * CL;
  hipMemset(flags, 0, 4096 * 128); hipMemset(fail, 0, 4);
  hipLaunchKernelGGL((k<CL, UC>), dim3(blocks), dim3(256), 0, 0, iters, same_xcd, buf, flags, out, ticks, fail);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks); int f = 0;
  hipMemcpy(h.data(), ticks, 8 * blocks, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
  double mx = 0, av = 0; for (auto t : h) { av += (double)t; if ((double)t > mx) mx = (double)t; } av /= blocks;
  printf("%s CL=%d clusters=%3d %-9s: %.2f us per exchange (mean), %.2f us (slowest workgroup)%s\n", UC ? "uncached" : "cached+fences", CL, clusters, same_xcd ? "same-XCD" : "spread",
         av / 2000.0 / 100.0, mx / 2000.0 / 100.0, f ? "  [SPIN LIMIT HIT]" : "");   // wall_clock64: 100 MHz
}

int main() {
  float *buf, *out, *ubuf; unsigned *flags, *uflags; long long* ticks; int* fail;
  hipMalloc(&buf, (size_t)512 * 2 * 8192 * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&flags, 4096 * 128); hipMalloc(&ticks, 4096 * 8); hipMalloc(&fail, 4);
  if (hipExtMallocWithFlags((void**)&ubuf, (size_t)512 * 2 * 8192 * 4, hipDeviceMallocUncached) != hipSuccess ||
      hipExtMallocWithFlags((void**)&uflags, 4096 * 128, hipDeviceMallocUncached) != hipSuccess) { printf("uncached malloc failed\n"); return 1; }
  for (int same = 1; same >= 0; --same) {
    run<4, 0>(8, same, buf, flags, out, ticks, fail);
    run<4, 0>(64, same, buf, flags, out, ticks, fail);
    run<2, 1>(8, same, ubuf, uflags, out, ticks, fail);
    run<4, 1>(8, same, ubuf, uflags, out, ticks, fail);
    run<8, 1>(8, same, ubuf, uflags, out, ticks, fail);
    run<4, 1>(16, same, ubuf, uflags, out, ticks, fail);
    run<4, 1>(64, same, ubuf, uflags, out, ticks, fail);
    run<4, 1>(256, same, ubuf, uflags, out, ticks, fail);
    run<8, 1>(128, same, ubuf, uflags, out, ticks, fail);
  }
  return 0;
}
