// Microbenchmark (round 5): one "layer exchange" of a cluster tile (distr_mlp.hpp, layer_cl) under the hand-off forms of
// MI355X_MICROARCH.md "inter-workgroup visibility": every member of a cluster of CL workgroups publishes its slice of a
// 512 x 16 f32 activation block and ends up with the whole block in LDS.
//   PROTO 0  flag:     payload stores -> vmcnt(0) -> __syncthreads -> per-member epoch word -> one wave polls the CL words ->
//                      bulk read (what round 2..4 shipped, on uncached memory)
//   PROTO 1  granule8: the data is the flag, 8-byte {value, tag = epoch} granules, relaxed agent atomics both sides
//   PROTO 2  granule16: the same granules moved two per 16-byte access (8-byte halves individually tagged)
//   PROTO 3  signbit:  post-ReLU values have a free sign bit; it carries the parity of the slot's use count, 16-byte accesses
//                      (slots zero-initialised by the host; every position is rewritten on every use)
// MEM 0 = hipMalloc (cached), 1 = hipDeviceMallocUncached.  STF 0 = plain stores, 1 = sc1 (write-through) stores.
// Loads of polled data are always sc1 (L1-bypassing). Every received word is checked; spins are bounded.
// Build: hipcc -O3 --offload-arch=gfx950 cluster_exchange2.hip -o cluster_exchange2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long long wall() { return (long long)wall_clock64(); }
__device__ __forceinline__ void ld16_sc1(u32x4& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void ld8_sc1(u32x2& d, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory"); }
template <int STF> __device__ __forceinline__ void st16(void* p, u32x4 v) {
  if (STF) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int STF> __device__ __forceinline__ void st8(void* p, u32x2 v) {
  if (STF) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void waitvm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N> __device__ __forceinline__ void landed(u32x4 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}
template <int N> __device__ __forceinline__ void landed(u32x2 (&v)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

__device__ __forceinline__ unsigned pat(int member_of_idx, int it, int idx, unsigned salt) {
  return ((unsigned)(it * 131 + idx * 7 + member_of_idx * 3 + salt) & 0x007fffffu) | 0x3f000000u;   // a positive float pattern
}

constexpr long long T_SPIN = 200 * 100;   // 200 us (100 MHz ticks)

template <int CL, int PROTO, int STF>
__global__ void __launch_bounds__(256) k(int iters, int same_xcd, unsigned epoch0, unsigned* buf, unsigned* flags, float* out, long long* ticks,
                                         int* fail, int* errs, int* xcc) {
  __shared__ unsigned X[8192];
  __shared__ int sfail;
  const int tid = threadIdx.x;
  int cluster, member;
  const int nclusters = gridDim.x / CL;
  if (same_xcd) { const int g = blockIdx.x / (8 * CL), r = blockIdx.x % (8 * CL); cluster = g * 8 + (r % 8); member = r / 8; }
  else { cluster = blockIdx.x / CL; member = blockIdx.x % CL; }
  if (cluster >= nclusters) return;
  if (tid == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = (int)(id & 0xf);
    sfail = 0;
  }
  constexpr int GR = (PROTO == 1 || PROTO == 2) ? 2 : 1;       // dwords per value
  unsigned* base = buf + (size_t)cluster * 2 * 8192 * 2;        // (sized for granules)
  unsigned* flag = flags + cluster * 128;                       // [16][8] words per cluster
  constexpr int SL = 8192 / CL;                                 // values per slice
  unsigned salt = 0;
  long long t0 = 0;
  int nerr = 0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    if (it == 8) t0 = wall();
    const unsigned epoch = epoch0 + (unsigned)it;
    unsigned* slot = base + (size_t)(it & 1) * 8192 * GR;
    const unsigned par = (PROTO == 3) ? ((((unsigned)(it >> 1) + 1u) & 1u) << 31) : 0u;
    // ---- publish the own slice (SL values; 4 values per thread-iteration), own copy straight to LDS
    for (int i = tid * 4; i < SL; i += 1024) {
      const int idx = member * SL + i;
      u32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = pat(member, it, idx + r, salt); X[idx + r] = v[r]; }
      if (PROTO == 0 || PROTO == 3) {
        u32x4 w = v;
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] |= par;
        st16<STF>(slot + idx, w);
      } else if (PROTO == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { u32x2 g; g[0] = v[r]; g[1] = epoch; st8<STF>(slot + (size_t)(idx + r) * 2, g); }
      } else {
        u32x4 a, b;
        a[0] = v[0]; a[1] = epoch; a[2] = v[1]; a[3] = epoch; b[0] = v[2]; b[1] = epoch; b[2] = v[3]; b[3] = epoch;
        st16<STF>(slot + (size_t)idx * 2, a);
        st16<STF>(slot + (size_t)idx * 2 + 4, b);
      }
    }
    if (PROTO == 0) {
      waitvm0();
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag + (it & 15) * 8 + member, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid < 64) {
        const long long ts = wall();
        for (;;) {
          const unsigned v = (tid < CL) ? __hip_atomic_load(flag + (it & 15) * 8 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
          if (__ballot(v != epoch) == 0ull) break;
          if (wall() - ts > T_SPIN) { if (tid == 0) sfail = 1; break; }
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      if (sfail) break;
      // bulk read of the other members' slices
      constexpr int OTHER = (8192 - SL) / 4, NLD = (OTHER + 255) / 256;
      u32x4 v[NLD];
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        int i = tid + q * 256;
        if (i < OTHER) { int e = i * 4; if (e >= member * SL) e += SL; ld16_sc1(v[q], slot + e); }
      }
      waitvm0();
      landed(v);
#pragma unroll
      for (int q = 0; q < NLD; ++q) {
        int i = tid + q * 256;
        if (i < OTHER) { int e = i * 4; if (e >= member * SL) e += SL;
#pragma unroll
          for (int r = 0; r < 4; ++r) X[e + r] = v[q][r]; }
      }
    } else if (PROTO == 3) {
      constexpr int OTHER = (8192 - SL) / 4, NLD = (OTHER + 255) / 256;
      u32x4 v[NLD];
      const long long ts = wall();
      unsigned pending = 0;
#pragma unroll
      for (int q = 0; q < NLD; ++q) if (tid + q * 256 < OTHER) pending |= 1u << q;
      for (;;) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          int i = tid + q * 256;
          if (pending & (1u << q)) { int e = i * 4; if (e >= member * SL) e += SL; ld16_sc1(v[q], slot + e); }
        }
        waitvm0();
        landed(v);
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          if (pending & (1u << q)) {
            const bool ok = ((v[q][0] ^ par) >> 31) == 0 && ((v[q][1] ^ par) >> 31) == 0 && ((v[q][2] ^ par) >> 31) == 0 && ((v[q][3] ^ par) >> 31) == 0;
            if (ok) {
              int i = tid + q * 256; int e = i * 4; if (e >= member * SL) e += SL;
#pragma unroll
              for (int r = 0; r < 4; ++r) X[e + r] = v[q][r] & 0x7fffffffu;
              pending &= ~(1u << q);
            }
          }
        }
        if (__ballot(pending != 0) == 0ull) break;
        if (wall() - ts > T_SPIN) { sfail = 1; break; }
      }
    } else {
      // granules: OTHER values, 2 per 16-byte load (PROTO 2) or 1 per 8-byte load (PROTO 1)
      constexpr int PERLD = (PROTO == 2) ? 2 : 1;
      constexpr int OTHER = (8192 - SL) / PERLD, NLD = (OTHER + 255) / 256;
      const long long ts = wall();
      if (PROTO == 2) {
        u32x4 v[NLD];
        unsigned pending = 0;
#pragma unroll
        for (int q = 0; q < NLD; ++q) if (tid + q * 256 < OTHER) pending |= 1u << q;
        for (;;) {
#pragma unroll
          for (int q = 0; q < NLD; ++q) {
            int i = tid + q * 256;
            if (pending & (1u << q)) { int e = i * 2; if (e >= member * SL) e += SL; ld16_sc1(v[q], slot + (size_t)e * 2); }
          }
          waitvm0();
          landed(v);
#pragma unroll
          for (int q = 0; q < NLD; ++q) {
            if (pending & (1u << q)) {
              if (v[q][1] == epoch && v[q][3] == epoch) {
                int i = tid + q * 256; int e = i * 2; if (e >= member * SL) e += SL;
                X[e] = v[q][0]; X[e + 1] = v[q][2];
                pending &= ~(1u << q);
              }
            }
          }
          if (__ballot(pending != 0) == 0ull) break;
          if (wall() - ts > T_SPIN) { sfail = 1; break; }
        }
      } else {
        u32x2 v[NLD];
        unsigned pending = 0;
#pragma unroll
        for (int q = 0; q < NLD; ++q) if (tid + q * 256 < OTHER) pending |= 1u << q;
        for (;;) {
#pragma unroll
          for (int q = 0; q < NLD; ++q) {
            int i = tid + q * 256;
            if (pending & (1u << q)) { int e = i; if (e >= member * SL) e += SL; ld8_sc1(v[q], slot + (size_t)e * 2); }
          }
          waitvm0();
          landed(v);
#pragma unroll
          for (int q = 0; q < NLD; ++q) {
            if (pending & (1u << q)) {
              if (v[q][1] == epoch) {
                int i = tid + q * 256; int e = i; if (e >= member * SL) e += SL;
                X[e] = v[q][0];
                pending &= ~(1u << q);
              }
            }
          }
          if (__ballot(pending != 0) == 0ull) break;
          if (wall() - ts > T_SPIN) { sfail = 1; break; }
        }
      }
    }
    __syncthreads();
    if (sfail) break;
    // check every word of the block, then derive the next iteration's salt from it (data dependence between exchanges)
    for (int i = tid; i < 8192; i += 256) nerr += (X[i] != pat(i / SL, it, i, salt)) ? 1 : 0;
    salt = X[(it * 37 + 11) & 8191] & 0xffu;
    __syncthreads();
  }
  const long long t1 = wall();
  out[blockIdx.x * 256 + tid] = (float)salt;
  if (nerr) atomicAdd(errs, nerr);
  if (tid == 0) { ticks[blockIdx.x] = t1 - t0; if (sfail) *fail = 1; }
}

static unsigned g_epoch = 1;

template <int CL, int PROTO, int STF>
void run(int mem, int clusters, int same_xcd, unsigned* buf, unsigned* flags, float* out, long long* ticks, int* fail, int* errs, int* xcc) {
  const int iters = 1008, blocks = clusters * CL;
  hipMemset(flags, 0, 4096 * 128); hipMemset(fail, 0, 4); hipMemset(errs, 0, 4);
  if (PROTO == 3) hipMemset(buf, 0, (size_t)512 * 2 * 8192 * 2 * 4);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((k<CL, PROTO, STF>), dim3(blocks), dim3(256), 0, 0, iters, same_xcd, g_epoch, buf, flags, out, ticks, fail, errs, xcc);
  g_epoch += iters + 16;
  hipDeviceSynchronize();
  std::vector<long long> h(blocks); std::vector<int> hx(blocks); int f = 0, e = 0;
  hipMemcpy(h.data(), ticks, 8 * blocks, hipMemcpyDeviceToHost); hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), xcc, 4 * blocks, hipMemcpyDeviceToHost);
  double mx = 0, av = 0; for (auto t : h) { av += (double)t; if ((double)t > mx) mx = (double)t; } av /= blocks;
  int mixed = 0;   // clusters whose members sit on more than one XCD
  for (int c = 0; c < clusters; ++c) {
    int first = -1; bool mix = false;
    for (int m = 0; m < CL; ++m) { const int b = same_xcd ? (c / 8) * 8 * CL + m * 8 + (c % 8) : c * CL + m; if (first < 0) first = hx[b]; else if (hx[b] != first) mix = true; }
    mixed += mix ? 1 : 0;
  }
  static const char* pn[] = {"flag     ", "granule8 ", "granule16", "signbit  "};
  printf("%s %s st=%s CL=%d clusters=%3d %-8s: %.2f us per exchange (mean), %.2f (slowest wg); wrong words %d; clusters on >1 XCD %d%s\n", pn[PROTO], mem ? "uncached" : "cached  ",
         STF ? "sc1  " : "plain", CL, clusters, same_xcd ? "same-XCD" : "spread", av / 1000.0 / 100.0, mx / 1000.0 / 100.0, e, mixed, f ? "  [SPIN LIMIT HIT]" : "");
  fflush(stdout);
}

int main() {
  unsigned *buf, *ubuf, *flags, *uflags; float* out; long long* ticks; int *fail, *errs, *xcc;
  const size_t bb = (size_t)512 * 2 * 8192 * 2 * 4;
  hipMalloc(&buf, bb); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&flags, 4096 * 128); hipMalloc(&ticks, 4096 * 8); hipMalloc(&fail, 4); hipMalloc(&errs, 4); hipMalloc(&xcc, 4096 * 4);
  if (hipExtMallocWithFlags((void**)&ubuf, bb, hipDeviceMallocUncached) != hipSuccess || hipExtMallocWithFlags((void**)&uflags, 4096 * 128, hipDeviceMallocUncached) != hipSuccess) {
    printf("uncached malloc failed\n"); return 1; }
  hipMemset(buf, 0, bb); hipMemset(ubuf, 0, bb);
  for (int same = 1; same >= 0; --same) {
    for (int ncl : {8, 24}) {
      // what ships today: flag protocol on uncached memory
      run<8, 0, 0>(1, ncl, same, ubuf, uflags, out, ticks, fail, errs, xcc);
      // granules, cached memory, plain / sc1 stores
      run<8, 1, 0>(0, ncl, same, buf, flags, out, ticks, fail, errs, xcc);
      run<8, 1, 1>(0, ncl, same, buf, flags, out, ticks, fail, errs, xcc);
      run<8, 2, 0>(0, ncl, same, buf, flags, out, ticks, fail, errs, xcc);
      run<8, 2, 1>(0, ncl, same, buf, flags, out, ticks, fail, errs, xcc);
      run<8, 3, 0>(0, ncl, same, buf, flags, out, ticks, fail, errs, xcc);
      run<8, 3, 1>(0, ncl, same, buf, flags, out, ticks, fail, errs, xcc);
      // the same on uncached memory
      run<8, 2, 1>(1, ncl, same, ubuf, uflags, out, ticks, fail, errs, xcc);
      run<8, 3, 1>(1, ncl, same, ubuf, uflags, out, ticks, fail, errs, xcc);
      // flag protocol on cached memory with sc1 payload (handoff-flag R1 without the acquire: loads are sc1)
      run<8, 0, 1>(0, ncl, same, buf, uflags, out, ticks, fail, errs, xcc);
    }
    run<4, 2, 1>(0, 56, same, buf, flags, out, ticks, fail, errs, xcc);
    run<4, 3, 1>(0, 56, same, buf, flags, out, ticks, fail, errs, xcc);
    run<4, 3, 0>(0, 56, same, buf, flags, out, ticks, fail, errs, xcc);
    run<2, 2, 1>(0, 120, same, buf, flags, out, ticks, fail, errs, xcc);
    run<2, 3, 1>(0, 120, same, buf, flags, out, ticks, fail, errs, xcc);
    run<2, 3, 0>(0, 120, same, buf, flags, out, ticks, fail, errs, xcc);
  }
  return 0;
}
