/* Plain-C program that RENDERS through include/distr.h: device memory from the HIP runtime's C API, the decoder weights, the
 * render configuration (the raw bytes of a distr_render_cfg written by the ctypes mirror), latent and camera from a blob file;
 * forward + backward; outputs written back to a file for the caller to compare byte for byte with the ctypes path.
 * Build: gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include abi_render.c -L /opt/rocm/lib -lamdhip64 -ldl
 * Run:   abi_render libdistr.so in.bin out.bin     (tests/test_gpu_parity.py::test_plain_c_program_renders) */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "distr.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d (%s)\n", #x, rc_, ctx ? last_error(ctx) : ""); return 3; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); return 4; } } while (0)
#define SYM(name) *(void**)(&name) = dlsym(h, "distr_" #name); if (!name) { fprintf(stderr, "missing distr_%s\n", #name); return 2; }

static void* dev_copy(const void* src, size_t n) {
  void* d = NULL;
  if (hipMalloc(&d, n ? n : 4) != hipSuccess) return NULL;
  if (n && hipMemcpy(d, src, n, hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}

int main(int argc, char** argv) {
  if (argc < 4) return 1;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  int (*create_abi)(distr_ctx**, int, uint32_t);
  void (*destroy)(distr_ctx*);
  const char* (*last_error)(const distr_ctx*);
  int (*set_decoder)(distr_ctx*, const distr_decoder_desc*, const float*, size_t);
  int (*workspace_bytes)(distr_ctx*, const distr_render_cfg*, size_t*, size_t*);
  int (*render_forward)(distr_ctx*, const distr_render_cfg*, const float*, const float*, const float*, float*, uint8_t*, float*, float*,
                        float*, void*, size_t, void*);
  int (*render_backward)(distr_ctx*, const distr_render_cfg*, const void*, size_t, const float*, const float*, const float*, const float*,
                         float*, float*, float*, void*, size_t, void*);
  SYM(create_abi) SYM(destroy) SYM(last_error) SYM(set_decoder) SYM(workspace_bytes) SYM(render_forward) SYM(render_backward)

  /* in.bin: int64 n_weights | distr_render_cfg bytes | weights | latent[256] | R[9] | T[3] | g_depth[P] | g_min_sdf[P] | g_normal[3P] */
  FILE* f = fopen(argv[2], "rb");
  if (!f) return 1;
  int64_t nw = 0;
  distr_render_cfg cfg;
  if (fread(&nw, sizeof(nw), 1, f) != 1 || fread(&cfg, sizeof(cfg), 1, f) != 1) return 1;
  const size_t P = (size_t)cfg.H * cfg.W;
  float* w = (float*)malloc(sizeof(float) * (size_t)nw);
  float cam[256 + 9 + 3];
  float* g = (float*)malloc(sizeof(float) * 5 * P);
  if (fread(w, sizeof(float), (size_t)nw, f) != (size_t)nw || fread(cam, sizeof(float), 268, f) != 268 || fread(g, sizeof(float), 5 * P, f) != 5 * P) return 1;
  fclose(f);

  distr_ctx* ctx = NULL;
  CHECK(create_abi(&ctx, 0, DISTR_ABI_VERSION));     /* = the distr_create macro, through dlsym */
  distr_decoder_desc desc = {sizeof(distr_decoder_desc), 256, 512, 9, 4};
  CHECK(set_decoder(ctx, &desc, w, (size_t)nw));
  size_t fwd = 0, bwd = 0;
  CHECK(workspace_bytes(ctx, &cfg, &fwd, &bwd));
  void *ws = NULL, *wsb = NULL;
  HIP(hipMalloc(&ws, fwd)); HIP(hipMalloc(&wsb, bwd));
  float* d_cam = (float*)dev_copy(cam, sizeof(cam));
  float* d_g = (float*)dev_copy(g, sizeof(float) * 5 * P);
  float* d_out = NULL;     /* zdepth[P] min_sdf[P] depth[P] normal[3P] g_latent[256] g_R[9] g_T[3] */
  uint8_t* d_mask = NULL;
  const size_t nout = 6 * P + 268;
  HIP(hipMalloc((void**)&d_out, sizeof(float) * nout)); HIP(hipMalloc((void**)&d_mask, P));
  if (!d_cam || !d_g) return 4;
  hipStream_t s = NULL;
  HIP(hipStreamCreate(&s));
  CHECK(render_forward(ctx, &cfg, d_cam, d_cam + 256, d_cam + 265, d_out, d_mask, d_out + P, d_out + 2 * P, d_out + 3 * P, ws, fwd, (void*)s));
  CHECK(render_backward(ctx, &cfg, ws, fwd, NULL, d_g + P, d_g, d_g + 2 * P, d_out + 6 * P, d_out + 6 * P + 256, d_out + 6 * P + 265, wsb, bwd, (void*)s));
  HIP(hipStreamSynchronize(s));
  float* out = (float*)malloc(sizeof(float) * nout);
  uint8_t* mask = (uint8_t*)malloc(P);
  HIP(hipMemcpy(out, d_out, sizeof(float) * nout, hipMemcpyDeviceToHost));
  HIP(hipMemcpy(mask, d_mask, P, hipMemcpyDeviceToHost));
  f = fopen(argv[3], "wb");
  if (!f || fwrite(out, sizeof(float), nout, f) != nout || fwrite(mask, 1, P, f) != P) return 1;
  fclose(f);
  destroy(ctx);
  printf("rendered %dx%d\n", cfg.H, cfg.W);
  return 0;
}
