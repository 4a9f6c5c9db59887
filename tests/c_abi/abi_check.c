/* Plain-C consumer of include/distr.h: proves the boundary is a C ABI (no C++ / torch types). Links nothing at build
 * time: the library is opened with dlopen, every entry point the header declares is resolved, and the calls that need no
 * GPU are exercised (distr_version, workspace-size helpers, distr_create on a machine without a device -> error string).
 * Build + run: see tests/test_host_logic.py::test_c_abi_from_plain_c. */
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include "distr.h"

#define RESOLVE(name) do { void* p_ = dlsym(h, #name); if (!p_) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } ++n; } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  int n = 0;
  RESOLVE(distr_create_abi); RESOLVE(distr_abi_version); RESOLVE(distr_destroy); RESOLVE(distr_last_error); RESOLVE(distr_version); RESOLVE(distr_set_decoder);
  RESOLVE(distr_workspace_bytes); RESOLVE(distr_render_forward); RESOLVE(distr_render_backward); RESOLVE(distr_render_normal);
  RESOLVE(distr_mlp_workspace_bytes); RESOLVE(distr_mlp_eval); RESOLVE(distr_mlp_grad); RESOLVE(distr_mlp_backward);
  RESOLVE(distr_mlp_backward_workspace_bytes); RESOLVE(distr_get_render_stats); RESOLVE(distr_profile_enable); RESOLVE(distr_profile_read); RESOLVE(distr_profile_read_list); RESOLVE(distr_get_live_counts);
  RESOLVE(distr_loss_workspace_bytes); RESOLVE(distr_single_loss_forward); RESOLVE(distr_single_loss_backward);
  RESOLVE(distr_warp_loss_forward); RESOLVE(distr_warp_loss_backward); RESOLVE(distr_set_color_decoder); RESOLVE(distr_color_eval); RESOLVE(distr_color_backward);
  RESOLVE(distr_debug_mlp_layer); RESOLVE(distr_debug_tile_timing); RESOLVE(distr_debug_xchg_ts);
  RESOLVE(distr_render_forward_batch); RESOLVE(distr_render_backward_batch); RESOLVE(distr_render_normal_batch);
  RESOLVE(distr_mlp_eval_bf16x6);
  RESOLVE(distr_mlp_eval_f16x3);
  const char* (*version)(void);
  size_t (*mlp_ws)(int64_t);
  size_t (*loss_ws)(int32_t, int32_t);
  int (*create)(distr_ctx**, int, uint32_t);
  uint32_t (*abi)(void);
  int (*ws_bytes)(distr_ctx*, const distr_render_cfg*, size_t*, size_t*);
  const char* (*last_error)(const distr_ctx*);
  void (*destroy)(distr_ctx*);
  *(void**)(&version) = dlsym(h, "distr_version");          /* POSIX idiom for object -> function pointer */
  *(void**)(&mlp_ws) = dlsym(h, "distr_mlp_workspace_bytes");
  *(void**)(&loss_ws) = dlsym(h, "distr_loss_workspace_bytes");
  *(void**)(&create) = dlsym(h, "distr_create_abi");
  *(void**)(&abi) = dlsym(h, "distr_abi_version");
  *(void**)(&ws_bytes) = dlsym(h, "distr_workspace_bytes");
  *(void**)(&last_error) = dlsym(h, "distr_last_error");
  *(void**)(&destroy) = dlsym(h, "distr_destroy");
  distr_render_cfg cfg;
  distr_ctx* ctx = NULL;
  /* ABI handshake: a caller built for another version is refused with a message; so is a struct whose size was never announced */
  if (abi() != DISTR_ABI_VERSION) { fprintf(stderr, "library ABI %u != header ABI %u\n", abi(), DISTR_ABI_VERSION); return 3; }
  if (create(&ctx, 0, DISTR_ABI_VERSION + 1u) != DISTR_ERR_INVALID_ARG || !ctx || !strstr(last_error(ctx), "ABI")) { fprintf(stderr, "wrong ABI version accepted\n"); return 3; }
  destroy(ctx);
  ctx = NULL;
  const int rc = create(&ctx, 0, DISTR_ABI_VERSION);          /* what the distr_create macro expands to */
  char create_err[512];
  snprintf(create_err, sizeof(create_err), "%s", rc ? last_error(ctx) : "");
  {
    size_t fwd = 0, bwd = 0;
    memset(&cfg, 0, sizeof(cfg));                             /* struct_size never set */
    cfg.H = cfg.W = 64; cfg.march_step = 20; cfg.buffer_size = 3; cfg.radius = 1.f; cfg.marcher = DISTR_MARCH_RECURSIVE;
    if (!ctx || ws_bytes(ctx, &cfg, &fwd, &bwd) != DISTR_ERR_INVALID_ARG || !strstr(last_error(ctx), "struct_size")) { fprintf(stderr, "unsized cfg accepted\n"); return 3; }
    DISTR_INIT(cfg);
    cfg.H = cfg.W = 64; cfg.march_step = 20; cfg.buffer_size = 3; cfg.radius = 1.f; cfg.marcher = DISTR_MARCH_RECURSIVE;
    if (ws_bytes(ctx, &cfg, &fwd, &bwd) != DISTR_OK || fwd == 0) { fprintf(stderr, "sized cfg refused: %s\n", last_error(ctx)); return 3; }
    printf("handshake ok fwd_ws=%zu\n", fwd);
  }
  printf("symbols=%d version=\"%s\" sizeof(cfg)=%zu mlp_ws=%zu loss_ws=%zu create_rc=%d err=\"%s\"\n", n, version(), sizeof(cfg),
         mlp_ws(1000), loss_ws(64, 64), rc, create_err);
  /* struct layouts, field by field, for the ctypes mirror to be compared with (tests/test_host_logic.py) */
#define OFF(T, f) printf("offset %s.%s %zu %zu\n", #T, #f, offsetof(T, f), sizeof(((T*)0)->f))
  OFF(distr_render_cfg, struct_size); OFF(distr_render_cfg, H); OFF(distr_render_cfg, W); OFF(distr_render_cfg, K_inv); OFF(distr_render_cfg, fx); OFF(distr_render_cfg, fy);
  OFF(distr_render_cfg, M); OFF(distr_render_cfg, M_normal); OFF(distr_render_cfg, march_step); OFF(distr_render_cfg, buffer_size); OFF(distr_render_cfg, ratio);
  OFF(distr_render_cfg, threshold); OFF(distr_render_cfg, radius); OFF(distr_render_cfg, clamp_dist); OFF(distr_render_cfg, marcher);
  OFF(distr_render_cfg, coarse_steps); OFF(distr_render_cfg, use_depth2normal); OFF(distr_render_cfg, normalize_normal);
  OFF(distr_render_cfg, want_normal); OFF(distr_render_cfg, grad_depth); OFF(distr_render_cfg, grad_mask); OFF(distr_render_cfg, grad_camera);
  OFF(distr_render_cfg, save_for_backward); OFF(distr_render_cfg, row0); OFF(distr_render_cfg, rows); OFF(distr_render_cfg, arith); OFF(distr_render_cfg, concurrent);
  OFF(distr_render_cfg, num_levels); OFF(distr_render_cfg, level_scale); OFF(distr_render_cfg, level_steps);
  OFF(distr_decoder_desc, struct_size); OFF(distr_decoder_desc, latent_size); OFF(distr_decoder_desc, hidden); OFF(distr_decoder_desc, num_linear); OFF(distr_decoder_desc, latent_in);
  OFF(distr_render_stats, struct_size); OFF(distr_render_stats, reserved); OFF(distr_render_stats, num_in_sphere); OFF(distr_render_stats, num_march_launches); OFF(distr_render_stats, num_point_evals);
  OFF(distr_render_stats, num_valid); OFF(distr_render_stats, num_grad_samples); OFF(distr_render_stats, cluster_fallbacks); OFF(distr_render_stats, f16_overflows);
  OFF(distr_render_stats, tail_from); OFF(distr_render_stats, tail_steals);
  OFF(distr_warp_cfg, struct_size); OFF(distr_warp_cfg, H); OFF(distr_warp_cfg, W); OFF(distr_warp_cfg, K); OFF(distr_warp_cfg, K_inv); OFF(distr_warp_cfg, thres_depth);
  printf("sizeof distr_decoder_desc %zu\nsizeof distr_render_stats %zu\nsizeof distr_warp_cfg %zu\n", sizeof(distr_decoder_desc),
         sizeof(distr_render_stats), sizeof(distr_warp_cfg));
  if (ctx) destroy(ctx);
  dlclose(h);
  return 0;
}
