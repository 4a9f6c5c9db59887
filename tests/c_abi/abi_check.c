/* Plain-C consumer of include/distr.h: proves the boundary is a C ABI (no C++ / torch types). Links nothing at build
 * time: the library is opened with dlopen, every entry point the header declares is resolved, and the calls that need no
 * GPU are exercised (distr_version, workspace-size helpers, distr_create on a machine without a device -> error string).
 * Build + run: see tests/test_host_logic.py::test_c_abi_from_plain_c. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include "distr.h"

#define RESOLVE(name) do { void* p_ = dlsym(h, #name); if (!p_) { fprintf(stderr, "missing symbol %s\n", #name); return 2; } ++n; } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
  int n = 0;
  RESOLVE(distr_create); RESOLVE(distr_destroy); RESOLVE(distr_last_error); RESOLVE(distr_version); RESOLVE(distr_set_decoder);
  RESOLVE(distr_workspace_bytes); RESOLVE(distr_render_forward); RESOLVE(distr_render_backward); RESOLVE(distr_render_normal);
  RESOLVE(distr_mlp_workspace_bytes); RESOLVE(distr_mlp_eval); RESOLVE(distr_mlp_grad); RESOLVE(distr_mlp_backward);
  RESOLVE(distr_mlp_backward_workspace_bytes); RESOLVE(distr_get_render_stats); RESOLVE(distr_profile_enable); RESOLVE(distr_profile_read); RESOLVE(distr_profile_read_list); RESOLVE(distr_get_live_counts);
  RESOLVE(distr_loss_workspace_bytes); RESOLVE(distr_single_loss_forward); RESOLVE(distr_single_loss_backward);
  RESOLVE(distr_warp_loss_forward); RESOLVE(distr_warp_loss_backward); RESOLVE(distr_set_color_decoder); RESOLVE(distr_color_eval); RESOLVE(distr_color_backward);
  RESOLVE(distr_debug_mlp_layer); RESOLVE(distr_debug_tile_timing); RESOLVE(distr_debug_xchg_ts);
  const char* (*version)(void);
  size_t (*mlp_ws)(int64_t);
  size_t (*loss_ws)(int32_t, int32_t);
  int (*create)(distr_ctx**, int);
  const char* (*last_error)(const distr_ctx*);
  void (*destroy)(distr_ctx*);
  *(void**)(&version) = dlsym(h, "distr_version");          /* POSIX idiom for object -> function pointer */
  *(void**)(&mlp_ws) = dlsym(h, "distr_mlp_workspace_bytes");
  *(void**)(&loss_ws) = dlsym(h, "distr_loss_workspace_bytes");
  *(void**)(&create) = dlsym(h, "distr_create");
  *(void**)(&last_error) = dlsym(h, "distr_last_error");
  *(void**)(&destroy) = dlsym(h, "distr_destroy");
  distr_render_cfg cfg;
  memset(&cfg, 0, sizeof(cfg));
  distr_ctx* ctx = NULL;
  const int rc = create(&ctx, 0);
  printf("symbols=%d version=\"%s\" sizeof(cfg)=%zu mlp_ws=%zu loss_ws=%zu create_rc=%d err=\"%s\"\n", n, version(), sizeof(cfg),
         mlp_ws(1000), loss_ws(64, 64), rc, rc ? last_error(ctx) : "");
  if (ctx) destroy(ctx);
  dlclose(h);
  return 0;
}
