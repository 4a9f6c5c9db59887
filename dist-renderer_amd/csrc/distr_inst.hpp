// distr_inst.hpp -- the big template kernels as explicit instantiations, one GROUP per translation unit, so that libdistr.so builds in
// parallel (distr.binding.build_library compiles distr_api.hip and the distr_inst_<group>.hip files side by side and links them: ~1.5 min
// instead of ~4.5 on 8 cores). distr_api.hip (DISTR_INST_GROUP undefined) sees every instantiation as `extern template`: it launches the
// kernels, the group's translation unit holds their code. A kernel missing from the lists below still works -- distr_api.hip then
// instantiates it itself, only slower to build.
#pragma once
#include "distr_kernels.hpp"

namespace distr {

#ifdef DISTR_INST_GROUP
#define DISTR_INST_DEF template
#else
#define DISTR_INST_DEF extern template
#endif

// in a group's translation unit only that group is instantiated; the launching unit declares all of them extern
#if defined(DISTR_INST_GROUP)
#define DISTR_GROUP_ON(g) (DISTR_INST_GROUP == (g))
#else
#define DISTR_GROUP_ON(g) 1
#endif

#define DISTR_K_STEP(K, AR) DISTR_INST_DEF __global__ void k_step<K, AR>(MarchArgs, DecoderDev, DecoderDev16, StepGrid);
#define DISTR_K_TAIL(K) DISTR_INST_DEF __global__ void k_tail<K>(MarchArgs, DecoderDev, DecoderDev16);
#define DISTR_K_MARCH(M, RB, K, AR) DISTR_INST_DEF __global__ void k_march<M, RB, K, AR>(MarchArgs, DecoderDev);
#define DISTR_K_MARCH16(M, K) DISTR_INST_DEF __global__ void k_march16<M, K>(MarchArgs, DecoderDev, DecoderDev16);
#define DISTR_K_BWD(M, RB, AR) DISTR_INST_DEF __global__ void k_bwd<M, RB, AR>(BwdArgs, DecoderDev);

#if DISTR_GROUP_ON(1)      // the full-resolution step of the exact-f32 march (the headline kernel)
DISTR_K_STEP(true, 0) DISTR_K_STEP(false, 0)
#endif
#if DISTR_GROUP_ON(2)      // the persistent tail launch
DISTR_K_TAIL(true) DISTR_K_TAIL(false)
#endif
#if DISTR_GROUP_ON(3)      // exact-f32 march tiles: coarse levels, 'trivial', point lists; 16-ray / cluster tiles
DISTR_K_MARCH(MODE_EVAL, 2, false, 0)
DISTR_K_MARCH(MODE_COARSE, 1, true, 0) DISTR_K_MARCH(MODE_COARSE, 1, false, 0) DISTR_K_MARCH(MODE_COARSE, 2, true, 0) DISTR_K_MARCH(MODE_COARSE, 2, false, 0)
DISTR_K_MARCH(MODE_FINE, 2, true, 0) DISTR_K_MARCH(MODE_FINE, 2, false, 0)
DISTR_K_MARCH16(MODE_EVAL, false) DISTR_K_MARCH16(MODE_COARSE, true) DISTR_K_MARCH16(MODE_COARSE, false)
#endif
#if DISTR_GROUP_ON(4)      // the step kernel in the two opt-in arithmetics
DISTR_K_STEP(true, 1) DISTR_K_STEP(false, 1) DISTR_K_STEP(true, 2) DISTR_K_STEP(false, 2)
#endif
#if DISTR_GROUP_ON(5)      // march tiles in the two opt-in arithmetics
DISTR_K_MARCH(MODE_COARSE, 1, true, 1) DISTR_K_MARCH(MODE_COARSE, 1, false, 1) DISTR_K_MARCH(MODE_COARSE, 2, true, 1) DISTR_K_MARCH(MODE_COARSE, 2, false, 1)
DISTR_K_MARCH(MODE_FINE, 2, true, 1) DISTR_K_MARCH(MODE_FINE, 2, false, 1)
DISTR_K_MARCH(MODE_COARSE, 1, true, 2) DISTR_K_MARCH(MODE_COARSE, 1, false, 2) DISTR_K_MARCH(MODE_COARSE, 2, true, 2) DISTR_K_MARCH(MODE_COARSE, 2, false, 2)
DISTR_K_MARCH(MODE_FINE, 2, true, 2) DISTR_K_MARCH(MODE_FINE, 2, false, 2)
#endif
#if DISTR_GROUP_ON(6)      // backward kernels
DISTR_K_BWD(BWD_FULL, 2, 0) DISTR_K_BWD(BWD_POINTGRAD, 2, 0)
DISTR_K_BWD(BWD_SAVED, 1, 0) DISTR_K_BWD(BWD_SAVED, 1, 1) DISTR_K_BWD(BWD_SAVED, 1, 2)
DISTR_K_BWD(BWD_SAVED, 2, 0) DISTR_K_BWD(BWD_SAVED, 2, 1) DISTR_K_BWD(BWD_SAVED, 2, 2)
#endif

constexpr int DISTR_NUM_INST_GROUPS = 6;

}  // namespace distr
