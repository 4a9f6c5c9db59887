// distr_inst.hip -- one group of explicit kernel instantiations (distr_inst.hpp); compiled once per group with -DDISTR_INST_GROUP=<n>
// by distr.binding.build_library, in parallel with distr_api.hip.
#ifndef DISTR_INST_GROUP
#error "compile with -DDISTR_INST_GROUP=<1..6> (distr.binding.build_library does)"
#endif
#include <hip/hip_runtime.h>
#define DISTR_GLOBAL static __global__
#include "distr_inst.hpp"
