// mz_inst.hip -- one translation unit per search-kernel shape: compiled with -DMZ_INST_F=KS1,JTP,G (k_search_fused) or
// -DMZ_INST_H=G (k_search_h2), it defines that shape's instantiations (mz_kernels.inc); mz_engine.hip launches them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mz_engine.h"
#define MZ_MAX_ACTIONS_K MZ_MAX_ACTIONS
#include "mz_common.h"
#include "mz_net.hip.h"
#include "mz_rng.h"
#include "mz_tree.hip.h"
#include "mz_selfplay.hip.h"
#include "mz_fused.hip.h"
#include "mz_root.hip.h"
#include "mz_fused_h2.hip.h"
#include "mz_kernels.inc"

#if defined(MZ_INST_F)
MZ_APPLY(MZ_INST_FUSED, , MZ_INST_F)
#ifdef MZ_INST_GAME      // (the <15, 1, 16> unit) whole moves of the device TicTacToe environment
template __global__ void k_search_fused<15, 1, 16, 2, false, false, true, true> MZ_KARGS;
#endif
#elif defined(MZ_INST_H)
MZ_APPLY(MZ_INST_H2, , MZ_INST_H)
#else
#error "compile with -DMZ_INST_F=KS1,JTP,G or -DMZ_INST_H=G"
#endif
