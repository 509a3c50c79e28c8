// hg_variants.h -- the instantiations of the step kernel (hg_step.hip), shared by the host side (hg_sim.hip), the
// per-variant translation units (hg_step_variant.hip) and the build script (humanoid_gym_amd/build.py reads the X rows).
#pragma once
#include <hip/hip_runtime.h>
#include "hg_device.h"

namespace hg {

// kernel instantiations: motor lanes, contact slots, tree depth, branch slots, waves per SIMD
struct Variant { int nmx, maxc, depth, nsl, wpe; };
#define HG_VARIANTS(X) \
    X(0, 18, 12, 6, 0, 4)  /* dancer (scratch-free at 4 waves/SIMD since round 2; 3 waves/SIMD is 18 % slower) */ \
    X(1, 26, 8, 6, 0, 4)   /* nao with the default 8 contact slots     */ \
    X(2, 28, 8, 6, 0, 4)   /* up to 28 dofs, shallow tree              */ \
    X(3, 28, 12, 6, 0, 3)  /* nao, 12 contact slots (64 rows: 632 B of scratch per lane at 4 waves/SIMD) */ \
    X(4, 20, 8, 9, 2, 4)   /* pepper (deep chain, branching torso)     */ \
    X(5, 20, 8, 12, 2, 4)  /* 20 dofs, deeper chains                   */ \
    X(6, 28, 12, 12, 2, 3) /* deep trees up to 28 dofs, 12 contacts    */ \
    X(7, 37, 8, 12, 2, 3)  /* romeo: 61 rows, 3 lanes left for limits  */ \
    X(8, 40, 8, 12, 2, 3)  /* up to 40 dofs                            */ \
    X(9, 46, 6, 12, 2, 3)  /* anything up to 46 dofs, 6 contacts       */ \
    X(10, 42, 8, 12, 2, 3) /* nao with finger joints (42 dofs): 66 rows = one row per lane + 2 overflow rows */ \
    X(11, 44, 8, 12, 2, 2) /* up to 44 dofs: 68 rows, two rows per lane */ \
    X(12, 62, 8, 12, 2, 2) /* romeo with hands, anything up to 62 dofs: 86 rows (1 wave/SIMD = no scratch, but 1.5x slower) */ \
    X(13, 62, 12, 12, 2, 2) /* catch-all: every model within the limits of hg_model.h has a kernel */
#define HG_VARIANT_ROW(i, a, b, c, d, e) {a, b, c, d, e},
constexpr Variant kVariants[] = {HG_VARIANTS(HG_VARIANT_ROW)};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

// One launch function per variant.  Split build (-DHG_SPLIT_TU, what build.py does): each is defined in its own
// translation unit (hg_step_variant.hip -DHG_VI=<index>), compiled in parallel.  Single-TU build (scripts that compile
// hg_sim.hip alone): hg_sim.hip includes hg_step.hip and the generic definition below is used.
template <int V>
hipError_t launch_step_v(int n_envs, int lds_bytes, hipStream_t st, const DevModel *M, const StepParams &P, float *state,
                         const float *actions, const StepOut &O);
#ifdef HG_PHASE_TIMING
template <int V> int read_ticks_v(unsigned long long *host_out, int n_envs);      // diagnostics build only
#endif

}  // namespace hg
