// hg_step_launch.h -- defines launch_step_v<V> (and, in the diagnostics build, read_ticks_v<V>) for the step kernel.
// Included by hg_sim.hip in a single-TU build (all variants) and by hg_step_variant.hip (one variant per TU).
#pragma once
#include "hg_variants.h"
#include "hg_step.hip"

namespace hg {

#ifdef HG_VI
#define HG_LAUNCH_TEMPLATE template <>
#define HG_LAUNCH_V HG_VI
#else
#define HG_LAUNCH_TEMPLATE template <int V>
#define HG_LAUNCH_V V
#endif

HG_LAUNCH_TEMPLATE
hipError_t launch_step_v<HG_LAUNCH_V>(int n_envs, int lds_bytes, hipStream_t st, const DevModel *M, const StepParams &P, float *state,
                                      const float *actions, const StepOut &O) {
    constexpr Variant v = kVariants[HG_LAUNCH_V];
    hipLaunchKernelGGL((hg_step_kernel<v.nmx, v.maxc, v.depth, v.nsl, v.wpe>), dim3(n_envs), dim3(WAVE), lds_bytes, st,
                       M, P, state, actions, O);
    return hipGetLastError();
}
#ifdef HG_PHASE_TIMING
HG_LAUNCH_TEMPLATE
int read_ticks_v<HG_LAUNCH_V>(unsigned long long *host_out, int n_envs) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(hg_ticks), sizeof(unsigned long long) * HG_TICK_SLOTS * (size_t)n_envs) != hipSuccess) return -1;
    return HG_TICK_SLOTS;
}
#endif

}  // namespace hg
