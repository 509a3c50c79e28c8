// hg_step_variant.hip -- one instantiation of the step kernel per translation unit: hipcc -DHG_SPLIT_TU -DHG_VI=<index>
// (row <index> of HG_VARIANTS, hg_variants.h).  build.py compiles these in parallel and links them with hg_sim.hip.
#ifndef HG_VI
#error "compile with -DHG_VI=<variant index>"
#endif
#include "hg_step_launch.h"
