// hg_device.h -- device-side data layout of libhg_sim (gfx950 only).
//
// One wavefront (64 lanes) simulates one env.  The robot description is a
// single read-only DevModel in HBM (~75 KB, L2 resident); per-env state
// is an env-major float record (include/hg_model.h, HG_STATE_SIZE; padded in the library's own
// buffer to a whole number of 64-byte lines) so that the 64 lanes of the env's wave read and write
// it as one contiguous, line-aligned burst.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hg_model.h"

#ifdef HG_PHASE_TIMING      // diagnostics build (tools/phase_ticks.py)
#define HG_TICK_ENVS 4096
#define HG_TICK_SLOTS 24
#endif

namespace hg {

constexpr int WAVE  = 64;
#define HG_XCDS 8           // accelerator complex dies of an MI355X: workgroups are dealt to them round-robin
constexpr int DMAX  = 12;   // deepest supported link (base children are depth 1)
constexpr int NSLOT = 2;    // saved-acceleration slots = deepest nesting of branching links (slots are recycled)
constexpr int LINKC = 32;   // floats per link in DevModel::link
// floats per link in the LDS articulated-body cache: 22 used, records 16-byte aligned.  28 (stride of 8 distinct LDS
// banks for the lane = link accesses) unless the model is so large that LDS bounds the occupancy: 24 lets a Romeo with
// hands (61 links) run 8 instead of 6 waves per CU (+10 %) and a NAO with fingers (42 links) 12 instead of 11 -- what its
// registers admit (+4 %); on Pepper 24 is 1.4 % slower than 28, on the 26-dof NAO 0.7 % faster.
__host__ __device__ constexpr int cache_stride(int nmx) { return nmx > 40 ? 24 : 28; }
constexpr int PUB   = 20;   // floats per body in the LDS kinematics exchange
constexpr int UP    = 28;   // floats per link in the LDS child->parent exchange
constexpr int CON   = 12;   // floats per contact record
constexpr int SCEN  = 8;    // floats per link-link primitive in LDS: world centre, radius, world half axis, pad
#define HG_DEV_SPHERES 2048 // device slots for HG_MAX_SPHERES primitives + padding
#define HG_DEV_SELF    256  // primitives that appear in link-link pairs (8-bit indices in pair_meta)
// the cache record of a link (world-aligned axes, origin at the link's COM; hg_step.hip, phase 1)
constexpr int C_RW = 0, C_DINV = 3, C_UW = 4, C_SW = 10, C_XW = 16, C_U = 19, C_MAXVEL = 20, C_MAXIMP = 21;

// DevModel::link[i][...]
enum { L_R0 = 0, L_AXIS = 9, L_S = 12, L_D = 15, L_E = 18, L_MASS = 21, L_INERTIA = 22,
       L_DAMP = 25, L_LOWER = 26, L_UPPER = 27, L_MAXIMP = 28, L_MAXVEL = 29, L_QINIT = 30, L_LIMITED = 31 };
// DevModel::base[...]
enum { B_MASS = 0, B_INERTIA = 1, B_POS0 = 4, B_QUAT0 = 7, B_FPOS = 11, B_FQUAT = 14 };

struct DevModel {
    int32_t n, n_spheres, n_pairs, n_act, maxlev, n_base_children, n_slots, n_self;
    int8_t  parent[HG_MAX_LINKS];
    int8_t  level[HG_MAX_LINKS];        // depth, base children = 1
    int8_t  first_child[HG_MAX_LINKS];
    int8_t  next_sibling[HG_MAX_LINKS];
    int8_t  par_mode[HG_MAX_LINKS];     // 0: parent is link i-1, 1: base, 2+s: saved slot s
    int8_t  save_slot[HG_MAX_LINKS];    // slot to keep this link's acceleration in, or -1
    int8_t  act_joint[HG_MAX_LINKS];
    int8_t  base_children[HG_MAX_LINKS];
    int8_t  path[HG_MAX_LINKS][DMAX];   // path[b][d-1]: ancestor-or-self of b at depth d, or -1
    // packed copies of the tables above, so that the kernel needs one load per lane instead
    // of one dependent load per tree level / per link:
    uint32_t linkinfo[HG_MAX_LINKS];    // par_mode | level << 8 | (save_slot + 1) << 16 | (parent + 1) << 24
    uint32_t kids[HG_MAX_LINKS];        // (first_child + 1) | (next_sibling + 1) << 8
    int32_t  lev_start[DMAX + 4];       // links of level l: lev_order[lev_start[l] .. lev_start[l + 1])
    int8_t   lev_order[HG_MAX_LINKS];   // link indices sorted by level
    uint32_t pathw[HG_MAX_LINKS][4];    // byte d-1 of the 16: path[b][d-1] + 1 (0 = none)
    // collision primitives in DEVICE order: the model's order, with padding slots (meta 0) inserted so that the primitives
    // of one body never straddle a trip of 64 slots (the wave ranks a body's ground points inside one 64-float LDS window)
    uint32_t sphere_meta[HG_DEV_SPHERES]; // (link + 1) | flags << 8 (bit 8 ground, 9 link-link, 15 appears in a pair) | body's first slot in its trip << 16 | one past its last << 24
    uint32_t pair_meta[HG_MAX_PAIRS];     // self index a | self index b << 8 | (link a + 1) << 16 | (link b + 1) << 24
    uint8_t  self_idx[HG_DEV_SPHERES];    // compact index of a primitive among those that take part in link-link pairs
    unsigned long long anc_mask[HG_MAX_LINKS];  // bit j: j is b or an ancestor of b
    float   link[HG_MAX_LINKS][LINKC];
    float   base[32];
    float   sphere[HG_DEV_SPHERES][4];  // x y z radius, internal frame of the body
    float   self_half[HG_DEV_SELF][4];  // half axis of the sphere-swept segments (by self index), internal frame; 0 = sphere
    float   frame_pos[HG_MAX_LINKS + 1][4];
    float   frame_quat[HG_MAX_LINKS + 1][4];
};

enum { F_GROUND = 1, F_SELF = 2, F_CLAMP = 4, F_ACTIONS = 8, F_BASECMD = 16, F_LIMITS = 32, F_PRIOFLIP = 64 };

struct StepParams {
    int32_t n_envs, n_substeps, iters, maxc, n_sel, flags;
    int32_t flip_from;    // envs >= flip_from flip their wave priority at solver entry (F_PRIOFLIP)
    int32_t state_stride; // floats between two envs' state records (HG_STATE_SIZE rounded up to a 64-byte line)
    // model sizes, repeated here so that they arrive with the kernel arguments (no dependent load from DevModel)
    int32_t n, n_spheres, n_pairs, n_act, maxlev, n_base_children, n_self;
    int8_t  lev_start[DMAX + 4];
    float   dt, inv_dt, gx, gy, gz, kp, kd, erp, mu, margin, lin_damp, ang_damp, vmax, limhi;
    int8_t  sel[HG_MAX_LINKS];
};

struct StepOut {
    float   *obs_q, *obs_qd, *base_xyyaw, *reward;
    uint8_t *done, *collision;
    unsigned long long *link_mask;   // internal, always written
    int32_t *pairs;                  // internal [n_envs][HG_MAX_CONTACTS][2]
    int32_t *counts;                 // internal [n_envs]
    const float *base_cmd;           // internal [n_envs][3] commanded base (vx, vy, yaw rate), used with F_BASECMD
    float   *cum;                    // internal [n_envs][4]: running {steps, reward, done, collision} since the last flush
    float   *torque;                 // internal [n_envs][n]: motor impulse / dt of the last physics step
};

// LDS carve-up (offsets in floats) for a model with n links / ns spheres.
// vel/F and lam/misc share storage (never live together); region AB is time-shared, see hg_step.hip.
struct LdsPlan {
    int cache, base, vel, F, lam, misc, con, Yd2, sdist, scen, tab, ys, AB, acc, total;
};
// With self-collision two more arrays exist, never live together, so they share one region: the world centres
// (+ radius, half axis) of the n_self primitives that appear in pairs (contact detection), and the second-path Y values of
// two-body rows, kept for the contact-row lanes only (maxlev x nc3, phase 4a).
__host__ __device__ inline int rup4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int imax(int a, int b) { return a > b ? a : b; }
// Variants whose impulse-response pass keeps the own-path `Y` values (one per depth and lane) in LDS instead of
// registers: DEPTH * WAVE floats laid over an array that is dead by then (the sphere distances).  For the shapes whose A-rows leave no registers for them: 12 contact slots (hipcc
// spilled them to scratch) and two rows per lane (NAO + fingers 6 % faster, Romeo + hands 4.5 %).  Measured on the
// others too: NAO and Pepper within 1 %, Romeo 10 % slower (the extra 1.3 KB of LDS costs it a wave per SIMD).
// Rows beyond the 64 lanes of a wave.  Up to HG_MAX_OVF of them (the friction pairs of the LAST contact slots) are
// "overflow rows": they have no lane; their candidate impulse is formed from a wave reduction once per sweep (hg_step.hip,
// 4b "overflow rows").  More rows than that: two rows per lane (DUAL).
#define HG_MAX_OVF 2
__host__ __device__ constexpr int ovf_rows(int nmx, int maxc) {
    return (nmx + 3 * maxc > 64 && nmx + 3 * maxc - 64 <= HG_MAX_OVF && (nmx + 3 * maxc) % 2 == 0) ? nmx + 3 * maxc - 64 : 0;
}
#ifndef HG_YS_LDS
#define HG_YS_LDS(nmx, maxc, depth) ((maxc) >= 12 || ((nmx) + 3 * (maxc) > 64 && ovf_rows(nmx, maxc) == 0))
#endif
__host__ __device__ constexpr int ys_floats(int nmx, int maxc, int depth) { return HG_YS_LDS(nmx, maxc, depth) ? depth * WAVE : 0; }
__host__ __device__ inline LdsPlan lds_plan(int n, int ns, int nc3, int maxlev, int n_self, int nrows, int ys, int cstride) {
    LdsPlan p;
    int o = 0;
    const int ndof = rup4(6 + n);
    p.cache = o; o += cstride * n;
    p.base = o;  o += 64;
    p.vel = o;   p.F = o;    o += imax(ndof, rup4(n));
    p.lam = o;   p.misc = o; o += nrows > WAVE ? WAVE + rup4(nc3) : WAVE;      // impulses of all rows (two rows per lane beyond 64)
    p.con = o;   o += CON * (nc3 / 3);
    p.Yd2 = o;   p.scen = o; o += n_self > 0 ? rup4(imax(maxlev * nc3, SCEN * n_self)) : 0;
    p.tab = o;   o += rup4(n);              // byte tables (4 n bytes): next_sibling, base_children, first_child, lev_order
    // the point - plane distances of one trip of 64 primitives (contact detection); dead once the contacts are found: the
    // Y values of the response pass are laid over them
    p.ys = o;
    p.sdist = o; o += ns > 0 ? WAVE : 0;
    o = imax(o, p.ys + ys);
    p.AB = o;    o += rup4(imax(imax(PUB * (n + 1), UP * n), imax((6 + n) * nc3, 16 * n + 8)));
    p.acc = o;   o += 8 * (n + 1);
    p.total = o;
    return p;
}

}  // namespace hg
