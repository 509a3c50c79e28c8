// hg_step.hip -- the fused env-step kernel: one 64-lane wavefront per env.
//
// Replaces, for every env at once, the PyBullet calls of
//   /root/reference/humanoid_gym/envs/dancer_env.py:19-35
//   /root/reference/humanoid_gym/envs/nao_env.py:70-73 (and pepper/romeo)
// with one launch: targets -> n_substeps x { kinematics, contact detection,
// articulated-body algorithm, motor + contact rows, projected Gauss-Seidel,
// integration } -> observations and contact flags.
//
// Every per-link quantity (velocity, inertia, bias force, U, S, acceleration, impulse response) is expressed in axes
// parallel to the world's with the origin at the link's own COM: parent <-> child transforms are translations.
// Lane roles change per phase (FP32 VALU throughout, no MFMA: the operands are
// 3x3 / 6x6 blocks of a tree recursion, not GEMM tiles):
//   * kinematics and the root -> leaves ABA pass: lane i owns link i; the tree is swept level by
//     level, parent <-> child data goes through LDS;
//   * the leaves -> root ABA pass: 7 lanes per link, one per column of X^T Ia X plus one for the
//     bias force;
//   * contact detection: trips of 64 collision primitives, lane = primitive (hull points against the plane ranked per body in
//     a 64-float LDS window; pairs of sphere-swept segments between links);
//   * constraint rows: lane r owns row r (motor rows first, then contact
//     normals, then friction pairs).  Each lane computes M^-1 J_r^T with its own
//     walk over the tree (path to the root, then a full descent reading the
//     broadcast ABA cache in LDS) and accumulates ITS ROW of the Delassus matrix
//     A = J M^-1 J^T in registers;
//   * PGS: the sweep over rows is unrolled; at row i every lane forms its own
//     candidate impulse, lane i's new impulse is broadcast with v_readlane (the
//     impulses live in scalar registers) and every lane j updates its candidate
//     with A[j][i].  This is the impulse-space form of Bullet's velocity-space
//     sweep (w = J dv), same iterates, no reductions; violated joint limits ride
//     along as extra rows on the wave's spare lanes.  Models with more than 64 rows (NAO / Romeo with
//     their finger joints): 65-66 rows keep one row per lane and run the friction pair of the last contact slot as
//     "overflow rows" from a wave reduction per sweep; beyond that every lane carries two rows -- motor row and contact
//     row.  Either way the impulses live on their lanes and the CHANGE of an impulse is broadcast (see "row layout"
//     in phase 4), the sweeps written as blocks of issue slots in assembly;
//   * the final M^-1 J^T lambda is one more level-by-level ABA response.
// Env order: XCD-aware (workgroup b runs on XCD b % 8; each XCD steps a contiguous block of envs).
// Scheduling: one wave per workgroup (a workgroup of several envs keeps its LDS until its slowest wave ends:
// measured slower); launches that fit the GPU at once reverse the SIMD arbiter's service order of their
// waves at solver entry with s_setprio (see 4b), so that the waves of a SIMD finish together.
#pragma once
#include "hg_device.h"
#include "hg_math.h"

namespace hg {

// Diagnostics build only (-DHG_PHASE_TIMING, tools/phase_ticks.py): lane 0 of the first envs writes the
// shader clock at every phase boundary.  The product library is built without it.
#ifdef HG_PHASE_TIMING
static __device__ unsigned long long hg_ticks[HG_TICK_ENVS][HG_TICK_SLOTS];
#define HG_TICK(K) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
        if (threadIdx.x == 0 && blockIdx.x < HG_TICK_ENVS) hg_ticks[blockIdx.x][K] = t_; } while (0)
// where the wave runs: HW_ID (wave, SIMD, CU, SH, SE) in slot 13, XCC_ID in slot 14
#define HG_TICK_HWID() do { unsigned h_, x_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(h_), "=s"(x_)); \
        if (threadIdx.x == 0 && blockIdx.x < HG_TICK_ENVS) { hg_ticks[blockIdx.x][13] = h_; hg_ticks[blockIdx.x][14] = x_; } } while (0)
#else
#define HG_TICK(K) do { } while (0)
#define HG_TICK_HWID() do { } while (0)
#endif

// The workgroup is ONE wavefront, whose LDS operations execute in program order, so lanes can
// hand data to each other through LDS without s_barrier (which would also drain vmcnt).  What is
// still needed is that hipcc does not move LDS accesses across the hand-off point.
// The lane index, recomputed where it is needed (two VALU instructions) instead of being kept in a
// register across the whole kernel: the volatile asm cannot be hoisted or merged, so nothing derived
// from one call site stays live until the next.
HG_FI int fresh_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// instruction-arbiter priority of this wave (s_setprio takes an immediate)
HG_FI void set_prio(const int p) {
    switch (p & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

HG_FI void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Y = Ys[d - 1] if link i (depth d) lies on the row's own path -- bit i of its ancestor mask, so the packed path
// words are dead once the walk is done -- else 0; one case per depth so that Ys is indexed by compile-time constants only.
#define HG_Y_CASE(K) case K + 1: if constexpr (K < DEPTH) { Y = on_path ? Ys[K < YSN ? K : 0] : 0.f; } break;
#define HG_Y_CASES HG_Y_CASE(0) HG_Y_CASE(1) HG_Y_CASE(2) HG_Y_CASE(3) HG_Y_CASE(4) HG_Y_CASE(5) \
                   HG_Y_CASE(6) HG_Y_CASE(7) HG_Y_CASE(8) HG_Y_CASE(9) HG_Y_CASE(10) HG_Y_CASE(11)

// One PGS row update of the lane-impulse form as assembly text (seven issue slots; see "4b, one row per lane plus overflow
// rows"): operands z, l (lane impulses), c, d (temporaries), s (SGPR), m (SGPR pair), lo, hi, a<K> (column), lane i0 SGN K.
#define HG_ROW_ASM(K, SGN) "v_med3_f32 %[c], %[z], %[lo], %[hi]\n\tv_sub_f32 %[d], %[c], %[l]\n\ts_lshl_b64 %[m], 1, %[i0]" SGN #K "\n\t" \
    "v_readlane_b32 %[s], %[d], %[i0]" SGN #K "\n\tv_cndmask_b32_e64 %[l], %[l], %[c], %[m]\n\ts_nop 0\n\tv_fmac_f32 %[z], %[s], %[a" #K "]\n\t"
// the same for two rows per lane (DUAL): the row lives in z / l (a motor row) and its change goes to both candidates
#define HG_DROW_ASM(K, SGN) "v_med3_f32 %[c], %[z], %[lo], %[hi]\n\tv_sub_f32 %[d], %[c], %[l]\n\ts_lshl_b64 %[m], 1, %[i0]" SGN #K "\n\t" \
    "v_readlane_b32 %[s], %[d], %[i0]" SGN #K "\n\tv_cndmask_b32_e64 %[l], %[l], %[c], %[m]\n\ts_nop 0\n\tv_fmac_f32 %[z], %[s], %[a" #K "]\n\t" \
    "v_fmac_f32 %[z1], %[s], %[b" #K "]\n\t"
#define HG_DROWS4_ASM(SGN) HG_DROW_ASM(0, SGN) HG_DROW_ASM(1, SGN) HG_DROW_ASM(2, SGN) HG_DROW_ASM(3, SGN)
#define HG_ROWS2_ASM(SGN) HG_ROW_ASM(0, SGN) HG_ROW_ASM(1, SGN)
#define HG_ROWS6_ASM(SGN) HG_ROW_ASM(0, SGN) HG_ROW_ASM(1, SGN) HG_ROW_ASM(2, SGN) HG_ROW_ASM(3, SGN) HG_ROW_ASM(4, SGN) HG_ROW_ASM(5, SGN)

// compile-time loop: f(IC<I>{}) for I = B .. E-1
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, typename F> HG_FI void static_for(F &&f) {
    if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}

// register vector of N floats (N <= 16): lets a wave-uniform run-time index address a VGPR directly
template <int N> struct RegVec { typedef float type __attribute__((ext_vector_type(N))); };
constexpr int clamp16(int x) { return x < 1 ? 1 : (x > 16 ? 16 : x); }

// What a lane knows about one constraint row it owns (set up in phase 4, consumed by the impulse response).
struct RowVars {
    int body, body2, col;                 // link the row acts on (-1 base), second body of a self-contact row (< -1: none), contact column
    float target_err, unit_f;             // desired change of J.v; the one Jacobian entry of a motor / limit row
    V3 Jang, Jlin;                        // base columns of a contact row
    unsigned long long myanc, myanc2;     // ancestor masks of body / body2
    unsigned pw[3], pw2[3];               // packed paths (ancestor-or-self per depth) of body / body2
    bool contact, motor, slot, two_body;
    HG_FI void clear() {
        body = -1; body2 = -3; col = -1; target_err = 0.f; unit_f = 0.f;
        Jang = v3(0, 0, 0); Jlin = v3(0, 0, 0); myanc = 0ull; myanc2 = 0ull;
        pw[0] = pw[1] = pw[2] = 0u; pw2[0] = pw2[1] = pw2[2] = 0u;
        contact = motor = slot = two_body = false;
    }
};

// NMX: motor-row lanes (>= n dofs), MAXC: contact slots, DEPTH: deepest link level handled,
// NSL: saved-acceleration slots for branching links, WPE: waves per SIMD the register budget targets.
// (A variant without the joint-limit rows for robots whose joints are all `continuous`, like Dancer, was measured:
// 1900 fewer static instructions, but hipcc then spills impulse SGPRs inside the sweep loop, 9 % slower.)
template <int NMX, int MAXC, int DEPTH, int NSL, int WPE>
__global__ __launch_bounds__(WAVE, WPE)
void hg_step_kernel(const DevModel *__restrict__ M, const StepParams P, float *__restrict__ state,
                    const float *__restrict__ actions, const StepOut O)
{
    constexpr int NC3 = 3 * MAXC;
    constexpr int NROWS = NMX + NC3;
    static_assert(NMX <= WAVE && NC3 <= WAVE, "motor rows and contact rows each fit one wavefront (two rows per lane beyond 64 rows)");
    static_assert((6 * NC3) % 4 == 0, "the joint part of Jc must start on a 16-byte boundary (even MAXC)");

    extern __shared__ __align__(16) float lds[];
    // XCD-aware env order.  Workgroups go to the 8 XCDs round-robin (workgroup b -> XCD b % 8), each XCD with its own L2:
    // with env = b the rows of the small per-env arrays (actions, running statistics, observations, flags: 4 ... 112
    // bytes) of eight consecutive envs would share 128-byte lines fetched by eight different L2s.  XCD x takes a
    // contiguous block of envs instead: q = n / 8 each, the first n % 8 XCDs one more.
    const int disp = blockIdx.x;           // dispatch order (the wave-priority schedule below goes by it)
    int env;
    {
        const int x = disp & (HG_XCDS - 1), q = P.n_envs / HG_XCDS, r = P.n_envs % HG_XCDS;
        env = x * q + (x < r ? x : r) + disp / HG_XCDS;
    }
    const int lane_o = threadIdx.x;
    const int n_o = P.n;
    const int lane = lane_o, n = n_o;
    const int ndof = 6 + n;
    const int ns = P.n_spheres;
    const int maxlev = P.maxlev;
    constexpr bool YSL = ys_floats(NMX, MAXC, DEPTH) > 0;      // own-path Y values of the response pass in LDS (hg_device.h)
    constexpr int CACHE = cache_stride(NMX);                   // floats per link in the ABA cache
    LdsPlan L = lds_plan(n, ns, NC3, maxlev, P.n_self, NROWS, ys_floats(NMX, MAXC, DEPTH), CACHE);
    // wave-uniform offsets: pin them to scalar registers (v_max3 and friends would otherwise
    // leave them in VGPRs that stay live for the whole kernel)
    L.AB = __builtin_amdgcn_readfirstlane(L.AB);
    L.acc = __builtin_amdgcn_readfirstlane(L.acc);
    static_assert(DEPTH <= DMAX, "path words hold DMAX levels");
    float *const cacheL = lds + L.cache;
    float *const baseL = lds + L.base;     // [0:9) Rw0  [9:12) xw0  [12:48) IA0^-1
    float *const velL = lds + L.vel;
    float *const FL = lds + L.F;
    float *const lamL = lds + L.lam;
    float *const miscL = lds + L.misc;
    float *const conL = lds + L.con;
    float *const sdistL = lds + L.sdist;
    float *const scenL = lds + L.scen;       // centres + radius of the spheres that appear in pairs (detection), with self-collision
    float *const Yd2L = lds + L.Yd2;         // second path of two-body rows, [maxlev][NC3] (phase 4a); shares scen's storage
    signed char *const nsibL = reinterpret_cast<signed char *>(lds + L.tab);     // next_sibling[n]   (bytes: indices are < 64)
    signed char *const bchL = nsibL + n;                                         // base_children[n]
    signed char *const fchL = nsibL + 2 * n;                                     // first_child[n]
    signed char *const ordL = nsibL + 3 * n;                                     // links sorted by level
    float *const ysL = lds + L.ys;
    // region AB is time-shared: kinematics exchange (pub) -> child-to-parent ABA exchange (up)
    // -> contact Jacobians (Jc) -> 6-float exchanges of the final response
    float *const pubL = lds + L.AB;                       // PUB * (n+1), entry n = base
    float *const upL = lds + L.AB;                        // UP * n, once pub's poses are consumed
    float *const JcL = lds + L.AB;                        // ndof * NC3 (after the ABA is done)
    float *const accL = lds + L.acc;                      // 8 * (n+1): accelerations root -> leaves

    HG_TICK(0);
    HG_TICK_HWID();
    const bool is_link = lane < n;
    const int S = HG_STATE_SIZE(n);
    float *const st = state + (size_t)env * P.state_stride;

    // ---- per-lane link constants --------------------------------------------------
    const int li = is_link ? lane : 0;
    const float *lk = M->link[li];
    // packed topology words of this lane's link: loaded once, decoded where needed; the
    // per-link words are read back by other lanes with v_readlane (no memory access)
    const unsigned info_w = is_link ? M->linkinfo[li] : 0u;
    const unsigned kids_w = is_link ? M->kids[li] : 0u;
    const int nbc = P.n_base_children;
    if (is_link) {
        nsibL[lane] = (signed char)((int)((kids_w >> 8) & 0xffu) - 1);
        bchL[lane] = lane < nbc ? M->base_children[lane] : (signed char)-1;
        fchL[lane] = (signed char)((int)(kids_w & 0xffu) - 1);
        ordL[lane] = M->lev_order[lane];
    }

    // The env's state record in HBM is the home of the state between substeps: each substep loads
    // it and stores it back (L2 hits after the first), so nothing is carried in registers around
    // the substep loop.

    // ---- targets (dancer_env.py:19 / nao_env.py:70) --------------------------------------
    // All four waves of a SIMD start together and would sit through every HBM latency of this prologue one after the
    // other: the independent loads (stored targets, action row, joint limits) are issued together, and the rest of the
    // state record and this lane's link constants are touched now so that the first substep finds them in the cache.
    float warm = 0.f;
    {
        const float w0 = st[lane < S ? lane : 0], w1 = st[WAVE + lane < S ? WAVE + lane : 0];
        const float w2 = lk[L_R0], w3 = lk[L_D + 1], w4 = lk[LINKC - 1];       // one word per 64-byte line of the link record
        warm = w0 + w1 + w2 + w3 + w4;
        if constexpr (NMX > 38) warm += st[2 * WAVE + lane < S ? 2 * WAVE + lane : 0];      // records beyond 128 floats
        if constexpr (NMX > 59) warm += st[3 * WAVE + lane < S ? 3 * WAVE + lane : 0];
    }
    if (P.flags & F_ACTIONS) {
        float tgt = is_link ? st[HG_STATE_BASE + 2 * n + lane] : 0.f;
        const float act = lane < P.n_sel ? actions[(size_t)env * P.n_sel + lane] : 0.f;
        const float lim_lo = lk[L_LOWER], lim_hi = lk[L_UPPER];
        if (is_link) miscL[lane] = tgt;
        wave_sync();
        int mysel = 0;
        #pragma unroll 1
        for (int k = 0; k < P.n_sel; ++k) if (lane == k) mysel = P.sel[k];
        if (lane < P.n_sel) miscL[mysel] = act;
        wave_sync();
        if (is_link) {
            tgt = miscL[lane];
            if (P.flags & F_CLAMP) tgt = fminf(fmaxf(tgt, lim_lo), lim_hi);
            st[HG_STATE_BASE + 2 * n + lane] = tgt;     // read back by the same lane below
        }
        wave_sync();
    }
    asm volatile("" :: "v"(warm));      // the touches complete here at the latest; their values are not used

    HG_TICK(1);
    int nc = 0;                       // contacts of the current physics step
    unsigned long long link_mask = 0;

    #pragma unroll 1
    for (int sub = 0; sub < P.n_substeps; ++sub) {
        // Re-derive everything lane- and size-dependent from opaque copies: without this hipcc
        // hoists hundreds of loop-invariant masks/addresses out of the substep loop and spills them.
        int lane_v = fresh_lane(), n_s = n_o;
        asm volatile("" : "+v"(lane_v), "+s"(n_s) :: "memory");
        const int lane = lane_v, n = n_s;
        const int ndof = 6 + n;
        const bool is_link = lane < n;
        const int li = is_link ? lane : 0;
        const float *lk = M->link[li];
        const int par = is_link ? (int)(info_w >> 24) - 1 : -1;
        const int pidx = par < 0 ? n : par;
        const int mylev = is_link ? (int)((info_w >> 8) & 0xffu) : 0;
        float q = 0.f, qd = 0.f;
        if (is_link) { q = st[HG_STATE_BASE + lane]; qd = st[HG_STATE_BASE + n + lane]; }
        // ============ 1. kinematics, root -> leaves ============================================
        // Every per-link quantity of the step (velocities, inertias, bias forces, the ABA cache, the impulse responses)
        // is expressed in axes PARALLEL TO THE WORLD'S with the origin at the link's own COM.  The transform between a
        // link and its parent is then a pure translation by r_w (6 VALU where a joint transform costs 24), gravity
        // needs no rotation, and origins stay at the COMs, so nothing cancels that would not cancel in link frames.
        // Cache record of a link: [0:3) r_w parent COM -> link COM, [3] 1/D, [4:10) U, [10:16) S, [16:19) COM, [19] u,
        // [20], [21] motor limits.
        const V3 ax = ld3(lk + L_AXIS);
        const V3 sl = ld3(lk + L_S);          // axis x e: linear part of the motion subspace (link frame)
        M3 E; V3 r;
        {
            float sq, cq;
            __sincosf(-q, &sq, &cq);     // v_sin/v_cos: |q| <= a few pi, absolute error ~4e-7
            E = mul(axis_rot(ax, sq, cq), ldm(lk + L_R0));
            r = ld3(lk + L_E) + mulv(E, ld3(lk + L_D));
        }
        if (is_link) {
            float *pc = cacheL + CACHE * lane;
            pc[C_MAXVEL] = lk[L_MAXVEL]; pc[C_MAXIMP] = lk[L_MAXIMP];      // motor-row limits, read in phase 4
        }
        Sv v0;      // base velocity: world axes, at the base COM (as the state record keeps it)
        {
            const V3 bpos = ld3(st);
            V3 bvel = ld3(st + 7), bang = ld3(st + 10);
            const float bq[4] = {st[3], st[4], st[5], st[6]};
            const M3 Rw0 = quat_to_mat(bq[0], bq[1], bq[2], bq[3]);
            if (P.flags & F_BASECMD) {
                // commanded planar base velocity (the three leading action slots of pepper_env.py:53), applied
                // like qibullet's resetBaseVelocity([vx, vy, 0], [0, 0, theta]) before every physics step
                const V3 cmd = ld3(O.base_cmd + 3 * (size_t)env);
                const M3 Rl = mul(Rw0, quat_to_mat(M->base[B_FQUAT], M->base[B_FQUAT + 1], M->base[B_FQUAT + 2], M->base[B_FQUAT + 3]));
                const float yaw = atan2f(Rl.m[3], Rl.m[0]);
                float sy, cy;
                sincosf(yaw, &sy, &cy);
                bvel = v3(cy * cmd.x - sy * cmd.y, sy * cmd.x + cy * cmd.y, 0.f);
                bang = v3(0.f, 0.f, cmd.z);
            }
            v0.a = bang; v0.l = bvel;
            if (lane == 0) {
                float *pb = pubL + PUB * n;
                stm(pb, Rw0); st3(pb + 9, bpos); stsv(pb + 12, v0);
                stm(baseL, Rw0); st3(baseL + 9, bpos); stsv(baseL + 52, v0);
                // the base state waits in LDS until the integration (registers are needed elsewhere)
                baseL[48] = bq[0]; baseL[49] = bq[1]; baseL[50] = bq[2]; baseL[51] = bq[3];
                st3(baseL + 58, bang); st3(baseL + 61, bvel);
            }
        }
        wave_sync();
        HG_TICK(2);
        Sv v = v0;                             // spatial velocity of this lane's link
        Sv Sw; Sw.a = v3(0, 0, 0); Sw.l = v3(0, 0, 0);     // its motion subspace
        V3 rw = v3(0, 0, 0), xw = v3(0, 0, 0); // parent COM -> COM, COM
        M3 Rw;                                 // world orientation
#pragma unroll
        for (int k = 0; k < 9; k++) Rw.m[k] = (k % 4 == 0) ? 1.f : 0.f;
        #pragma unroll 1
        for (int lev = 1; lev <= maxlev; ++lev) {      // the body is issued maxlev times: only what the children wait for
            if (is_link && mylev == lev) {
                const float *pp = pubL + PUB * pidx;
                Rw = mul_bt(ldm(pp), E);
                rw = mulv(Rw, r);
                xw = ld3(pp + 9) + rw;
                Sw.a = mulv(Rw, ax); Sw.l = mulv(Rw, sl);
                const Sv vp = ldsv(pp + 12);
                v.a = fma3(Sw.a, qd, vp.a);
                v.l = fma3(Sw.l, qd, vp.l + cross(vp.a, rw));
                float *po = pubL + PUB * lane;
                stm(po, Rw); st3(po + 9, xw); stsv(po + 12, v);
            }
            wave_sync();
        }
        if (is_link) {       // once per link, after the sweep
            float *pc = cacheL + CACHE * lane;
            st3(pc + C_RW, rw); stsv(pc + C_SW, Sw); st3(pc + C_XW, xw);
        }

        HG_TICK(3);
        // ============ 2. contact detection on the current pose =================================
        nc = 0;
        link_mask = 0;
        if (P.flags & (F_GROUND | F_SELF)) {
            // Trips of 64 device slots (hg_sim.hip build_dev_model: a body's primitives never straddle a trip).  Per trip:
            // world centre of this lane's primitive, its height over the plane into a 64-float window, rank among the
            // points of the same body inside the window (the 4 deepest within the margin become contacts, in slot
            // order), and the centre / radius / world half axis of the link-link primitives into `scen`.
            int count = 0;
            #pragma unroll 1
            for (int s0 = 0; s0 < ns; s0 += WAVE) {
                const int s = s0 + lane;
                const unsigned meta = s < ns ? M->sphere_meta[s] : 0u;      // 0: padding slot
                float d = 1e30f;
                V3 c = v3(0, 0, 0);
                const int b = (int)(meta & 0xffu) - 1;
                if (meta != 0u) {
                    const float *pp = pubL + PUB * (b < 0 ? n : b);
                    const float4 sp = *reinterpret_cast<const float4 *>(M->sphere[s]);
                    const M3 Rb = ldm(pp);
                    c = ld3(pp + 9) + mulv(Rb, v3(sp.x, sp.y, sp.z));
                    if ((meta & 0x100u) && (P.flags & F_GROUND)) d = c.z - sp.w;
                    if ((P.flags & F_SELF) && (meta & 0x8000u)) {      // appears in a pair: keep centre, radius, half axis
                        const int si = (int)M->self_idx[s];
                        const float4 hh = *reinterpret_cast<const float4 *>(M->self_half[si]);
                        float *pc8 = scenL + SCEN * si;
                        st3(pc8, c); pc8[3] = sp.w;
                        st3(pc8 + 4, mulv(Rb, v3(hh.x, hh.y, hh.z)));
                    }
                }
                if (P.flags & F_GROUND) {
                    sdistL[lane] = d;
                    wave_sync();
                    bool act = false;
                    if (d < P.margin) {
                        int rank = 0;
                        const int ge = (int)(meta >> 24);
                        #pragma unroll 1
                        for (int o = (int)((meta >> 16) & 0xffu); o < ge; ++o) {
                            const float od = sdistL[o];
                            rank += (o != lane && od < P.margin && (od < d || (od == d && o < lane))) ? 1 : 0;
                        }
                        act = rank < 4;
                    }
                    const unsigned long long bal = __ballot(act);
                    const int slot = count + __popcll(bal & ((1ull << lane) - 1ull));
                    if (act && slot < P.maxc) {
                        float *co = conL + CON * slot;
                        co[0] = __int_as_float(b); co[1] = __int_as_float(-2);
                        co[2] = c.x; co[3] = c.y; co[4] = d;
                        co[5] = 0.f; co[6] = 0.f; co[7] = 1.f; co[8] = d;
                    }
                    count += __popcll(bal);
                    wave_sync();      // the next trip rewrites the window
                }
            }
            wave_sync();
            if (P.flags & F_SELF) {
                // pairs of sphere-swept segments on bodies that are not ancestors of one another, in pair order
                const int np = P.n_pairs;
                #pragma unroll 1
                for (int p0 = 0; p0 < np && count < P.maxc; p0 += WAVE) {
                    const int p = p0 + lane;
                    bool act = false;
                    unsigned pr = 0;
                    V3 d = v3(0, 0, 0), c1 = d, c2 = d;
                    float len = 1.f, ds = 0.f, ra = 0.f, rb = 0.f;
                    if (p < np) {
                        pr = M->pair_meta[p];
                        const float *pa = scenL + SCEN * (pr & 0xffu), *pb = scenL + SCEN * ((pr >> 8) & 0xffu);
                        const float4 a0 = *reinterpret_cast<const float4 *>(pa), a1 = *reinterpret_cast<const float4 *>(pa + 4);
                        const float4 b0 = *reinterpret_cast<const float4 *>(pb), b1 = *reinterpret_cast<const float4 *>(pb + 4);
                        ra = a0.w; rb = b0.w;
                        seg_seg_closest(v3(a0.x, a0.y, a0.z), v3(a1.x, a1.y, a1.z), v3(b0.x, b0.y, b0.z), v3(b1.x, b1.y, b1.z), c1, c2);
                        d = c1 - c2;
                        len = sqrtf(dot(d, d));
                        ds = len - ra - rb;
                        act = ds < P.margin && len >= 1e-9f;
                    }
                    const unsigned long long bal = __ballot(act);
                    const int slot = count + __popcll(bal & ((1ull << lane) - 1ull));
                    if (act && slot < P.maxc) {
                        const V3 nrm = d * __frcp_rn(len);
                        const V3 pt = ((c1 - nrm * ra) + (c2 + nrm * rb)) * 0.5f;
                        float *co = conL + CON * slot;
                        co[0] = __int_as_float((int)((pr >> 16) & 0xffu) - 1);
                        co[1] = __int_as_float((int)((pr >> 24) & 0xffu) - 1);
                        st3(co + 2, pt); st3(co + 5, nrm); co[8] = ds;
                    }
                    count += __popcll(bal);
                }
            }
            nc = count < P.maxc ? count : P.maxc;
        }

        HG_TICK(4);
        // ============ 3. articulated-body algorithm ============================================
        wave_sync();      // pub (poses, velocities) is dead: region AB now holds the child -> parent records
        // Prelude, lane = link: velocity-product acceleration and bias force.  The link's own `up`
        // record carries its rigid-body inertia and bias force until the sweep overwrites it.
        if (is_link) {
            Sv vj; vj.a = Sw.a * qd; vj.l = Sw.l * qd;
            // v already contains vj; v x vj == (v - vj) x vj + 0, use v directly (vj x vj = 0 for the
            // angular part; the linear part keeps both terms)
            Sv cor, pA;
            cor.a = cross(v.a, vj.a);
            cor.l = cross(v.a, vj.l) + cross(v.l, vj.a);
            const float mass = lk[L_MASS];
            const V3 inertia = ld3(lk + L_INERTIA);
            // rotational inertia in world axes: Rw diag(I) Rw^T
            Sym3 Iw;
            {
                const V3 r0 = row(Rw, 0), r1 = row(Rw, 1), r2 = row(Rw, 2);
                const V3 s0 = v3(r0.x * inertia.x, r0.y * inertia.y, r0.z * inertia.z);
                const V3 s1 = v3(r1.x * inertia.x, r1.y * inertia.y, r1.z * inertia.z);
                const V3 s2 = v3(r2.x * inertia.x, r2.y * inertia.y, r2.z * inertia.z);
                Iw.xx = dot(s0, r0); Iw.xy = dot(s0, r1); Iw.xz = dot(s0, r2);
                Iw.yy = dot(s1, r1); Iw.yz = dot(s1, r2); Iw.zz = dot(s2, r2);
            }
            const V3 Iom = symv(Iw, v.a);
            const float wn = sqrtf(dot(v.a, v.a)), vn = sqrtf(dot(v.l, v.l));
            pA.a = cross(v.a, Iom) + Iom * (P.ang_damp + P.ang_damp * wn);
            pA.l = (cross(v.a, v.l) - v3(P.gx, P.gy, P.gz) + v.l * (P.lin_damp + P.lin_damp * vn)) * mass;
            float *pu = upL + UP * lane;
            pu[0] = Iw.xx; pu[1] = Iw.xy; pu[2] = Iw.xz; pu[3] = Iw.yy; pu[4] = Iw.yz; pu[5] = Iw.zz;
            pu[6] = mass; pu[7] = -lk[L_DAMP] * qd;
            stsv(pu + 21, pA);
            stsv(accL + 8 * lane, cor);      // read back in the downward pass
        }
        wave_sync();
        // leaves -> root, 7 lanes per link: lane (g, k) of a link at the current level applies the
        // link's articulated inertia to one vector -- k < 6: column k of the motion transform to the
        // parent (giving column k of X^T Ia X), k = 6: the velocity-product acceleration (giving the
        // bias force handed to the parent).  U, D and the child sums are recomputed by all 7 lanes.
        {
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));
            const int g = (lane_p * 37) >> 8;             // lane / 7 for lane < 64
            const int k = lane_p - 7 * g;
            #pragma unroll 1
            for (int lev = maxlev; lev >= 1; --lev) {
                const int s0 = P.lev_start[lev], cnt = P.lev_start[lev + 1] - s0;
                #pragma unroll 1
                for (int b0 = 0; b0 < cnt; b0 += 9) {
                    if (g < 9 && b0 + g < cnt) {
                        const int link = ordL[s0 + b0 + g];
                        float *pc = cacheL + CACHE * link;
                        float *own = upL + UP * link;
                        AI IA;
                        IA.A.xx = own[0]; IA.A.xy = own[1]; IA.A.xz = own[2]; IA.A.yy = own[3]; IA.A.yz = own[4]; IA.A.zz = own[5];
                        IA.C.xx = IA.C.yy = IA.C.zz = own[6]; IA.C.xy = IA.C.xz = IA.C.yz = 0.f;
#pragma unroll
                        for (int j = 0; j < 9; j++) IA.B.m[j] = 0.f;
                        const float tau0 = own[7];
                        Sv pA = ldsv(own + 21);
                        #pragma unroll 1
                        for (int c = fchL[link]; c >= 0; c = nsibL[c]) {
                            float pu[28];      // the child's record as seven 16-byte LDS reads (UP floats = 112 bytes, 16-byte aligned)
                            {
                                const float4 *p4 = reinterpret_cast<const float4 *>(upL + UP * c);
#pragma unroll
                                for (int j = 0; j < 7; j++) { const float4 t = p4[j]; pu[4 * j] = t.x; pu[4 * j + 1] = t.y; pu[4 * j + 2] = t.z; pu[4 * j + 3] = t.w; }
                            }
                            static_assert(UP == 28, "child -> parent record is seven float4");
                            IA.A.xx += pu[0]; IA.A.xy += pu[1]; IA.A.xz += pu[2]; IA.A.yy += pu[3]; IA.A.yz += pu[4]; IA.A.zz += pu[5];
#pragma unroll
                            for (int j = 0; j < 9; j++) IA.B.m[j] += pu[6 + j];
                            IA.C.xx += pu[15]; IA.C.xy += pu[16]; IA.C.xz += pu[17]; IA.C.yy += pu[18]; IA.C.yz += pu[19]; IA.C.zz += pu[20];
                            pA = pA + ldsv(pu + 21);
                        }
                        const V3 r = ld3(pc + C_RW);
                        const Sv Sm = ldsv(pc + C_SW);
                        const Sv U = ai_mul(IA, Sm);
                        const float Dinv = __builtin_amdgcn_rcpf(sdot(Sm, U));      // v_rcp_f32, 1 ulp (__frcp_rn is a ten-instruction division)
                        const float u = tau0 - sdot(Sm, pA);
                        // the vector this lane pushes through Ia = IA - U U^T / D
                        Sv w;
                        if (k == 6) {
                            w = ldsv(accL + 8 * link);
                        } else {
                            const int m = k < 3 ? k : k - 3;
                            const V3 col = v3(m == 0 ? 1.f : 0.f, m == 1 ? 1.f : 0.f, m == 2 ? 1.f : 0.f);      // the transform is a translation
                            if (k < 3) { w.a = col; w.l = cross(col, r); }
                            else { w.a = v3(0, 0, 0); w.l = col; }
                        }
                        const float t = sdot(U, w) * Dinv;
                        Sv f = sfma(U, -t, ai_mul(IA, w));
                        if (k == 6) f = sfma(U, u * Dinv, f + pA);
                        Sv o; o.a = f.a + cross(r, f.l); o.l = f.l;      // to the parent's origin
                        // all reads of the own record are issued: overwrite it with the parent-frame data
                        if (k == 6) {
                            stsv(own + 21, o);
                            stsv(pc + C_UW, U); pc[C_DINV] = Dinv; pc[C_U] = u;
                        } else {
                            // column k of X^T Ia X; the packed record keeps rows <= column of A and C, and B by rows
                            const int m = k < 3 ? k : k - 3;
                            const V3 sv = k < 3 ? o.a : o.l;
                            float *blk = own + (k < 3 ? 0 : 15);
                            blk[m] = sv.x;
                            if (m >= 1) blk[2 + m] = sv.y;
                            if (m == 2) blk[5] = sv.z;
                            if (k < 3) st3(own + 6 + 3 * k, o.l);
                        }
                    }
                    wave_sync();
                }
            }
        }
        HG_TICK(5);
        // base, in world axes at its COM (every lane computes the same thing; results stay in registers)
        Sv a0;
        {
            float A6[36];
#pragma unroll
            for (int k = 0; k < 36; k++) A6[k] = 0.f;
            const float bmass = M->base[B_MASS];
            const V3 binertia = ld3(M->base + B_INERTIA);
            const M3 Rw0 = ldm(baseL);
            const Sv v0 = ldsv(baseL + 52);
            {   // rotational inertia in world axes
                const V3 r0 = row(Rw0, 0), r1 = row(Rw0, 1), r2 = row(Rw0, 2);
                const V3 s0 = v3(r0.x * binertia.x, r0.y * binertia.y, r0.z * binertia.z);
                const V3 s1 = v3(r1.x * binertia.x, r1.y * binertia.y, r1.z * binertia.z);
                const V3 s2 = v3(r2.x * binertia.x, r2.y * binertia.y, r2.z * binertia.z);
                A6[0] = dot(s0, r0); A6[1] = dot(s0, r1); A6[2] = dot(s0, r2);
                A6[7] = dot(s1, r1); A6[8] = dot(s1, r2); A6[14] = dot(s2, r2);
            }
            A6[21] = A6[28] = A6[35] = bmass;
            const V3 Iw = v3(fmaf(A6[0], v0.a.x, fmaf(A6[1], v0.a.y, A6[2] * v0.a.z)),
                             fmaf(A6[1], v0.a.x, fmaf(A6[7], v0.a.y, A6[8] * v0.a.z)),
                             fmaf(A6[2], v0.a.x, fmaf(A6[8], v0.a.y, A6[14] * v0.a.z)));
            const V3 g = v3(P.gx, P.gy, P.gz);
            const float wn = sqrtf(dot(v0.a, v0.a)), vn = sqrtf(dot(v0.l, v0.l));
            Sv p0;
            p0.a = cross(v0.a, Iw) + Iw * (P.ang_damp + P.ang_damp * wn);
            p0.l = (cross(v0.a, v0.l) - g + v0.l * (P.lin_damp + P.lin_damp * vn)) * bmass;
            #pragma unroll 1
            for (int k = 0; k < nbc; ++k) {
                const float *pu = upL + UP * bchL[k];
                A6[0] += pu[0]; A6[1] += pu[1]; A6[2] += pu[2]; A6[7] += pu[3]; A6[8] += pu[4]; A6[14] += pu[5];
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) A6[6 * i + 3 + j] += pu[6 + 3 * i + j];
                A6[21] += pu[15]; A6[22] += pu[16]; A6[23] += pu[17]; A6[28] += pu[18]; A6[29] += pu[19]; A6[35] += pu[20];
                p0 = p0 + ldsv(pu + 21);
            }
            // mirror to the lower triangle
            A6[6] = A6[1]; A6[12] = A6[2]; A6[13] = A6[8];
#pragma unroll
            for (int i = 0; i < 3; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) A6[6 * (3 + j) + i] = A6[6 * i + 3 + j];
            A6[27] = A6[22]; A6[33] = A6[23]; A6[34] = A6[29];
            // One factorisation (replicated), then every lane solves ONE system: lanes 0-5 the unit vectors -- row `lane` of
            // the inverse, which the impulse responses stream from LDS -- and the other lanes -p0, the base acceleration.
            float Lc[36], rd[6], rhs[6], x[6];
            spd6_cholesky(A6, Lc, rd);
            const float p6[6] = {p0.a.x, p0.a.y, p0.a.z, p0.l.x, p0.l.y, p0.l.z};
#pragma unroll
            for (int k = 0; k < 6; k++) rhs[k] = lane < 6 ? (lane == k ? 1.f : 0.f) : -p6[k];
            chol6_solve(Lc, rd, rhs, x);
            if (lane < 6) {
#pragma unroll
                for (int k = 0; k < 6; k++) baseL[12 + 6 * lane + k] = x[k];
            }
#pragma unroll
            for (int k = 0; k < 6; k++) x[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x[k]), 6));      // lane 6 solved for -p0
            a0.a = v3(x[0], x[1], x[2]); a0.l = v3(x[3], x[4], x[5]);
            if (lane == 0) stsv(accL + 8 * n, a0);
        }
        wave_sync();
        // root -> leaves: accelerations
        float qdd = 0.f;
        {
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));      // phase-local copies: nothing derived earlier stays live
            unsigned info_p = info_w;
            asm volatile("" : "+v"(info_p));
            const bool is_link = lane_p < n;
            const int mylev = is_link ? (int)((info_p >> 8) & 0xffu) : 0;
            const int par = (int)(info_p >> 24) - 1;
            const int pidx = par < 0 ? n : par;
            #pragma unroll 1
            for (int lev = 1; lev <= maxlev; ++lev) {
                if (mylev == lev) {
                    const float *pc = cacheL + CACHE * lane_p;
                    float *pa = accL + 8 * lane_p;
                    Sv a = ldsv(accL + 8 * pidx);
                    a.l = a.l + cross(a.a, ld3(pc + C_RW));      // the parent's acceleration at this link's origin
                    a = a + ldsv(pa);
                    qdd = (pc[C_U] - sdot(ldsv(pc + C_UW), a)) * pc[C_DINV];
                    a = sfma(ldsv(pc + C_SW), qdd, a);
                    stsv(pa, a);
                }
                wave_sync();
            }
        }
        HG_TICK(6);
        // velocities += dt * unconstrained acceleration
        // (every velocity coordinate is clamped to +-vmax after each update, as btMultiBody does)
        qd = clampv(fmaf(P.dt, qdd, qd), P.vmax);
        {
            const Sv v0 = ldsv(baseL + 52);
            const V3 bang = clamp3(ld3(baseL + 58) + a0.a * P.dt, P.vmax);
            const V3 bvel = clamp3(ld3(baseL + 61) + (a0.l + cross(v0.a, v0.l)) * P.dt, P.vmax);      // classical acceleration of the COM
            // ============ 4. constraint rows ===================================================
            // current velocities for the contact rows
            if (lane == 0) { st3(velL, bang); st3(velL + 3, bvel); }
        }
        if (is_link) { velL[6 + lane] = qd; accL[lane] = qd; }     // accL: parked until the end of the substep (accelerations are dead)
        wave_sync();      // region AB (pub/up) is dead from here on: Jc may overwrite it

        // ---- row layout ---------------------------------------------------------------------------------
        // Up to 64 rows (every robot without finger joints): ONE row per lane -- motor rows on lanes < NMX, contact
        // rows on lanes NMX .. NROWS-1, up to four violated joint limits on the spare lanes.
        // More rows than lanes (DUAL: the 42-dof NAO with its fingers and 8 contacts has 66 rows, a Romeo with hands
        // 85): every lane carries TWO rows, row set 0 = the motor row of joint `lane`, row set 1 = contact row `lane`
        // (lanes < NC3).  Both sets run the same row set-up and impulse-response code below; the solver keeps both
        // candidates per lane and every row update touches both.
        // More rows than lanes by at most HG_MAX_OVF (OVF: the 42-dof NAO with 8 contact slots has 66 rows): ONE row per lane as
        // above, motor rows on lanes < NMX, contact rows 0 .. NCL-1 on lanes NMX .. 63; the last NOV contact rows -- the friction
        // pair of the last contact slot, live only when every slot is taken -- have no lane.  Their Jacobians are built by lanes
        // < NOV (row set 1, as in DUAL), their mutual entries of A by a response along their own paths only (4a'), and in
        // the sweep their candidate impulses come from a wave reduction (4b, "overflow rows").
        constexpr int NOV = ovf_rows(NMX, MAXC);
        constexpr bool DUAL = NROWS > WAVE && NOV == 0;
        constexpr int NCL = NC3 - NOV;             // contact rows that have a lane
        constexpr int OC0 = MAXC - NOV / 2;        // first contact slot whose friction pair overflows
        constexpr int CB = DUAL ? 0 : NMX;         // lane of contact column 0 (in row set 1 when DUAL)
        constexpr int CBL = DUAL ? WAVE : NMX;     // index of contact impulse 0 in lamL
        RowVars R0, R1;                             // R1 is dead code unless DUAL or OVF
        R0.clear(); R1.clear();
        RowVars &RC = *(DUAL ? &R1 : &R0);          // the set that carries the contact rows
        RC.col = (lane >= CB && lane < CB + NCL) ? lane - CB : -1;          // contact-row column
        if constexpr (NOV > 0) R1.col = lane < NOV ? NCL + lane : -1;
        const bool ovf_on = NOV > 0 && nc > OC0;                            // wave-uniform
        R0.motor = is_link;                              // NMX >= n by construction
        if (R0.motor) R0.body = lane;
        unsigned long long rowmask = 0ull;       // links with a non-zero row in Jc (wave-uniform)
        // One contact row: Jacobian column into Jc (joint part on the path(s) from the contact bodies to the base, base part),
        // desired change of J.v
        auto contact_setup = [&](RowVars &R) __attribute__((always_inline)) {
            const int col = R.col;
            const int cidx = col < 0 ? -1 : (col < MAXC ? col : (col - MAXC) >> 1);
            R.contact = cidx >= 0 && cidx < nc;
            V3 pt = v3(0, 0, 0), dir = v3(0, 0, 0);
            float dist = 0.f;
            if (R.contact) {
                const float *co = conL + CON * cidx;
                R.body = __float_as_int(co[0]);
                R.body2 = __float_as_int(co[1]);
                pt = ld3(co + 2);
                const V3 nrm = ld3(co + 5);
                dist = co[8];
                V3 t1, t2;
                plane_space(nrm, t1, t2);
                dir = col < MAXC ? nrm : (((col - MAXC) & 1) ? t2 : t1);
                if (R.body >= 0) {
                    R.myanc = M->anc_mask[R.body];
                    const uint4 t = *reinterpret_cast<const uint4 *>(M->pathw[R.body]);
                    R.pw[0] = t.x; R.pw[1] = t.y; R.pw[2] = t.z;
                }
                if (R.body2 >= -1) {          // both bodies belong to the robot: the base columns cancel
                    if (R.body2 >= 0) {
                        R.myanc2 = M->anc_mask[R.body2];
                        const uint4 t = *reinterpret_cast<const uint4 *>(M->pathw[R.body2]);
                        R.pw2[0] = t.x; R.pw2[1] = t.y; R.pw2[2] = t.z;
                    }
                } else {
                    R.Jang = cross(pt - ld3(baseL + 9), dir);
                    R.Jlin = dir;
                }
            }
            // joint part: non-zero only on the path(s) from the contact bodies to the base, at most DEPTH entries per column
            float rv = 0.f;
            const bool two = __ballot(R.myanc2 != 0ull) != 0ull;
            if (R.contact) {
#pragma unroll
                for (int dd = 1; dd <= DEPTH; ++dd) {
                    if (dd <= maxlev) {      // wave-uniform
                        const int l = (int)((R.pw[(dd - 1) >> 2] >> (8 * ((dd - 1) & 3))) & 0xffu) - 1;
                        if (l >= 0) {
                            const float *pcl = cacheL + CACHE * l;      // velocity of the point per unit joint rate: S.a x (pt - COM) + S.l
                            const Sv Sl = ldsv(pcl + C_SW);
                            const float j = dot(Sl.a, cross(pt - ld3(pcl + C_XW), dir)) + dot(Sl.l, dir);
                            JcL[(6 + l) * NC3 + col] = j;
                            rv = fmaf(j, velL[6 + l], rv);
                        }
                    }
                }
                if (two) {
                    // second body: opposite sign; entries of common ancestors cancel exactly (same axis, pivot, point)
#pragma unroll
                    for (int dd = 1; dd <= DEPTH; ++dd) {
                        if (dd <= maxlev) {
                            const int l = (int)((R.pw2[(dd - 1) >> 2] >> (8 * ((dd - 1) & 3))) & 0xffu) - 1;
                            if (l >= 0) {
                                const float *pcl = cacheL + CACHE * l;
                                const Sv Sl = ldsv(pcl + C_SW);
                                const float j = -(dot(Sl.a, cross(pt - ld3(pcl + C_XW), dir)) + dot(Sl.l, dir));
                                JcL[(6 + l) * NC3 + col] += j;
                                rv = fmaf(j, velL[6 + l], rv);
                            }
                        }
                    }
                }
            }
            if (col >= 0) {
                JcL[0 * NC3 + col] = R.Jang.x; JcL[1 * NC3 + col] = R.Jang.y; JcL[2 * NC3 + col] = R.Jang.z;
                JcL[3 * NC3 + col] = R.Jlin.x; JcL[4 * NC3 + col] = R.Jlin.y; JcL[5 * NC3 + col] = R.Jlin.z;
                rv += dot(R.Jang, ld3(velL)) + dot(R.Jlin, ld3(velL + 3));
            }
            if (R.contact) {
                if (col < MAXC) {
                    R.target_err = -rv;
                    if (dist > 0.f) R.target_err -= dist * P.inv_dt;
                    else R.target_err -= dist * P.erp * P.inv_dt;
                } else {
                    R.target_err = -rv;
                }
            }
        };
        if (nc > 0) {
            // joint part of Jc: zero everywhere ...
            {
                float4 *z4 = reinterpret_cast<float4 *>(JcL + 6 * NC3);
                float zero = 0.f;
                asm volatile("" : "+v"(zero));       // materialised here: a hoisted zero quad would be spilled
                #pragma unroll 1
                // whole quads, rounded UP: (n * NC3) % 4 may be 2 (odd n with NC3 = 18); 6 * NC3 is a multiple of 4 and the
                // region is sized rup4((6 + n) * NC3), so the last quad stays inside it
                for (int k = lane; k < (n * NC3 + 3) / 4; k += WAVE) z4[k] = make_float4(zero, zero, zero, zero);
            }
            wave_sync();
            // ... except on the contact paths
            contact_setup(RC);
            if constexpr (NOV > 0) contact_setup(R1);      // also when inactive: the base part of their Jc columns must be written (zeros)
            // which links have a non-zero row in Jc: OR of the ancestor masks of the contact bodies, read
            // from the normal rows (lane CB + c) on the scalar unit
            {
                const unsigned long long m = RC.myanc | RC.myanc2;
                const int mlo = (int)(unsigned)m, mhi = (int)(unsigned)(m >> 32);
                unsigned rlo = 0u, rhi = 0u;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < nc) {
                        rlo |= (unsigned)__builtin_amdgcn_readlane(mlo, CB + c);
                        if constexpr (NMX > 32) rhi |= (unsigned)__builtin_amdgcn_readlane(mhi, CB + c);
                    }
                rowmask = ((unsigned long long)rhi << 32) | rlo;
            }
        }
        if (R0.motor) {
            const float tgt = st[HG_STATE_BASE + 2 * n + lane];
            float tv = fmaf(P.kp * (tgt - q), P.inv_dt, qd) - P.kd * qd;
            const float maxvel = cacheL[CACHE * lane + C_MAXVEL];
            tv = fminf(fmaxf(tv, -maxvel), maxvel);
            R0.target_err = tv - qd;
        }
        // Joint-limit rows (btMultiBodyJointLimitConstraint): one per violated bound, Jacobian +-1 on the joint
        // (so its impulse response is +- the motor row's: the row shares the motor's lane), desired
        // velocity 0 with the Baumgarte term -penetration * erp / dt, impulse in [0, limhi].
        // (evaluated twice, before and after the impulse-response pass, from q of the state record and the parked qd:
        // two values fewer to keep alive across that pass, where the A-row leaves no register to spare)
        auto limit_row = [&](int ln, float ql, float qdl, float &ls, float &lb) __attribute__((always_inline)) {
            ls = 0.f; lb = 0.f;
            if (ln < n && (P.flags & F_LIMITS)) {
                const float *lkl = M->link[ln];
                if (lkl[L_LIMITED] != 0.f) {
                    const float pen_lo = ql - lkl[L_LOWER], pen_hi = lkl[L_UPPER] - ql;
                    float pen = 0.f;
                    if (pen_lo <= 0.f) { ls = 1.f; pen = pen_lo; }
                    else if (pen_hi <= 0.f) { ls = -1.f; pen = pen_hi; }
                    lb = -pen * P.erp * P.inv_dt - ls * qdl;
                }
            }
        };
        float lim_s, lim_b;
        limit_row(lane, q, qd, lim_s, lim_b);
        const unsigned long long limmask = __ballot(lim_s != 0.f);       // wave-uniform
        // Limit rows on spare lanes (one row per lane only).  The wave has 64 - NROWS lanes without a row; up to KL of
        // them take one violated joint limit each and become ordinary rows: lane NROWS + k (slot k) carries a row with
        // Jacobian slot_f = +-1 on joint slot_joint[k], so the impulse-response pass below computes its
        // A-row like any other lane's at no extra instruction.  Envs with more violated joints than
        // slots (and the variants with no spare lane, and the two-rows-per-lane variants) take the generic path in the solver.
        constexpr int KL = NROWS >= WAVE ? 0 : ((WAVE - NROWS) < 4 ? (WAVE - NROWS) : 4);
        const int nlim = __popcll(limmask);
        const bool use_slots = KL > 0 && nlim > 0 && nlim <= KL;      // wave-uniform
        int slot_joint[KL > 0 ? KL : 1];
        float slot_sign[KL > 0 ? KL : 1];
#pragma unroll
        for (int k = 0; k < (KL > 0 ? KL : 1); k++) { slot_joint[k] = 0; slot_sign[k] = 0.f; }
        if constexpr (KL > 0) {
            if (use_slots) {
                unsigned long long mk = limmask;
#pragma unroll
                for (int k = 0; k < KL; k++) {
                    if (k < nlim) {       // slot k <- k-th violated joint, ascending
                        const int i = __builtin_ctzll(mk);
                        mk &= mk - 1ull;
                        const float sg = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lim_s), i));
                        const float b0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lim_b), i));
                        slot_joint[k] = i; slot_sign[k] = sg;
                        if (lane == NROWS + k) { R0.slot = true; R0.unit_f = sg; R0.body = i; R0.target_err = b0; }
                    }
                }
            }
        }
        if (R0.motor) R0.unit_f = 1.f;           // the one non-zero Jacobian entry of a motor / limit row
        if (R0.motor || R0.slot) {     // motor and limit rows act on one link
            R0.myanc = M->anc_mask[R0.body];
            const uint4 t = *reinterpret_cast<const uint4 *>(M->pathw[R0.body]);
            R0.pw[0] = t.x; R0.pw[1] = t.y; R0.pw[2] = t.z;
        }
        RC.two_body = __ballot(RC.myanc2 != 0ull) != 0ull;   // wave-uniform
        if constexpr (NOV > 0) R1.two_body = __ballot(R1.myanc2 != 0ull) != 0ull;
        lamL[lane] = R0.target_err;       // parked across the impulse-response pass (the impulses' own storage, unused until the sweeps end)
        wave_sync();

        HG_TICK(7);
        // ---- 4a'. rows of row set 1 (DUAL: the contact rows; OVF: the overflow rows) -------------------------------------------
        // Their entries against the MOTOR rows are not computed by a second response over the whole tree: A is symmetric, and
        // the lane of motor row i accumulates all NC3 contact columns in 4a, so A[c][i] is lane i's own column c (DUAL: moved
        // to lane c through LDS before the sweeps; OVF: used in place by the wave reduction).  What is left are their entries
        // against the CONTACT rows, diagonal included: J_c' is non-zero only on the links of the contact paths (rowmask,
        // closed under ancestors), so M^-1 J_c^T is evaluated on those links only -- the own-path walk towards the root, the
        // base, then a descent that skips every link off the contact paths (`response` with PATHONLY).
        // OVF results as wave-uniform scalars: ov_b (desired change of J.v), ov_inv (1 / A_oo, 0 = row disabled), ov_a[k][k'].
        float ov_b[NOV > 0 ? NOV : 1], ov_inv[NOV > 0 ? NOV : 1], ov_a[NOV > 0 ? NOV : 1][NOV > 0 ? NOV : 1];
#pragma unroll
        for (int k = 0; k < (NOV > 0 ? NOV : 1); k++) {
            ov_b[k] = 0.f; ov_inv[k] = 0.f;
#pragma unroll
            for (int j = 0; j < (NOV > 0 ? NOV : 1); j++) ov_a[k][j] = 0.f;
        }
        // ---- 4a. lane r: u_r = M^-1 J_r^T, and row r of A = J M^-1 J^T ----------------------------
        // Motor columns of the row: register vectors of up to 16 columns, so that column i (wave-uniform, known only
        // at run time in the descent below) is written with one indexed move (s_set_gpr_idx + v_mov)
        // instead of a 64-way switch; contact columns: plain registers, compile-time indices only.
        typedef typename RegVec<clamp16(NMX)>::type VA0;
        typedef typename RegVec<clamp16(NMX - 16)>::type VA1;
        typedef typename RegVec<clamp16(NMX - 32)>::type VA2;
        typedef typename RegVec<clamp16(NMX - 48)>::type VA3;
        VA0 AM0 = (VA0)(0.f); VA1 AM1 = (VA1)(0.f); VA2 AM2 = (VA2)(0.f); VA3 AM3 = (VA3)(0.f);     // columns 0-15, 16-31, 32-47, 48-63
        float AC[NC3 > 0 ? NC3 : 1];
#pragma unroll
        for (int k = 0; k < NC3; k++) AC[k] = 0.f;
        float diag = 0.f;
        // the same for the contact rows of row set 1 (DUAL only; dead code otherwise)
        VA0 BM0 = (VA0)(0.f); VA1 BM1 = (VA1)(0.f); VA2 BM2 = (VA2)(0.f); VA3 BM3 = (VA3)(0.f);
        float BC[NC3 > 0 ? NC3 : 1];
#pragma unroll
        for (int k = 0; k < NC3; k++) BC[k] = 0.f;
        float diagB = 0.f;
        // One row per lane + overflow rows (the 42-dof NAO): lane i keeps the cache record of ITS link -- r_w, 1/D, U, S, 16 floats
        // -- in VGPRs through the response passes, and the descent fetches link i's record with 16 v_readlane into SGPRs
        // instead of four broadcast LDS reads: every use of the record is scalar x vector, which is what an SGPR operand is for,
        // and the LDS reads of the contact Jacobian rows no longer queue behind them.  Same-box A/B (tools/recreg_ab.sh): the
        // bench workload (8 contacts: two descents per step) +3.9 %, the same robot under random targets (4 contacts) -1.1 %;
        // Romeo's 37-dof shape -4 % and the two-rows-per-lane shape -0.5 % (registers), so they keep the LDS reads.
#ifndef HG_RECREG
#define HG_RECREG 1        // 0: the LDS reads everywhere (A/B builds, tools/recreg_ab.sh)
#endif
        constexpr bool RECREG = HG_RECREG && NOV > 0;
        float4 rq0 = {0.f, 0.f, 0.f, 0.f}, rq1 = rq0, rq2 = rq0, rq3 = rq0;
        if constexpr (RECREG) {
            static_assert(C_RW == 0 && C_DINV == 3 && C_UW == 4 && C_SW == 10, "cache record layout");
            const float *pc = cacheL + CACHE * (lane < n ? lane : 0);
            rq0 = *reinterpret_cast<const float4 *>(pc); rq1 = *reinterpret_cast<const float4 *>(pc + 4);
            rq2 = *reinterpret_cast<const float4 *>(pc + 8); rq3 = *reinterpret_cast<const float4 *>(pc + 12);
        }
        unsigned long long ovmask = 0ull;      // links on the path(s) of the overflow rows' contact (wave-uniform), and the deeper path's depth
        int ovlev = 0;
        if constexpr (NOV > 0) {
            const unsigned long long m = R1.myanc | R1.myanc2;
            const int p1 = __popcll(R1.myanc), p2 = __popcll(R1.myanc2);
            ovlev = __builtin_amdgcn_readfirstlane(p1 > p2 ? p1 : p2);
            ovmask = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m) |
                     ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m >> 32)) << 32);
        }
        auto response = [&](auto PO_, const RowVars &R, VA0 &XM0, VA1 &XM1, VA2 &XM2, VA3 &XM3, float (&XC)[NC3 > 0 ? NC3 : 1], float &xdiag)
                            __attribute__((always_inline)) {
            constexpr bool PATHONLY = decltype(PO_)::value != 0;      // contact rows of row set 1: links on the contact paths only, no motor columns
            // overflow rows (the friction pair of ONE contact): only their mutual entries are wanted, so only the links of that
            // contact's own path(s) are descended and only the overflow columns accumulated
            constexpr bool OVP = PATHONLY && NOV > 0;
            constexpr int C0 = OVP ? NCL : 0;
            const unsigned long long dmask = OVP ? ovmask : rowmask;
            const int wlev = OVP ? ovlev : maxlev;
            const int col = R.col, body = R.body;
            const bool unit_row = R.motor || R.slot;
            const unsigned pw[3] = {R.pw[0], R.pw[1], R.pw[2]};
            // (i) walk the own path towards the root, accumulating the zero-acceleration force
            Sv Z; Z.a = v3(0, 0, 0); Z.l = v3(0, 0, 0);
            constexpr int YSN = YSL ? 1 : DEPTH;
            float Ys[YSN];               // Y of the own-path link at each depth (registers; LDS for the YSL variants)
#pragma unroll
            for (int dd = DEPTH; dd >= 1; --dd) {
                float Y = 0.f;
                if (dd <= wlev) {        // wave-uniform
                    const int l = (int)((pw[(dd - 1) >> 2] >> (8 * ((dd - 1) & 3))) & 0xffu) - 1;
                    if (l >= 0) {
                        const float *pc = cacheL + CACHE * l;
                        const float f = unit_row ? (l == body ? R.unit_f : 0.f) : JcL[(6 + l) * NC3 + col];
                        const Sv Sl = ldsv(pc + C_SW);
                        const Sv Ul = ldsv(pc + C_UW);
                        Y = f - sdot(Sl, Z);
                        Z = sfma(Ul, Y * pc[C_DINV], Z);
                        Z.a = Z.a + cross(ld3(pc + C_RW), Z.l);        // to the parent's origin
                    }
                }
                if constexpr (YSL) ysL[(dd - 1) * WAVE + lane] = Y; else Ys[dd - 1] = Y;
            }
            if (R.two_body) {
                // The impulse of a self-contact row acts on two branches.  By linearity the second
                // branch is walked on its own (its force entries are already in Jc with their sign);
                // the zero-acceleration forces meet at the base.
                Sv Z2; Z2.a = v3(0, 0, 0); Z2.l = v3(0, 0, 0);
#pragma unroll
                for (int dd = DEPTH; dd >= 1; --dd) {
                    float Y = 0.f;
                    if (dd <= wlev) {
                        const int l = (int)((R.pw2[(dd - 1) >> 2] >> (8 * ((dd - 1) & 3))) & 0xffu) - 1;
                        if (l >= 0) {
                            const float *pc = cacheL + CACHE * l;
                            // entries of links shared with the first branch are zero in Jc (they cancel)
                            const float f = ((R.myanc >> l) & 1ull) ? 0.f : JcL[(6 + l) * NC3 + col];
                            Y = f - sdot(ldsv(pc + C_SW), Z2);
                            Z2 = sfma(ldsv(pc + C_UW), Y * pc[C_DINV], Z2);
                            Z2.a = Z2.a + cross(ld3(pc + C_RW), Z2.l);
                        }
                        if (col >= 0) Yd2L[(dd - 1) * NC3 + col] = Y;
                    }
                }
                Z = Z + Z2;
            }
            HG_TICK(16);      // (the last call's stamps stay: the rows that have a lane)
            // (ii) base
            Z.a = Z.a - R.Jang;
            Z.l = Z.l - R.Jlin;
            const Sv ab = neg_mul6(baseL + 12, Z);       // IA0^-1 (world axes, at the base COM) streamed from LDS
            const V3 uba = ab.a, ubl = ab.l;
            xdiag = dot(uba, R.Jang) + dot(ubl, R.Jlin);
            if (nc > 0) {
                const float ub[6] = {uba.x, uba.y, uba.z, ubl.x, ubl.y, ubl.z};
#pragma unroll
                for (int k = 0; k < 6; k++)
#pragma unroll
                    for (int c = C0; c < NC3; c++) XC[c] = fmaf(ub[k], JcL[k * NC3 + c], XC[c]);
            }
            // (iii) full descent: joint accelerations of every link for this row's impulse.  Unrolled over the links: the link
            // index is a compile-time constant (the motor column is a named register, the cache record a constant LDS
            // offset, its topology word one v_readlane); what stays run-time is the topology itself.
            // Parent acceleration of a link: the previous link's (mode 0: `a` simply stays in its registers), the base's, or that
            // of a saved branching link -- picked with selects under scalar conditions inside ONE uniform branch, so that a
            // chain link (the common case) pays nothing and nothing is copied at a merge point.
            static_assert(NSL <= 2, "two saved-acceleration slots");
            HG_TICK(17);
            // (Reading a link's cache record one link AHEAD was tried: every link is its own basic block, so hipcc does not
            // hoist the four LDS reads and a lone wave waits for them once per link.  LDS returns in order, so the reads of the
            // contact Jacobian rows behind the prefetch wait for it anyway; 16 more live VGPRs spill the 4-waves-per-SIMD
            // shapes; the 42-dof descent got 20 % slower.  Not kept.)
            Sv a = ab, slot0 = ab, slot1 = ab;
            // own-path Y values by depth in one register vector, read with the link's depth as index
            typedef float V16f __attribute__((ext_vector_type(16)));
            V16f Yv = (V16f)(0.f);
            if constexpr (!YSL) {
#pragma unroll
                for (int k = 0; k < DEPTH; k++) Yv[k] = Ys[k < YSN ? k : 0];
            }
#pragma unroll
            for (int i = 0; i < NMX; ++i) {      // (a plain loop, fully unrolled: a closure per link sent the slots to scratch)
                if (i < n && (!PATHONLY || ((dmask >> i) & 1ull))) {
                const unsigned inf = (unsigned)__builtin_amdgcn_readlane((int)info_w, i);
                const int mode = (int)(inf & 0x3u);
                const float *pc = cacheL + CACHE * i;
                if (mode != 0) {
                    const bool m2 = NSL > 0 && mode == 2, m3 = NSL > 1 && mode == 3;
                    a.a.x = m3 ? slot1.a.x : (m2 ? slot0.a.x : ab.a.x); a.a.y = m3 ? slot1.a.y : (m2 ? slot0.a.y : ab.a.y);
                    a.a.z = m3 ? slot1.a.z : (m2 ? slot0.a.z : ab.a.z); a.l.x = m3 ? slot1.l.x : (m2 ? slot0.l.x : ab.l.x);
                    a.l.y = m3 ? slot1.l.y : (m2 ? slot0.l.y : ab.l.y); a.l.z = m3 ? slot1.l.z : (m2 ? slot0.l.z : ab.l.z);
                }
                // the cache record [0:16): r_w, 1/D, U, S -- 16 readlanes of lane i's copy, or four 16-byte LDS reads
                float4 c0, c1, c2, c3;
                if constexpr (RECREG) {
                    auto rl = [&](const float v) -> float { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), i)); };
                    c0 = {rl(rq0.x), rl(rq0.y), rl(rq0.z), rl(rq0.w)}; c1 = {rl(rq1.x), rl(rq1.y), rl(rq1.z), rl(rq1.w)};
                    c2 = {rl(rq2.x), rl(rq2.y), rl(rq2.z), rl(rq2.w)}; c3 = {rl(rq3.x), rl(rq3.y), rl(rq3.z), rl(rq3.w)};
                } else {
                    c0 = *reinterpret_cast<const float4 *>(pc); c1 = *reinterpret_cast<const float4 *>(pc + 4);
                    c2 = *reinterpret_cast<const float4 *>(pc + 8); c3 = *reinterpret_cast<const float4 *>(pc + 12);
                }
                static_assert(C_RW == 0 && C_DINV == 3 && C_UW == 4 && C_SW == 10, "cache record layout");
                a.l = a.l + cross(a.a, v3(c0.x, c0.y, c0.z));        // the same acceleration at this link's origin
                // Y of this link: non-zero only if it is the own-path link of its depth d (wave-uniform)
                const int d = (int)((inf >> 8) & 0xffu);
                const bool on_path = (((i < 32 ? (unsigned)R.myanc : (unsigned)(R.myanc >> 32)) >> (i & 31)) & 1u) != 0u;
                float Y = 0.f;
                if constexpr (YSL) { const float y = ysL[(d - 1) * WAVE + lane]; Y = on_path ? y : 0.f; }      // own lane's value: no sync
                else { const float y = Yv[(d - 1) & 15]; Y = on_path ? y : 0.f; }
                if (R.two_body) {          // myanc2 != 0 only on contact-row lanes
                    const bool on2 = (((i < 32 ? (unsigned)R.myanc2 : (unsigned)(R.myanc2 >> 32)) >> (i & 31)) & 1u) != 0u;
                    if (on2) Y += Yd2L[(d - 1) * NC3 + col];
                }
                Sv Uw, Swv;
                Uw.a = v3(c1.x, c1.y, c1.z); Uw.l = v3(c1.w, c2.x, c2.y);
                Swv.a = v3(c2.z, c2.w, c3.x); Swv.l = v3(c3.y, c3.z, c3.w);
                const float qa = (Y - sdot(Uw, a)) * c0.w;
                a = sfma(Swv, qa, a);
                {
                    const int sv = (int)((inf >> 16) & 0xffu);      // slot + 1
                    if (NSL > 0 && sv != 0) {
                        if (sv == 1) slot0 = a;
                        else if (NSL > 1) slot1 = a;
                    }
                }
                if constexpr (!PATHONLY) {
                    if (i < 16) XM0[i % clamp16(NMX)] = qa;
                    else if (i < 32) XM1[(i - 16) % clamp16(NMX - 16)] = qa;
                    else if (i < 48) XM2[(i - 32) % clamp16(NMX - 32)] = qa;
                    else XM3[(i - 48) % clamp16(NMX - 48)] = qa;
                }
                if (unit_row && i == body) xdiag = R.unit_f * qa;
                if (nc > 0 && ((rowmask >> i) & 1ull)) {      // wave-uniform: rows of Jc off every contact path are zero
                    const float *jr = JcL + (6 + i) * NC3;
                    if constexpr (OVP) {
#pragma unroll
                        for (int c = C0; c < NC3; c++) XC[c] = fmaf(qa, jr[c], XC[c]);
                    } else if constexpr (NC3 % 4 == 0) {
#pragma unroll
                        for (int c = 0; c < NC3; c += 4) {
                            const float4 j4 = *reinterpret_cast<const float4 *>(jr + c);
                            XC[c] = fmaf(qa, j4.x, XC[c]); XC[c + 1] = fmaf(qa, j4.y, XC[c + 1]);
                            XC[c + 2] = fmaf(qa, j4.z, XC[c + 2]); XC[c + 3] = fmaf(qa, j4.w, XC[c + 3]);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < NC3; c++) XC[c] = fmaf(qa, jr[c], XC[c]);
                    }
                    if (col >= 0) xdiag = fmaf(qa, jr[col], xdiag);
                }
                }
            }
        };
        // row set 1 first -- its set-up values die here -- then the rows that have a lane
        if constexpr (DUAL) { if (nc > 0) response(IC<1>{}, R1, BM0, BM1, BM2, BM3, BC, diagB); }
        if constexpr (NOV > 0) {
            if (ovf_on) {
                response(IC<1>{}, R1, BM0, BM1, BM2, BM3, BC, diagB);
                const float inv = (R1.contact && diagB > (float)HG_ROW_DEN_EPS) ? __builtin_amdgcn_rcpf(diagB) : 0.f;      // else: row disabled
#pragma unroll
                for (int k = 0; k < NOV; k++) {
                    ov_b[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(R1.target_err), k));
                    ov_inv[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv), k));
#pragma unroll
                    for (int j = 0; j < NOV; j++)
                        ov_a[k][j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(BC[NOV > 0 ? NCL + j : 0]), k));
                }
            }
        }
        HG_TICK(15);      // end of row set 1's path-only response (4a')
        response(IC<0>{}, R0, AM0, AM1, AM2, AM3, AC, diag);
        if constexpr (DUAL) {
            // A[contact c][motor i] = A[motor i][contact c]: lane i's column c goes to lane c through LDS, in chunks of CH contact
            // columns that fit the own-path Y storage (dead since the responses ended): T[(c - c0) * NMX + i]
            static_assert(YSL, "two rows per lane keeps the own-path Y values in LDS: the transposition buffer");
            constexpr int CH = (DEPTH * WAVE) / NMX < NC3 ? (DEPTH * WAVE) / NMX : NC3;
            static_assert(CH >= 1, "transposition chunk");
            if (nc > 0) {
#pragma unroll
                for (int c0 = 0; c0 < NC3; c0 += CH) {
                    wave_sync();
                    if (lane < NMX) {
#pragma unroll
                        for (int c = 0; c < CH; c++) if (c0 + c < NC3) ysL[c * NMX + lane] = AC[c0 + c < NC3 ? c0 + c : 0];
                    }
                    wave_sync();
                    if (lane >= c0 && lane < c0 + CH && lane < NC3) {
                        const float *t = ysL + (lane - c0) * NMX;
#pragma unroll
                        for (int i = 0; i < NMX; i++) {
                            const float v = t[i];
                            if (i < 16) BM0[i % clamp16(NMX)] = v;
                            else if (i < 32) BM1[(i - 16) % clamp16(NMX - 16)] = v;
                            else if (i < 48) BM2[(i - 32) % clamp16(NMX - 32)] = v;
                            else BM3[(i - 48) % clamp16(NMX - 48)] = v;
                        }
                    }
                }
            }
        }
        const bool row_motor = R0.motor, row_slot = R0.slot;
        const bool row_contact = DUAL ? false : R0.contact;      // set-0 view (one row per lane): the solver below
        const int col = DUAL ? -1 : R0.col;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));          // fresh address arithmetic
        const float target_err = lamL[lane_e];
        {   // same values as before the pass
            const bool lk_e = lane_e < n;
            limit_row(lane_e, lk_e ? st[HG_STATE_BASE + lane_e] : 0.f, lk_e ? accL[lane_e] : 0.f, lim_s, lim_b);
        }
        const bool row_active = row_motor || row_contact || row_slot;

        HG_TICK(8);
        // The SIMD's instruction arbiter serves its resident waves in strict order: equal user priority ->
        // lowest wave slot first.  With one resident round (4096 NAO envs = 4 waves on every SIMD) the wave in
        // slot 0 runs at single-wave speed and the one in slot 3 finishes 1.5x later, alone on an idle SIMD.
        // Each wave therefore flips the order when it enters the solver here: slot 0 leads through the tree
        // phases, slot 3 through the sweeps, the four finish within 10 % of each other, and their phases stay
        // apart (tree sweeps of one wave under the solver of another).  Measured with tools/phase_ticks.py.
        // Launches of several resident rounds mix their phases by themselves and lose 2 % to the flip, but their tail has the
        // same problem: the host sets flip_from to 0 for launches that fit the GPU at once and to the last half round otherwise.
        if ((P.flags & F_PRIOFLIP) && disp >= P.flip_from) {
            const int wslot = __builtin_amdgcn_s_getreg(6148) & 3;      // HW_ID.wave_id[1:0]: slot of this wave on its SIMD
            // priority of slot w: bits [2w, 2w + 2) of the pattern byte in flags (0xE4 = 0, 1, 2, 3); odd substeps mirror it
            set_prio((int)((unsigned)P.flags >> (8 + 2 * ((sub & 1) ? 3 - wslot : wslot))));      // the next substep's tree phases keep this order, its solver flips it back
        }
        // ---- 4b. projected Gauss-Seidel on impulses ------------------------------------------
        // Per lane j: z_j = lam_j + (b_j - w_j) / A_jj, the unclamped candidate impulse of row j
        // (w = A lam).  Updating row i by d changes z_j by -A_ji d / A_jj for j != i and leaves
        // z_i unchanged, so the sweep needs one med3, one readlane and two fmas (- old, + new impulse) per
        // row.  Same iterates as Bullet's velocity-space sweep over deltaV.
        float z, lo = 0.f, hi = 0.f;
        float AOr[NOV > 0 ? NOV : 1];       // OVF: this lane's raw entries against the overflow rows, A[lane row][o]
#pragma unroll
        for (int k = 0; k < (NOV > 0 ? NOV : 1); k++) AOr[k] = NOV > 0 ? AC[NOV > 0 ? NCL + k : 0] : 0.f;
        {
            const float dinv = (row_active && diag > (float)HG_ROW_DEN_EPS) ? __builtin_amdgcn_rcpf(diag) : 0.f;      // else: row disabled (hg_model.h)
            z = target_err * dinv;
#pragma unroll
            for (int k = 0; k < NMX; k++) {
                if (k < 16) AM0[k] = (k == lane) ? 0.f : -AM0[k] * dinv;
                else if (k < 32) AM1[k - 16] = (k == lane) ? 0.f : -AM1[k - 16] * dinv;
                else if (k < 48) AM2[k - 32] = (k == lane) ? 0.f : -AM2[k - 32] * dinv;
                else AM3[k - 48] = (k == lane) ? 0.f : -AM3[k - 48] * dinv;
            }
#pragma unroll
            for (int k = 0; k < NC3; k++) AC[k] = (!DUAL && NMX + k == lane) ? 0.f : -AC[k] * dinv;
            if (row_motor) { hi = cacheL[CACHE * lane + C_MAXIMP]; lo = -hi; }
            else if (row_contact && col < MAXC) { lo = 0.f; hi = 1e10f; }
            lim_b *= dinv;          // start value of the limit row's candidate: b_l / A_ii
        }
        float lam_l = 0.f;
        float lam = 0.f;            // impulse of this lane's (set-0) row once the sweeps are done

        if constexpr (NROWS <= WAVE) {
        // The slot rows bring one more column each: aL[k][j] = -A[j][L_k] / A[j][j] = slot_sign * (column i_k),
        // except on the joint's own motor lane (-slot_sign: same Jacobian entry, A[i][L] = sign * A[i][i]) and on
        // the slot lane itself (0, the diagonal).
        float aL[KL > 0 ? KL : 1];
        int slamL[KL > 0 ? KL : 1];
#pragma unroll
        for (int k = 0; k < (KL > 0 ? KL : 1); k++) { aL[k] = 0.f; slamL[k] = 0; }
        if constexpr (KL > 0) {
            if (use_slots) {
                if (row_slot) { lo = 0.f; hi = P.limhi; }
#pragma unroll
                for (int k = 0; k < KL; k++) {
                    if (k < nlim) {
                        const int i = slot_joint[k];
                        const float sg = slot_sign[k];
                        float col;
                        if constexpr (NMX <= 16) col = AM0[i];
                        else if constexpr (NMX <= 32) col = i < 16 ? AM0[i & 15] : AM1[(i - 16) % clamp16(NMX - 16)];
                        else col = i < 16 ? AM0[i & 15] : (i < 32 ? AM1[(i - 16) & 15] : AM2[(i - 32) % clamp16(NMX - 32)]);
                        aL[k] = lane == i ? -sg : (lane == NROWS + k ? 0.f : sg * col);
                    }
                }
            }
        }

        // The impulses live in scalar registers (slam[i] = bits of lam_i): a row update is then
        //   cand = med3(z, lo, hi);  s_new = readlane(cand, i);  z -= A_i * s_old;  z += A_i * s_new
        // -- four vector instructions and no write to a single lane.
        int slam[NROWS];
#pragma unroll
        for (int k = 0; k < NROWS; k++) slam[k] = 0;
        float mu_s = P.mu;
        asm volatile("" : "+s"(mu_s));          // opaque: the copy below must not be hoisted out of the substep loop
        float mu_v = mu_s;
        asm volatile("" : "+v"(mu_v));          // keep mu in a VGPR: v_mul may take only one SGPR operand
        auto row_update = [&](const int i, const float rlo, const float rhi) {
            const float cand = __builtin_amdgcn_fmed3f(z, rlo, rhi);
            const int s_new = __builtin_amdgcn_readlane(__float_as_int(cand), i);
            float a_i;          // i is a compile-time constant at every call site (unrolled sweeps)
            if (i >= NMX) a_i = AC[i >= NMX ? i - NMX : 0];
            else if (i < 16) a_i = AM0[i < 16 ? i : 0];
            else if (i < 32) a_i = AM1[i >= 16 && i < 32 && i < NMX ? i - 16 : 0];
            else a_i = AM2[i >= 32 && i < NMX ? i - 32 : 0];
            z = fmaf(-a_i, __int_as_float(slam[i]), z);
            z = fmaf(a_i, __int_as_float(s_new), z);
            slam[i] = s_new;
        };
        auto contact_rows = [&]() {
            if (nc > 0) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < nc) row_update(NMX + c, lo, hi);
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < nc) {
                        const int tot_i = slam[NMX + c];
                        if (tot_i > 0) {                      // tot > 0 on the scalar unit (positive float bits)
                            const float lim = mu_v * __int_as_float(tot_i);
                            row_update(NMX + MAXC + 2 * c, -lim, lim);
                            row_update(NMX + MAXC + 2 * c + 1, -lim, lim);
                        }
                    }
            }
        };
        // Bullet's row order: motor rows back to front on even sweeps and front to back on odd ones,
        // then the contact normals, then the friction pairs of contacts that carry a normal impulse.
        auto slot_update = [&](const int k) {              // limit row of slot k, lane NROWS + k
            const float cand = __builtin_amdgcn_fmed3f(z, lo, hi);
            const int s_new = __builtin_amdgcn_readlane(__float_as_int(cand), NROWS + (k < KL ? k : 0));
            z = fmaf(-aL[k], __int_as_float(slamL[k]), z);
            z = fmaf(aL[k], __int_as_float(s_new), z);
            slamL[k] = s_new;
        };
        // (Bullet's non-contact list is [limit rows..., motor rows...], walked back to front on even sweeps)
        auto sweep = [&](const bool forward) {
            if (forward) {
                if constexpr (KL > 0) {
                    if (use_slots) {
#pragma unroll
                        for (int k = 0; k < KL; ++k) if (k < nlim) slot_update(k);
                    }
                }
#pragma unroll
                for (int i = 0; i < NMX; ++i) row_update(i, lo, hi);
            } else {
#pragma unroll
                for (int i = NMX - 1; i >= 0; --i) row_update(i, lo, hi);
                if constexpr (KL > 0) {
                    if (use_slots) {
#pragma unroll
                        for (int k = KL - 1; k >= 0; --k) if (k < nlim) slot_update(k);
                    }
                }
            }
            contact_rows();
        };
        // two sweeps per trip, so that the loop body is straight-line code in the motor rows
        if (limmask == 0ull || use_slots) {
            #pragma unroll 1
            for (int it = 0; it + 1 < P.iters; it += 2) { sweep(false); sweep(true); }
            if (P.iters & 1) sweep(false);
        } else {
            // Some joint of this env is beyond a bound (rare, so not tuned).  A limit row lives on its
            // joint's motor lane.  With d the change of the JOINT impulse (lim_s * change of lam_l): every
            // other row sees it like a motor impulse on column i, the motor row of joint i itself loses d
            // from its candidate.  The motor impulses are mirrored in a VGPR (lane i: lam of motor i) so
            // that a run-time loop over the violated joints can read them.
            float lamv = 0.f;
            const float lim_k = lim_b - lim_s * z;      // z_l = lam_l + lim_k + lim_s * (z - lam_motor); z is still b_m / A_ii here
            auto motor_update_v = [&](const int i) {
                const float cand = __builtin_amdgcn_fmed3f(z, lo, hi);
                const int s_new = __builtin_amdgcn_readlane(__float_as_int(cand), i);
                float a_i;
                if (i < 16) a_i = AM0[i < 16 ? i : 0];
                else if (i < 32) a_i = AM1[i >= 16 && i < 32 && i < NMX ? i - 16 : 0];
                else a_i = AM2[i >= 32 && i < NMX ? i - 32 : 0];
                z = fmaf(-a_i, __int_as_float(slam[i]), z);
                z = fmaf(a_i, __int_as_float(s_new), z);
                slam[i] = s_new;
                asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %3\n\tv_mov_b32 %0, %1\n\ts_mov_b64 exec, -1"
                             : "+v"(lamv) : "v"(cand), "i"(i < 32 ? (int)(1u << (i & 31)) : 0), "i"(i >= 32 ? (int)(1u << (i & 31)) : 0));
            };
            auto limit_update = [&](const int i) {             // i: wave-uniform, known at run time
                const float zl = lam_l + lim_k + lim_s * (z - lamv);
                const float cand = __builtin_amdgcn_fmed3f(zl, 0.f, P.limhi);
                const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lim_s * (cand - lam_l)), i));
                float a_i;
                if constexpr (NMX <= 16) a_i = AM0[i];
                else if constexpr (NMX <= 32) a_i = i < 16 ? AM0[i & 15] : AM1[(i - 16) % clamp16(NMX - 16)];
                else a_i = i < 16 ? AM0[i & 15] : (i < 32 ? AM1[(i - 16) & 15] : AM2[(i - 32) % clamp16(NMX - 32)]);
                z = fmaf(a_i, d, z);                            // a_ii = 0: lane i is handled next
                if (lane == i) { lam_l = cand; z -= d; }
            };
            // Bullet's non-contact list is [limit rows..., motor rows...], walked back to front on even sweeps
            auto sweep_limits = [&](const bool forward) {
                if (forward) {
                    #pragma unroll 1
                    for (unsigned long long mk = limmask; mk != 0ull; mk &= mk - 1ull) limit_update(__builtin_ctzll(mk));
#pragma unroll
                    for (int i = 0; i < NMX; ++i) motor_update_v(i);
                } else {
#pragma unroll
                    for (int i = NMX - 1; i >= 0; --i) motor_update_v(i);
                    #pragma unroll 1
                    for (unsigned long long mk = limmask; mk != 0ull;) {
                        const int i = 63 - __builtin_clzll(mk);
                        limit_update(i);
                        mk &= ~(1ull << i);
                    }
                }
                contact_rows();
            };
            #pragma unroll 1
            for (int it = 0; it + 1 < P.iters; it += 2) { sweep_limits(false); sweep_limits(true); }
            if (P.iters & 1) sweep_limits(false);
        }
        HG_TICK(9);
#pragma unroll
        for (int k = 0; k < NROWS; k++)       // lane k <- lam_k: one v_mov under a one-lane EXEC (all 64 lanes are active here)
            asm volatile("s_mov_b32 exec_lo, %2\n\ts_mov_b32 exec_hi, %3\n\tv_mov_b32 %0, %1\n\ts_mov_b64 exec, -1"
                         : "+v"(lam) : "s"(slam[k]), "i"(k < 32 ? (int)(1u << (k & 31)) : 0), "i"(k >= 32 ? (int)(1u << (k & 31)) : 0));
        if constexpr (KL > 0) {
            if (use_slots) {
#pragma unroll
                for (int k = 0; k < KL; k++)
                    if (k < nlim && lane == slot_joint[k]) lam_l = __int_as_float(slamL[k]);
            }
        }
        lamL[lane] = fmaf(lim_s, lam_l, lam);      // joint impulse = motor row + (signed) limit row; contact lanes: their own impulse
        } else if constexpr (NOV > 0) {
        // ---- 4b, one row per lane plus overflow rows.  The impulses of the rows that have a lane live in a VGPR (lane j: the
        // impulse of its row; 64 of them do not fit the scalar register file) and a row update broadcasts the CHANGE of its impulse:
        //   cand = med3(z, lo, hi);  s_d = readlane(cand - lam, i);  z += A_i s_d;  lam_i = cand  (one-lane EXEC move)
        // Overflow rows: an overflow row o is updated from
        //   z_o = (b_o - sum_j A[j][o] lam_j - sum_{o' != o} A[o][o'] lam_o') / A[o][o],
        // the first sum over the rows with a lane -- one multiply per lane and a wave reduction (DPP) -- and its change is
        // then applied to every lane's candidate like any other row's.  Same iterates, same row order as Bullet's sweep (the
        // friction pair of the last contact slot is the last pair of the sweep).  Violated joint limits take the run-time
        // loop (a limit row lives on its joint's motor lane, see the one-row-per-lane solver above for the algebra).
        float lamv0 = 0.f;
        float lamo[NOV];
#pragma unroll
        for (int k = 0; k < NOV; k++) lamo[k] = 0.f;
        float mu_s = P.mu;
        asm volatile("" : "+s"(mu_s));
        float mu_v = mu_s;
        asm volatile("" : "+v"(mu_v));
        const float lim_k = lim_b - lim_s * z;      // z is still b_m / A_ii here
        auto colA = [&](const int i) -> float {     // compile-time i
            if (i < 16) return AM0[i < 16 ? i : 0];
            else if (i < 32) return AM1[i >= 16 && i < 32 && i < NMX ? i - 16 : 0];
            else if (i < 48) return AM2[i >= 32 && i < 48 && i < NMX ? i - 32 : 0];
            else return AM3[i >= 48 && i < NMX ? i - 48 : 0];
        };
        // One row update as one block of seven issue slots (a lone wave issues an instruction every 5-6 clocks whatever its
        // kind, and the waves of these shapes run nearly alone: slots are what counts).  Wait states of gfx950 respected by
        // hand: one between a VALU write of a VGPR and v_readlane of it, two between v_readlane's SGPR and its VALU use;
        // lam_i = cand on lane i alone is a select under a one-lane mask built on the scalar unit, placed in those gaps.
        auto lane_row_update = [&](const int i, const float a_i, const float rlo, const float rhi) {      // row on lane i (compile-time)
            float cand, d; int sd; unsigned long long m1;
            asm("v_med3_f32 %2, %0, %7, %8\n\tv_sub_f32 %3, %2, %1\n\ts_lshl_b64 %5, 1, %9\n\tv_readlane_b32 %4, %3, %9\n\t"
                "v_cndmask_b32_e64 %1, %1, %2, %5\n\ts_nop 0\n\tv_fmac_f32 %0, %4, %6"
                : "+v"(z), "+v"(lamv0), "=&v"(cand), "=&v"(d), "=&s"(sd), "=&s"(m1) : "v"(a_i), "v"(rlo), "v"(rhi), "n"(i) : "scc");      // s_lshl writes SCC
        };
        auto lane_row_update_sym = [&](const int i, const float a_i, const float lim) {      // bounds -lim, lim
            float cand, d; int sd; unsigned long long m1;
            asm("v_med3_f32 %2, %0, -%7, %7\n\tv_sub_f32 %3, %2, %1\n\ts_lshl_b64 %5, 1, %8\n\tv_readlane_b32 %4, %3, %8\n\t"
                "v_cndmask_b32_e64 %1, %1, %2, %5\n\ts_nop 0\n\tv_fmac_f32 %0, %4, %6"
                : "+v"(z), "+v"(lamv0), "=&v"(cand), "=&v"(d), "=&s"(sd), "=&s"(m1) : "v"(a_i), "v"(lim), "n"(i) : "scc");
        };
        // Several rows on consecutive lanes as ONE block of assembly (hipcc puts a wait state between two asm statements):
        // lanes i0, i0 +- 1, ... with the columns a0 .. ; forward or backward.
        auto rows6 = [&](const int i0, const bool fwd, const float a0, const float a1, const float a2, const float a3, const float a4,
                         const float a5, const float rlo, const float rhi) {
            float cand, d; int sd; unsigned long long m1;
            if (fwd)
                asm(HG_ROWS6_ASM("+") : [z] "+v"(z), [l] "+v"(lamv0), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                    : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5), [lo] "v"(rlo), [hi] "v"(rhi), [i0] "n"(i0) : "scc");
            else
                asm(HG_ROWS6_ASM("-") : [z] "+v"(z), [l] "+v"(lamv0), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                    : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [a4] "v"(a4), [a5] "v"(a5), [lo] "v"(rlo), [hi] "v"(rhi), [i0] "n"(i0) : "scc");
        };
        auto rows2 = [&](const int i0, const bool fwd, const float a0, const float a1, const float rlo, const float rhi) {
            float cand, d; int sd; unsigned long long m1;
            if (fwd)
                asm(HG_ROWS2_ASM("+") : [z] "+v"(z), [l] "+v"(lamv0), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                    : [a0] "v"(a0), [a1] "v"(a1), [lo] "v"(rlo), [hi] "v"(rhi), [i0] "n"(i0) : "scc");
            else
                asm(HG_ROWS2_ASM("-") : [z] "+v"(z), [l] "+v"(lamv0), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                    : [a0] "v"(a0), [a1] "v"(a1), [lo] "v"(rlo), [hi] "v"(rhi), [i0] "n"(i0) : "scc");
        };
        // rows on lanes [b, b + cnt) with the columns col(k), k = 0 .. cnt - 1, in blocks of 6, 2 and 1; ascending or descending
        auto row_run = [&](const int b, const int cnt, const bool fwd, auto col) __attribute__((always_inline)) {
            int k = 0;
#pragma unroll
            for (; k + 6 <= cnt; k += 6) {
                const int j = fwd ? k : cnt - 1 - k, sg = fwd ? 1 : -1;
                rows6(b + j, fwd, col(j), col(j + sg), col(j + 2 * sg), col(j + 3 * sg), col(j + 4 * sg), col(j + 5 * sg), lo, hi);
            }
#pragma unroll
            for (; k + 2 <= cnt; k += 2) {
                const int j = fwd ? k : cnt - 1 - k, sg = fwd ? 1 : -1;
                rows2(b + j, fwd, col(j), col(j + sg), lo, hi);
            }
#pragma unroll
            for (; k < cnt; k += 1) {
                const int j = fwd ? k : cnt - 1 - k;
                lane_row_update(b + j, col(j), lo, hi);
            }
        };
        auto overflow_pair = [&](const int k, const float lim) {      // overflow rows k, k + 1: the friction pair of one contact
            const float lt = fmaf(lim_s, lam_l, lamv0);               // joint impulse on motor lanes, the row's own elsewhere
            // both sums in one reduction: v_permlane32_swap puts the partial products of row k on lanes 0-31 and those of
            // row k + 1 on lanes 32-63, each half is then summed on its own
            float S0, S1;
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(AOr[k] * lt), __float_as_uint(AOr[k + 1] * lt), false, false);
                float x = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x128, 0xf, 0xf, false));      // row_ror:8
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x124, 0xf, 0xf, false));      // row_ror:4
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x122, 0xf, 0xf, false));      // row_ror:2
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x121, 0xf, 0xf, false));      // row_ror:1
                x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x142, 0xa, 0xf, false));      // row_bcast:15 into rows 1, 3
                S0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 31));
                S1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
            }
#pragma unroll
            for (int r = 0; r < 2; r++) {
                float acc = ov_b[k + r] - (r ? S1 : S0);
#pragma unroll
                for (int j = 0; j < NOV; j++) if (j != k + r) acc = fmaf(-ov_a[k + r][j], lamo[j], acc);
                const float cand = __builtin_amdgcn_fmed3f(acc * ov_inv[k + r], -lim, lim);
                const float d = cand - lamo[k + r];
                lamo[k + r] = cand;
                z = fmaf(AC[NCL + k + r], d, z);
            }
        };
        auto friction_pair = [&](const int c, const float lim) __attribute__((always_inline)) {      // compile-time c
            if (MAXC + 2 * c + 1 < NCL) {
                lane_row_update_sym(NMX + MAXC + 2 * c, AC[MAXC + 2 * c < NCL ? MAXC + 2 * c : 0], lim);
                lane_row_update_sym(NMX + MAXC + 2 * c + 1, AC[MAXC + 2 * c + 1 < NCL ? MAXC + 2 * c + 1 : 0], lim);
            } else {
                overflow_pair(MAXC + 2 * c >= NCL ? MAXC + 2 * c - NCL : 0, lim);
            }
        };
        auto contact_rows = [&]() {
            if (nc == MAXC) {
                // every slot taken (a robot standing on two feet): the normals as one run, no slot tests.  (The test of the
                // normal impulse stays: a friction pair under a normal impulse of 0 KEEPS its impulse, as in Bullet's sweep.)
                row_run(NMX, MAXC, true, [&](const int c) -> float { return AC[c >= 0 && c < MAXC ? c : 0]; });
#pragma unroll
                for (int c = 0; c < MAXC; ++c) {
                    const int tot_i = __builtin_amdgcn_readlane(__float_as_int(lamv0), NMX + c);
                    if (tot_i > 0) friction_pair(c, mu_v * __int_as_float(tot_i));
                }
            } else if (nc > 0) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < nc) lane_row_update(NMX + c, AC[c], lo, hi);
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < nc) {
                        const int tot_i = __builtin_amdgcn_readlane(__float_as_int(lamv0), NMX + c);
                        if (tot_i > 0)                        // tot > 0 on the scalar unit (positive float bits)
                            friction_pair(c, mu_v * __int_as_float(tot_i));
                    }
            }
        };
        auto limit_update_idx = [&](const int i) __attribute__((always_inline)) {   // i: wave-uniform, known at run time
            const float zl = lam_l + lim_k + lim_s * (z - lamv0);
            const float cand = __builtin_amdgcn_fmed3f(zl, 0.f, P.limhi);
            const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lim_s * (cand - lam_l)), i));
            const int j = i & 15;
            float a_i;
            if (i < 16) a_i = AM0[j % clamp16(NMX)];
            else if (i < 32) a_i = AM1[j % clamp16(NMX - 16)];
            else if (i < 48) a_i = AM2[j % clamp16(NMX - 32)];
            else a_i = AM3[j % clamp16(NMX - 48)];
            z = fmaf(a_i, d, z);                            // a_ii = 0: lane i is handled next
            if (lane == i) { lam_l = cand; z -= d; }
        };
        // Up to KLG violated joint limits (the finger-joint models rest a few fingers ON a bound: the common case): the column
        // of limit row k -- that of its joint's motor row, with -1 on the joint's own lane, which takes the change of the
        // JOINT impulse off the motor row's candidate -- is fetched ONCE per step with an indexed register read; the sweep then
        // runs the rows with compile-time k.  More violated joints than that: the run-time loop with indexed reads per sweep.
        constexpr int KLG = 4;
        const int nlim_g = __popcll(limmask);
        const bool lim_gathered = nlim_g > 0 && nlim_g <= KLG;        // wave-uniform
        float alim[KLG];
        int lim_joint[KLG];
        unsigned long long lim_bit[KLG];
#pragma unroll
        for (int k = 0; k < KLG; k++) { alim[k] = 0.f; lim_joint[k] = 0; lim_bit[k] = 0ull; }
        if (lim_gathered) {
            unsigned long long mk = limmask;
#pragma unroll
            for (int k = 0; k < KLG; k++) {
                if (k < nlim_g) {
                    const int i = __builtin_ctzll(mk);
                    mk &= mk - 1ull;
                    const int j = i & 15;
                    float a_i;
                    if (i < 16) a_i = AM0[j % clamp16(NMX)];
                    else if (i < 32) a_i = AM1[j % clamp16(NMX - 16)];
                    else if (i < 48) a_i = AM2[j % clamp16(NMX - 32)];
                    else a_i = AM3[j % clamp16(NMX - 48)];
                    alim[k] = lane == i ? -1.f : a_i;
                    lim_joint[k] = i; lim_bit[k] = 1ull << i;
                }
            }
        }
        auto limit_update_g = [&](const int k) {           // compile-time k; the row lives on lane lim_joint[k]
            const float zl = fmaf(lim_s, z - lamv0, lam_l) + lim_k;
            const float cand = __builtin_amdgcn_fmed3f(zl, 0.f, P.limhi);
            const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lim_s * (cand - lam_l)), lim_joint[k]));
            z = fmaf(alim[k], d, z);
            asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(lam_l) : "v"(cand), "s"(lim_bit[k]));
        };
        auto sweep3 = [&](const bool forward) {
            if (forward) {
                if (lim_gathered) {
#pragma unroll
                    for (int k = 0; k < KLG; ++k) if (k < nlim_g) limit_update_g(k);
                } else {
                    #pragma unroll 1
                    for (unsigned long long mk = limmask; mk != 0ull; mk &= mk - 1ull) limit_update_idx(__builtin_ctzll(mk));
                }
                row_run(0, NMX, true, colA);
            } else {
                row_run(0, NMX, false, colA);
                if (lim_gathered) {
#pragma unroll
                    for (int k = KLG - 1; k >= 0; --k) if (k < nlim_g) limit_update_g(k);
                } else {
                    #pragma unroll 1
                    for (unsigned long long mk = limmask; mk != 0ull;) {
                        const int i = 63 - __builtin_clzll(mk);
                        limit_update_idx(i);
                        mk &= ~(1ull << i);
                    }
                }
            }
            contact_rows();
        };
        #pragma unroll 1
        for (int it = 0; it + 1 < P.iters; it += 2) { sweep3(false); sweep3(true); }
        if (P.iters & 1) sweep3(false);
        HG_TICK(9);
        lam = lamv0;
        lamL[lane] = fmaf(lim_s, lam_l, lamv0);      // joint impulse = motor row + (signed) limit row; contact lanes: their own impulse
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NOV; k++) lamL[NMX + NCL + k] = lamo[k];
        }
        } else {
        // ---- 4b, two rows per lane.  The impulses live in VGPRs (lane i of lamv0: motor i, lane c of lamv1: contact
        // row c; 66-98 of them would not fit the scalar register file), and a row update broadcasts the CHANGE of its impulse:
        //   cand = med3(z, lo, hi);  d = cand - lam;  s_d = readlane(d, i);  z0 += A0_i s_d;  z1 += A1_i s_d;  lam_i = cand
        // Same iterates, same row order.  Violated joint limits always take the run-time loop (a limit row lives on its
        // joint's motor lane, see the one-row-per-lane solver above for the algebra).
        float z1, lo1 = 0.f, hi1 = 0.f;
        {
            const float dinvB = (R1.contact && diagB > (float)HG_ROW_DEN_EPS) ? __builtin_amdgcn_rcpf(diagB) : 0.f;
            z1 = R1.target_err * dinvB;
#pragma unroll
            for (int k = 0; k < NMX; k++) {
                if (k < 16) BM0[k] = -BM0[k] * dinvB;
                else if (k < 32) BM1[k - 16] = -BM1[k - 16] * dinvB;
                else if (k < 48) BM2[k - 32] = -BM2[k - 32] * dinvB;
                else BM3[k - 48] = -BM3[k - 48] * dinvB;
            }
#pragma unroll
            for (int k = 0; k < NC3; k++) BC[k] = (k == lane) ? 0.f : -BC[k] * dinvB;
            if (R1.contact && R1.col < MAXC) { lo1 = 0.f; hi1 = 1e10f; }
        }
        float lamv0 = 0.f, lamv1 = 0.f;
        float mu_s = P.mu;
        asm volatile("" : "+s"(mu_s));
        float mu_v = mu_s;
        asm volatile("" : "+v"(mu_v));
        const float lim_k = lim_b - lim_s * z;      // z is still b_m / A_ii here
        auto colA = [&](const int i) -> float {     // compile-time i
            if (i < 16) return AM0[i < 16 ? i : 0];
            else if (i < 32) return AM1[i >= 16 && i < 32 && i < NMX ? i - 16 : 0];
            else if (i < 48) return AM2[i >= 32 && i < 48 && i < NMX ? i - 32 : 0];
            else return AM3[i >= 48 && i < NMX ? i - 48 : 0];
        };
        auto colB = [&](const int i) -> float {
            if (i < 16) return BM0[i < 16 ? i : 0];
            else if (i < 32) return BM1[i >= 16 && i < 32 && i < NMX ? i - 16 : 0];
            else if (i < 48) return BM2[i >= 32 && i < 48 && i < NMX ? i - 32 : 0];
            else return BM3[i >= 48 && i < NMX ? i - 48 : 0];
        };
        // Row updates as blocks of assembly, eight issue slots per row (see the one-row-per-lane solver with overflow rows
        // above for the wait states and the one-lane select): the row's change goes to both candidates.
        auto motor_update = [&](const int i) {
            float cand, d; int sd; unsigned long long m1;
            asm(HG_DROW_ASM(0, "+") : [z] "+v"(z), [z1] "+v"(z1), [l] "+v"(lamv0), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                : [a0] "v"(colA(i)), [b0] "v"(colB(i)), [lo] "v"(lo), [hi] "v"(hi), [i0] "n"(i) : "scc");
        };
        auto motor_rows4 = [&](const int i0, const bool fwd) {      // rows i0, i0 +- 1, i0 +- 2, i0 +- 3
            float cand, d; int sd; unsigned long long m1;
            const int sg = fwd ? 1 : -1;
            if (fwd)
                asm(HG_DROWS4_ASM("+") : [z] "+v"(z), [z1] "+v"(z1), [l] "+v"(lamv0), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                    : [a0] "v"(colA(i0)), [a1] "v"(colA(i0 + sg)), [a2] "v"(colA(i0 + 2 * sg)), [a3] "v"(colA(i0 + 3 * sg)),
                      [b0] "v"(colB(i0)), [b1] "v"(colB(i0 + sg)), [b2] "v"(colB(i0 + 2 * sg)), [b3] "v"(colB(i0 + 3 * sg)),
                      [lo] "v"(lo), [hi] "v"(hi), [i0] "n"(i0) : "scc");
            else
                asm(HG_DROWS4_ASM("-") : [z] "+v"(z), [z1] "+v"(z1), [l] "+v"(lamv0), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                    : [a0] "v"(colA(i0)), [a1] "v"(colA(i0 + sg)), [a2] "v"(colA(i0 + 2 * sg)), [a3] "v"(colA(i0 + 3 * sg)),
                      [b0] "v"(colB(i0)), [b1] "v"(colB(i0 + sg)), [b2] "v"(colB(i0 + 2 * sg)), [b3] "v"(colB(i0 + 3 * sg)),
                      [lo] "v"(lo), [hi] "v"(hi), [i0] "n"(i0) : "scc");
        };
        auto motor_run = [&](const bool fwd) __attribute__((always_inline)) {      // all NMX motor rows, ascending or descending
            int k = 0;
#pragma unroll
            for (; k + 4 <= NMX; k += 4) motor_rows4(fwd ? k : NMX - 1 - k, fwd);
#pragma unroll
            for (; k < NMX; k += 1) motor_update(fwd ? k : NMX - 1 - k);
        };
        auto contact_update = [&](const int c, const float rlo, const float rhi) {       // contact row c: lane c of row set 1
            float cand, d; int sd; unsigned long long m1;
            asm("v_med3_f32 %[c], %[z1], %[lo], %[hi]\n\tv_sub_f32 %[d], %[c], %[l]\n\ts_lshl_b64 %[m], 1, %[i0]\n\tv_readlane_b32 %[s], %[d], %[i0]\n\t"
                "v_cndmask_b32_e64 %[l], %[l], %[c], %[m]\n\ts_nop 0\n\tv_fmac_f32 %[z], %[s], %[a0]\n\tv_fmac_f32 %[z1], %[s], %[b0]"
                : [z] "+v"(z), [z1] "+v"(z1), [l] "+v"(lamv1), [c] "=&v"(cand), [d] "=&v"(d), [s] "=&s"(sd), [m] "=&s"(m1)
                : [a0] "v"(AC[c]), [b0] "v"(BC[c]), [lo] "v"(rlo), [hi] "v"(rhi), [i0] "n"(c) : "scc");
        };
        auto contact_rows = [&]() {
            if (nc > 0) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < nc) contact_update(c, lo1, hi1);
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < nc) {
                        const int tot_i = __builtin_amdgcn_readlane(__float_as_int(lamv1), c);
                        if (tot_i > 0) {                      // tot > 0 on the scalar unit (positive float bits)
                            const float lim = mu_v * __int_as_float(tot_i);
                            contact_update(MAXC + 2 * c, -lim, lim);
                            contact_update(MAXC + 2 * c + 1, -lim, lim);
                        }
                    }
            }
        };
        // The finger-joint models rest most of their fingers ON a bound (q = 0 = lower): 2 (NAO) to 7 (Romeo) limit rows per
        // env and sweep are the common case here.  Measured on one box (tools/hands_limits_ab.py; HEAD = indexed reads only):
        //   61-dof Romeo, 7.2 such rows per env: indexed reads 1.368 ms per step, scan 1.231 ms, compare tree into the per-joint code 1.372 ms
        //   42-dof NAO,   2.0 such rows per env: indexed reads 0.374 ms,          scan 0.427 ms
        // "scan" walks ALL joints with compile-time indices and a scalar test of each joint's mask bit (columns in named
        // registers, one-lane writes under an immediate EXEC); "indexed reads" loops over the set bits and fetches the joint's
        // column with an indexed register read.  Same rows, same order either way: scan beyond 48 joints, indexed reads up to 48.
        auto limit_update_at = [&](const int i) {          // compile-time i
            const float zl = fmaf(lim_s, z - lamv0, lam_l + lim_k);
            const float cand = __builtin_amdgcn_fmed3f(zl, 0.f, P.limhi);
            const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lim_s * (cand - lam_l)), i));
            z = fmaf(colA(i), d, z);                        // a_ii = 0: lane i is handled below
            z1 = fmaf(colB(i), d, z1);
            asm volatile("s_mov_b32 exec_lo, %4\n\ts_mov_b32 exec_hi, %5\n\tv_mov_b32 %0, %2\n\tv_subrev_f32 %1, %3, %1\n\ts_mov_b64 exec, -1"
                         : "+v"(lam_l), "+v"(z) : "v"(cand), "s"(d), "i"(i < 32 ? (int)(1u << (i & 31)) : 0), "i"(i >= 32 ? (int)(1u << (i & 31)) : 0));
        };
        auto limit_update_idx = [&](const int i) __attribute__((always_inline)) {   // i: wave-uniform, known at run time
            const float zl = lam_l + lim_k + lim_s * (z - lamv0);
            const float cand = __builtin_amdgcn_fmed3f(zl, 0.f, P.limhi);
            const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lim_s * (cand - lam_l)), i));
            const int j = i & 15;
            float a_i, b_i;
            if (i < 16) { a_i = AM0[j % clamp16(NMX)]; b_i = BM0[j % clamp16(NMX)]; }
            else if (i < 32) { a_i = AM1[j % clamp16(NMX - 16)]; b_i = BM1[j % clamp16(NMX - 16)]; }
            else if (i < 48) { a_i = AM2[j % clamp16(NMX - 32)]; b_i = BM2[j % clamp16(NMX - 32)]; }
            else { a_i = AM3[j % clamp16(NMX - 48)]; b_i = BM3[j % clamp16(NMX - 48)]; }
            z = fmaf(a_i, d, z);                            // a_ii = 0: lane i is handled next
            z1 = fmaf(b_i, d, z1);
            if (lane == i) { lam_l = cand; z -= d; }
        };
        static_assert(NMX <= 64, "limit mask");
        constexpr bool LIM_SCAN = NMX > 48;
        const unsigned lim_lo32 = (unsigned)limmask, lim_hi32 = (unsigned)(limmask >> 32);
        auto sweep2 = [&](const bool forward) {
            // the mask words are made opaque per sweep: hipcc would otherwise hoist every joint's bit test out of the sweep loop as a
            // 64-bit condition mask, 61 SGPR pairs that it then parks in VGPR lanes and reads back with two v_readlane per test
            unsigned lo32 = lim_lo32, hi32 = lim_hi32;
            asm volatile("" : "+s"(lo32), "+s"(hi32));
            if (forward) {
                if constexpr (LIM_SCAN) {
                    if (limmask != 0ull) {
                        // (groups of 8 joints under one test of their mask byte: the violated joints are the fingers, neighbours in
                        //  the joint order, so most groups are skipped with one test instead of eight)
#pragma unroll
                        for (int g = 0; g < NMX; g += 8) {
                            if (((g < 32 ? lo32 : hi32) >> (g & 31)) & 0xffu) {
#pragma unroll
                                for (int i = g; i < g + 8; ++i)
                                    if (i < NMX && (((i < 32 ? lo32 : hi32) >> (i & 31)) & 1u)) limit_update_at(i);
                            }
                        }
                    }
                } else {
                    #pragma unroll 1
                    for (unsigned long long mk = limmask; mk != 0ull; mk &= mk - 1ull) limit_update_idx(__builtin_ctzll(mk));
                }
                motor_run(true);
            } else {
                motor_run(false);
                if constexpr (LIM_SCAN) {
                    if (limmask != 0ull) {
#pragma unroll
                        for (int g = (NMX - 1) / 8 * 8; g >= 0; g -= 8) {
                            if (((g < 32 ? lo32 : hi32) >> (g & 31)) & 0xffu) {
#pragma unroll
                                for (int i = g + 7; i >= g; --i)
                                    if (i < NMX && (((i < 32 ? lo32 : hi32) >> (i & 31)) & 1u)) limit_update_at(i);
                            }
                        }
                    }
                } else {
                    #pragma unroll 1
                    for (unsigned long long mk = limmask; mk != 0ull;) {
                        const int i = 63 - __builtin_clzll(mk);
                        limit_update_idx(i);
                        mk &= ~(1ull << i);
                    }
                }
            }
            contact_rows();
        };
        #pragma unroll 1
        for (int it = 0; it + 1 < P.iters; it += 2) { sweep2(false); sweep2(true); }
        if (P.iters & 1) sweep2(false);
        HG_TICK(9);
        lam = lamv0;
        lamL[lane] = fmaf(lim_s, lam_l, lamv0);      // joint impulse = motor row + (signed) limit row
        if (lane < NC3) lamL[WAVE + lane] = lamv1;   // contact impulses
        }

        {   // getJointState()[3]: the motor torque applied in this physics step (dancer_env.py:24)
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));      // fresh address arithmetic
            if (O.torque && lane_t < n) O.torque[(size_t)env * n + lane_t] = lam * P.inv_dt;
        }

        // ============ 5. velocities += M^-1 J^T lambda =========================================
        wave_sync();      // all reads of Jc by other lanes' Arow accumulation are done as well
        V3 bang = ld3(velL), bvel = ld3(velL + 3), bpos = ld3(baseL + 9);
        float bq[4] = {baseL[48], baseL[49], baseL[50], baseL[51]};
        wave_sync();      // F overwrites the velocity vector below
        #pragma unroll 1
        for (int k = lane; k < ndof; k += WAVE) {      // ndof = 6 + n may exceed the 64 lanes
            float f = k >= 6 ? lamL[k - 6] : 0.f;
            if (nc > 0) {
                const float *jr = JcL + k * NC3;
#pragma unroll
                for (int c = 0; c < NC3; c++) f = fmaf(jr[c], lamL[CBL + c], f);
            }
            FL[k] = f;
        }
        wave_sync();      // Jc is dead: region AB is reused for the 6-float exchanges below
        {
            float *const up6 = lds + L.AB;             // 8 * n
            float *const pub6 = lds + L.AB + 8 * n;    // 8 * (n + 1)
            // phase-local topology (fresh from the packed words: nothing from the kinematics stays live)
            unsigned info_p = info_w;
            asm volatile("" : "+v"(info_p));
            const int mylev = is_link ? (int)((info_p >> 8) & 0xffu) : 0;
            const int par = is_link ? (int)(info_p >> 24) - 1 : -1;
            const int pidx = par < 0 ? n : par;
            const int first_child = is_link ? fchL[lane] : -1;
            const float Fi = is_link ? FL[6 + lane] : 0.f;
            float Y = 0.f;
            int lane_f = lane;
            asm volatile("" : "+v"(lane_f));      // fresh: the substep's `li` would stay live (spilled) across the solver
            const float *pcm = cacheL + CACHE * (lane_f < n ? lane_f : 0);
            const V3 rw = ld3(pcm + C_RW);
            const Sv U = ldsv(pcm + C_UW);
            const float Dinv = pcm[C_DINV];
            const Sv Sm = ldsv(pcm + C_SW);
            #pragma unroll 1
            for (int lev = maxlev; lev >= 1; --lev) {
                if (is_link && mylev == lev) {
                    Sv Z; Z.a = v3(0, 0, 0); Z.l = v3(0, 0, 0);
                    #pragma unroll 1
                    for (int c = first_child; c >= 0; c = nsibL[c]) Z = Z + ldsv(up6 + 8 * c);
                    Y = Fi - sdot(Sm, Z);
                    Z = sfma(U, Y * Dinv, Z);
                    Z.a = Z.a + cross(rw, Z.l);
                    stsv(up6 + 8 * lane, Z);
                }
                wave_sync();
            }
            Sv Z0; Z0.a = v3(0, 0, 0); Z0.l = v3(0, 0, 0);
            #pragma unroll 1
            for (int k = 0; k < nbc; ++k) Z0 = Z0 + ldsv(up6 + 8 * bchL[k]);
            Z0.a = Z0.a - ld3(FL);
            Z0.l = Z0.l - ld3(FL + 3);
            const Sv ab = neg_mul6(baseL + 12, Z0);
            // (The sweep back down the tree could be skipped: the motor row of joint j has the Jacobian e_j, so the change of
            // its velocity is (A lam)_j = b_j - (z_j - lam_j) A_jj, which the sweeps leave on lane j.  Built and measured: +2 %
            // on the 42-dof NAO -- and the base quaternion of a free-floating Dancer drifts 2.6e-4 in 240 steps instead of
            // < 1e-4 (test_dancer_contact_free_240_steps): z_j carries the rounding of 150 x 54 row updates, the tree sweep
            // starts afresh from the impulses, and joints and base must come from ONE solve to conserve momentum.  Not kept.)
            if (lane == 0) stsv(pub6 + 8 * n, ab);
            wave_sync();
            float dq = 0.f;
            #pragma unroll 1
            for (int lev = 1; lev <= maxlev; ++lev) {
                if (is_link && mylev == lev) {
                    Sv a = ldsv(pub6 + 8 * pidx);
                    a.l = a.l + cross(a.a, rw);
                    dq = (Y - sdot(U, a)) * Dinv;
                    a = sfma(Sm, dq, a);
                    stsv(pub6 + 8 * lane, a);
                }
                wave_sync();
            }
            qd = clampv((is_link ? accL[lane] : 0.f) + dq, P.vmax);   // qd was parked in LDS, q is re-read from the record
            bang = clamp3(bang + ab.a, P.vmax);
            bvel = clamp3(bvel + ab.l, P.vmax);
        }

        HG_TICK(10);
        // ============ 6. integrate positions ==================================================
        float dt = P.dt, inv_dt = P.inv_dt;
        asm volatile("" : "+s"(dt), "+s"(inv_dt));   // opaque: keeps dt-derived constants inside the loop
        {
            int lane_q = lane;
            asm volatile("" : "+v"(lane_q));      // fresh address arithmetic (see the store below)
            q = fmaf(dt, qd, lane_q < n ? st[HG_STATE_BASE + lane_q] : 0.f);
        }
        bpos = bpos + bvel * dt;
        {
            float ang = sqrtf(dot(bang, bang));
            if (ang * dt > 0.78539816f) ang = 0.78539816f * inv_dt;
            float f;
            if (ang < 0.001f) f = 0.5f * dt - dt * dt * dt * 0.020833333333f * ang * ang;
            else f = __sinf(0.5f * ang * dt) / ang;
            const V3 a = bang * f;
            const float cw = __cosf(0.5f * ang * dt);
            const float x = bq[0], y = bq[1], z = bq[2], s = bq[3];
            float nq[4] = {cw * x + a.x * s + a.y * z - a.z * y,
                           cw * y - a.x * z + a.y * s + a.z * x,
                           cw * z + a.x * y - a.y * x + a.z * s,
                           cw * s - a.x * x - a.y * y - a.z * z};
            const float inv = rsqrtf(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
#pragma unroll
            for (int k = 0; k < 4; k++) bq[k] = nq[k] * inv;
        }
        // ---- store state ---------------------------------------------------------------------
        {
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));      // fresh address arithmetic: nothing kept from the loads
            if (lane_p < n) { st[HG_STATE_BASE + lane_p] = q; st[HG_STATE_BASE + n + lane_p] = qd; }
        }
        if (lane == 0) {
            st3(st, bpos); st[3] = bq[0]; st[4] = bq[1]; st[5] = bq[2]; st[6] = bq[3];
            st3(st + 7, bvel); st3(st + 10, bang);
        }
        wave_sync();      // the record is re-read by every lane (next substep, guard below)
    HG_TICK(11);
    }  // substeps

    {   // epilogue, with its own lane index (the kernel-level one would otherwise stay live across the substep loop)
    const int lane = fresh_lane();
    const bool is_link = lane < n;
    // ---- non-finite guard: reset the env in place and raise done ---------------------------------
    float q = 0.f, qd = 0.f, tgt = 0.f;
    if (is_link) { q = st[HG_STATE_BASE + lane]; qd = st[HG_STATE_BASE + n + lane]; tgt = st[HG_STATE_BASE + 2 * n + lane]; }
    V3 bpos = ld3(st);
    float bq[4] = {st[3], st[4], st[5], st[6]};
    // (the stored motor target counts: an infinite action is clamped into a finite target velocity and would otherwise stay
    // in the record for ever without making anything else non-finite)
    bool bad = !(isfinite(q) && isfinite(qd) && isfinite(tgt));
    {
        const V3 bvel = ld3(st + 7), bang = ld3(st + 10);
        bad = bad || !(isfinite(bpos.x + bpos.y + bpos.z + bvel.x + bvel.y + bvel.z + bang.x + bang.y + bang.z + bq[0] + bq[1] + bq[2] + bq[3]));
    }
    const bool any_bad = __ballot(bad) != 0ull;
    if (any_bad) {
        q = is_link ? M->link[lane][L_QINIT] : 0.f; qd = 0.f;
        bpos = ld3(M->base + B_POS0);
#pragma unroll
        for (int k = 0; k < 4; k++) bq[k] = M->base[B_QUAT0 + k];
        if (is_link) {
            st[HG_STATE_BASE + lane] = q; st[HG_STATE_BASE + n + lane] = 0.f; st[HG_STATE_BASE + 2 * n + lane] = q;
            if (O.torque) O.torque[(size_t)env * n + lane] = 0.f;
        }
        if (lane == 0) {
            st3(st, bpos); st[3] = bq[0]; st[4] = bq[1]; st[5] = bq[2]; st[6] = bq[3];
            st3(st + 7, v3(0, 0, 0)); st3(st + 10, v3(0, 0, 0));
        }
    }

    // ---- observations and flags (dancer_env.py:32-35,51-52; pepper_env.py:64-68) -----------------
    if (is_link) { miscL[lane] = q; velL[lane] = qd; }
    wave_sync();
    const int na = P.n_act;
    if (lane < na) {
        const int j = M->act_joint[lane];
        if (O.obs_q) O.obs_q[(size_t)env * na + lane] = miscL[j];
        if (O.obs_qd) O.obs_qd[(size_t)env * na + lane] = velL[j];
    }
    {
        int a = -3, b = -3;
        if (lane < nc) { a = __float_as_int(conL[CON * lane]); b = __float_as_int(conL[CON * lane + 1]); }
        if (O.pairs && lane < HG_MAX_CONTACTS) {
            O.pairs[((size_t)env * HG_MAX_CONTACTS + lane) * 2] = a;
            O.pairs[((size_t)env * HG_MAX_CONTACTS + lane) * 2 + 1] = b;
        }
        unsigned long long bits = 0;
        if (lane < nc) { bits |= 1ull << (a + 1); if (b >= -1) bits |= 1ull << (b + 1); }
        unsigned int blo = (unsigned int)bits, bhi = (unsigned int)(bits >> 32);
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) { blo |= __shfl_xor(blo, o); bhi |= __shfl_xor(bhi, o); }
        link_mask = ((unsigned long long)bhi << 32) | blo;   // valid on lanes 0..15
    }
    if (lane == 0) {
        if (O.pairs) { O.link_mask[env] = link_mask; O.counts[env] = nc; }      // (store_contacts = 0: none of the three)
        if (O.collision) O.collision[env] = (link_mask >> 1) != 0ull ? 1 : 0;
        if (O.reward) O.reward[env] = 0.f;
        if (O.done) O.done[env] = any_bad ? 1 : 0;
        {   // per-env running episode statistics, reduced and cleared by hg_stats_flush
            float4 c = *reinterpret_cast<float4 *>(O.cum + 4 * (size_t)env);
            c.x += 1.f; c.z += any_bad ? 1.f : 0.f; c.w += (link_mask >> 1) != 0ull ? 1.f : 0.f;
            *reinterpret_cast<float4 *>(O.cum + 4 * (size_t)env) = c;
        }
        if (O.base_xyyaw) {
            const M3 Rw0 = quat_to_mat(bq[0], bq[1], bq[2], bq[3]);
            const V3 fp = bpos + mulv(Rw0, ld3(M->base + B_FPOS));
            const M3 Rl = mul(Rw0, quat_to_mat(M->base[B_FQUAT], M->base[B_FQUAT + 1], M->base[B_FQUAT + 2], M->base[B_FQUAT + 3]));
            O.base_xyyaw[3 * (size_t)env] = fp.x;
            O.base_xyyaw[3 * (size_t)env + 1] = fp.y;
            O.base_xyyaw[3 * (size_t)env + 2] = atan2f(Rl.m[3], Rl.m[0]);
        }
    }
    HG_TICK(12);
    }
}

}  // namespace hg
