// hg_sim.hip -- host side of libhg_sim.so: the C-ABI of include/hg_sim.h, the
// device model builder and the small helper kernels (reset, link states,
// joint states, episode statistics).  The fused step kernel is in hg_step.hip.
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC hg_sim.hip -o libhg_sim.so
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/hg_sim.h"
#include "hg_device.h"
#include "hg_math.h"
#include "hg_variants.h"
#ifndef HG_SPLIT_TU
#include "hg_step_launch.h"      // single translation unit: the step kernel and its launchers are compiled here
#endif

namespace hg {

static thread_local std::string g_err;

static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HG_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(e_ == hipErrorOutOfMemory ? HG_ENOMEM : HG_ENODEVICE,          \
                        std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

// Makes `dev` current for the duration of one entry point and restores the caller's device afterwards:
// torch reads its current device from the HIP runtime, so a library call must not change it behind its back.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define HG_DEVICE(s) DeviceGuard guard_((s)->device); HG_HIP(guard_.err)

// ---------------------------------------------------------------------------------------
// helper kernels
// ---------------------------------------------------------------------------------------
__global__ void reset_kernel(const DevModel *__restrict__ M, float *__restrict__ state, float *__restrict__ torque,
                             const uint8_t *__restrict__ mask, int n_envs, int stride) {
    const int env = blockIdx.x;
    if (env >= n_envs || (mask && !mask[env])) return;
    const int n = M->n, S = HG_STATE_SIZE(n);
    float *st = state + (size_t)env * stride;
    if (torque) for (int k = threadIdx.x; k < n; k += blockDim.x) torque[(size_t)env * n + k] = 0.f;
    for (int k = threadIdx.x; k < S; k += blockDim.x) {
        float v = 0.f;
        if (k < 3) v = M->base[B_POS0 + k];
        else if (k < 7) v = M->base[B_QUAT0 + k - 3];
        else if (k >= HG_STATE_BASE && k < HG_STATE_BASE + n) v = M->link[k - HG_STATE_BASE][L_QINIT];
        else if (k >= HG_STATE_BASE + 2 * n) v = M->link[k - HG_STATE_BASE - 2 * n][L_QINIT];
        st[k] = v;
    }
}

__global__ void joint_state_kernel(const float *__restrict__ state, float *__restrict__ q,
                                   float *__restrict__ qd, int n, int n_envs, int stride) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_envs * n) return;
    const int env = (int)(i / n), j = (int)(i % n);
    const float *st = state + (size_t)env * stride;
    if (q) q[i] = st[HG_STATE_BASE + j];
    if (qd) qd[i] = st[HG_STATE_BASE + n + j];
}

// getLinkState for every body: one wave per env, lane i = link i, level sweep through LDS
__global__ __launch_bounds__(WAVE)
void link_state_kernel(const DevModel *__restrict__ M, const float *__restrict__ state,
                       float *__restrict__ com_pos, float *__restrict__ com_quat,
                       float *__restrict__ frame_pos, float *__restrict__ frame_quat, int stride) {
    __shared__ float pub[(HG_MAX_LINKS + 1) * 12];
    const int env = blockIdx.x, lane = threadIdx.x;
    const int n = M->n;
    const float *st = state + (size_t)env * stride;
    const bool is_link = lane < n;
    const int li = is_link ? lane : 0;
    const float *lk = M->link[li];
    const int par = is_link ? M->parent[li] : -1;
    const int pidx = par < 0 ? n : par;
    const int mylev = is_link ? M->level[li] : 0;
    M3 Rw = quat_to_mat(st[3], st[4], st[5], st[6]);
    V3 xw = ld3(st);
    if (lane == 0) { stm(pub + 12 * n, Rw); st3(pub + 12 * n + 9, xw); }
    __syncthreads();
    const float q = is_link ? st[HG_STATE_BASE + lane] : 0.f;
    float sq, cq;
    sincosf(-q, &sq, &cq);
    const M3 E = mul(axis_rot(ld3(lk + L_AXIS), sq, cq), ldm(lk + L_R0));
    const V3 r = ld3(lk + L_E) + mulv(E, ld3(lk + L_D));
    for (int lev = 1; lev <= M->maxlev; ++lev) {
        if (is_link && mylev == lev) {
            Rw = mul_bt(ldm(pub + 12 * pidx), E);
            xw = ld3(pub + 12 * pidx + 9) + mulv(Rw, r);
            stm(pub + 12 * lane, Rw); st3(pub + 12 * lane + 9, xw);
        }
        __syncthreads();
    }
    // body index: 0 = base (written by lane n), i + 1 = link i
    int body = -1;
    if (is_link) body = lane + 1;
    else if (lane == n) { body = 0; Rw = ldm(pub + 12 * n); xw = ld3(pub + 12 * n + 9); }
    if (body < 0) return;
    const size_t o = (size_t)env * (n + 1) + body;
    float qq[4];
    if (com_pos) st3(com_pos + 3 * o, xw);
    if (com_quat) { mat_to_quat(Rw, qq); for (int k = 0; k < 4; k++) com_quat[4 * o + k] = qq[k]; }
    if (frame_pos) st3(frame_pos + 3 * o, xw + mulv(Rw, ld3(M->frame_pos[body])));
    if (frame_quat) {
        const float *fq = M->frame_quat[body];
        mat_to_quat(mul(Rw, quat_to_mat(fq[0], fq[1], fq[2], fq[3])), qq);
        for (int k = 0; k < 4; k++) frame_quat[4 * o + k] = qq[k];
    }
}

// state records between the caller's dense layout and the library's line-padded one: one thread per float
__global__ void copy_records_kernel(float *__restrict__ dst, int dstride, const float *__restrict__ src, int sstride,
                                    int width, int n_envs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_envs * width) return;
    const size_t env = i / width;
    const int k = (int)(i - env * width);
    dst[env * dstride + k] = src[env * sstride + k];
}

__global__ void copy_contacts_kernel(const int32_t *__restrict__ pairs_in, const int32_t *__restrict__ counts_in,
                                     const unsigned long long *__restrict__ mask_in,
                                     int32_t *__restrict__ pairs, int32_t *__restrict__ counts,
                                     unsigned long long *__restrict__ mask, int n_envs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)n_envs * HG_MAX_CONTACTS * 2 && pairs) pairs[i] = pairs_in[i];
    if (i < (size_t)n_envs) {
        if (counts) counts[i] = counts_in[i];
        if (mask) mask[i] = mask_in[i];
    }
}

__global__ void stats_kernel(const float *__restrict__ reward, const uint8_t *__restrict__ done,
                             const uint8_t *__restrict__ collision, float *__restrict__ stats, int n_envs) {
    float r = 0.f, d = 0.f, c = 0.f, e = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_envs; i += gridDim.x * blockDim.x) {
        e += 1.f;
        if (reward) r += reward[i];
        if (done) d += done[i] ? 1.f : 0.f;
        if (collision) c += collision[i] ? 1.f : 0.f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        e += __shfl_xor(e, o); r += __shfl_xor(r, o); d += __shfl_xor(d, o); c += __shfl_xor(c, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(stats + 0, e); atomicAdd(stats + 1, r); atomicAdd(stats + 2, d); atomicAdd(stats + 3, c);
    }
}

// stats[0..4) += sum over envs of the running per-env counters, which are cleared
__global__ void stats_flush_kernel(float *__restrict__ cum, float *__restrict__ stats, int n_envs) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_envs; i += gridDim.x * blockDim.x) {
        float4 *p = reinterpret_cast<float4 *>(cum) + i;
        const float4 c = *p;
        a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w;
        *p = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(stats + 0, a.x); atomicAdd(stats + 1, a.y); atomicAdd(stats + 2, a.z); atomicAdd(stats + 3, a.w);
    }
}

// ---- hg_stats_accumulate_noalloc: fixed-order two-stage reduction through a static device array --------------
// Stage 1 writes one partial {envs, reward, done, collision} per 1024 envs into `scratch`, stage 2 (one workgroup)
// adds them up in slot order and adds the totals to the caller's vector.  No atomics: the sum does not depend on
// which workgroup finishes first.  The array is a file-scope device symbol, so no call ever allocates; every GPU
// the library runs on holds its own copy (16 KB) until the process ends.
constexpr int NOALLOC_ENVS_PER_SLOT = 1024;                        // 256 threads x 4 envs
constexpr int NOALLOC_SLOTS = HG_NOALLOC_MAX_ENVS / NOALLOC_ENVS_PER_SLOT;
static __device__ __align__(256) float scratch[NOALLOC_SLOTS * 4];

__device__ inline float4 block_sum4(float4 a, float4 *sh) {        // 256 threads = 4 waves, fixed order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a.x += __shfl_xor(a.x, o); a.y += __shfl_xor(a.y, o); a.z += __shfl_xor(a.z, o); a.w += __shfl_xor(a.w, o);
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
    __syncthreads();
    float4 r = sh[0];
    for (int w = 1; w < 4; w++) { r.x += sh[w].x; r.y += sh[w].y; r.z += sh[w].z; r.w += sh[w].w; }
    return r;
}

__global__ __launch_bounds__(256)
void stats_partial_kernel(const float *__restrict__ reward, const uint8_t *__restrict__ done,
                          const uint8_t *__restrict__ collision, float *__restrict__ part, int n_envs) {
    __shared__ float4 sh[4];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    const int base = blockIdx.x * NOALLOC_ENVS_PER_SLOT;
    for (int k = 0; k < 4; k++) {
        const int i = base + k * 256 + threadIdx.x;
        if (i < n_envs) {
            a.x += 1.f;
            if (reward) a.y += reward[i];
            if (done) a.z += done[i] ? 1.f : 0.f;
            if (collision) a.w += collision[i] ? 1.f : 0.f;
        }
    }
    const float4 r = block_sum4(a, sh);
    if (threadIdx.x == 0) reinterpret_cast<float4 *>(part)[blockIdx.x] = r;
}

__global__ __launch_bounds__(256)
void stats_final_kernel(const float *__restrict__ part, int n_slots, float *__restrict__ stats) {
    __shared__ float4 sh[4];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = threadIdx.x; k < n_slots; k += 256) {             // only the slots stage 1 of THIS call wrote
        const float4 p = reinterpret_cast<const float4 *>(part)[k];
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    const float4 r = block_sum4(a, sh);
    if (threadIdx.x == 0) { stats[0] += r.x; stats[1] += r.y; stats[2] += r.z; stats[3] += r.w; }
}

}  // namespace hg

#define HG_LDS_PER_CU  (160 * 1024)   // gfx950
#define HG_LDS_GRANULE 1280           // LDS is handed out in blocks of this size

// ---------------------------------------------------------------------------------------
// the handle
// ---------------------------------------------------------------------------------------
struct hg_sim {
    int device = 0;
    int n_envs = 0, n = 0, n_act = 0, state_size = 0;
    int state_stride = 0;       // floats between two envs' records in d_state: state_size rounded up to a 64-byte line
    int variant = 0;            // index into the (NMX, MAXC) instantiation table
    int n_spheres = 0, n_pairs = 0, maxlev = 0, n_base_children = 0, n_self = 0;
    int8_t lev_start[hg::DMAX + 4]{};
    int lds_bytes = 0;
    int waves_per_cu = 16;      // resident waves (envs) per CU: 4 SIMDs x the variant's waves per SIMD, or fewer (hg_create)
    int n_cu = 256;             // compute units of the device (MI355X: 256)
    hg_config cfg{};
    int32_t act_joint[HG_MAX_LINKS]{};
    hg::DevModel *d_model = nullptr;
    float *d_state = nullptr;
    unsigned long long *d_link_mask = nullptr;
    int32_t *d_pairs = nullptr, *d_counts = nullptr;
    float *d_cum = nullptr;            // [n_envs][4] running episode statistics
    float *d_torque = nullptr;         // [n_envs][n] applied motor torque of the last physics step
    float *d_base_cmd = nullptr;       // [n_envs][3] commanded base velocity, allocated on first use
    bool base_cmd_on = false;
    float *d_noalloc_scratch = nullptr; // this device's copy of the static scratch array (hg_stats_accumulate_noalloc)
};

namespace hg {


// Numbers the step divides by, takes square roots of or indexes with: checked once here, because a NaN mass or a zero
// inertia does not fault -- it makes every env of the handle "non-finite" and the guard resets it on every step, silently.
static int check_numbers(const hg_model *m, const hg_config *c, std::string *why) {
    auto fin = [](const double *v, int k) { for (int i = 0; i < k; i++) if (!std::isfinite(v[i])) return false; return true; };
    auto pos = [](const double *v, int k) { for (int i = 0; i < k; i++) if (!(v[i] > 0) || !std::isfinite(v[i])) return false; return true; };
    auto bad = [&](const char *what, int i) { *why = std::string(what) + (i >= 0 ? " (index " + std::to_string(i) + ")" : ""); return HG_EINVAL; };
    if (!pos(&m->base_mass, 1) || !pos(m->base_inertia, 3)) return bad("base mass and principal inertia must be positive and finite", -1);
    if (!fin(m->base_pos0, 3) || !fin(m->base_quat0, 4)) return bad("base reset pose must be finite", -1);
    {
        const double *q = m->base_quat0;
        if (std::fabs(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3] - 1.0) > 1e-3) return bad("base reset quaternion must be a unit quaternion", -1);
    }
    for (int i = 0; i < m->n_links; i++) {
        if (!pos(&m->mass[i], 1) || !pos(m->inertia[i], 3)) return bad("link mass and principal inertia must be positive and finite", i);
        if (!fin(m->R0[i], 9) || !fin(m->d[i], 3) || !fin(m->e[i], 3) || !fin(&m->q_init[i], 1)) return bad("link frame constants must be finite", i);
        const double *a = m->axis[i];
        if (!fin(a, 3) || std::fabs(a[0] * a[0] + a[1] * a[1] + a[2] * a[2] - 1.0) > 1e-3) return bad("joint axis must be a unit vector", i);
        if (!(m->joint_damping[i] >= 0) || !std::isfinite(m->joint_damping[i])) return bad("joint damping must be finite and >= 0", i);
        if (!(m->max_force[i] >= 0) || !(m->max_vel[i] >= 0)) return bad("motor force and velocity bounds must be >= 0 (may be infinite)", i);
        if (std::isnan(m->lower[i]) || std::isnan(m->upper[i])) return bad("joint bounds must not be NaN", i);
        // (lower > upper on a joint without a limit constraint is the URDF "no limit" convention: nothing to enforce but the
        //  target clamp, which then needs an ordered interval)
        if (m->lower[i] > m->upper[i] && (m->joint_limited[i] || c->clamp_targets)) return bad("joint bounds must satisfy lower <= upper where they are used (limit rows, target clamp)", i);
    }
    for (int b = 0; b <= m->n_links; b++)
        if (!fin(m->frame_pos[b], 3) || !fin(m->frame_quat[b], 4)) return bad("URDF frame offsets must be finite", b);
    for (int s = 0; s < m->n_spheres; s++)
        if (!fin(m->sphere_pos[s], 3) || !fin(m->sphere_half[s], 3) || !(m->sphere_radius[s] >= 0) || !std::isfinite(m->sphere_radius[s])) return bad("collision primitive centre / half axis / radius must be finite, radius >= 0", s);
    if (!fin(c->gravity, 3) || !(c->kp >= 0) || !(c->kd >= 0) || !(c->contact_erp >= 0) || !(c->friction >= 0) || !(c->lin_damping >= 0) ||
        !(c->ang_damping >= 0) || !(c->limit_max_impulse >= 0) || !std::isfinite(c->contact_margin) || std::isnan(c->max_coord_velocity) ||
        !std::isfinite(c->kp) || !std::isfinite(c->kd) || !std::isfinite(c->contact_erp) || !std::isfinite(c->friction) ||
        !std::isfinite(c->lin_damping) || !std::isfinite(c->ang_damping) || !std::isfinite(c->dt))
        return bad("config: gravity, gains, erp, friction, damping and limit_max_impulse must be finite and >= 0", -1);
    return HG_OK;
}

static int build_dev_model(const hg_model *m, const hg_config *c, DevModel *D, std::string *why) {
    const int n = m->n_links;
    if (int rc = check_numbers(m, c, why)) return rc;
    memset(D, 0, sizeof(DevModel));
    D->n = n; D->n_spheres = m->n_spheres; D->n_pairs = m->n_pairs; D->n_act = m->n_act;
    int nchild[HG_MAX_LINKS] = {0}, last_child[HG_MAX_LINKS];
    int nbc = 0, nslots = 0, maxlev = 0;
    for (int i = 0; i < n; i++) { D->first_child[i] = -1; D->next_sibling[i] = -1; D->save_slot[i] = -1; last_child[i] = -1; }
    for (int i = 0; i < n; i++) {
        const int p = m->parent[i];
        if (p < -1 || p >= i) { *why = "links must be ordered parents-first"; return HG_EINVAL; }
        D->parent[i] = (int8_t)p;
        D->level[i] = (int8_t)(p < 0 ? 1 : D->level[p] + 1);
        if (D->level[i] > DMAX) { *why = "kinematic tree deeper than DMAX"; return HG_EUNSUPPORTED; }
        if (D->level[i] > maxlev) maxlev = D->level[i];
        if (p < 0) D->base_children[nbc++] = (int8_t)i;
        else {
            if (last_child[p] < 0) D->first_child[p] = (int8_t)i; else D->next_sibling[last_child[p]] = (int8_t)i;
            last_child[p] = i; nchild[p]++;
        }
        D->anc_mask[i] = (1ull << i) | (p >= 0 ? D->anc_mask[p] : 0ull);
        for (int d = 0; d < DMAX; d++) D->path[i][d] = -1;
        for (int a = i; a >= 0; a = m->parent[a]) D->path[i][D->level[a] - 1] = (int8_t)a;
    }
    // How the full-descent walk (links in index order) finds the parent's acceleration: the previous
    // link (mode 0), the base (mode 1), or a saved copy (mode 2 + slot).  A link with a child
    // that does not directly follow it keeps its acceleration in a slot from its own visit until its last child has been
    // visited; slots are recycled, so NSLOT bounds the NESTING of branching links (torso -> hand ->
    // fingers is 2), not their number.
    {
        int owner[NSLOT], last_kid[HG_MAX_LINKS];
        for (int k = 0; k < NSLOT; k++) owner[k] = -1;
        for (int i = 0; i < n; i++) last_kid[i] = last_child[i];
        for (int i = 0; i < n; i++) {
            const int p = m->parent[i];
            for (int k = 0; k < NSLOT; k++)
                if (owner[k] >= 0 && last_kid[owner[k]] < i) owner[k] = -1;
            if (p < 0) D->par_mode[i] = 1;
            else if (p == i - 1) D->par_mode[i] = 0;
            else {
                if (D->save_slot[p] < 0) { *why = "internal: branching link without a slot"; return HG_EINVAL; }
                D->par_mode[i] = (int8_t)(2 + D->save_slot[p]);
            }
            bool needs = false;        // some child does not follow its parent directly in index order
            for (int c = i + 2; c < n; c++) needs = needs || m->parent[c] == i;
            if (needs) {
                int k = 0;
                while (k < NSLOT && owner[k] >= 0) k++;
                if (k == NSLOT) { *why = "branching links nested deeper than 2"; return HG_EUNSUPPORTED; }
                owner[k] = i; D->save_slot[i] = (int8_t)k;
                if (k + 1 > nslots) nslots = k + 1;
            }
        }
    }
    D->maxlev = maxlev; D->n_base_children = nbc; D->n_slots = nslots;
    {   // links grouped by level, for the column-parallel leaves -> root sweep
        int k = 0;
        for (int l = 0; l < DMAX + 4; l++) D->lev_start[l] = 0;
        for (int l = 1; l <= maxlev; l++) {
            D->lev_start[l] = k;
            for (int i = 0; i < n; i++) if (D->level[i] == l) D->lev_order[k++] = (int8_t)i;
        }
        for (int l = maxlev + 1; l < DMAX + 4; l++) D->lev_start[l] = k;
    }
    for (int i = 0; i < n; i++) {
        D->linkinfo[i] = (uint32_t)(uint8_t)D->par_mode[i] | ((uint32_t)(uint8_t)D->level[i] << 8) |
                         ((uint32_t)(uint8_t)(D->save_slot[i] + 1) << 16) | ((uint32_t)(uint8_t)(D->parent[i] + 1) << 24);
        D->kids[i] = (uint32_t)(uint8_t)(D->first_child[i] + 1) | ((uint32_t)(uint8_t)(D->next_sibling[i] + 1) << 8);
        for (int d = 0; d < DMAX; d++)
            D->pathw[i][d >> 2] |= (uint32_t)(uint8_t)(D->path[i][d] + 1) << (8 * (d & 3));
    }
    for (int k = 0; k < m->n_act; k++) {
        if (m->act_joint[k] < 0 || m->act_joint[k] >= n) { *why = "act_joint out of range"; return HG_EINVAL; }
        D->act_joint[k] = (int8_t)m->act_joint[k];
    }
    for (int i = 0; i < n; i++) {
        float *lk = D->link[i];
        for (int k = 0; k < 9; k++) lk[L_R0 + k] = (float)m->R0[i][k];
        const double *a = m->axis[i], *e = m->e[i];
        const double s[3] = {a[1] * e[2] - a[2] * e[1], a[2] * e[0] - a[0] * e[2], a[0] * e[1] - a[1] * e[0]};
        for (int k = 0; k < 3; k++) {
            lk[L_AXIS + k] = (float)a[k]; lk[L_S + k] = (float)s[k];
            lk[L_D + k] = (float)m->d[i][k]; lk[L_E + k] = (float)e[k];
            lk[L_INERTIA + k] = (float)m->inertia[i][k];
        }
        lk[L_MASS] = (float)m->mass[i];
        lk[L_DAMP] = (float)m->joint_damping[i];
        lk[L_LOWER] = (float)m->lower[i]; lk[L_UPPER] = (float)m->upper[i];
        lk[L_MAXIMP] = (float)fmin(m->max_force[i] * c->dt, 1e30);
        lk[L_MAXVEL] = (float)fmin(m->max_vel[i], 1e30);
        lk[L_QINIT] = (float)m->q_init[i];
        lk[L_LIMITED] = m->joint_limited[i] ? 1.0f : 0.0f;
    }
    D->base[B_MASS] = (float)m->base_mass;
    for (int k = 0; k < 3; k++) {
        D->base[B_INERTIA + k] = (float)m->base_inertia[k];
        D->base[B_POS0 + k] = (float)m->base_pos0[k];
        D->base[B_FPOS + k] = (float)m->frame_pos[0][k];
    }
    for (int k = 0; k < 4; k++) { D->base[B_QUAT0 + k] = (float)m->base_quat0[k]; D->base[B_FQUAT + k] = (float)m->frame_quat[0][k]; }
    for (int b = 0; b <= n; b++) {
        for (int k = 0; k < 3; k++) D->frame_pos[b][k] = (float)m->frame_pos[b][k];
        for (int k = 0; k < 4; k++) D->frame_quat[b][k] = (float)m->frame_quat[b][k];
    }
    // Collision primitives -> device slots.  The "4 deepest points per body" rule ranks inside a body's group, which the
    // kernel does in a 64-float LDS window per trip of 64 slots: the primitives of a body must be contiguous (verified, not
    // assumed), at most 64, and padding slots are inserted so that no group straddles a trip.  The order of the model is
    // kept -- it is the order in which contacts take their slots (oracle/hg_oracle.c detect_contacts).
    static_assert(HG_DEV_SELF <= 256 && WAVE == 64, "pair_meta packs 8-bit self indices; sphere_meta 8-bit window offsets");
    int dev_of[HG_MAX_SPHERES], nd = 0;
    for (int s = 0; s < m->n_spheres;) {
        const int link = m->sphere_link[s];
        if (link < -1 || link >= n) { *why = "sphere_link out of range"; return HG_EINVAL; }
        int e = s;
        while (e < m->n_spheres && m->sphere_link[e] == link) e++;
        for (int o = e; o < m->n_spheres; o++)
            if (m->sphere_link[o] == link) { *why = "collision primitives of one body must be contiguous"; return HG_EINVAL; }
        const int cnt = e - s;
        if (cnt > WAVE) { *why = "more than 64 collision primitives on one body"; return HG_EUNSUPPORTED; }
        if (nd % WAVE + cnt > WAVE) nd = (nd + WAVE - 1) / WAVE * WAVE;
        if (nd + cnt > HG_DEV_SPHERES) { *why = "collision primitives (with trip padding) exceed the device table"; return HG_EUNSUPPORTED; }
        const int g0 = nd % WAVE;
        for (int k = s; k < e; k++, nd++) {
            const double *h = m->sphere_half[k];
            if ((m->sphere_flags[k] & 1) && (h[0] != 0 || h[1] != 0 || h[2] != 0)) { *why = "a ground primitive must be a point or a sphere (sphere_half = 0)"; return HG_EINVAL; }
            dev_of[k] = nd;
            D->sphere_meta[nd] = (uint32_t)(uint8_t)(link + 1) | ((uint32_t)(m->sphere_flags[k] & 0x3) << 8) |
                                 ((uint32_t)g0 << 16) | ((uint32_t)(g0 + cnt) << 24);
            for (int j = 0; j < 3; j++) D->sphere[nd][j] = (float)m->sphere_pos[k][j];
            D->sphere[nd][3] = (float)m->sphere_radius[k];
        }
        s = e;
    }
    D->n_spheres = nd;
    // primitives that appear in some pair get a compact index: only their world centres / half axes are kept in LDS
    int self_of[HG_MAX_SPHERES], n_self = 0;
    for (int s = 0; s < m->n_spheres; s++) self_of[s] = -1;
    for (int k = 0; k < m->n_pairs; k++) {
        if (m->pair_a[k] < 0 || m->pair_a[k] >= m->n_spheres || m->pair_b[k] < 0 || m->pair_b[k] >= m->n_spheres) {
            *why = "self-collision pair out of range"; return HG_EINVAL;
        }
        if (m->sphere_link[m->pair_a[k]] == m->sphere_link[m->pair_b[k]]) { *why = "self-collision pair within one body"; return HG_EINVAL; }
        self_of[m->pair_a[k]] = self_of[m->pair_b[k]] = 0;
    }
    for (int s = 0; s < m->n_spheres; s++)
        if (self_of[s] == 0) {
            if (n_self >= HG_DEV_SELF) { *why = "more than 256 primitives in link-link pairs"; return HG_EUNSUPPORTED; }
            self_of[s] = n_self; D->self_idx[dev_of[s]] = (uint8_t)n_self; D->sphere_meta[dev_of[s]] |= 0x8000u;
            for (int j = 0; j < 3; j++) D->self_half[n_self][j] = (float)m->sphere_half[s][j];
            n_self++;
        }
    D->n_self = c->enable_self ? n_self : 0;
    for (int k = 0; k < m->n_pairs; k++)
        D->pair_meta[k] = (uint32_t)self_of[m->pair_a[k]] | ((uint32_t)self_of[m->pair_b[k]] << 8) |
                          ((uint32_t)(uint8_t)(m->sphere_link[m->pair_a[k]] + 1) << 16) |
                          ((uint32_t)(uint8_t)(m->sphere_link[m->pair_b[k]] + 1) << 24);
    return HG_OK;
}

}  // namespace hg

using namespace hg;

// ---------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------
extern "C" {

const char *hg_last_error(void) { return g_err.c_str(); }
int hg_abi_version(void) { return 8; }

#ifdef HG_PHASE_TIMING
// diagnostics build only (tools/phase_ticks.py): the shader-clock stamps of the last launch (of the variant launched last)
static int g_last_variant = -1;
extern "C" int hg_debug_phase_ticks(unsigned long long *host_out, int n_envs) {
    if (n_envs > HG_TICK_ENVS) n_envs = HG_TICK_ENVS;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    switch (g_last_variant) {
#define HG_TICKS_CASE(i, a, b, c, d, w) case i: return read_ticks_v<i>(host_out, n_envs);
        HG_VARIANTS(HG_TICKS_CASE)
        default: return -1;
    }
}
#endif

// Device address of `scratch` (hg_stats_accumulate_noalloc) on each GPU, resolved by hg_create -- which allocates anyway
// and has the device current -- so that the entry point itself is a bounds check and two launches (safe under graph
// capture).  In host code the symbol's name is only a handle for hipGetSymbolAddress, never a pointer a kernel could use.
static float *g_scratch_ptr[HG_NOALLOC_MAX_DEVICES];
static std::mutex g_scratch_mu;

int hg_create(const hg_model *model, const hg_config *config, int n_envs, int device_id, hg_sim **out) {
    if (!model || !config || !out) return fail(HG_EINVAL, "hg_create: null argument");
    *out = nullptr;
    if (n_envs <= 0) return fail(HG_EINVAL, "hg_create: n_envs must be positive");
    const int n = model->n_links;
    if (n <= 0 || n > HG_MAX_LINKS) return fail(HG_EINVAL, "hg_create: n_links out of range");
    if (model->n_spheres < 0 || model->n_spheres > HG_MAX_SPHERES || model->n_pairs < 0 || model->n_pairs > HG_MAX_PAIRS ||
        model->n_act < 0 || model->n_act > n)
        return fail(HG_EINVAL, "hg_create: model sizes out of range");
    if (config->max_contacts < 0 || config->max_contacts > HG_MAX_CONTACTS || config->solver_iters < 0 || !(config->dt > 0))
        return fail(HG_EINVAL, "hg_create: bad config (max_contacts, solver_iters or dt)");
    if (config->residual_threshold != 0)       // measured: the exit is not reached within the configured sweeps (profiles/round3_residual_histogram.txt)
        return fail(HG_EUNSUPPORTED, "hg_create: residual_threshold != 0 is implemented by the CPU oracle only; the HIP solver runs every sweep");

    // the model is flattened and matched to a kernel variant on the host first, so an unsupported robot is
    // reported as such with or without a GPU
    DevModel *hm = new (std::nothrow) DevModel;
    if (!hm) return fail(HG_ENOMEM, "hg_create: host allocation failed");
    std::string why;
    int rc = build_dev_model(model, config, hm, &why);
    if (rc) { delete hm; return fail(rc, "hg_create: " + why); }
    int variant = -1;
    for (int v = 0; v < kNumVariants; v++)
        if (kVariants[v].nmx >= n && kVariants[v].maxc >= config->max_contacts &&
            kVariants[v].depth >= hm->maxlev && kVariants[v].nsl >= hm->n_slots) { variant = v; break; }
    if (const char *ov = getenv("HG_VARIANT")) {      // diagnostics: force an instantiation that fits the model (tools/config_table.py)
        const int v = atoi(ov);
        if (v >= 0 && v < kNumVariants && kVariants[v].nmx >= n && kVariants[v].maxc >= config->max_contacts &&
            kVariants[v].depth >= hm->maxlev && kVariants[v].nsl >= hm->n_slots) variant = v;
    }
    if (variant < 0) {
        delete hm;
        return fail(HG_EUNSUPPORTED, "hg_create: no kernel variant for this (n_links, max_contacts, tree depth, branching)");
    }

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (device_id < 0 || device_id >= ndev) { delete hm; return fail(HG_ENODEVICE, "hg_create: no such HIP device"); }
    DeviceGuard guard_(device_id);
    if (guard_.err != hipSuccess) { delete hm; return fail(HG_ENODEVICE, "hg_create: hipSetDevice failed"); }

    hg_sim *s = new (std::nothrow) hg_sim;
    if (!s) { delete hm; return fail(HG_ENOMEM, "hg_create: host allocation failed"); }
    s->device = device_id; s->n_envs = n_envs; s->n = n; s->n_act = model->n_act;
    { int cu = 0; if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && cu > 0) s->n_cu = cu; }
    s->n_spheres = hm->n_spheres; s->n_pairs = hm->n_pairs; s->maxlev = hm->maxlev; s->n_base_children = hm->n_base_children;
    s->n_self = hm->n_self;
    for (int l = 0; l < DMAX + 4; l++) s->lev_start[l] = (int8_t)hm->lev_start[l];
    s->state_size = HG_STATE_SIZE(n); s->variant = variant; s->cfg = *config;
    s->state_stride = (s->state_size + 15) & ~15;      // records start on 64-byte lines: no partial-line traffic
    memcpy(s->act_joint, model->act_joint, sizeof(s->act_joint));
    s->lds_bytes = (int)sizeof(float) * lds_plan(n, hm->n_spheres, 3 * kVariants[variant].maxc, hm->maxlev, hm->n_self,
                                                       kVariants[variant].nmx + 3 * kVariants[variant].maxc,
                                                       ys_floats(kVariants[variant].nmx, kVariants[variant].maxc, kVariants[variant].depth),
                                                       cache_stride(kVariants[variant].nmx)).total;

    s->waves_per_cu = 4 * kVariants[variant].wpe;
    {   // ... or what LDS admits (NAO with fingers: 16.1 KB per wave = 9 per CU where the registers would admit 12)
        const int by_lds = HG_LDS_PER_CU / ((s->lds_bytes + HG_LDS_GRANULE - 1) / HG_LDS_GRANULE * HG_LDS_GRANULE);
        if (by_lds >= 1 && by_lds < s->waves_per_cu) s->waves_per_cu = by_lds;
    }
    if (const char *w = getenv("HG_WAVES_PER_CU")) {      // diagnostics: fewer resident waves per CU, by asking for more LDS per wave
        const int k = atoi(w);
        if (k >= 1 && k < s->waves_per_cu && HG_LDS_PER_CU / k >= s->lds_bytes) {
            s->waves_per_cu = k;
            s->lds_bytes = HG_LDS_PER_CU / k / HG_LDS_GRANULE * HG_LDS_GRANULE;
        }
    }

    hipError_t e = hipMalloc(&s->d_model, sizeof(DevModel));
    if (e == hipSuccess) e = hipMalloc(&s->d_state, sizeof(float) * (size_t)n_envs * s->state_stride);
    if (e == hipSuccess) e = hipMemset(s->d_state, 0, sizeof(float) * (size_t)n_envs * s->state_stride);
    if (config->store_contacts) {
        if (e == hipSuccess) e = hipMalloc(&s->d_link_mask, sizeof(unsigned long long) * (size_t)n_envs);
        if (e == hipSuccess) e = hipMalloc(&s->d_pairs, sizeof(int32_t) * (size_t)n_envs * HG_MAX_CONTACTS * 2);
        if (e == hipSuccess) e = hipMalloc(&s->d_counts, sizeof(int32_t) * (size_t)n_envs);
    }
    if (e == hipSuccess) e = hipMalloc(&s->d_cum, sizeof(float) * 4 * (size_t)n_envs);
    if (e == hipSuccess && config->store_torques) e = hipMalloc(&s->d_torque, sizeof(float) * (size_t)n_envs * n);
    if (e == hipSuccess) e = hipMemcpy(s->d_model, hm, sizeof(DevModel), hipMemcpyHostToDevice);
    if (e == hipSuccess && s->d_pairs) e = hipMemset(s->d_link_mask, 0, sizeof(unsigned long long) * (size_t)n_envs);
    if (e == hipSuccess && s->d_pairs) e = hipMemset(s->d_pairs, 0xFF, sizeof(int32_t) * (size_t)n_envs * HG_MAX_CONTACTS * 2);
    if (e == hipSuccess && s->d_pairs) e = hipMemset(s->d_counts, 0, sizeof(int32_t) * (size_t)n_envs);
    if (e == hipSuccess) e = hipMemset(s->d_cum, 0, sizeof(float) * 4 * (size_t)n_envs);
    if (e == hipSuccess && s->d_torque) e = hipMemset(s->d_torque, 0, sizeof(float) * (size_t)n_envs * n);
    delete hm;
    if (e != hipSuccess) {
        const std::string msg = std::string("hg_create: ") + hipGetErrorString(e);
        hg_destroy(s);
        return fail(e == hipErrorOutOfMemory ? HG_ENOMEM : HG_ENODEVICE, msg);
    }
    if (device_id < HG_NOALLOC_MAX_DEVICES) {      // this device's copy of the static scratch array (the device is current)
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch_ptr[device_id]) {
            void *p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(hg::scratch)) != hipSuccess) { hg_destroy(s); return fail(HG_ENODEVICE, "hg_create: hipGetSymbolAddress(scratch) failed"); }
            g_scratch_ptr[device_id] = static_cast<float *>(p);
        }
        s->d_noalloc_scratch = g_scratch_ptr[device_id];
    }
    rc = hg_reset(s, nullptr, nullptr);
    if (rc == HG_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(HG_ENODEVICE, "hg_create: reset kernel failed");
    if (rc) { hg_destroy(s); return rc; }
    *out = s;
    return HG_OK;
}

int hg_destroy(hg_sim *s) {
    if (!s) return HG_OK;
    DeviceGuard guard_(s->device);
    (void)hipFree(s->d_model); (void)hipFree(s->d_state); (void)hipFree(s->d_link_mask);
    (void)hipFree(s->d_pairs); (void)hipFree(s->d_counts); (void)hipFree(s->d_cum); (void)hipFree(s->d_base_cmd);
    (void)hipFree(s->d_torque);
    delete s;
    return HG_OK;
}

int hg_reset(hg_sim *s, const uint8_t *env_mask_dev, void *stream) {
    if (!s) return fail(HG_EINVAL, "hg_reset: null handle");
    HG_DEVICE(s);
    hipLaunchKernelGGL(reset_kernel, dim3(s->n_envs), dim3(WAVE), 0, (hipStream_t)stream,
                       s->d_model, s->d_state, s->d_torque, env_mask_dev, s->n_envs, s->state_stride);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_step(hg_sim *s, const float *actions_dev, const int32_t *joint_sel, int n_sel, int n_substeps,
            float *obs_q_dev, float *obs_qd_dev, float *base_xyyaw_dev,
            float *reward_dev, uint8_t *done_dev, uint8_t *collision_dev, void *stream) {
    if (!s) return fail(HG_EINVAL, "hg_step: null handle");
    if (n_substeps < 1 || n_substeps > 1024) return fail(HG_EINVAL, "hg_step: n_substeps must be in 1..1024");
    StepParams P;
    memset(&P, 0, sizeof P);
    if (joint_sel == nullptr) { joint_sel = s->act_joint; n_sel = s->n_act; }
    if (n_sel < 0 || n_sel > s->n) return fail(HG_EINVAL, "hg_step: n_sel out of range");
    unsigned long long seen = 0;
    for (int k = 0; k < n_sel; k++) {
        if (joint_sel[k] < 0 || joint_sel[k] >= s->n) return fail(HG_EINVAL, "hg_step: joint_sel entry out of range");
        if (seen & (1ull << joint_sel[k])) return fail(HG_EINVAL, "hg_step: joint_sel names a joint twice");
        seen |= 1ull << joint_sel[k];
        P.sel[k] = (int8_t)joint_sel[k];
    }
    const hg_config &c = s->cfg;
    P.n_envs = s->n_envs; P.n_substeps = n_substeps; P.iters = c.solver_iters; P.maxc = c.max_contacts;
    P.n_sel = n_sel; P.state_stride = s->state_stride;
    P.n = s->n; P.n_spheres = s->n_spheres; P.n_pairs = s->n_pairs; P.n_act = s->n_act; P.maxlev = s->maxlev;
    P.n_base_children = s->n_base_children; P.n_self = s->n_self;
    memcpy(P.lev_start, s->lev_start, sizeof P.lev_start);
    P.flags = (c.enable_ground ? F_GROUND : 0) | (c.enable_self ? F_SELF : 0) | (c.clamp_targets ? F_CLAMP : 0) |
              ((actions_dev && n_sel > 0) ? F_ACTIONS : 0) | (s->base_cmd_on ? F_BASECMD : 0) |
              (c.enable_limits ? F_LIMITS : 0);
    {   // Wave-priority flip (hg_step.hip, 4b): every env of a launch that fits the GPU at once; of a longer launch only the
        // last half resident round (the tail is where waves are left alone on their SIMD; in the middle of the launch the
        // phases mix by themselves and the flip costs 2 %).  Diagnostics: HG_PRIO_FLIP=0 never, 1 every env, a value > 1
        // (hex) every env with that byte as the per-slot priority pattern (default 0xE4).
        static const char *ov = getenv("HG_PRIO_FLIP");
        static const long ovv = ov ? strtol(ov, nullptr, 16) : -1;
        const long resident = (long)s->n_cu * s->waves_per_cu;
        P.flip_from = (!ov && (long)s->n_envs > resident) ? (int)(s->n_envs - resident / 2) : 0;
        static const char *ff = getenv("HG_FLIP_FROM");      // diagnostics (tools/prio_ab.sh)
        if (ff) P.flip_from = atoi(ff);
        if (!ov || ovv != 0) P.flags |= F_PRIOFLIP | (int)(((ovv > 1 ? ovv : 0xE4) & 0xff) << 8);
    }
    P.dt = (float)c.dt; P.inv_dt = (float)(1.0 / c.dt);
    P.gx = (float)c.gravity[0]; P.gy = (float)c.gravity[1]; P.gz = (float)c.gravity[2];
    P.kp = (float)c.kp; P.kd = (float)c.kd; P.erp = (float)c.contact_erp; P.mu = (float)c.friction;
    P.margin = (float)c.contact_margin; P.lin_damp = (float)c.lin_damping; P.ang_damp = (float)c.ang_damping;
    P.limhi = (float)c.limit_max_impulse;
    P.vmax = c.max_coord_velocity > 0 ? (float)c.max_coord_velocity : 3.0e38f;
    StepOut O;
    O.obs_q = obs_q_dev; O.obs_qd = obs_qd_dev; O.base_xyyaw = base_xyyaw_dev; O.reward = reward_dev;
    O.done = done_dev; O.collision = collision_dev;
    O.link_mask = s->d_link_mask; O.pairs = s->d_pairs; O.counts = s->d_counts; O.cum = s->d_cum; O.base_cmd = s->d_base_cmd;
    O.torque = s->d_torque;
    HG_DEVICE(s);
    hipError_t e;
    hipStream_t st = (hipStream_t)stream;
#ifdef HG_PHASE_TIMING
    g_last_variant = s->variant;
#endif
    switch (s->variant) {
#define HG_VARIANT_CASE(i, a, b, c, d, w) case i: e = launch_step_v<i>(s->n_envs, s->lds_bytes, st, s->d_model, P, s->d_state, actions_dev, O); break;
        HG_VARIANTS(HG_VARIANT_CASE)
        default: return fail(HG_EUNSUPPORTED, "hg_step: bad kernel variant");
    }
    if (e != hipSuccess) return fail(HG_ENODEVICE, std::string("hg_step: launch failed: ") + hipGetErrorString(e));
    return HG_OK;
}

int hg_get_joint_states(hg_sim *s, float *q_dev, float *qd_dev, void *stream) {
    if (!s) return fail(HG_EINVAL, "hg_get_joint_states: null handle");
    HG_DEVICE(s);
    const size_t tot = (size_t)s->n_envs * s->n;
    hipLaunchKernelGGL(joint_state_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       s->d_state, q_dev, qd_dev, s->n, s->n_envs, s->state_stride);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_get_joint_torques(hg_sim *s, float *tau_dev, void *stream) {
    if (!s || !tau_dev) return fail(HG_EINVAL, "hg_get_joint_torques: null argument");
    HG_DEVICE(s);
    if (!s->d_torque) return fail(HG_EINVAL, "hg_get_joint_torques: this handle was created with store_torques = 0");
    HG_HIP(hipMemcpyAsync(tau_dev, s->d_torque, sizeof(float) * (size_t)s->n_envs * s->n, hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    return HG_OK;
}

int hg_get_link_states(hg_sim *s, float *com_pos_dev, float *com_quat_dev, float *frame_pos_dev,
                       float *frame_quat_dev, void *stream) {
    if (!s) return fail(HG_EINVAL, "hg_get_link_states: null handle");
    HG_DEVICE(s);
    hipLaunchKernelGGL(link_state_kernel, dim3(s->n_envs), dim3(WAVE), 0, (hipStream_t)stream,
                       s->d_model, s->d_state, com_pos_dev, com_quat_dev, frame_pos_dev, frame_quat_dev, s->state_stride);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_get_contact_pairs(hg_sim *s, int32_t *pairs_dev, int32_t *counts_dev, uint64_t *link_mask_dev, void *stream) {
    if (!s) return fail(HG_EINVAL, "hg_get_contact_pairs: null handle");
    HG_DEVICE(s);
    if (!s->d_pairs) return fail(HG_EINVAL, "hg_get_contact_pairs: this handle was created with store_contacts = 0");
    const size_t tot = (size_t)s->n_envs * HG_MAX_CONTACTS * 2;
    hipLaunchKernelGGL(copy_contacts_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       s->d_pairs, s->d_counts, s->d_link_mask, pairs_dev, counts_dev,
                       (unsigned long long *)link_mask_dev, s->n_envs);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_get_state(hg_sim *s, float *state_dev, void *stream) {
    if (!s || !state_dev) return fail(HG_EINVAL, "hg_get_state: null argument");
    HG_DEVICE(s);
    // the caller's records are dense (HG_STATE_SIZE floats), the library's are padded to whole 64-byte lines
    const size_t tot = (size_t)s->n_envs * s->state_size;
    hipLaunchKernelGGL(copy_records_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       state_dev, s->state_size, s->d_state, s->state_stride, s->state_size, s->n_envs);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_set_state(hg_sim *s, const float *state_dev, void *stream) {
    if (!s || !state_dev) return fail(HG_EINVAL, "hg_set_state: null argument");
    HG_DEVICE(s);
    const size_t tot = (size_t)s->n_envs * s->state_size;
    hipLaunchKernelGGL(copy_records_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       s->d_state, s->state_stride, state_dev, s->state_size, s->state_size, s->n_envs);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_get_layout(const hg_sim *s, int32_t *n_envs, int32_t *n_links, int32_t *n_act, int32_t *state_size,
                  int32_t *max_contacts) {
    if (!s) return fail(HG_EINVAL, "hg_get_layout: null handle");
    if (n_envs) *n_envs = s->n_envs;
    if (n_links) *n_links = s->n;
    if (n_act) *n_act = s->n_act;
    if (state_size) *state_size = s->state_size;
    if (max_contacts) *max_contacts = s->cfg.max_contacts;
    return HG_OK;
}

int hg_stats_accumulate(hg_sim *s, const float *reward_dev, const uint8_t *done_dev, const uint8_t *collision_dev,
                        float *stats_dev, void *stream) {
    if (!s || !stats_dev) return fail(HG_EINVAL, "hg_stats_accumulate: null argument");
    HG_DEVICE(s);
    int blocks = (s->n_envs + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(stats_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       reward_dev, done_dev, collision_dev, stats_dev, s->n_envs);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_stats_accumulate_noalloc(hg_sim *s, const float *reward_dev, const uint8_t *done_dev, const uint8_t *collision_dev,
                                float *stats_dev, void *stream) {
    if (!s || !stats_dev) return fail(HG_EINVAL, "hg_stats_accumulate_noalloc: null argument");
    if (s->n_envs > HG_NOALLOC_MAX_ENVS)
        return fail(HG_EUNSUPPORTED, "hg_stats_accumulate_noalloc: n_envs = " + std::to_string(s->n_envs) + " exceeds HG_NOALLOC_MAX_ENVS = " +
                                     std::to_string(HG_NOALLOC_MAX_ENVS) + " (the static scratch array holds one partial per 1024 envs)");
    if (s->device < 0 || s->device >= HG_NOALLOC_MAX_DEVICES)
        return fail(HG_EUNSUPPORTED, "hg_stats_accumulate_noalloc: device index beyond HG_NOALLOC_MAX_DEVICES");
    HG_DEVICE(s);
    float *const part = s->d_noalloc_scratch;      // resolved once per device by hg_create
    if (!part) return fail(HG_EUNSUPPORTED, "hg_stats_accumulate_noalloc: no scratch address for this device");
    // Nothing in the array has to start at zero: stage 2 reads exactly the n_slots partials that stage 1 of this
    // call has just written on the same stream; slots beyond them keep an earlier call's contents and are not read.
    const int n_slots = (s->n_envs + NOALLOC_ENVS_PER_SLOT - 1) / NOALLOC_ENVS_PER_SLOT;
    hipLaunchKernelGGL(stats_partial_kernel, dim3(n_slots), dim3(256), 0, (hipStream_t)stream,
                       reward_dev, done_dev, collision_dev, part, s->n_envs);
    HG_HIP(hipGetLastError());
    hipLaunchKernelGGL(stats_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, n_slots, stats_dev);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

int hg_set_base_velocity(hg_sim *s, const float *cmd_dev, void *stream) {
    if (!s) return fail(HG_EINVAL, "hg_set_base_velocity: null handle");
    HG_DEVICE(s);
    if (!cmd_dev) { s->base_cmd_on = false; return HG_OK; }
    if (!s->d_base_cmd) {
        if (hipMalloc(&s->d_base_cmd, sizeof(float) * 3 * (size_t)s->n_envs) != hipSuccess) {
            s->d_base_cmd = nullptr;
            return fail(HG_ENOMEM, "hg_set_base_velocity: device allocation failed");
        }
    }
    HG_HIP(hipMemcpyAsync(s->d_base_cmd, cmd_dev, sizeof(float) * 3 * (size_t)s->n_envs, hipMemcpyDeviceToDevice,
                          (hipStream_t)stream));
    s->base_cmd_on = true;
    return HG_OK;
}

int hg_stats_flush(hg_sim *s, float *stats_dev, void *stream) {
    if (!s || !stats_dev) return fail(HG_EINVAL, "hg_stats_flush: null argument");
    HG_DEVICE(s);
    int blocks = (s->n_envs + 255) / 256;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(stats_flush_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s->d_cum, stats_dev, s->n_envs);
    HG_HIP(hipGetLastError());
    return HG_OK;
}

}  // extern "C"
