/*
 * hg_oracle.c -- CPU restatement (plain C, FP64, one env at a time) of the
 * physics step the reference delegates to PyBullet.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in humanoid_gym_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do, and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of the path lives in `pybullet` (version not
 * pinned, /root/reference/setup.py:18), which is absent from /root/reference
 * and from this image; the reference holds no tests, golden vectors or
 * recorded trajectories.  This file therefore restates the published
 * algorithm of Bullet's btMultiBodyDynamicsWorld single step as called from
 *   /root/reference/humanoid_gym/envs/dancer_env.py:19   setJointMotorControlArray(POSITION_CONTROL)
 *   /root/reference/humanoid_gym/envs/dancer_env.py:20   stepSimulation()
 *   /root/reference/humanoid_gym/envs/dancer_env.py:24   getJointState
 *   /root/reference/humanoid_gym/envs/dancer_env.py:27   getLinkState
 *   /root/reference/humanoid_gym/envs/dancer_env.py:34   getContactPoints
 *   /root/reference/humanoid_gym/envs/nao_env.py:70,73   setAngles / getAnglesPosition
 * and is pinned only by physical invariants (tests/test_oracle_physics.py) and
 * by the integer fixtures derived from the reference's URDF (tests/golden/).
 *
 * One physics step (order of btMultiBodyDynamicsWorld::solveConstraints +
 * integrateTransforms [upstream recall]):
 *   1. forward kinematics, contact detection on the CURRENT pose
 *   2. articulated-body algorithm with gravity and Bullet's link damping,
 *      velocities += dt * unconstrained acceleration
 *   3. one velocity-level constraint row per motor (btMultiBodyJointMotor:
 *      target velocity kp*(q*-q)/dt + (1-kd)*qd clamped to +-max_vel, impulse
 *      bounded by +-max_force*dt), then per contact one normal row and two
 *      friction rows; each row needs M^-1 J^T (a unit-impulse ABA response)
 *   4. projected Gauss-Seidel, fixed iteration count, Bullet's row order
 *      (motor rows sweep backwards on even iterations, forwards on odd ones;
 *      then normals; then frictions bounded by mu * current normal impulse)
 *   5. velocities += M^-1 J^T lambda; positions integrated with the new
 *      velocities (semi-implicit Euler, quaternion by exponential map).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#if defined(__SSE2__)
#include <xmmintrin.h>
#include <pmmintrin.h>
#endif

#include "../include/hg_model.h"

/* Flop counting ("replace the estimated constants with counted values from the oracle's
 * instrumented build").  With -DHGO_COUNT_FLOPS (make libhg_oracle_flops.so) every arithmetic block adds the
 * multiplies + adds / subtracts it executes (a divide, sqrt, sin or cos counts 1) to a thread-local counter;
 * hgo_flops_reset / hgo_flops_read expose it.  This counts what THIS formulation executes (dense 6x6 products in
 * the ABA included), so it is an upper estimate of the algorithmic work, not of what the kernel issues.
 * The normal build compiles the macro away. */
#ifdef HGO_COUNT_FLOPS
static _Thread_local double hgo_flop_counter;
#define HGO_FLOPS(k) (hgo_flop_counter += (double)(k))
#else
#define HGO_FLOPS(k) ((void)0)
#endif

#define NL HG_MAX_LINKS
#define NDOF_MAX (6 + HG_MAX_LINKS)
#define NROWS_MAX (2 * HG_MAX_LINKS + 3 * HG_MAX_CONTACTS)

/* ------------------------------------------------------------------ */
/* small linear algebra                                                */
/* ------------------------------------------------------------------ */
static void v3_cross(const double a[3], const double b[3], double o[3]) {
    HGO_FLOPS(9);
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static double v3_dot(const double a[3], const double b[3]) { HGO_FLOPS(5); return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void m3_mulv(const double M[9], const double v[3], double o[3]) {
    HGO_FLOPS(15);
    double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2];
    double y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2];
    double z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void m3_tmulv(const double M[9], const double v[3], double o[3]) { /* M^T v */
    HGO_FLOPS(15);
    double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2];
    double y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2];
    double z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void m3_mul(const double A[9], const double B[9], double O[9]) {
    double T[9];
    HGO_FLOPS(45);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(O, T, sizeof T);
}
static void m3_mul_bt(const double A[9], const double B[9], double O[9]) { /* A B^T */
    double T[9];
    HGO_FLOPS(45);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            T[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
    memcpy(O, T, sizeof T);
}
static void quat_to_mat(const double q[4], double R[9]) { /* xyzw, body->world */
    double x = q[0], y = q[1], z = q[2], w = q[3];
    HGO_FLOPS(30);
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
static void mat_to_quat(const double m[9], double q[4]) {
    double t = m[0] + m[4] + m[8], s;
    if (t > 0) { s = sqrt(t + 1.0) * 2; q[3] = 0.25 * s; q[0] = (m[7] - m[5]) / s; q[1] = (m[2] - m[6]) / s; q[2] = (m[3] - m[1]) / s; }
    else if (m[0] > m[4] && m[0] > m[8]) { s = sqrt(1.0 + m[0] - m[4] - m[8]) * 2; q[3] = (m[7] - m[5]) / s; q[0] = 0.25 * s; q[1] = (m[1] + m[3]) / s; q[2] = (m[2] + m[6]) / s; }
    else if (m[4] > m[8]) { s = sqrt(1.0 + m[4] - m[0] - m[8]) * 2; q[3] = (m[2] - m[6]) / s; q[0] = (m[1] + m[3]) / s; q[1] = 0.25 * s; q[2] = (m[5] + m[7]) / s; }
    else { s = sqrt(1.0 + m[8] - m[0] - m[4]) * 2; q[3] = (m[3] - m[1]) / s; q[0] = (m[2] + m[6]) / s; q[1] = (m[5] + m[7]) / s; q[2] = 0.25 * s; }
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
}
/* rotation about unit axis a by angle th (active, right-handed) */
static void axis_angle(const double a[3], double th, double R[9]) {
    double c = cos(th), s = sin(th), t = 1 - c;
    HGO_FLOPS(3 + 27);
    R[0] = t * a[0] * a[0] + c;        R[1] = t * a[0] * a[1] - s * a[2]; R[2] = t * a[0] * a[2] + s * a[1];
    R[3] = t * a[0] * a[1] + s * a[2]; R[4] = t * a[1] * a[1] + c;        R[5] = t * a[1] * a[2] - s * a[0];
    R[6] = t * a[0] * a[2] - s * a[1]; R[7] = t * a[1] * a[2] + s * a[0]; R[8] = t * a[2] * a[2] + c;
}

/* spatial vectors are [angular(3); linear(3)] in the body's internal frame */

/* ------------------------------------------------------------------ */
/* per-env scratch                                                     */
/* ------------------------------------------------------------------ */
typedef struct {
    /* kinematics */
    double E[NL][9];      /* parent frame -> link frame                     */
    double r[NL][3];      /* parent COM -> link COM, link frame             */
    double S[NL][6];      /* joint motion subspace, link frame              */
    double Rw[NL + 1][9]; /* body frame -> world (index 0 = base)           */
    double xw[NL + 1][3]; /* body COM in world                              */
    /* ABA cache (kept for the unit-impulse responses) */
    double U[NL][6];
    double Dinv[NL];
    double IAbase_inv[36];
} scratch_t;

static void motion_xform(const double E[9], const double r[3], const double vp[6], double vc[6]) {
    /* child <- parent:  w' = E w ; v' = E v + w' x r */
    double w[3], v[3], c[3];
    HGO_FLOPS(3);
    m3_mulv(E, vp, w);
    m3_mulv(E, vp + 3, v);
    v3_cross(w, r, c);
    vc[0] = w[0]; vc[1] = w[1]; vc[2] = w[2];
    vc[3] = v[0] + c[0]; vc[4] = v[1] + c[1]; vc[5] = v[2] + c[2];
}
static void force_xform_to_parent(const double E[9], const double r[3], const double fc[6], double fp[6]) {
    /* parent <- child:  f = E^T f_c ; n = E^T (n_c + r x f_c) */
    double c[3], n[3];
    HGO_FLOPS(3);
    v3_cross(r, fc + 3, c);
    n[0] = fc[0] + c[0]; n[1] = fc[1] + c[1]; n[2] = fc[2] + c[2];
    m3_tmulv(E, n, fp);
    m3_tmulv(E, fc + 3, fp + 3);
}
static void build_X(const double E[9], const double r[3], double X[36]) {
    /* 6x6 motion transform child <- parent: [[E,0],[-[r]x E, E]] */
    double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0}, B[9];
    m3_mul(rx, E, B);
    memset(X, 0, 36 * sizeof(double));
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            X[6 * i + j] = E[3 * i + j];
            X[6 * (i + 3) + j + 3] = E[3 * i + j];
            X[6 * (i + 3) + j] = -B[3 * i + j];
        }
}

/* symmetric positive definite 6x6 inverse via Cholesky */
static int spd6_inverse(const double A[36], double Ainv[36]) {
    double L[36] = {0};
    HGO_FLOPS(70 + 6 + 15 + 6 * 72);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j <= i; j++) {
            double s = A[6 * i + j];
            for (int k = 0; k < j; k++) s -= L[6 * i + k] * L[6 * j + k];
            if (i == j) { if (s <= 0) return -1; L[6 * i + i] = sqrt(s); }
            else L[6 * i + j] = s / L[6 * j + j];
        }
    for (int c = 0; c < 6; c++) {
        double y[6], x[6];
        for (int i = 0; i < 6; i++) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = 0; k < i; k++) s -= L[6 * i + k] * y[k];
            y[i] = s / L[6 * i + i];
        }
        for (int i = 5; i >= 0; i--) {
            double s = y[i];
            for (int k = i + 1; k < 6; k++) s -= L[6 * k + i] * x[k];
            x[i] = s / L[6 * i + i];
        }
        for (int i = 0; i < 6; i++) Ainv[6 * i + c] = x[i];
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* forward kinematics                                                  */
/* ------------------------------------------------------------------ */
static void kinematics(const hg_model *m, const double *st, scratch_t *w) {
    const int n = m->n_links;
    const double *q = st + HG_STATE_BASE;
    quat_to_mat(st + 3, w->Rw[0]);
    memcpy(w->xw[0], st, 3 * sizeof(double));
    for (int i = 0; i < n; i++) {
        double Rq[9], t[3];
        axis_angle(m->axis[i], -q[i], Rq);
        m3_mul(Rq, m->R0[i], w->E[i]);
        m3_mulv(w->E[i], m->d[i], t);
        for (int k = 0; k < 3; k++) w->r[i][k] = m->e[i][k] + t[k];
        HGO_FLOPS(6);
        memcpy(w->S[i], m->axis[i], 3 * sizeof(double));
        v3_cross(m->axis[i], m->e[i], w->S[i] + 3);
        const int p = m->parent[i] + 1;
        m3_mul_bt(w->Rw[p], w->E[i], w->Rw[i + 1]);          /* Rw_child = Rw_parent E^T */
        m3_mulv(w->Rw[i + 1], w->r[i], t);
        for (int k = 0; k < 3; k++) w->xw[i + 1][k] = w->xw[p][k] + t[k];
    }
}

/* ------------------------------------------------------------------ */
/* articulated-body algorithm: unconstrained accelerations             */
/* acc = [base angular (world), base linear (world, of the COM), qdd]  */
/* ------------------------------------------------------------------ */
static double hgo_noise_eps;            /* the noise probe of the tests, defined with hgo_set_row_noise below */
static double noise_factor(void);

static int aba(const hg_model *m, const hg_config *c, const double *st, const double *tau,
               scratch_t *w, double *acc) {
    const int n = m->n_links;
    const double *qd = st + HG_STATE_BASE + n;
    double v[NL + 1][6], cor[NL][6], pA[NL + 1][6], IA[NL + 1][36], u[NL], a[NL + 1][6];

    m3_tmulv(w->Rw[0], st + 10, v[0]);        /* base angular velocity, base frame */
    m3_tmulv(w->Rw[0], st + 7, v[0] + 3);     /* base linear velocity, base frame  */

    for (int b = 0; b <= n; b++) {
        const int i = b - 1;
        if (b > 0) {
            double vj[6];
            motion_xform(w->E[i], w->r[i], v[m->parent[i] + 1], v[b]);
            for (int k = 0; k < 6; k++) vj[k] = w->S[i][k] * qd[i];
            /* c = v x vj (spatial motion cross product) */
            v3_cross(v[b], vj, cor[i]);
            double t1[3], t2[3];
            v3_cross(v[b], vj + 3, t1);
            v3_cross(v[b] + 3, vj, t2);
            for (int k = 0; k < 3; k++) cor[i][3 + k] = t1[k] + t2[k];
            for (int k = 0; k < 6; k++) v[b][k] += vj[k];
        }
        HGO_FLOPS(b ? 6 + 3 + 6 : 0);       /* vj, cor linear part, v += vj */
        HGO_FLOPS(3 + 2 + 12 + 24);         /* Iw, two sqrt, bias force */
        const double mass = b ? m->mass[i] : m->base_mass;
        const double *I = b ? m->inertia[i] : m->base_inertia;
        memset(IA[b], 0, sizeof IA[b]);
        for (int k = 0; k < 3; k++) { IA[b][7 * k] = I[k]; IA[b][7 * (k + 3)] = mass; }
        /* bias force: v x* (I v) - gravity + Bullet's velocity damping */
        double Iw[3] = {I[0] * v[b][0], I[1] * v[b][1], I[2] * v[b][2]}, g[3], t[3];
        v3_cross(v[b], Iw, pA[b]);
        v3_cross(v[b], v[b] + 3, t);
        m3_tmulv(w->Rw[b], c->gravity, g);
        const double wn = sqrt(v3_dot(v[b], v[b])), vn = sqrt(v3_dot(v[b] + 3, v[b] + 3));
        for (int k = 0; k < 3; k++) {
            pA[b][k] += Iw[k] * (c->ang_damping + c->ang_damping * wn);
            pA[b][3 + k] = mass * t[k] - mass * g[k] + mass * v[b][3 + k] * (c->lin_damping + c->lin_damping * vn);
        }
        if (hgo_noise_eps > 0)      /* noise probe: the velocity-product terms as a float32 recursion over the chain leaves them */
            for (int k = 0; k < 6; k++) { if (b > 0) cor[i][k] *= noise_factor(); pA[b][k] *= noise_factor(); }
    }
    for (int i = n - 1; i >= 0; i--) {
        const int b = i + 1, p = m->parent[i] + 1;
        double D = 0, Ia[36], pa[6], X[36], T[36];
        HGO_FLOPS(72 + 12 + 1 + 14 + 108 + 90 + 432 + 468 + 6);   /* U, D, 1/D, u, Ia, pa, Ia X, X^T (Ia X) +=, pA += */
        if (hgo_noise_eps > 0)      /* noise probe: the accumulated articulated inertia as float32 sums leave it (the rank-one
                                       downdate below cancels: a light link under a heavy subtree amplifies this) */
            for (int r = 0; r < 6; r++)
                for (int k = r; k < 6; k++) { IA[b][6 * r + k] *= noise_factor(); IA[b][6 * k + r] = IA[b][6 * r + k]; }
        for (int r = 0; r < 6; r++) {
            double s = 0;
            for (int k = 0; k < 6; k++) s += IA[b][6 * r + k] * w->S[i][k];
            w->U[i][r] = s;
        }
        for (int k = 0; k < 6; k++) D += w->S[i][k] * w->U[i][k];
        w->Dinv[i] = 1.0 / D;
        u[i] = (tau ? tau[i] : 0.0) - m->joint_damping[i] * qd[i];
        for (int k = 0; k < 6; k++) u[i] -= w->S[i][k] * pA[b][k];
        for (int r = 0; r < 6; r++)
            for (int k = 0; k < 6; k++) Ia[6 * r + k] = IA[b][6 * r + k] - w->U[i][r] * w->U[i][k] * w->Dinv[i];
        for (int r = 0; r < 6; r++) {
            double s = pA[b][r] + w->U[i][r] * u[i] * w->Dinv[i];
            for (int k = 0; k < 6; k++) s += Ia[6 * r + k] * cor[i][k];
            pa[r] = s;
        }
        build_X(w->E[i], w->r[i], X);
        for (int r = 0; r < 6; r++)
            for (int k = 0; k < 6; k++) {
                double s = 0;
                for (int j = 0; j < 6; j++) s += Ia[6 * r + j] * X[6 * j + k];
                T[6 * r + k] = s;
            }
        for (int r = 0; r < 6; r++)
            for (int k = 0; k < 6; k++) {
                double s = 0;
                for (int j = 0; j < 6; j++) s += X[6 * j + r] * T[6 * j + k];
                IA[p][6 * r + k] += s;
            }
        double fp[6];
        force_xform_to_parent(w->E[i], w->r[i], pa, fp);
        for (int k = 0; k < 6; k++) pA[p][k] += fp[k];
    }
    if (spd6_inverse(IA[0], w->IAbase_inv)) return -1;
    HGO_FLOPS(72 + 31 * n + 3);            /* base acceleration; per link of the root -> leaves pass: + cor, U.a, 1/D, S qdd */
    for (int r = 0; r < 6; r++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s -= w->IAbase_inv[6 * r + k] * pA[0][k];
        a[0][r] = s;
    }
    for (int i = 0; i < n; i++) {
        const int b = i + 1;
        motion_xform(w->E[i], w->r[i], a[m->parent[i] + 1], a[b]);
        double s = u[i];
        for (int k = 0; k < 6; k++) { a[b][k] += cor[i][k]; }
        for (int k = 0; k < 6; k++) s -= w->U[i][k] * a[b][k];
        acc[6 + i] = s * w->Dinv[i];
        for (int k = 0; k < 6; k++) a[b][k] += w->S[i][k] * acc[6 + i];
    }
    /* base: world frame, classical acceleration of the COM */
    double t[3], lin[3];
    v3_cross(v[0], v[0] + 3, t);
    for (int k = 0; k < 3; k++) lin[k] = a[0][3 + k] + t[k];
    m3_mulv(w->Rw[0], a[0], acc);
    m3_mulv(w->Rw[0], lin, acc + 3);
    return 0;
}

/* response to a generalized impulse f (same layout as acc): out = M^-1 f */
static void impulse_response(const hg_model *m, const scratch_t *w, const double *f, double *out) {
    const int n = m->n_links;
    double Z[NL + 1][6], Y[NL], a[NL + 1][6];
    memset(Z, 0, sizeof(double) * 6 * (n + 1));
    double t[3];
    HGO_FLOPS((36 + 25) * n + 72);         /* per link: Y, Z + U Y / D, Z[p] +=  |  U.a, 1/D, S qdd;  base: 6x6 product */
    m3_tmulv(w->Rw[0], f, t);
    for (int k = 0; k < 3; k++) Z[0][k] = -t[k];
    m3_tmulv(w->Rw[0], f + 3, t);
    for (int k = 0; k < 3; k++) Z[0][3 + k] = -t[k];
    for (int i = n - 1; i >= 0; i--) {
        const int b = i + 1, p = m->parent[i] + 1;
        double y = f[6 + i], zt[6], fp[6];
        for (int k = 0; k < 6; k++) y -= w->S[i][k] * Z[b][k];
        Y[i] = y;
        for (int k = 0; k < 6; k++) zt[k] = Z[b][k] + w->U[i][k] * y * w->Dinv[i];
        force_xform_to_parent(w->E[i], w->r[i], zt, fp);
        for (int k = 0; k < 6; k++) Z[p][k] += fp[k];
    }
    for (int r = 0; r < 6; r++) {
        double s = 0;
        for (int k = 0; k < 6; k++) s -= w->IAbase_inv[6 * r + k] * Z[0][k];
        a[0][r] = s;
    }
    for (int i = 0; i < n; i++) {
        const int b = i + 1;
        motion_xform(w->E[i], w->r[i], a[m->parent[i] + 1], a[b]);
        double s = Y[i];
        for (int k = 0; k < 6; k++) s -= w->U[i][k] * a[b][k];
        out[6 + i] = s * w->Dinv[i];
        for (int k = 0; k < 6; k++) a[b][k] += w->S[i][k] * out[6 + i];
    }
    m3_mulv(w->Rw[0], a[0], out);
    m3_mulv(w->Rw[0], a[0] + 3, out + 3);
}

/* Jacobian row: velocity of world point pt on body `body` (-1 = base) along dir */
static void point_jacobian(const hg_model *m, const scratch_t *w, int body, const double pt[3],
                           const double dir[3], double sign, double *J) {
    double rel[3], c[3];
    HGO_FLOPS(3 + 12);
    for (int k = 0; k < 3; k++) rel[k] = pt[k] - w->xw[0][k];
    v3_cross(rel, dir, c);
    for (int k = 0; k < 3; k++) { J[k] += sign * c[k]; J[3 + k] += sign * dir[k]; }
    for (int i = body; i >= 0; i = m->parent[i]) {
        double aw[3], ew[3], piv[3];
        HGO_FLOPS(6 + 2);
        m3_mulv(w->Rw[i + 1], m->axis[i], aw);
        m3_mulv(w->Rw[i + 1], m->e[i], ew);
        for (int k = 0; k < 3; k++) piv[k] = pt[k] - (w->xw[i + 1][k] - ew[k]);
        v3_cross(piv, dir, c);
        J[6 + i] += sign * v3_dot(aw, c);
    }
}

/* Bullet's btPlaneSpace1: two unit tangents p, q orthogonal to n */
static void plane_space(const double n[3], double p[3], double q[3]) {
    HGO_FLOPS(12);
    if (fabs(n[2]) > 0.7071067811865475244) {
        double a = n[1] * n[1] + n[2] * n[2], k = 1.0 / sqrt(a);
        p[0] = 0; p[1] = -n[2] * k; p[2] = n[1] * k;
        q[0] = a * k; q[1] = -n[0] * p[2]; q[2] = n[0] * p[1];
    } else {
        double a = n[0] * n[0] + n[1] * n[1], k = 1.0 / sqrt(a);
        p[0] = -n[1] * k; p[1] = n[0] * k; p[2] = 0;
        q[0] = -n[2] * p[1]; q[1] = n[2] * p[0]; q[2] = a * k;
    }
}

/* Test instrumentation (tests/_oracle.py: OracleSim.noise_spread): "what would an FP32 implementation of this same step
 * answer?"  With eps > 0 every constraint row is perturbed before the sweeps -- each entry of its impulse response
 * M^-1 J^T, its right-hand side and its diagonal by an independent relative error of standard deviation eps -- which is
 * what the roundings of a float32 ABA / Jacobian chain leave in those numbers (eps ~ 1e-6 = 16 ulp) -- and every
 * collision sphere's world centre is moved by eps * 0.3 m per coordinate (3e-7 m: float32 positions of O(1 m) after a
 * chain of transforms), which reaches contact points, normals, distances and through them the tangent basis.  The same
 * relative error goes into the articulated-body pass itself: every link's velocity-product acceleration and bias force,
 * and its accumulated articulated inertia before the rank-one downdate -- that downdate cancels, so a light link under a
 * heavy subtree (mass ratios of 1000 in the fuzzers' trees) amplifies float32 rounding a thousandfold, which noise on the
 * finished rows alone does not show (tools/fuzz_fail/r3final_tree411).  The noise stream
 * of env e is seeded with (seed, e), so a batch of identical envs samples the spread of the answer. */
static double hgo_noise_eps = 0.0;
static uint64_t hgo_noise_seed = 0;
static _Thread_local uint64_t hgo_rng_state;
void hgo_set_row_noise(double eps, uint64_t seed) { hgo_noise_eps = eps; hgo_noise_seed = seed; }
static void noise_begin(int env) {
    hgo_rng_state = (hgo_noise_seed + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)(env + 1) * 0xD1B54A32D192ED03ull;
}
static double noise_factor(void) {       /* 1 + eps * u, u uniform with unit variance */
    uint64_t x = hgo_rng_state;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    hgo_rng_state = x;
    const double u01 = (double)((x * 0x2545F4914F6CDD1Dull) >> 11) * (1.0 / 9007199254740992.0);
    return 1.0 + hgo_noise_eps * 3.4641016151377544 * (u01 - 0.5);
}

/* ------------------------------------------------------------------ */
/* contact detection                                                   */
/* ------------------------------------------------------------------ */
typedef struct {
    int body_a, body_b;     /* body_b = -2 for the ground plane */
    int sphere_a, sphere_b;
    double pt[3], n[3], dist;
} contact_t;

/* Closest points of two sphere-swept segments (capsules; a sphere is a segment of length 0): centres ca, cb, half axes
 * ha, hb (world), on the segments p + s d, s in [0, 1] (Ericson, Real-Time Collision Detection 5.1.9, for the exact
 * solution).  Two nearly parallel segments have a whole interval of (nearly) closest points and the exact solution jumps
 * between its ends with the sign of a tiny angle: a contact point that does so makes two bodies chatter.  So the
 * parameter on a is blended from the centre of the overlap interval (parallel) to the exact solution (sin^2 of the angle
 * >= HG_SEG_BLEND), which is continuous in the pose; the point on b is then the closest to that point, and the point on a
 * the closest to the point on b -- for non-parallel segments exactly Ericson's result.  Writes the closest points c1 (on
 * a), c2 (on b).  The kernel (hg_step.hip, contact detection) runs the same formulas in float32. */
#define HG_SEG_EPS 1e-12
#define HG_SEG_BLEND 0.03
static double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
static void seg_seg_closest(const double *ca, const double *ha, const double *cb, const double *hb, double *c1, double *c2) {
    double p1[3], p2[3], d1[3], d2[3], r[3];
    for (int k = 0; k < 3; k++) { p1[k] = ca[k] - ha[k]; d1[k] = 2.0 * ha[k]; p2[k] = cb[k] - hb[k]; d2[k] = 2.0 * hb[k]; r[k] = p1[k] - p2[k]; }
    const double a = v3_dot(d1, d1), e = v3_dot(d2, d2), f = v3_dot(d2, r), c = v3_dot(d1, r), b = v3_dot(d1, d2);
    const double ia = a > HG_SEG_EPS ? 1.0 / a : 0.0, ie = e > HG_SEG_EPS ? 1.0 / e : 0.0;
    double nx[3], rx[3];
    v3_cross(d1, d2, nx); v3_cross(r, d2, rx);
    const double ae = a * e, denom = v3_dot(nx, nx);      /* = a e - b^2, without its cancellation */
    const double w = denom < HG_SEG_BLEND * ae ? denom / (HG_SEG_BLEND * ae) : 1.0;      /* ae == 0: 1 (no axis, no blend) */
    const double s_exact = denom > 0.0 ? clamp01(-v3_dot(rx, nx) / denom) : 0.0;         /* = (b f - c e) / denom */
    const double sb0 = -c * ia, sb1 = (b - c) * ia;                                      /* the ends of b projected on a */
    const double s_mid = 0.5 * (clamp01(sb0 < sb1 ? sb0 : sb1) + clamp01(sb0 < sb1 ? sb1 : sb0));
    double s = s_mid + w * (s_exact - s_mid);
    const double t = clamp01((b * s + f) * ie);
    s = clamp01((b * t - c) * ia);
    for (int k = 0; k < 3; k++) { c1[k] = p1[k] + d1[k] * s; c2[k] = p2[k] + d2[k] * t; }
}

/* test hook (tests/test_oracle_physics.py): the closest points of two sphere-swept segments as the step computes them */
void hgo_seg_seg_closest(const double *ca, const double *ha, const double *cb, const double *hb, double *c1, double *c2) {
    seg_seg_closest(ca, ha, cb, hb, c1, c2);
}

static int detect_contacts(const hg_model *m, const hg_config *c, const scratch_t *w, contact_t *out) {
    int nc = 0;
    const int maxc = c->max_contacts < HG_MAX_CONTACTS ? c->max_contacts : HG_MAX_CONTACTS;
    static _Thread_local double cw[HG_MAX_SPHERES][3], hw[HG_MAX_SPHERES][3], dist[HG_MAX_SPHERES];
    for (int s = 0; s < m->n_spheres; s++) {
        const int b = m->sphere_link[s] + 1;
        double t[3];
        m3_mulv(w->Rw[b], m->sphere_pos[s], t);
        m3_mulv(w->Rw[b], m->sphere_half[s], hw[s]);
        HGO_FLOPS(4);
        for (int k = 0; k < 3; k++) cw[s][k] = w->xw[b][k] + t[k];
        if (hgo_noise_eps > 0) for (int k = 0; k < 3; k++) cw[s][k] += 0.3 * (noise_factor() - 1.0);
        dist[s] = cw[s][2] - m->sphere_radius[s];
    }
    if (c->enable_ground) {
        for (int s = 0; s < m->n_spheres && nc < maxc; s++) {
            if (!(m->sphere_flags[s] & 1) || !(dist[s] < c->contact_margin)) continue;
            /* at most 4 points per body: keep the 4 smallest (dist, index) */
            int rank = 0;
            for (int o = 0; o < m->n_spheres; o++) {
                if (o == s || m->sphere_link[o] != m->sphere_link[s] || !(m->sphere_flags[o] & 1)) continue;
                if (!(dist[o] < c->contact_margin)) continue;
                if (dist[o] < dist[s] || (dist[o] == dist[s] && o < s)) rank++;
            }
            if (rank >= 4) continue;
            contact_t *k = &out[nc++];
            k->body_a = m->sphere_link[s]; k->body_b = -2; k->sphere_a = s; k->sphere_b = -1;
            k->n[0] = 0; k->n[1] = 0; k->n[2] = 1;
            k->pt[0] = cw[s][0]; k->pt[1] = cw[s][1]; k->pt[2] = dist[s];
            k->dist = dist[s];
        }
    }
    if (c->enable_self) {
        for (int p = 0; p < m->n_pairs && nc < maxc; p++) {
            const int a = m->pair_a[p], b = m->pair_b[p];
            double c1[3], c2[3];
            seg_seg_closest(cw[a], hw[a], cw[b], hw[b], c1, c2);
            double d[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
            HGO_FLOPS(60 + 3 + 1 + 2);
            const double len = sqrt(v3_dot(d, d));
            const double ds = len - m->sphere_radius[a] - m->sphere_radius[b];
            if (!(ds < c->contact_margin) || len < 1e-9) continue;
            HGO_FLOPS(3 + 3 * 6);
            contact_t *k = &out[nc++];
            k->body_a = m->sphere_link[a]; k->body_b = m->sphere_link[b]; k->sphere_a = a; k->sphere_b = b;
            for (int j = 0; j < 3; j++) {
                k->n[j] = d[j] / len;
                k->pt[j] = 0.5 * ((c1[j] - m->sphere_radius[a] * k->n[j]) + (c2[j] + m->sphere_radius[b] * k->n[j]));
            }
            k->dist = ds;
        }
    }
    return nc;
}

/* ------------------------------------------------------------------ */
/* one physics step                                                    */
/* ------------------------------------------------------------------ */
typedef struct {
    double J[NDOF_MAX], u[NDOF_MAX];
    double rhs, diag_inv, lo, hi, lambda;
    int normal_row;     /* for friction rows: index of the normal row, else -1 */
} row_t;

/* 1 / (J M^-1 J^T), or 0 = row disabled when the denominator vanishes (HG_ROW_DEN_EPS, hg_model.h): rhs and every
 * update of such a row are 0, its impulse stays 0 */
static double row_diag_inv(double den) { return den > HG_ROW_DEN_EPS ? 1.0 / den : 0.0; }

/* Returns the velocity residual of the update, deltaImpulse / m_jacDiagABInv as
 * btMultiBodyConstraintSolver::resolveSingleConstraintRowGeneric does [UPSTREAM-RECALL]. */
static double resolve_row(row_t *r, const row_t *rows, double mu, double *dv, int ndof) {
    double lo = r->lo, hi = r->hi;
    if (r->normal_row >= 0) {
        const double tot = rows[r->normal_row].lambda;
        if (!(tot > 0)) return 0.0;
        lo = -mu * tot; hi = mu * tot;
    }
    double jv = 0;
    HGO_FLOPS(4 * ndof + 4 + (r->normal_row >= 0 ? 2 : 0));
    for (int k = 0; k < ndof; k++) jv += r->J[k] * dv[k];
    double dl = r->rhs - jv * r->diag_inv;
    double sum = r->lambda + dl;
    if (sum < lo) sum = lo;
    if (sum > hi) sum = hi;
    dl = sum - r->lambda;
    r->lambda = sum;
    for (int k = 0; k < ndof; k++) dv[k] += r->u[k] * dl;
    return r->diag_inv > 0 ? dl / r->diag_inv : 0.0;
}

/* Test instrumentation (tools/residual_histogram.py): with a probe threshold set, every physics step records the first
 * sweep after which the largest squared residual was <= probe (solver_iters + 1: never), whatever the config says. */
static double hgo_probe_threshold = 0.0;
static long hgo_sweep_hist[1026];
void hgo_residual_probe(double threshold) { hgo_probe_threshold = threshold; memset(hgo_sweep_hist, 0, sizeof hgo_sweep_hist); }
void hgo_residual_histogram(long *out, int n) { for (int k = 0; k < n && k < 1026; k++) out[k] = hgo_sweep_hist[k]; }



typedef struct {
    int n_contacts;
    uint64_t link_mask;                    /* bit i+1: link i has a contact, bit 0: base */
    int pair_a[HG_MAX_CONTACTS], pair_b[HG_MAX_CONTACTS];  /* body indices, -2 = ground */
    double lambda_n[HG_MAX_CONTACTS];
    double motor_torque[HG_MAX_LINKS];     /* motor row impulse / dt of the last physics step (getJointState()[3]) */
} step_info_t;

/* btMultiBody::applyDeltaVeeMultiDof clamps every velocity coordinate to +-m_maxCoordinateVelocity
 * (100 by default) each time velocities change [UPSTREAM-RECALL]; it is what keeps a robot that is
 * driven by 1e5 N m motors towards random targets from gaining energy without bound.  Bullet clamps
 * the base components in the base frame, this restatement (and the kernel) in the world frame. */
static void clamp_velocities(const hg_config *c, double *st, int n) {
    const double vmax = c->max_coord_velocity;
    if (!(vmax > 0)) return;
    for (int k = 7; k < 13; k++) { if (st[k] > vmax) st[k] = vmax; if (st[k] < -vmax) st[k] = -vmax; }
    double *qd = st + HG_STATE_BASE + n;
    for (int i = 0; i < n; i++) { if (qd[i] > vmax) qd[i] = vmax; if (qd[i] < -vmax) qd[i] = -vmax; }
}

static int physics_step(const hg_model *m, const hg_config *c, double *st, step_info_t *info) {
    const int n = m->n_links, ndof = 6 + n;
    const double dt = c->dt;
    double *q = st + HG_STATE_BASE, *qd = q + n, *tgt = qd + n;
    static _Thread_local scratch_t W;
    static _Thread_local row_t rows[NROWS_MAX];
    contact_t con[HG_MAX_CONTACTS];
    double acc[NDOF_MAX], vel[NDOF_MAX], dv[NDOF_MAX];

    kinematics(m, st, &W);
    const int nc = detect_contacts(m, c, &W, con);
    if (aba(m, c, st, NULL, &W, acc)) return -1;

    HGO_FLOPS(2 * ndof);
    for (int k = 0; k < 3; k++) { st[10 + k] += dt * acc[k]; st[7 + k] += dt * acc[3 + k]; }
    for (int i = 0; i < n; i++) qd[i] += dt * acc[6 + i];
    clamp_velocities(c, st, n);
    for (int k = 0; k < 3; k++) { vel[k] = st[10 + k]; vel[3 + k] = st[7 + k]; }
    for (int i = 0; i < n; i++) vel[6 + i] = qd[i];

    /* Non-contact rows in the order PyBullet creates them: joint-limit constraints while the URDF is
     * converted, motors afterwards [UPSTREAM-RECALL].  btMultiBodyJointLimitConstraint: one row per
     * violated bound (penetration = q - lower or upper - q; rows with penetration > 0 are skipped),
     * Jacobian +-1 on the joint, desired velocity 0, Baumgarte term -penetration * erp / dt, impulse in
     * [0, m_maxAppliedImpulse].  The response of such a row is +- that of the joint's motor row. */
    int nr = 0;
    int lim_joint[HG_MAX_LINKS];
    double lim_sign[HG_MAX_LINKS], lim_pen[HG_MAX_LINKS];
    int n_lim = 0;
    if (c->enable_limits)
        for (int i = 0; i < n; i++) {
            if (!m->joint_limited[i]) continue;
            if (q[i] - m->lower[i] <= 0) { lim_joint[n_lim] = i; lim_sign[n_lim] = 1.0; lim_pen[n_lim++] = q[i] - m->lower[i]; }
            else if (m->upper[i] - q[i] <= 0) { lim_joint[n_lim] = i; lim_sign[n_lim] = -1.0; lim_pen[n_lim++] = m->upper[i] - q[i]; }
        }
    for (int k = 0; k < n_lim; k++) {
        row_t *r = &rows[nr++];
        const int i = lim_joint[k];
        memset(r->J, 0, sizeof(double) * ndof);
        r->J[6 + i] = lim_sign[k];
        impulse_response(m, &W, r->J, r->u);
        HGO_FLOPS(8);
        r->diag_inv = row_diag_inv(lim_sign[k] * r->u[6 + i]);
        r->rhs = (-lim_pen[k] * c->contact_erp / dt - lim_sign[k] * qd[i]) * r->diag_inv;
        r->lo = 0; r->hi = c->limit_max_impulse; r->lambda = 0; r->normal_row = -1;
    }
    const int n_noncontact = n_lim + n;
    for (int i = 0; i < n; i++) {
        row_t *r = &rows[nr++];
        memset(r->J, 0, sizeof(double) * ndof);
        r->J[6 + i] = 1.0;
        HGO_FLOPS(11);
        double target = c->kp * (tgt[i] - q[i]) / dt + qd[i] + c->kd * (0.0 - qd[i]);
        if (target > m->max_vel[i]) target = m->max_vel[i];
        if (target < -m->max_vel[i]) target = -m->max_vel[i];
        impulse_response(m, &W, r->J, r->u);
        r->diag_inv = row_diag_inv(r->u[6 + i]);
        r->rhs = (target - qd[i]) * r->diag_inv;
        r->lo = -m->max_force[i] * dt; r->hi = m->max_force[i] * dt;
        r->lambda = 0; r->normal_row = -1;
    }
    const int first_normal = nr;
    for (int k = 0; k < nc; k++) {
        row_t *r = &rows[nr++];
        memset(r->J, 0, sizeof(double) * ndof);
        point_jacobian(m, &W, con[k].body_a, con[k].pt, con[k].n, 1.0, r->J);
        if (con[k].body_b >= -1) point_jacobian(m, &W, con[k].body_b, con[k].pt, con[k].n, -1.0, r->J);
        impulse_response(m, &W, r->J, r->u);
        double den = 0, rv = 0;
        HGO_FLOPS(4 * ndof + 6);
        for (int j = 0; j < ndof; j++) { den += r->J[j] * r->u[j]; rv += r->J[j] * vel[j]; }
        r->diag_inv = row_diag_inv(den);
        double pos_err = 0, vel_err = -rv;
        if (con[k].dist > 0) vel_err -= con[k].dist / dt;
        else pos_err = -con[k].dist * c->contact_erp / dt;
        r->rhs = (pos_err + vel_err) * r->diag_inv;
        r->lo = 0; r->hi = 1e10; r->lambda = 0; r->normal_row = -1;
    }
    const int first_friction = nr;
    for (int k = 0; k < nc; k++) {
        double t[2][3];
        plane_space(con[k].n, t[0], t[1]);
        for (int d = 0; d < 2; d++) {
            row_t *r = &rows[nr++];
            memset(r->J, 0, sizeof(double) * ndof);
            point_jacobian(m, &W, con[k].body_a, con[k].pt, t[d], 1.0, r->J);
            if (con[k].body_b >= -1) point_jacobian(m, &W, con[k].body_b, con[k].pt, t[d], -1.0, r->J);
            impulse_response(m, &W, r->J, r->u);
            double den = 0, rv = 0;
            HGO_FLOPS(4 * ndof + 2);
            for (int j = 0; j < ndof; j++) { den += r->J[j] * r->u[j]; rv += r->J[j] * vel[j]; }
            r->diag_inv = row_diag_inv(den);
            r->rhs = -rv * r->diag_inv;
            r->lo = 0; r->hi = 0; r->lambda = 0; r->normal_row = first_normal + k;
        }
    }

    if (hgo_noise_eps > 0)
        for (int j = 0; j < nr; j++) {
            for (int k = 0; k < ndof; k++) rows[j].u[k] *= noise_factor();
            rows[j].rhs *= noise_factor(); rows[j].diag_inv *= noise_factor();
        }
    memset(dv, 0, sizeof(double) * ndof);
    /* btSequentialImpulseConstraintSolver::solveGroupCacheFriendlyIterations [UPSTREAM-RECALL]: the residual of a sweep is
     * the largest squared velocity residual of its row updates; the loop ends after the first sweep with
     * residual <= m_leastSquaresResidualThreshold (config residual_threshold; 0 = run every sweep). */
    int first_below = -1;
    for (int it = 0; it < c->solver_iters; it++) {
        double lsr = 0.0, rr;
        for (int j = 0; j < n_noncontact; j++) {
            const int idx = (it & 1) ? j : n_noncontact - 1 - j;
            rr = resolve_row(&rows[idx], rows, c->friction, dv, ndof); if (rr * rr > lsr) lsr = rr * rr;
        }
        for (int j = first_normal; j < first_friction; j++) { rr = resolve_row(&rows[j], rows, c->friction, dv, ndof); if (rr * rr > lsr) lsr = rr * rr; }
        for (int j = first_friction; j < nr; j++) { rr = resolve_row(&rows[j], rows, c->friction, dv, ndof); if (rr * rr > lsr) lsr = rr * rr; }
        if (hgo_probe_threshold > 0 && first_below < 0 && lsr <= hgo_probe_threshold) first_below = it + 1;
        if (c->residual_threshold > 0 && lsr <= c->residual_threshold) break;
    }
    if (hgo_probe_threshold > 0) {
        const int slot = first_below < 0 ? (c->solver_iters + 1 > 1025 ? 1025 : c->solver_iters + 1) : (first_below > 1025 ? 1025 : first_below);
#pragma omp atomic
        hgo_sweep_hist[slot]++;
    }

    HGO_FLOPS(ndof + 2 * (3 + n) + 60);    /* v += dv, positions, quaternion update */
    for (int k = 0; k < 3; k++) { st[10 + k] += dv[k]; st[7 + k] += dv[3 + k]; }
    for (int i = 0; i < n; i++) qd[i] += dv[6 + i];
    clamp_velocities(c, st, n);

    /* integrate positions with the new velocities */
    for (int k = 0; k < 3; k++) st[k] += dt * st[7 + k];
    for (int i = 0; i < n; i++) q[i] += dt * qd[i];
    {
        const double *wv = st + 10;
        double ang = sqrt(v3_dot(wv, wv)), ax[3], cw;
        if (ang * dt > 0.7853981633974483) ang = 0.7853981633974483 / dt;  /* Bullet's ANGULAR_MOTION_THRESHOLD */
        if (ang < 0.001) {
            const double f = 0.5 * dt - dt * dt * dt * 0.020833333333 * ang * ang;
            for (int k = 0; k < 3; k++) ax[k] = wv[k] * f;
        } else {
            const double f = sin(0.5 * ang * dt) / ang;
            for (int k = 0; k < 3; k++) ax[k] = wv[k] * f;
        }
        cw = cos(0.5 * ang * dt);
        /* q_new = dq * q  (world-frame angular velocity) */
        const double x = st[3], y = st[4], z = st[5], s = st[6];
        double nq[4] = {
            cw * x + ax[0] * s + ax[1] * z - ax[2] * y,
            cw * y - ax[0] * z + ax[1] * s + ax[2] * x,
            cw * z + ax[0] * y - ax[1] * x + ax[2] * s,
            cw * s - ax[0] * x - ax[1] * y - ax[2] * z};
        const double nn = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
        for (int k = 0; k < 4; k++) st[3 + k] = nq[k] / nn;
    }

    if (info) {
        for (int i = 0; i < n; i++) info->motor_torque[i] = rows[n_lim + i].lambda / dt;
        info->n_contacts = nc;
        info->link_mask = 0;
        for (int k = 0; k < nc; k++) {
            info->pair_a[k] = con[k].body_a; info->pair_b[k] = con[k].body_b;
            info->lambda_n[k] = rows[first_normal + k].lambda;
            info->link_mask |= 1ull << (con[k].body_a + 1);
            if (con[k].body_b >= -1) info->link_mask |= 1ull << (con[k].body_b + 1);
        }
    }
    return 0;
}

/*
 * Commanded planar base velocity (the three leading action slots that
 * pepper_env.py:53 declares): cmd = (vx, vy) in the heading frame of the base LINK frame [m/s] and
 * the yaw rate [rad/s].  Applied like qibullet's base controller drives Pepper [UPSTREAM-RECALL:
 * PepperBaseController.move calls resetBaseVelocity([vx_world, vy_world, 0], [0, 0, theta]) every
 * control cycle]: before every physics step the base velocity is overwritten -- horizontal linear
 * velocity as commanded, no vertical velocity, yaw rate as commanded, no roll/pitch rate.
 */
static void apply_base_cmd(const hg_model *m, double *st, const double *cmd) {
    double R[9], Rf[9], Rl[9];
    quat_to_mat(st + 3, R);
    quat_to_mat(m->frame_quat[0], Rf);
    m3_mul(R, Rf, Rl);
    const double yaw = atan2(Rl[3], Rl[0]), cy = cos(yaw), sy = sin(yaw);
    st[7] = cy * cmd[0] - sy * cmd[1];
    st[8] = sy * cmd[0] + cy * cmd[1];
    st[9] = 0.0;
    st[10] = 0.0; st[11] = 0.0; st[12] = cmd[2];
}

/* ================================================================== */
/* exported API                                                        */
/* ================================================================== */
#ifdef HGO_COUNT_FLOPS
void hgo_flops_reset(void) { hgo_flop_counter = 0.0; }
double hgo_flops_read(void) { return hgo_flop_counter; }      /* of the calling thread: use n_envs <= 3 (no OpenMP team) */
#endif
void hgo_reset(const hg_model *m, double *state, int n_envs) {
    const int n = m->n_links, S = HG_STATE_SIZE(n);
    for (int e = 0; e < n_envs; e++) {
        double *st = state + (size_t)e * S;
        memset(st, 0, sizeof(double) * S);
        memcpy(st, m->base_pos0, 3 * sizeof(double));
        memcpy(st + 3, m->base_quat0, 4 * sizeof(double));
        for (int i = 0; i < n; i++) { st[HG_STATE_BASE + i] = m->q_init[i]; st[HG_STATE_BASE + 2 * n + i] = m->q_init[i]; }
    }
}

/*
 * One env step for n_envs envs: write targets, n_substeps physics steps, read
 * back (dancer_env.py:19-53 / nao_env.py:70-73).
 *   actions   [n_envs][n_sel]   target angles (rad)
 *   joint_sel [n_sel] dof index of each action column; NULL = model act_joint
 *   obs_q / obs_qd [n_envs][m->n_act]  (nullable)
 *   base_xyyaw [n_envs][3] base LINK frame x, y, yaw (nullable)
 *   collision [n_envs] any non-base link in contact (dancer_env.py:32-35)
 *   link_mask [n_envs] bit b: body b-1 touched something in the LAST substep
 *   pairs [n_envs][HG_MAX_CONTACTS][2], n_contacts [n_envs]  (nullable)
 */
int hgo_step_full(const hg_model *m, const hg_config *c, int n_envs, double *state,
                  const double *actions, const int32_t *joint_sel, int n_sel, int n_substeps,
                  double *obs_q, double *obs_qd, double *base_xyyaw,
                  uint8_t *collision, uint64_t *link_mask, int32_t *pairs, int32_t *n_contacts,
                  const double *base_cmd /* [n_envs][3] or NULL */,
                  double *motor_torque /* [n_envs][n_links] or NULL: getJointState()[3] */) {
    const int n = m->n_links, S = HG_STATE_SIZE(n);
    if (joint_sel == NULL) { joint_sel = m->act_joint; n_sel = m->n_act; }
    for (int k = 0; k < n_sel; k++)
        if (joint_sel[k] < 0 || joint_sel[k] >= n) return -2;
    int err = 0;
#pragma omp parallel for schedule(static) reduction(|:err) if (n_envs > 3)
    for (int e = 0; e < n_envs; e++) {
#if defined(__SSE2__)
        /* impulses that decay towards zero otherwise run the solver on subnormals (40x slower) */
        _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
        _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
#endif
        double *st = state + (size_t)e * S;
        double *q = st + HG_STATE_BASE, *qd = q + n, *tgt = qd + n;
        step_info_t info;
        memset(&info, 0, sizeof info);
        if (hgo_noise_eps > 0) noise_begin(e);
        if (actions)
            for (int k = 0; k < n_sel; k++) {
                const int j = joint_sel[k];
                double a = actions[(size_t)e * n_sel + k];
                if (c->clamp_targets) { if (a < m->lower[j]) a = m->lower[j]; if (a > m->upper[j]) a = m->upper[j]; }
                tgt[j] = a;
            }
        for (int s = 0; s < n_substeps; s++) {
            if (base_cmd) apply_base_cmd(m, st, base_cmd + 3 * (size_t)e);
            if (physics_step(m, c, st, &info)) err |= 1;
        }
        for (int k = 0; k < m->n_act; k++) {
            if (obs_q) obs_q[(size_t)e * m->n_act + k] = q[m->act_joint[k]];
            if (obs_qd) obs_qd[(size_t)e * m->n_act + k] = qd[m->act_joint[k]];
        }
        if (base_xyyaw) {
            double R[9], Rf[9], Rl[9], t[3];
            quat_to_mat(st + 3, R);
            quat_to_mat(m->frame_quat[0], Rf);
            m3_mul(R, Rf, Rl);
            m3_mulv(R, m->frame_pos[0], t);
            base_xyyaw[3 * e] = st[0] + t[0];
            base_xyyaw[3 * e + 1] = st[1] + t[1];
            base_xyyaw[3 * e + 2] = atan2(Rl[3], Rl[0]);
        }
        if (motor_torque) for (int i = 0; i < n; i++) motor_torque[(size_t)e * n + i] = info.motor_torque[i];
        if (collision) collision[e] = (info.link_mask >> 1) != 0;
        if (link_mask) link_mask[e] = info.link_mask;
        if (n_contacts) n_contacts[e] = info.n_contacts;
        if (pairs)
            for (int k = 0; k < HG_MAX_CONTACTS; k++) {
                pairs[((size_t)e * HG_MAX_CONTACTS + k) * 2] = k < info.n_contacts ? info.pair_a[k] : -3;
                pairs[((size_t)e * HG_MAX_CONTACTS + k) * 2 + 1] = k < info.n_contacts ? info.pair_b[k] : -3;
            }
    }
    return err ? -1 : 0;
}

int hgo_step_cmd(const hg_model *m, const hg_config *c, int n_envs, double *state,
                 const double *actions, const int32_t *joint_sel, int n_sel, int n_substeps,
                 double *obs_q, double *obs_qd, double *base_xyyaw,
                 uint8_t *collision, uint64_t *link_mask, int32_t *pairs, int32_t *n_contacts,
                 const double *base_cmd) {
    return hgo_step_full(m, c, n_envs, state, actions, joint_sel, n_sel, n_substeps, obs_q, obs_qd, base_xyyaw,
                         collision, link_mask, pairs, n_contacts, base_cmd, NULL);
}

int hgo_step(const hg_model *m, const hg_config *c, int n_envs, double *state,
             const double *actions, const int32_t *joint_sel, int n_sel, int n_substeps,
             double *obs_q, double *obs_qd, double *base_xyyaw,
             uint8_t *collision, uint64_t *link_mask, int32_t *pairs, int32_t *n_contacts) {
    return hgo_step_cmd(m, c, n_envs, state, actions, joint_sel, n_sel, n_substeps, obs_q, obs_qd, base_xyyaw,
                        collision, link_mask, pairs, n_contacts, NULL);
}

/* getLinkState (dancer_env.py:27): per body b = 0..n (0 = base)
 *   com_pos[3], com_quat[4]: internal (inertial) frame in world
 *   frame_pos[3], frame_quat[4]: URDF link frame in world                 */
void hgo_link_states(const hg_model *m, const double *state, int n_envs,
                     double *com_pos, double *com_quat, double *frame_pos, double *frame_quat) {
    const int n = m->n_links, S = HG_STATE_SIZE(n);
    static _Thread_local scratch_t W;
    for (int e = 0; e < n_envs; e++) {
        kinematics(m, state + (size_t)e * S, &W);
        for (int b = 0; b <= n; b++) {
            const size_t o = (size_t)e * (n + 1) + b;
            double Rf[9], Rl[9], t[3];
            memcpy(com_pos + 3 * o, W.xw[b], 3 * sizeof(double));
            mat_to_quat(W.Rw[b], com_quat + 4 * o);
            quat_to_mat(m->frame_quat[b], Rf);
            m3_mul(W.Rw[b], Rf, Rl);
            mat_to_quat(Rl, frame_quat + 4 * o);
            m3_mulv(W.Rw[b], m->frame_pos[b], t);
            for (int k = 0; k < 3; k++) frame_pos[3 * o + k] = W.xw[b][k] + t[k];
        }
    }
}

/* ---- hooks for the physics self-consistency tests ------------------ */

/* unconstrained forward dynamics of one env: acc[6+n] */
int hgo_forward_dynamics(const hg_model *m, const hg_config *c, const double *state, const double *tau, double *acc) {
    static _Thread_local scratch_t W;
    kinematics(m, state, &W);
    return aba(m, c, state, tau, &W, acc);
}

/* M^-1 f for one env (f, out: [6+n]) */
int hgo_apply_minv(const hg_model *m, const hg_config *c, const double *state, const double *f, double *out) {
    static _Thread_local scratch_t W;
    double acc[NDOF_MAX];
    kinematics(m, state, &W);
    if (aba(m, c, state, NULL, &W, acc)) return -1;
    impulse_response(m, &W, f, out);
    return 0;
}

/* kinetic energy, potential energy, linear momentum[3], angular momentum about origin[3] */
void hgo_energy_momentum(const hg_model *m, const hg_config *c, const double *state, double *out8) {
    static _Thread_local scratch_t W;
    const int n = m->n_links;
    const double *qd = state + HG_STATE_BASE + n;
    double v[NL + 1][6];
    kinematics(m, state, &W);
    m3_tmulv(W.Rw[0], state + 10, v[0]);
    m3_tmulv(W.Rw[0], state + 7, v[0] + 3);
    memset(out8, 0, 8 * sizeof(double));
    for (int b = 0; b <= n; b++) {
        const int i = b - 1;
        if (b > 0) {
            motion_xform(W.E[i], W.r[i], v[m->parent[i] + 1], v[b]);
            for (int k = 0; k < 6; k++) v[b][k] += W.S[i][k] * qd[i];
        }
        const double mass = b ? m->mass[i] : m->base_mass;
        const double *I = b ? m->inertia[i] : m->base_inertia;
        double Iw[3] = {I[0] * v[b][0], I[1] * v[b][1], I[2] * v[b][2]}, Lw[3], pw[3], t[3];
        out8[0] += 0.5 * (v3_dot(v[b], Iw) + mass * v3_dot(v[b] + 3, v[b] + 3));
        out8[1] -= mass * v3_dot(c->gravity, W.xw[b]);
        m3_mulv(W.Rw[b], v[b] + 3, pw);
        m3_mulv(W.Rw[b], Iw, Lw);
        for (int k = 0; k < 3; k++) pw[k] *= mass;
        v3_cross(W.xw[b], pw, t);
        for (int k = 0; k < 3; k++) { out8[2 + k] += pw[k]; out8[5 + k] += Lw[k] + t[k]; }
    }
}

/*
 * How close the contact SET of one env's pose is to changing (tests/test_gpu_parity.py: the integer
 * outputs of the FP32 kernel may differ from this restatement's only where this is within FP32
 * rounding of a position).  out[0]: smallest |distance - contact_margin| over all ground-tested spheres
 * and (with enable_self) all candidate sphere pairs; out[1]: smallest gap between the 4th and the
 * 5th smallest in-margin distance of one body (the "4 deepest per body" rule flips there); 1e30 when
 * there is no such candidate.
 */
void hgo_contact_slack(const hg_model *m, const hg_config *c, const double *state, double *out2) {
    static _Thread_local scratch_t W;
    static _Thread_local double cw[HG_MAX_SPHERES][3], hw[HG_MAX_SPHERES][3], dist[HG_MAX_SPHERES];
    kinematics(m, state, &W);
    out2[0] = 1e30; out2[1] = 1e30;
    for (int s = 0; s < m->n_spheres; s++) {
        const int b = m->sphere_link[s] + 1;
        double t[3];
        m3_mulv(W.Rw[b], m->sphere_pos[s], t);
        m3_mulv(W.Rw[b], m->sphere_half[s], hw[s]);
        for (int k = 0; k < 3; k++) cw[s][k] = W.xw[b][k] + t[k];
        dist[s] = cw[s][2] - m->sphere_radius[s];
        if (c->enable_ground && (m->sphere_flags[s] & 1)) {
            const double g = fabs(dist[s] - c->contact_margin);
            if (g < out2[0]) out2[0] = g;
        }
    }
    if (c->enable_ground)
        for (int s = 0; s < m->n_spheres; s++) {
            if (!(m->sphere_flags[s] & 1) || !(dist[s] < c->contact_margin)) continue;
            /* in-margin distances of this sphere's body, sorted: gap between ranks 3 and 4 (0-based) */
            static _Thread_local double d[HG_MAX_SPHERES]; int k = 0;
            for (int o = 0; o < m->n_spheres; o++)
                if (m->sphere_link[o] == m->sphere_link[s] && (m->sphere_flags[o] & 1) && dist[o] < c->contact_margin) d[k++] = dist[o];
            if (k <= 4) continue;
            for (int i = 1; i < k; i++) { double v = d[i]; int j = i - 1; while (j >= 0 && d[j] > v) { d[j + 1] = d[j]; j--; } d[j + 1] = v; }
            if (d[4] - d[3] < out2[1]) out2[1] = d[4] - d[3];
        }
    if (c->enable_self)
        for (int p = 0; p < m->n_pairs; p++) {
            const int a = m->pair_a[p], b = m->pair_b[p];
            double c1[3], c2[3];
            seg_seg_closest(cw[a], hw[a], cw[b], hw[b], c1, c2);
            double dd[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
            const double ds = sqrt(v3_dot(dd, dd)) - m->sphere_radius[a] - m->sphere_radius[b];
            const double g = fabs(ds - c->contact_margin);
            if (g < out2[0]) out2[0] = g;
        }
}

/* world-frame pose of every body, for kinematics checks: Rw[(n+1)*9], xw[(n+1)*3] */
void hgo_kinematics(const hg_model *m, const double *state, double *Rw, double *xw) {
    static _Thread_local scratch_t W;
    kinematics(m, state, &W);
    memcpy(Rw, W.Rw, sizeof(double) * 9 * (m->n_links + 1));
    memcpy(xw, W.xw, sizeof(double) * 3 * (m->n_links + 1));
}
