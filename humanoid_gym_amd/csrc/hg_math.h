// hg_math.h -- register-resident 3-vector / 3x3 / spatial helpers (FP32 VALU).
// Everything is forced inline with compile-time indices only, so that hipcc
// keeps all of it in VGPRs (no private-memory arrays).
#pragma once
#include <hip/hip_runtime.h>

namespace hg {

#define HG_FI __device__ __forceinline__

struct V3 { float x, y, z; };
HG_FI V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
HG_FI V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
HG_FI V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
HG_FI V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
HG_FI V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
HG_FI float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
HG_FI V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// clamp to [-m, m] that lets NaN through (v_med3 / fmin / fmax would turn it into a bound and hide it from the
// non-finite guard at the end of the step)
HG_FI float clampv(float x, float m) { x = x > m ? m : x; return x < -m ? -m : x; }
HG_FI V3 clamp3(V3 a, float m) { return v3(clampv(a.x, m), clampv(a.y, m), clampv(a.z, m)); }
HG_FI V3 fma3(V3 a, float s, V3 c) { return v3(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z)); }

struct M3 { float m[9]; };  // row-major
HG_FI V3 row(const M3 &A, int i) { return v3(A.m[3 * i], A.m[3 * i + 1], A.m[3 * i + 2]); }
HG_FI V3 mulv(const M3 &A, V3 v) { return v3(dot(row(A, 0), v), dot(row(A, 1), v), dot(row(A, 2), v)); }
HG_FI V3 tmulv(const M3 &A, V3 v) {
    return v3(fmaf(A.m[0], v.x, fmaf(A.m[3], v.y, A.m[6] * v.z)),
              fmaf(A.m[1], v.x, fmaf(A.m[4], v.y, A.m[7] * v.z)),
              fmaf(A.m[2], v.x, fmaf(A.m[5], v.y, A.m[8] * v.z)));
}
HG_FI M3 mul(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C.m[3 * i + j] = fmaf(A.m[3 * i], B.m[j], fmaf(A.m[3 * i + 1], B.m[3 + j], A.m[3 * i + 2] * B.m[6 + j]));
    return C;
}
HG_FI M3 mul_bt(const M3 &A, const M3 &B) {  // A * B^T
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C.m[3 * i + j] = fmaf(A.m[3 * i], B.m[3 * j], fmaf(A.m[3 * i + 1], B.m[3 * j + 1], A.m[3 * i + 2] * B.m[3 * j + 2]));
    return C;
}
HG_FI M3 tmul(const M3 &A, const M3 &B) {    // A^T * B
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            C.m[3 * i + j] = fmaf(A.m[i], B.m[j], fmaf(A.m[3 + i], B.m[3 + j], A.m[6 + i] * B.m[6 + j]));
    return C;
}

struct Sym3 { float xx, xy, xz, yy, yz, zz; };
HG_FI M3 full(Sym3 s) {
    M3 A;
    A.m[0] = s.xx; A.m[1] = s.xy; A.m[2] = s.xz;
    A.m[3] = s.xy; A.m[4] = s.yy; A.m[5] = s.yz;
    A.m[6] = s.xz; A.m[7] = s.yz; A.m[8] = s.zz;
    return A;
}
HG_FI Sym3 symm(const M3 &A) {
    Sym3 s; s.xx = A.m[0]; s.xy = A.m[1]; s.xz = A.m[2]; s.yy = A.m[4]; s.yz = A.m[5]; s.zz = A.m[8];
    return s;
}
HG_FI V3 symv(Sym3 s, V3 v) {
    return v3(fmaf(s.xx, v.x, fmaf(s.xy, v.y, s.xz * v.z)),
              fmaf(s.xy, v.x, fmaf(s.yy, v.y, s.yz * v.z)),
              fmaf(s.xz, v.x, fmaf(s.yz, v.y, s.zz * v.z)));
}
// s -= u u^T * k
HG_FI void sym_rank1_sub(Sym3 &s, V3 u, float k) {
    s.xx = fmaf(-u.x * k, u.x, s.xx); s.xy = fmaf(-u.x * k, u.y, s.xy); s.xz = fmaf(-u.x * k, u.z, s.xz);
    s.yy = fmaf(-u.y * k, u.y, s.yy); s.yz = fmaf(-u.y * k, u.z, s.yz); s.zz = fmaf(-u.z * k, u.z, s.zz);
}

// spatial vector: angular part a, linear part l (axes parallel to the world's, origin at a body's COM)
struct Sv { V3 a, l; };
HG_FI Sv operator+(Sv p, Sv q) { Sv r; r.a = p.a + q.a; r.l = p.l + q.l; return r; }
HG_FI float sdot(Sv p, Sv q) { return dot(p.a, q.a) + dot(p.l, q.l); }
HG_FI Sv sfma(Sv p, float s, Sv c) { Sv r; r.a = fma3(p.a, s, c.a); r.l = fma3(p.l, s, c.l); return r; }

// articulated inertia [[A, B], [B^T, C]], A and C symmetric
struct AI { Sym3 A; M3 B; Sym3 C; };
HG_FI Sv ai_mul(const AI &I, Sv v) {
    Sv f;
    f.a = symv(I.A, v.a) + mulv(I.B, v.l);
    f.l = tmulv(I.B, v.a) + symv(I.C, v.l);
    return f;
}
HG_FI M3 quat_to_mat(float x, float y, float z, float w) {
    M3 R;
    R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - z * w);     R.m[2] = 2 * (x * z + y * w);
    R.m[3] = 2 * (x * y + z * w);     R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - x * w);
    R.m[6] = 2 * (x * z - y * w);     R.m[7] = 2 * (y * z + x * w);     R.m[8] = 1 - 2 * (x * x + y * y);
    return R;
}
HG_FI void mat_to_quat(const M3 &R, float q[4]) {
    const float *m = R.m;
    float t = m[0] + m[4] + m[8], s;
    if (t > 0) { s = sqrtf(t + 1.0f) * 2; q[3] = 0.25f * s; q[0] = (m[7] - m[5]) / s; q[1] = (m[2] - m[6]) / s; q[2] = (m[3] - m[1]) / s; }
    else if (m[0] > m[4] && m[0] > m[8]) { s = sqrtf(1.0f + m[0] - m[4] - m[8]) * 2; q[3] = (m[7] - m[5]) / s; q[0] = 0.25f * s; q[1] = (m[1] + m[3]) / s; q[2] = (m[2] + m[6]) / s; }
    else if (m[4] > m[8]) { s = sqrtf(1.0f + m[4] - m[0] - m[8]) * 2; q[3] = (m[2] - m[6]) / s; q[0] = (m[1] + m[3]) / s; q[1] = 0.25f * s; q[2] = (m[5] + m[7]) / s; }
    else { s = sqrtf(1.0f + m[8] - m[0] - m[4]) * 2; q[3] = (m[3] - m[1]) / s; q[0] = (m[2] + m[6]) / s; q[1] = (m[5] + m[7]) / s; q[2] = 0.25f * s; }
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
}
// rotation about unit axis a by angle with sine s and cosine c
HG_FI M3 axis_rot(V3 a, float s, float c) {
    const float t = 1 - c;
    M3 R;
    R.m[0] = t * a.x * a.x + c;       R.m[1] = t * a.x * a.y - s * a.z; R.m[2] = t * a.x * a.z + s * a.y;
    R.m[3] = t * a.x * a.y + s * a.z; R.m[4] = t * a.y * a.y + c;       R.m[5] = t * a.y * a.z - s * a.x;
    R.m[6] = t * a.x * a.z - s * a.y; R.m[7] = t * a.y * a.z + s * a.x; R.m[8] = t * a.z * a.z + c;
    return R;
}

// Bullet's btPlaneSpace1: unit tangents p, q orthogonal to unit n
HG_FI void plane_space(V3 n, V3 &p, V3 &q) {
    if (fabsf(n.z) > 0.70710678f) {
        const float a = n.y * n.y + n.z * n.z, k = rsqrtf(a);
        p = v3(0.f, -n.z * k, n.y * k);
        q = v3(a * k, -n.x * p.z, n.x * p.y);
    } else {
        const float a = n.x * n.x + n.y * n.y, k = rsqrtf(a);
        p = v3(-n.y * k, n.x * k, 0.f);
        q = v3(-n.z * p.y, n.z * p.x, a * k);
    }
}

// LDS / global helpers
HG_FI V3 ld3(const float *p) { return v3(p[0], p[1], p[2]); }
HG_FI void st3(float *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
HG_FI M3 ldm(const float *p) { M3 A;
#pragma unroll
    for (int i = 0; i < 9; i++) A.m[i] = p[i];
    return A; }
HG_FI void stm(float *p, const M3 &A) {
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = A.m[i]; }
HG_FI Sv ldsv(const float *p) { Sv s; s.a = ld3(p); s.l = ld3(p + 3); return s; }
HG_FI void stsv(float *p, Sv s) { st3(p, s.a); st3(p + 3, s.l); }

// Cholesky factor of a symmetric positive definite 6x6 (full storage, row-major; only the lower triangle of A is read):
// L lower triangular in full storage, rd = reciprocals of its diagonal
HG_FI void spd6_cholesky(const float A[36], float L[36], float rd[6]) {
#pragma unroll
    for (int i = 0; i < 36; i++) L[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            float s = A[6 * i + j];
#pragma unroll
            for (int k = 0; k < j; k++) s = fmaf(-L[6 * i + k], L[6 * j + k], s);
            if (i == j) { rd[i] = rsqrtf(fmaxf(s, 1e-30f)); L[6 * i + i] = fmaxf(s, 1e-30f) * rd[i]; }    // v_rsq_f32, 1 ulp
            else L[6 * i + j] = s * rd[j];
        }
    }
}
// x = (L L^T)^-1 b: one forward and one backward substitution (30 fma + 12 mul)
HG_FI void chol6_solve(const float L[36], const float rd[6], const float b[6], float x[6]) {
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s = fmaf(-L[6 * i + k], y[k], s);
        y[i] = s * rd[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        float s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s = fmaf(-L[6 * k + i], x[k], s);
        x[i] = s * rd[i];
    }
}
// y = -Inv * z for spatial vectors
HG_FI Sv neg_mul6(const float *Inv, Sv z) {
    const float zz[6] = {z.a.x, z.a.y, z.a.z, z.l.x, z.l.y, z.l.z};
    float y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) s = fmaf(-Inv[6 * i + k], zz[k], s);
        y[i] = s;
    }
    Sv r; r.a = v3(y[0], y[1], y[2]); r.l = v3(y[3], y[4], y[5]);
    return r;
}

// Closest points of two sphere-swept segments (centre c, half axis h; h = 0: a sphere), the float32 statement of
// oracle/hg_oracle.c seg_seg_closest: exact solution (cross-product form: no cancellation in the denominator) blended
// towards the centre of the overlap interval as the segments become parallel, so that the contact point is continuous in
// the pose.  c1 on a, c2 on b.
#define HG_SEG_EPS   1e-12f
#define HG_SEG_BLEND 0.03f
HG_FI float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
HG_FI void seg_seg_closest(V3 ca, V3 ha, V3 cb, V3 hb, V3 &c1, V3 &c2) {
    const V3 p1 = ca - ha, d1 = ha * 2.f, p2 = cb - hb, d2 = hb * 2.f, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r), c = dot(d1, r), b = dot(d1, d2);
    const float ia = a > HG_SEG_EPS ? 1.f / a : 0.f, ie = e > HG_SEG_EPS ? 1.f / e : 0.f;
    const V3 nx = cross(d1, d2);
    const float ae = a * e, denom = dot(nx, nx);
    const float w = denom < HG_SEG_BLEND * ae ? denom / (HG_SEG_BLEND * ae) : 1.f;
    const float s_exact = denom > 0.f ? clamp01(-dot(cross(r, d2), nx) / denom) : 0.f;
    const float sb0 = -c * ia, sb1 = (b - c) * ia;
    const float s_mid = 0.5f * (clamp01(fminf(sb0, sb1)) + clamp01(fmaxf(sb0, sb1)));
    float s = s_mid + w * (s_exact - s_mid);
    const float t = clamp01((b * s + f) * ie);
    s = clamp01((b * t - c) * ia);
    c1 = p1 + d1 * s; c2 = p2 + d2 * t;
}

}  // namespace hg
