// ubench_valu.hip -- what does one VALU instruction cost in the shapes the step kernel uses?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_valu tools/ubench_valu.hip     Run on the GPU box.
// Each kernel runs ITERS trips of a 16-row body on every wave; grid = 1024 SIMDs x W waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 16;

// (a) 8 independent fma chains
__global__ __launch_bounds__(64) void k_indep(float *out, int iters, float s) {
    float x[8];
    for (int k = 0; k < 8; k++) x[k] = threadIdx.x + k;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < ROWS * 4 / 8; r++)
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = fmaf(x[k], s, 1.0f);
    }
    float t = 0; for (int k = 0; k < 8; k++) t += x[k];
    out[blockIdx.x * 64 + threadIdx.x] = t;
}

// (b) one dependent fma chain
__global__ __launch_bounds__(64) void k_dep(float *out, int iters, float s) {
    float x = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < ROWS * 4; r++) x = fmaf(x, s, 1.0f);
    }
    out[blockIdx.x * 64 + threadIdx.x] = x;
}

// (c) the PGS row as the step kernel has it: med3, readlane, fma, fma (impulses in SGPRs)
__global__ __launch_bounds__(64) void k_pgs4(float *out, const float *in, int iters) {
    float a[ROWS];
    for (int k = 0; k < ROWS; k++) a[k] = in[k * 64 + threadIdx.x];
    float z = in[ROWS * 64 + threadIdx.x];
    const float hi = in[(ROWS + 1) * 64 + threadIdx.x], lo = -hi;
    int slam[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) slam[k] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const float cand = __builtin_amdgcn_fmed3f(z, lo, hi);
            const int s_new = __builtin_amdgcn_readlane(__float_as_int(cand), i);
            z = fmaf(-a[i], __int_as_float(slam[i]), z);
            z = fmaf(a[i], __int_as_float(s_new), z);
            slam[i] = s_new;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = z + __int_as_float(slam[3]);
}

// (d) clamp on the scalar unit: readlane, 4 SALU, fma, fma (bounds +-1 after scaling)
__global__ __launch_bounds__(64) void k_pgs3(float *out, const float *in, int iters) {
    float a[ROWS];
    for (int k = 0; k < ROWS; k++) a[k] = in[k * 64 + threadIdx.x];
    float z = in[ROWS * 64 + threadIdx.x];
    int slam[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) slam[k] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const unsigned s = (unsigned)__builtin_amdgcn_readlane(__float_as_int(z), i);
            int s_new, tmp;            // |x| clamp to 1.0 with integer compares, sign put back: all on the scalar unit
            asm volatile("s_and_b32 %0, %2, 0x7fffffff\n\ts_and_b32 %1, %2, 0x80000000\n\ts_min_u32 %0, %0, 0x3f800000\n\ts_or_b32 %0, %0, %1"
                         : "=&s"(s_new), "=&s"(tmp) : "s"(s) : "scc");
            z = fmaf(-a[i], __int_as_float(slam[i]), z);
            z = fmaf(a[i], __int_as_float(s_new), z);
            slam[i] = s_new;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = z + __int_as_float(slam[3]);
}

// (e) one-sided clamp on the scalar unit: readlane, s_max_i32, fma, fma
__global__ __launch_bounds__(64) void k_pgs3n(float *out, const float *in, int iters) {
    float a[ROWS];
    for (int k = 0; k < ROWS; k++) a[k] = in[k * 64 + threadIdx.x];
    float z = in[ROWS * 64 + threadIdx.x];
    int slam[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) slam[k] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const int s = __builtin_amdgcn_readlane(__float_as_int(z), i);
            const int s_new = s > 0 ? s : 0;
            z = fmaf(-a[i], __int_as_float(slam[i]), z);
            z = fmaf(a[i], __int_as_float(s_new), z);
            slam[i] = s_new;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = z + __int_as_float(slam[3]);
}

// (f) delta form with the impulse vector in a VGPR: med3, sub, readlane, fma + masked move
__global__ __launch_bounds__(64) void k_pgs_delta(float *out, const float *in, int iters) {
    float a[ROWS];
    for (int k = 0; k < ROWS; k++) a[k] = in[k * 64 + threadIdx.x];
    float z = in[ROWS * 64 + threadIdx.x];           // z_j = lam_j + r_j with a_jj = 0
    const float hi = in[(ROWS + 1) * 64 + threadIdx.x], lo = -hi;
    float lam = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const float cand = __builtin_amdgcn_fmed3f(z, lo, hi);
            const float d = cand - lam;
            const float di = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), i));
            z = fmaf(a[i], di, z);
            lam = (threadIdx.x == i) ? cand : lam;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = z + lam;
}


// (g) two envs in one wave, rows interleaved: every instruction's neighbour belongs to the other env
__global__ __launch_bounds__(64) void k_pgs4x2(float *out, const float *in, int iters) {
    float a[ROWS], b[ROWS];
    for (int k = 0; k < ROWS; k++) { a[k] = in[k * 64 + threadIdx.x]; b[k] = 0.5f * in[k * 64 + (threadIdx.x ^ 1)]; }
    float za = in[ROWS * 64 + threadIdx.x], zb = 0.7f * za;
    const float hi = in[(ROWS + 1) * 64 + threadIdx.x], lo = -hi;
    int sa[ROWS], sb[ROWS];
#pragma unroll
    for (int k = 0; k < ROWS; k++) { sa[k] = 0; sb[k] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const float ca = __builtin_amdgcn_fmed3f(za, lo, hi);
            const float cb = __builtin_amdgcn_fmed3f(zb, lo, hi);
            const int na = __builtin_amdgcn_readlane(__float_as_int(ca), i);
            const int nb = __builtin_amdgcn_readlane(__float_as_int(cb), i);
            za = fmaf(-a[i], __int_as_float(sa[i]), za);
            zb = fmaf(-b[i], __int_as_float(sb[i]), zb);
            za = fmaf(a[i], __int_as_float(na), za);
            zb = fmaf(b[i], __int_as_float(nb), zb);
            sa[i] = na; sb[i] = nb;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = za + zb + __int_as_float(sa[3]) + __int_as_float(sb[5]);
}

// (h) two envs, delta form: med3, sub, readlane, fma, one-lane move of the new impulse (no scalar storage)
__global__ __launch_bounds__(64) void k_deltax2(float *out, const float *in, int iters) {
    float a[ROWS], b[ROWS];
    for (int k = 0; k < ROWS; k++) { a[k] = in[k * 64 + threadIdx.x]; b[k] = 0.5f * in[k * 64 + (threadIdx.x ^ 1)]; }
    float za = in[ROWS * 64 + threadIdx.x], zb = 0.7f * za;
    const float hi = in[(ROWS + 1) * 64 + threadIdx.x], lo = -hi;
    float la = 0.f, lb = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ROWS; i++) {
            const float ca = __builtin_amdgcn_fmed3f(za, lo, hi);
            const float cb = __builtin_amdgcn_fmed3f(zb, lo, hi);
            const float da = ca - la, db = cb - lb;
            const float ra = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(da), i));
            const float rb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(db), i));
            za = fmaf(a[i], ra, za);
            zb = fmaf(b[i], rb, zb);
            asm volatile("s_mov_b64 exec, %4\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\ts_mov_b64 exec, -1"
                         : "+v"(la), "+v"(lb) : "v"(ca), "v"(cb), "i"(1 << i));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = za + zb + la + lb;
}

template <typename F> static void run(const char *name, F launch, int waves_per_simd, int iters, int vinst_per_row) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 1024 * waves_per_simd;
    launch(grid, iters);                       // warm
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    launch(grid, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double rows = (double)iters * ROWS;
    const double clk = 2.4e9 * ms * 1e-3;
    printf("%-12s W=%2d  %8.3f ms  %7.2f clk/row/SIMD-slot  %6.2f clk per VALU inst per SIMD (%d VALU/row)\n",
           name, waves_per_simd, ms, clk / rows, clk / (rows * waves_per_simd * vinst_per_row), vinst_per_row);
}

int main() {
    const int iters = 20000;
    float *out, *in;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * 1024 * 16));
    CHECK(hipMalloc(&in, sizeof(float) * 64 * (ROWS + 2)));
    float h[64 * (ROWS + 2)];
    for (int i = 0; i < 64 * (ROWS + 2); i++) h[i] = 0.001f * ((i * 37) % 101 - 50);
    for (int i = 0; i < 64; i++) { h[ROWS * 64 + i] = 0.3f + 0.01f * i; h[(ROWS + 1) * 64 + i] = 0.5f; }
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int w : {1, 2, 4}) {
        run("indep8", [&](int g, int it) { hipLaunchKernelGGL(k_indep, dim3(g), dim3(64), 0, 0, out, it, 0.999f); }, w, iters, 4);
        run("dep", [&](int g, int it) { hipLaunchKernelGGL(k_dep, dim3(g), dim3(64), 0, 0, out, it, 0.999f); }, w, iters, 4);
        run("pgs4", [&](int g, int it) { hipLaunchKernelGGL(k_pgs4, dim3(g), dim3(64), 0, 0, out, in, it); }, w, iters, 4);
        run("pgs3(salu4)", [&](int g, int it) { hipLaunchKernelGGL(k_pgs3, dim3(g), dim3(64), 0, 0, out, in, it); }, w, iters, 3);
        run("pgs3n(smax)", [&](int g, int it) { hipLaunchKernelGGL(k_pgs3n, dim3(g), dim3(64), 0, 0, out, in, it); }, w, iters, 3);
        run("pgs4x2", [&](int g, int it) { hipLaunchKernelGGL(k_pgs4x2, dim3(g), dim3(64), 0, 0, out, in, it); }, w, iters, 8);
        run("deltax2", [&](int g, int it) { hipLaunchKernelGGL(k_deltax2, dim3(g), dim3(64), 0, 0, out, in, it); }, w, iters, 10);
        run("pgs_delta", [&](int g, int it) { hipLaunchKernelGGL(k_pgs_delta, dim3(g), dim3(64), 0, 0, out, in, it); }, w, iters, 5);
    }
    return 0;
}
