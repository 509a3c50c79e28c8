// ubench_inst.hip -- issue cost of single VALU instruction kinds, 8 independent streams per wave.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_inst tools/ubench_inst.hip     Run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ long long g_clk[2];
#define REP8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
#define BODY4(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)        // 32 instructions

#define KERNEL(NAME, ASM8)                                                               \
__global__ __launch_bounds__(64) void NAME(float *out, const float *in, int iters) {     \
    float x[8]; float c = in[threadIdx.x], d = in[64 + threadIdx.x];                     \
    for (int k = 0; k < 8; k++) x[k] = in[128 + 8 * k + (threadIdx.x & 7)];              \
    int s[8]; for (int k = 0; k < 8; k++) s[k] = __builtin_amdgcn_readfirstlane(__float_as_int(x[k])); \
    const long long t0 = clock64(), w0 = wall_clock64();                                  \
    for (int it = 0; it < iters; ++it) { ASM8 ASM8 ASM8 ASM8 }                            \
    const long long t1 = clock64(), w1 = wall_clock64();                                  \
    float t = 0; for (int k = 0; k < 8; k++) t += x[k] + __int_as_float(s[k]);            \
    out[blockIdx.x * 64 + threadIdx.x] = t;                                               \
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = t1 - t0; g_clk[1] = w1 - w0; }  \
}

#define A_FMA3 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c), "v"(d));
#define A_FMA3S asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n" \
    "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c), "s"(s[0]));
#define A_FMAC asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n" \
    "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c), "v"(d));
#define A_FMACS asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n" \
    "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "s"(s[0]), "v"(d));
#define A_MUL asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n" \
    "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c));
#define A_MED3 asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n" \
    "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c), "v"(d));
#define A_MAXMIN asm volatile("v_max_f32 %0, %0, %8\n v_min_f32 %1, %1, %9\n v_max_f32 %2, %2, %8\n v_min_f32 %3, %3, %9\n" \
    "v_max_f32 %4, %4, %8\n v_min_f32 %5, %5, %9\n v_max_f32 %6, %6, %8\n v_min_f32 %7, %7, %9" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c), "v"(d));
#define A_RDLANE asm volatile("v_readlane_b32 %0, %8, 3\n v_readlane_b32 %1, %9, 5\n v_readlane_b32 %2, %10, 7\n v_readlane_b32 %3, %11, 9\n" \
    "v_readlane_b32 %4, %12, 11\n v_readlane_b32 %5, %13, 13\n v_readlane_b32 %6, %14, 15\n v_readlane_b32 %7, %15, 17" \
    : "=s"(s[0]), "=s"(s[1]), "=s"(s[2]), "=s"(s[3]), "=s"(s[4]), "=s"(s[5]), "=s"(s[6]), "=s"(s[7]) \
    : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
#define A_MOV asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n" \
    "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
#define A_PKFMA asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n" \
    "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" \
    : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(pc), "v"(pd));

KERNEL(k_fma3, A_FMA3)
KERNEL(k_fma3s, A_FMA3S)
KERNEL(k_fmac, A_FMAC)
KERNEL(k_fmacs, A_FMACS)
KERNEL(k_mul, A_MUL)
KERNEL(k_med3, A_MED3)
KERNEL(k_maxmin, A_MAXMIN)
KERNEL(k_rdlane, A_RDLANE)
KERNEL(k_mov, A_MOV)


#define A_FMACS8 asm volatile("v_fmac_f32 %0, %8, %16\n v_fmac_f32 %1, %9, %16\n v_fmac_f32 %2, %10, %16\n v_fmac_f32 %3, %11, %16\n" \
    "v_fmac_f32 %4, %12, %16\n v_fmac_f32 %5, %13, %16\n v_fmac_f32 %6, %14, %16\n v_fmac_f32 %7, %15, %16" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) \
    : "s"(s[0]), "s"(s[1]), "s"(s[2]), "s"(s[3]), "s"(s[4]), "s"(s[5]), "s"(s[6]), "s"(s[7]), "v"(d));
#define A_MAX0 asm volatile("v_max_f32 %0, 0, %0\n v_max_f32 %1, 0, %1\n v_max_f32 %2, 0, %2\n v_max_f32 %3, 0, %3\n" \
    "v_max_f32 %4, 0, %4\n v_max_f32 %5, 0, %5\n v_max_f32 %6, 0, %6\n v_max_f32 %7, 0, %7" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
#define A_ADD asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n" \
    "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c));
#define A_MAXI asm volatile("v_max_i32 %0, %0, %8\n v_max_i32 %1, %1, %8\n v_max_i32 %2, %2, %8\n v_max_i32 %3, %3, %8\n" \
    "v_max_i32 %4, %4, %8\n v_max_i32 %5, %5, %8\n v_max_i32 %6, %6, %8\n v_max_i32 %7, %7, %8" \
    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c));
KERNEL(k_fmacs8, A_FMACS8)
KERNEL(k_max0, A_MAX0)
KERNEL(k_add, A_ADD)
KERNEL(k_maxi, A_MAXI)

typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k_pkfma(float *out, const float *in, int iters) {
    f2 p[4]; f2 pc = {in[threadIdx.x], in[threadIdx.x + 1]}, pd = {in[64 + threadIdx.x], in[65 + threadIdx.x]};
    for (int k = 0; k < 4; k++) p[k] = f2{in[128 + k], in[140 + k]};
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) { A_PKFMA A_PKFMA A_PKFMA A_PKFMA }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = t1 - t0; g_clk[1] = w1 - w0; }
    float t = 0; for (int k = 0; k < 4; k++) t += p[k].x + p[k].y;
    out[blockIdx.x * 64 + threadIdx.x] = t;
}

typedef void (*kern_t)(float *, const float *, int);
static void run(const char *name, kern_t k, int w, int iters, float *out, float *in) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(1024 * w), dim3(64), 0, 0, out, in, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(1024 * w), dim3(64), 0, 0, out, in, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    long long hc[2]; CHECK(hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), sizeof(hc)));
    printf("%-10s W=%d  %7.3f ms  %5.2f clk/inst/SIMD at 2.4 GHz | wave 0: memtime %lld ticks, realtime %lld ticks (100 MHz) -> %.0f ns/inst/wave\n", name, w, ms,
           2.4e9 * ms * 1e-3 / ((double)iters * 32 * w), hc[0], hc[1], hc[1] * 10.0 / ((double)iters * 32));
}

int main() {
    const int iters = 40000;
    float *out, *in;
    CHECK(hipMalloc(&out, sizeof(float) * 64 * 1024 * 8));
    CHECK(hipMalloc(&in, sizeof(float) * 512));
    float h[512];
    for (int i = 0; i < 512; i++) h[i] = 0.9f + 0.0001f * (i % 17);
    CHECK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
    for (int w : {1, 4}) {
        run("fma3 vvv", k_fma3, w, iters, out, in);
        run("fma3 vs", k_fma3s, w, iters, out, in);
        run("fmac vv", k_fmac, w, iters, out, in);
        run("fmac sv", k_fmacs, w, iters, out, in);
        run("mul", k_mul, w, iters, out, in);
        run("med3", k_med3, w, iters, out, in);
        run("max/min", k_maxmin, w, iters, out, in);
        run("readlane", k_rdlane, w, iters, out, in);
        run("mov", k_mov, w, iters, out, in);
        run("pk_fma", k_pkfma, w, iters, out, in);
        run("fmac s8v", k_fmacs8, w, iters, out, in);
        run("max 0", k_max0, w, iters, out, in);
        run("add", k_add, w, iters, out, in);
        run("max_i32", k_maxi, w, iters, out, in);
    }
    return 0;
}
