// ubench_fp64.hip -- issue cost of the fp64 VALU instructions the Cooper-Frye hot loop uses (gfx950).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench_fp64.hip -o /tmp/ub && /tmp/ub
// Each kernel runs ITER x 16 instructions on 8 independent registers per lane; all SIMDs loaded with
// WAVES waves.  Prints lane-ops/s and cycles per wave-instruction per SIMD at the measured clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 2000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFKERNEL(NAME, ASM)                                                              \
    __global__ void __launch_bounds__(256) NAME(double *out, double c, int n)             \
    {                                                                                     \
        double x0 = 1.0 + threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;      \
        double x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;                        \
        for (int i = 0; i < ITER; i++) {                                                  \
            ASM ASM                                                                       \
        }                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; \
    }

#define A_FMA asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n" \
                           "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n" \
                           : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
#define A_OP2(OP) asm volatile(OP " %0, %0, %8\n " OP " %1, %1, %8\n " OP " %2, %2, %8\n " OP " %3, %3, %8\n" \
                               OP " %4, %4, %8\n " OP " %5, %5, %8\n " OP " %6, %6, %8\n " OP " %7, %7, %8\n" \
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
#define A_OP1(OP) asm volatile(OP " %0, %0\n " OP " %1, %1\n " OP " %2, %2\n " OP " %3, %3\n" \
                               OP " %4, %4\n " OP " %5, %5\n " OP " %6, %6\n " OP " %7, %7\n" \
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
#define A_LDEXP asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n" \
                             "v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8\n" \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(n));
#define A_MOV asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n" \
                           "v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8\n" \
                           : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
#define A_FMA32 asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n" \
                             "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n" \
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fc));

DEFKERNEL(k_fma, A_FMA)
DEFKERNEL(k_mul, A_OP2("v_mul_f64"))
DEFKERNEL(k_add, A_OP2("v_add_f64"))
DEFKERNEL(k_max, A_OP2("v_max_f64"))
DEFKERNEL(k_rcp, A_OP1("v_rcp_f64"))
DEFKERNEL(k_rndne, A_OP1("v_rndne_f64"))
DEFKERNEL(k_ldexp, A_LDEXP)
DEFKERNEL(k_mov, A_MOV)

__global__ void __launch_bounds__(256) k_fma32(double *out, double c, int n)
{
    float f0 = 1.0f + threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    float fc = (float)c;
    for (int i = 0; i < ITER; i++) { A_FMA32 A_FMA32 }
    out[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
__global__ void __launch_bounds__(256) k_cvt(double *out, double c, int n)
{
    double x0 = 1.0 + threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0, i4 = 0, i5 = 0, i6 = 0, i7 = 0;
    for (int i = 0; i < ITER; i++) {
        asm volatile("v_cvt_i32_f64 %0, %8\n v_cvt_i32_f64 %1, %9\n v_cvt_i32_f64 %2, %10\n v_cvt_i32_f64 %3, %11\n"
                     "v_cvt_i32_f64 %4, %12\n v_cvt_i32_f64 %5, %13\n v_cvt_i32_f64 %6, %14\n v_cvt_i32_f64 %7, %15\n"
                     "v_cvt_i32_f64 %0, %8\n v_cvt_i32_f64 %1, %9\n v_cvt_i32_f64 %2, %10\n v_cvt_i32_f64 %3, %11\n"
                     "v_cvt_i32_f64 %4, %12\n v_cvt_i32_f64 %5, %13\n v_cvt_i32_f64 %6, %14\n v_cvt_i32_f64 %7, %15\n"
                     : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7)
                     : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}
// accuracy of v_rcp_f64 and of 1 / 2 Newton steps
__global__ void k_rcp_acc(const double *in, double *o0, double *o1, double *o2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = in[i];
    double r = __builtin_amdgcn_rcp(d);
    o0[i] = r;
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    o1[i] = r;
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    o2[i] = r;
}

typedef void (*kern_t)(double *, double, int);
static void run(const char *name, kern_t k, double *d_out, double clk_ghz, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double winstr_per_simd = (double)ITER * 16 * waves_per_simd;  // wave-instructions issued on each SIMD
    double cyc = ms * 1e-3 * clk_ghz * 1e9 / winstr_per_simd;
    printf("%-10s waves/SIMD=%d  %8.3f ms  %6.2f cycles/wave-instr/SIMD (at %.2f GHz)\n", name, waves_per_simd, ms, cyc, clk_ghz);
}

// Shader clock under load: idle waves (one per XCD) read s_memtime (shader clock) and s_memrealtime (constant rate) `ref_ticks` apart.
__global__ void __launch_bounds__(64) k_clock_probe(unsigned long long ref_ticks, unsigned long long *out)
{
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1;
    do { __builtin_amdgcn_s_sleep(64); r1 = wall_clock64(); } while (r1 - r0 < ref_ticks);
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

// The same instruction stream for ~0.6 s with the probe running beside it for 0.25 s: the clock the chip sustains under that
// load, and the issue cost in cycles of THAT clock.
static void sustained(const char *name, kern_t k, double *d_out, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd, wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    unsigned long long *d_p, h_p[16];
    hipMalloc(&d_p, sizeof h_p);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, sa, d_out, 1.0000001, 0);
    hipStreamSynchronize(sa);
    hipEventRecord(a, sa);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, sa, d_out, 1.0000001, 0);
    hipEventRecord(b, sa);
    hipEventSynchronize(b);
    float ms1;
    hipEventElapsedTime(&ms1, a, b);
    const int reps = (int)(600.0 / ms1) + 1;
    hipEventRecord(a, sa);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, sa, d_out, 1.0000001, 0);
    hipEventRecord(b, sa);
    hipLaunchKernelGGL(k_clock_probe, dim3(8), dim3(64), 0, sb, (unsigned long long)(0.25 * wall_khz * 1e3), d_p);
    hipMemcpyAsync(h_p, d_p, sizeof h_p, hipMemcpyDeviceToHost, sb);
    hipStreamSynchronize(sb);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double ratio = 0;
    for (int i = 0; i < 8; i++) ratio += (double)h_p[2 * i] / (double)h_p[2 * i + 1] / 8.0;
    const double ghz = ratio * wall_khz * 1e-6;
    const double winstr_per_simd = (double)ITER * 16 * waves_per_simd * reps;
    printf("sustained %-10s waves/SIMD=%d  %7.1f ms  shader clock %.3f GHz  %5.2f cycles/wave-instr/SIMD at that clock\n", name,
           waves_per_simd, ms, ghz, ms * 1e-3 * ghz * 1e9 / winstr_per_simd);
    hipFree(d_p);
    hipStreamDestroy(sa); hipStreamDestroy(sb);
}

int main()
{
    double *d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    double ghz = clk_khz * 1e-6;
    printf("reported max clock %.3f GHz\n", ghz);
    for (int w = 1; w <= 4; w *= 2) {
        run("fma_f32", k_fma32, d_out, ghz, w);
        run("fma_f64", k_fma, d_out, ghz, w);
        run("mul_f64", k_mul, d_out, ghz, w);
        run("add_f64", k_add, d_out, ghz, w);
        run("max_f64", k_max, d_out, ghz, w);
        run("rcp_f64", k_rcp, d_out, ghz, w);
        run("rndne_f64", k_rndne, d_out, ghz, w);
        run("ldexp_f64", k_ldexp, d_out, ghz, w);
        run("cvt_i32", k_cvt, d_out, ghz, w);
        run("mov_b64", k_mov, d_out, ghz, w);
    }
    for (int w = 2; w <= 4; w *= 2) {
        sustained("fma_f64", k_fma, d_out, w);
        sustained("mul_f64", k_mul, d_out, w);
        sustained("add_f64", k_add, d_out, w);
        sustained("rcp_f64", k_rcp, d_out, w);
        sustained("fma_f32", k_fma32, d_out, w);
    }
    // rcp accuracy
    const int n = 1 << 16;
    std::vector<double> h(n), r0(n), r1(n), r2(n);
    srand(1);
    for (int i = 0; i < n; i++) h[i] = 0.5 + 1.5 * (rand() / (double)RAND_MAX) * (i % 3 == 0 ? 1e10 : 1.0);
    double *di, *d0, *d1, *d2;
    hipMalloc(&di, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(di, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_rcp_acc, dim3(n / 256), dim3(256), 0, 0, di, d0, d1, d2, n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0;
    for (int i = 0; i < n; i++) {
        long double t = 1.0L / (long double)h[i];
        e0 = fmax(e0, fabs((double)((r0[i] - t) / t)));
        e1 = fmax(e1, fabs((double)((r1[i] - t) / t)));
        e2 = fmax(e2, fabs((double)((r2[i] - t) / t)));
    }
    printf("v_rcp_f64 max rel err: raw %.3e, +1 Newton %.3e, +2 Newton %.3e\n", e0, e1, e2);
    return 0;
}
