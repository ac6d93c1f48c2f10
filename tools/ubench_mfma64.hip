// ubench_mfma64.hip -- does the fp64 matrix pipe of gfx950 run BESIDE the fp64 VALU?  
//
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench_mfma64.hip -o /tmp/ubm && /tmp/ubm
//
// cf_main_tile is bound by fp64 VALU issue.  Its per-evaluation multiply-adds are K = 2-3 bilinear
// forms (br = mT^2 alpha_k + mT pT beta_jk + pT^2 gamma_j etc.) that v_mfma_f64_16x16x4_f64 could form on the matrix
// pipe -- worth it only if an fp64 MFMA leaves the VALU issue of its SIMD (nearly) alone.  This measures exactly that:
// every SIMD of the chip runs W waves (W = 2: the occupancy of cf_main_tile) whose instruction stream is NF independent
// v_fma_f64 per loop trip with NM v_mfma_f64_16x16x4_f64 spread evenly between them, for ~0.5 s, with the shader clock
// probed beside it (idle waves, s_memtime against s_memrealtime).  Reported per configuration:
//   cycles per loop trip per SIMD, cycles per VALU instruction (all cycles charged to the FMAs), and
//   "MFMA cost" = (cycles of the trip - cycles of the same trip without MFMAs) / NM = what one MFMA takes away from the VALU.
// NF = 0 gives the back-to-back MFMA issue rate (the matrix pipe's own ceiling).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITER 1000
typedef double double4_t __attribute__((ext_vector_type(4)));

#define A_FMA8 asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n" \
                            "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8\n" \
                            : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));

// NG groups of 8 FMAs per trip, NM MFMAs per trip (NM divides NG, or NG == 0).  ACC: 0 = accumulators in VGPRs, 1 = in AGPRs.
// DEP: 1 = every MFMA accumulates into the same registers (dependent chain), 0 = four independent accumulators in rotation.
template <int NG, int NM, int ACC, int DEP>
__global__ void __launch_bounds__(256) k_mix(double *out, double c, double a, double b)
{
    double x0 = 1.0 + threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double4_t m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    a += threadIdx.x * 1e-6;
    b -= threadIdx.x * 1e-6;
    for (int i = 0; i < ITER; i++) {
        constexpr int SLOTS = NM > 0 ? NM : 1;
        constexpr int GPS = NG / SLOTS;   // FMA groups per slot
#pragma unroll
        for (int s = 0; s < SLOTS; s++) {
            if constexpr (NM > 0) {
                const int w = DEP ? 0 : (s & 3);
                if constexpr (ACC == 0) {
                    if (w == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m0) : "v"(a), "v"(b));
                    else if (w == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m1) : "v"(a), "v"(b));
                    else if (w == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m2) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(m3) : "v"(a), "v"(b));
                } else {
                    if (w == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(m0) : "v"(a), "v"(b));
                    else if (w == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(m1) : "v"(a), "v"(b));
                    else if (w == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(m2) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(m3) : "v"(a), "v"(b));
                }
            }
#pragma unroll
            for (int g = 0; g < GPS; g++) { A_FMA8 }
        }
    }
    const double4_t ms = m0 + m1 + m2 + m3;
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + ms.x + ms.y + ms.z + ms.w;
}

// numerical check of the operand layout (cdna_hip_programming.md section 3): D = A (16x4) B (4x16), lane l holds A[l & 15][l >> 4],
// B[l >> 4][l & 15] and D[(l >> 4) + 4 i][l & 15], i = 0..3
__global__ void __launch_bounds__(64) k_layout(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    double4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[((l >> 4) + 4 * i) * 16 + (l & 15)] = acc[i];
}

__global__ void __launch_bounds__(64) k_clock_probe(unsigned long long ref_ticks, unsigned long long *out)
{
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1;
    do { __builtin_amdgcn_s_sleep(64); r1 = wall_clock64(); } while (r1 - r0 < ref_ticks);
    const unsigned long long c1 = clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
}

typedef void (*kern_t)(double *, double, double, double);

struct Result { double cyc_trip, ghz, ms; };

static Result sustained(kern_t k, double *d_out, int waves_per_simd)
{
    int blocks = 256 * waves_per_simd, wall_khz = 0;   // 256 threads = 4 waves = one per SIMD of a CU; 256 CUs
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    unsigned long long *d_p, h_p[16];
    hipMalloc(&d_p, sizeof h_p);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, sa, d_out, 1.0000001, 0.5, 0.25);
    hipStreamSynchronize(sa);
    hipEventRecord(a, sa);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, sa, d_out, 1.0000001, 0.5, 0.25);
    hipEventRecord(b, sa);
    hipEventSynchronize(b);
    float ms1;
    hipEventElapsedTime(&ms1, a, b);
    const int reps = (int)(500.0 / ms1) + 1;
    hipEventRecord(a, sa);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, sa, d_out, 1.0000001, 0.5, 0.25);
    hipEventRecord(b, sa);
    hipLaunchKernelGGL(k_clock_probe, dim3(8), dim3(64), 0, sb, (unsigned long long)(0.2 * wall_khz * 1e3), d_p);
    hipMemcpyAsync(h_p, d_p, sizeof h_p, hipMemcpyDeviceToHost, sb);
    hipStreamSynchronize(sb);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double ratio = 0;
    for (int i = 0; i < 8; i++) ratio += (double)h_p[2 * i] / (double)h_p[2 * i + 1] / 8.0;
    const double ghz = ratio * wall_khz * 1e-6;
    Result r;
    r.ghz = ghz;
    r.ms = ms;
    r.cyc_trip = ms * 1e-3 * ghz * 1e9 / ((double)ITER * reps * waves_per_simd);   // SIMD cycles per loop trip of ONE wave's stream
    hipFree(d_p);
    hipStreamDestroy(sa); hipStreamDestroy(sb);
    hipEventDestroy(a); hipEventDestroy(b);
    return r;
}

template <int NG, int NM, int ACC, int DEP>
static double report(double *d_out, int W, double base_trip)
{
    Result r = sustained(k_mix<NG, NM, ACC, DEP>, d_out, W);
    printf("waves/SIMD=%d  FMA/trip=%3d  MFMA/trip=%d  acc=%s %s  %7.1f ms  clock %.3f GHz  %8.2f cycles/trip", W, NG * 8, NM,
           ACC ? "AGPR" : "VGPR", DEP ? "dependent  " : "independent", r.ms, r.ghz, r.cyc_trip);
    if (NG > 0) printf("  %5.2f cycles/FMA", r.cyc_trip / (NG * 8));
    if (NM > 0 && NG == 0) printf("  %6.2f cycles/MFMA (back-to-back issue)", r.cyc_trip / NM);
    if (NM > 0 && base_trip > 0) printf("  MFMA cost to the VALU stream: %6.2f cycles each", (r.cyc_trip - base_trip) / NM);
    printf("\n");
    fflush(stdout);
    return r.cyc_trip;
}

int main()
{
    double *d_out;
    hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256);
    // ---- layout check ----
    {
        std::vector<double> A(64), B(64), D(256), R(256, 0.0);
        srand(7);
        for (auto &v : A) v = rand() / (double)RAND_MAX - 0.5;
        for (auto &v : B) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++)
                for (int k = 0; k < 4; k++) R[i * 16 + j] = __builtin_fma(A[i * 4 + k], B[k * 16 + j], R[i * 16 + j]);
        double *dA, *dB, *dD;
        hipMalloc(&dA, 64 * 8); hipMalloc(&dB, 64 * 8); hipMalloc(&dD, 256 * 8);
        hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        double e = 0;
        int nbit = 0;
        for (int i = 0; i < 256; i++) { e = fmax(e, fabs(D[i] - R[i])); nbit += D[i] == R[i]; }
        printf("v_mfma_f64_16x16x4_f64 layout check: max |D - fma chain over k ascending| = %.3e, %d of 256 bitwise equal\n", e, nbit);
    }
    for (int W = 1; W <= 4; W *= 2) {
        // FMA-only baselines at the trip lengths used below
        const double b16 = report<2, 0, 0, 0>(d_out, W, 0);
        const double b32 = report<4, 0, 0, 0>(d_out, W, 0);
        const double b64 = report<8, 0, 0, 0>(d_out, W, 0);
        const double b128 = report<16, 0, 0, 0>(d_out, W, 0);
        // matrix pipe alone
        report<0, 4, 0, 0>(d_out, W, 0);
        report<0, 4, 0, 1>(d_out, W, 0);
        report<0, 4, 1, 0>(d_out, W, 0);
        // one MFMA per 16 / 32 / 64 / 128 FMAs
        report<2, 1, 0, 0>(d_out, W, b16);
        report<4, 1, 0, 0>(d_out, W, b32);
        report<8, 1, 0, 0>(d_out, W, b64);
        report<16, 1, 0, 0>(d_out, W, b128);
        report<16, 2, 0, 0>(d_out, W, b128);
        report<16, 4, 0, 0>(d_out, W, b128);
        report<16, 4, 0, 1>(d_out, W, b128);
        report<16, 4, 1, 0>(d_out, W, b128);
    }
    return 0;
}
