// tools/ubench_smem.hip -- can the wave-uniform operands of cf_main_tile3e's row loop travel on the SCALAR path (s_load_dwordx8/x16 from the
// record stream into SGPRs, fed to v_fma_f64 as its one scalar source) instead of global -> LDS -> ds_read_b128 -> VGPRs, without stalling the
// fp64 pipe?  Dev tool (LABBOOK round 5); not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_smem.hip -o /tmp/ubs && /tmp/ubs
// Every wave walks its own stream of 96-byte "rows" (12 doubles: the 3+1D row record {A_k, C_k, alpha_k, W_k, beta_jk x 8}); per row it runs
// NJ = 8 "evaluations" of FPE fp64 FMAs each (dependent chains over 8 accumulators, one FMA per evaluation takes the row's beta_j as an operand,
// two take A_k / alpha_k) -- the shape and the instruction count of the kernel's row (8 x ~14 + exponential).  Variants:
//   const   operands are kernel-lifetime constants (no memory traffic): the VALU bound of this loop
//   smem0   s_load the row, wait, compute (no prefetch): exposes the scalar-load latency once per row
//   smem1   s_load row r+1 before computing row r (two SGPR sets, 48 SGPRs): what a kernel variant would do
//   lds     the rows sit in LDS (filled once, no staging cost), read as wave-uniform ds_read_b128 into VGPRs: the cost of the LDS reads alone
// Streams: `reuse` waves of a CU walk the SAME stream (the kernel's G lane-wave groups share a record stream through one L2 / scalar cache);
// footprint per stream = rows x 96 B, total far above the 16-KiB scalar caches, L2-resident (streams x rows x 96 B <= 32 MB) or not.
// Reported: cycles per row per wave (s_memtime, summed over waves) and the ratio to `const`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(4))) double cdouble;   // constant address space: uniform reads become s_load

constexpr int NJ = 8, ROWD = 12;

template <int FPE>
__device__ __forceinline__ void row_work(double (&acc)[NJ], const double (&op)[ROWD], double x, double y)
{
    // per evaluation: FPE FMAs, dependent inside the evaluation, independent across the 8 evaluations
    const double mTA = x * op[0], mT2a = y * op[2];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        double t = __builtin_fma(x, op[4 + j], mT2a);        // fma(mTpTs, beta_j, ...): the scalar operand
        double u = __builtin_fma(t, y, mTA);
#pragma unroll
        for (int i = 0; i < FPE - 3; i++) u = __builtin_fma(u, 0.999 + 1e-3 * i, t);
        acc[j] = __builtin_fma(u, t, acc[j]);
    }
}

// MODE 0 const, 1 smem no prefetch, 2 smem prefetch, 3 lds
template <int MODE, int FPE>
__global__ void __launch_bounds__(512) k(const double *streams, int rows, int reuse, unsigned long long *out, double *sink)
{
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int stream = __builtin_amdgcn_readfirstlane(wave_g / reuse);
    cdouble *S = (cdouble *)(streams + (size_t)stream * rows * ROWD);
    double acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[j] = 0.0;
    const double x = 1.0 + 1e-3 * lane, y = 0.5 + 1e-4 * lane;
    if (MODE == 3) {
        // 128 rows of the stream in LDS (12 KB per workgroup: four two-wave workgroups per CU as in the kernel), filled once; the row index wraps
        for (int i = threadIdx.x; i < 128 * ROWD; i += blockDim.x) lds[i] = streams[(size_t)stream * rows * ROWD + i];
        __syncthreads();
    }
    const unsigned long long t0 = clock64();
    if (MODE == 0) {
        double op[ROWD];
#pragma unroll
        for (int i = 0; i < ROWD; i++) op[i] = S[i];
        for (int r = 0; r < rows; r++) {
#pragma unroll
            for (int i = 0; i < ROWD; i++) asm volatile("" : "+s"(op[i]));   // opaque per row: the loop-invariant work must not be hoisted
            row_work<FPE>(acc, op, x, y);
        }
    } else if (MODE == 1) {
        for (int r = 0; r < rows; r++) {
            double op[ROWD];
#pragma unroll
            for (int i = 0; i < ROWD; i++) op[i] = S[(size_t)r * ROWD + i];
            row_work<FPE>(acc, op, x, y);
        }
    } else if (MODE == 2) {
        double cur[ROWD], nxt[ROWD];
#pragma unroll
        for (int i = 0; i < ROWD; i++) cur[i] = S[i];
        for (int r = 0; r < rows; r++) {
            const size_t rn = (size_t)(r + 1 < rows ? r + 1 : r) * ROWD;
#pragma unroll
            for (int i = 0; i < ROWD; i++) nxt[i] = S[rn + i];
            __builtin_amdgcn_sched_barrier(0);   // the loads issue here, at the head of the row, not where the scheduler finds free SGPRs
            row_work<FPE>(acc, cur, x, y);
#pragma unroll
            for (int i = 0; i < ROWD; i++) cur[i] = nxt[i];
        }
    } else {
        for (int r = 0; r < rows; r++) {
            const double2 *row = (const double2 *)(lds + (size_t)(r & 127) * ROWD);
            double op[ROWD];
#pragma unroll
            for (int i = 0; i < ROWD / 2; i++) { const double2 v = row[i]; op[2 * i] = v.x; op[2 * i + 1] = v.y; }
            row_work<FPE>(acc, op, x, y);
        }
    }
    const unsigned long long t1 = clock64();
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; j++) s += acc[j];
    if (s == 12345.678 && sink) sink[0] = s;
    if (lane == 0) {
        atomicAdd(&out[0], t1 - t0);
        atomicAdd(&out[1], (unsigned long long)rows);
    }
}

template <int MODE, int FPE>
static double run(const char *label, const double *streams, unsigned long long *d_out, double *sink, int wpb, int grid, int rows, int reuse, double ref = 0.0)
{
    unsigned long long h[2] = {0, 0};
    CK(hipMemset(d_out, 0, sizeof h));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t shm = MODE == 3 ? 128 * ROWD * sizeof(double) : 0;
    hipLaunchKernelGGL((k<MODE, FPE>), dim3(grid), dim3(wpb * 64), shm, 0, streams, rows, reuse, d_out, sink);   // warm (code, L2)
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_out, 0, sizeof h));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, FPE>), dim3(grid), dim3(wpb * 64), shm, 0, streams, rows, reuse, d_out, sink);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
    const double cpr = (double)h[0] / (double)h[1];
    printf("%-8s FMAs/eval=%2d waves/wg=%d grid=%5d rows/wave=%6d waves/stream=%d : %8.1f cycles per row per wave  %7.3f ms%s", label, FPE, wpb, grid, rows, reuse, cpr, ms,
           ref > 0 ? "" : "\n");
    if (ref > 0) printf("   x %.3f of const\n", cpr / ref);
    return cpr;
}

template <int FPE>
static void suite(const double *streams, unsigned long long *d_out, double *sink, int wpb, int grid, int rows, int reuse)
{
    const double c = run<0, FPE>("const", streams, d_out, sink, wpb, grid, rows, reuse);
    run<1, FPE>("smem0", streams, d_out, sink, wpb, grid, rows, reuse, c);
    run<2, FPE>("smem1", streams, d_out, sink, wpb, grid, rows, reuse, c);
    run<3, FPE>("lds", streams, d_out, sink, wpb, grid, rows, reuse, c);
}

int main()
{
    // 2048 waves (256 CUs x 4 SIMDs x 2 waves) each with up to 16384 rows of 96 B: 3 GB if every wave has its own stream
    const int waves = 2048;
    const int rows_max = 16384;
    double *streams;
    unsigned long long *d_out;
    double *sink;
    const size_t n = (size_t)waves * rows_max * ROWD;
    CK(hipMalloc(&streams, n * sizeof(double)));
    std::vector<double> h(1 << 20);
    for (size_t i = 0; i < h.size(); i++) h[i] = 0.5 + 1e-6 * (double)(i % 977);
    for (size_t off = 0; off < n; off += h.size()) CK(hipMemcpy(streams + off, h.data(), std::min(h.size(), n - off) * sizeof(double), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, 64));
    CK(hipMalloc(&sink, 64));
    printf("# two waves per SIMD (grid 1024 x 2 waves = one resident round), every wave its own stream (HBM-resident: 3 GB), then 8 waves per stream\n");
    suite<14>(streams, d_out, sink, 2, 1024, rows_max, 1);
    suite<14>(streams, d_out, sink, 2, 1024, rows_max, 8);
    printf("# L2-resident streams: 2048 rows per wave (192 KB per stream), 8 waves per stream\n");
    suite<14>(streams, d_out, sink, 2, 1024, 2048, 8);
    printf("# a lighter row (6 FMAs per evaluation): less cover per row\n");
    suite<6>(streams, d_out, sink, 2, 1024, rows_max, 8);
    printf("# one wave per SIMD\n");
    suite<14>(streams, d_out, sink, 1, 1024, rows_max, 8);
    return 0;
}
