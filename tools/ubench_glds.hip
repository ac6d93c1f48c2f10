// tools/ubench_glds.hip -- what does a wave pay to ISSUE direct-to-LDS loads (global_load_lds_dwordx4, 1 KiB per
// wave-instruction) on gfx950, and how does that compare with register loads (global_load_dwordx4) of the same bytes?
// Dev tool (staging of cf_main_tile3e); not part of the library.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_glds.hip -o /tmp/ubg && /tmp/ubg
// Each wave issues NL loads back to back from an L2-resident 1 MiB buffer, reads s_memtime before the first, after the last
// issue and after s_waitcnt vmcnt(0).  Reported per load: issue cycles (wave blocked in issue) and completion cycles.
// `busy` waves per SIMD run an independent fp64 FMA stream beside the loader wave (the situation in the kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const void *g, void *l)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// mode 0: direct-to-LDS; mode 1: register loads + ds_write afterwards; loaders = waves of the workgroup that load (the others
// run FMAs); NL loads per loader wave per round, `rounds` rounds separated by a vmcnt(0)
template <int MODE, int NL>
__global__ void __launch_bounds__(512) k(const char *src, unsigned long long *out, int loaders, int rounds, double *sink, unsigned nblk)
{
    static_assert(NL <= 18, "the source buffer holds 2^18 blocks of at most 18 KiB");
    extern __shared__ char lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < loaders) {
        unsigned long long t_issue = 0, t_done = 0;
        char *dst = lds + wave * NL * 1024;
        for (int r = 0; r < rounds; r++) {
            // a fresh NL-KiB block every round: nblk blocks of the buffer (64: L1/L2 hits; 2^18: HBM)
            const char *s = src + (size_t)(((unsigned)(blockIdx.x * 8 + wave) * 2654435761u + (unsigned)r * 40503u) % nblk) * (NL * 1024) + lane * 16;
            const unsigned long long a = clock64();
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < NL; i++) glds16(s + i * 1024, dst + i * 1024);
                const unsigned long long b = clock64();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long c = clock64();
                t_issue += b - a; t_done += c - a;
            } else {
                double2 v[NL];
#pragma unroll
                for (int i = 0; i < NL; i++) v[i] = *(const double2 *)(s + i * 1024);
                const unsigned long long b = clock64();
#pragma unroll
                for (int i = 0; i < NL; i++) *(double2 *)(dst + i * 1024 + lane * 16) = v[i];
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long c = clock64();
                t_issue += b - a; t_done += c - a;
            }
        }
        if (lane == 0) {
            atomicAdd(&out[0], t_issue);
            atomicAdd(&out[1], t_done);
            atomicAdd(&out[2], (unsigned long long)rounds * NL);
        }
        if (lds[threadIdx.x] == 77 && sink) sink[0] = 1.0;
    } else {
        double a0 = 1.0 + lane, a1 = 2.0, a2 = 3.0, a3 = 4.0, a4 = 5.0, a5 = 6.0, a6 = 7.0, a7 = 8.0;
        const double m = 1.0000001, c = 1e-9;
        for (int r = 0; r < rounds * 40; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
                a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
            }
        }
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.0 && sink) sink[1] = a0;
    }
}

template <int MODE, int NL>
static void run(const char *label, const char *src, unsigned long long *d_out, double *sink, int wpb, int loaders, int grid, unsigned nblk = 64)
{
    unsigned long long h[3] = {0, 0, 0};
    CK(hipMemset(d_out, 0, sizeof h));
    hipLaunchKernelGGL((k<MODE, NL>), dim3(grid), dim3(wpb * 64), 8 * NL * 1024, 0, src, d_out, loaders, 200, sink, nblk);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost));
    printf("%-34s blocks=%6u NL=%2d waves/wg=%d loaders=%d grid=%4d : issue %7.1f cycles/load   issue+complete %7.1f cycles/load\n", label, nblk, NL, wpb, loaders,
           grid, (double)h[0] / h[2], (double)h[1] / h[2]);
}

int main()
{
    char *src;
    unsigned long long *d_out;
    double *sink;
    const size_t big = (size_t)18 << 28;   // 2^18 blocks of up to 18 KiB (NL <= 18)
    CK(hipMalloc(&src, big));
    CK(hipMemset(src, 0, big));
    CK(hipMalloc(&d_out, 64));
    CK(hipMalloc(&sink, 64));
    // one loader wave alone on a CU
    run<0, 9>("glds, 1 wave per CU", src, d_out, sink, 1, 1, 256);
    run<1, 9>("regs, 1 wave per CU", src, d_out, sink, 1, 1, 256);
    run<0, 1>("glds, 1 wave per CU", src, d_out, sink, 1, 1, 256);
    run<0, 3>("glds, 1 wave per CU", src, d_out, sink, 1, 1, 256);
    // 8 loader waves per CU (every wave loads)
    run<0, 9>("glds, 8 loader waves per CU", src, d_out, sink, 8, 8, 256);
    run<1, 9>("regs, 8 loader waves per CU", src, d_out, sink, 8, 8, 256);
    // 1 loader wave + 7 FMA waves per CU (2 waves per SIMD: the loader shares its SIMD with an FMA wave)
    run<0, 9>("glds, 1 loader + 7 FMA waves", src, d_out, sink, 8, 1, 256);
    run<1, 9>("regs, 1 loader + 7 FMA waves", src, d_out, sink, 8, 1, 256);
    run<0, 9>("glds, 2 loaders + 6 FMA waves", src, d_out, sink, 8, 2, 256);
    run<0, 9>("glds, 4 loaders + 4 FMA waves", src, d_out, sink, 8, 4, 256);
    for (unsigned nb : {1024u, 1u << 18}) {
        run<0, 9>("glds, 1 wave per CU", src, d_out, sink, 1, 1, 256, nb);
        run<0, 9>("glds, 8 loader waves per CU", src, d_out, sink, 8, 8, 256, nb);
        run<0, 9>("glds, 1 loader + 7 FMA waves", src, d_out, sink, 8, 1, 256, nb);
        run<0, 9>("glds, 2 loaders + 6 FMA waves", src, d_out, sink, 8, 2, 256, nb);
        run<0, 3>("glds, 2 loaders + 6 FMA waves", src, d_out, sink, 8, 2, 256, nb);
        run<0, 18>("glds, 2 loaders + 6 FMA waves", src, d_out, sink, 8, 2, 256, nb);
        run<1, 9>("regs, 2 loaders + 6 FMA waves", src, d_out, sink, 8, 2, 256, nb);
    }
    return 0;
}
