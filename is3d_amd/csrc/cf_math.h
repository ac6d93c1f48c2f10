// cf_math.h -- fp64 device building blocks shared by the HIP translation units (gfx950 has no v_exp_f64 / exact
// v_rcp_f64 / v_sqrt_f64: these are the hand-rolled versions the hot loops use).
#pragma once
#include <hip/hip_runtime.h>

#include "cf_device.h"

namespace is3d {

#define IS3D_LOG2E 1.44269504088896338700e+00
#define IS3D_LN2_HI 6.93147180369123816490e-01 /* 0x3fe62e42fee00000: n*LN2_HI exact for |n| < 2^21 */
#define IS3D_LN2_LO 1.90821492927058770002e-10 /* 0x3dea39ef35793c76 */

// 16 bytes per lane global -> LDS without a register round trip (global_load_lds_dwordx4): the wave writes 1 KiB contiguously at
// the wave-uniform LDS byte address l32 (M0; lane i lands at l32 + 16 i), each lane reads its own global address g; OFF (< 4096)
// advances both addresses.  Asynchronous: it is retired by s_waitcnt vmcnt.
// The instruction is written out so that the compiler does not know LDS is being written behind its back: with a
// __builtin_amdgcn_global_load_lds in flight its waitcnt pass answers every LDS read of the kernel with s_waitcnt lgkmcnt(0) (no
// counted waits at all in cf_main_tile3e; 692 -> 678 ms with culling off); the caller orders reads against these loads itself
// (s_waitcnt vmcnt + barrier), as it has to anyway.
template <int OFF>
__device__ __forceinline__ void glds16a(const void *g, unsigned l32)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" : : "v"(g), "s"(l32), "n"(OFF) : "memory");
    // M0 is a RESERVED register to the compiler: it is never allocated, a value is materialised into it immediately in front of each
    // compiler-generated user (movrel, sendmsg, LDS-direct), and naming it in the clobber list is refused ("clobber list contains reserved
    // registers ... not modelled").  tools/count_isa.py records every instruction that reads M0 in the kernels that use this helper and
    // tests/test_isa_counts.py asserts there is none besides global_load_lds itself.
}
__device__ __forceinline__ unsigned lds_addr32(const void *p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)p;
}

// The waves of a workgroup copy NP whole 1-KiB pieces global -> LDS (contiguous on both sides): a contiguous range of pieces per
// wave, four pieces per address through the immediate offset, nothing predicated -- the caller pads the LDS buffer to NP KiB and
// leaves slack behind the global stream for the over-read of the last piece.  Retired by s_waitcnt vmcnt(0) + barrier.
template <int NP>
__device__ __forceinline__ void stage_pieces(const char *gsrc, const void *lds_dst, int tid, int nthr)
{
    const int nw = nthr >> 6, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = (NP * wave) / nw, hi = (NP * (wave + 1)) / nw;
    const unsigned lane16 = (unsigned)(tid & 63) * 16u;
    for (int p = lo; p < hi; p += 4) {
        const char *gp = gsrc + p * 1024 + lane16;
        const unsigned lp = __builtin_amdgcn_readfirstlane(lds_addr32((const char *)lds_dst + p * 1024));
        const int n = hi - p;
        glds16a<0>(gp, lp);
        if (n > 1) glds16a<1024>(gp, lp);
        if (n > 2) glds16a<2048>(gp, lp);
        if (n > 3) glds16a<3072>(gp, lp);
    }
}

// e^v = f * 2^n with f in [0.70, 1.42].  Cody-Waite reduction + degree-10 near-minimax polynomial of e^r, |r| <= ln2/2.  Splitting mantissa and exponent lets two exponentials be
// multiplied without overflow: e^(a-b) = (f_a f_b) 2^(n_a+n_b)  (used by the factorised kernel).
// e^r on |r| <= ln2/2 as a degree-10 polynomial (Chebyshev-node interpolant computed in long double, rounded to
// double: max relative error 1.4e-15 over +-1.02 ln2/2); c10 .. c1, c0 = 1.
// Weak, externally visible on purpose: with internal linkage the optimizer folds the table back into literals.
__attribute__((weak)) __constant__ double kExpC[10] = {2.76263718333300477554e-07, 2.76401815125786182983e-06, 2.48015043178771474428e-05,
    1.98411702685802050468e-04, 1.38888889325031097272e-03, 8.33333338566940792780e-03,
    4.16666666665731558195e-02, 1.66666666665543999892e-01, 5.00000000000000555112e-01, 1.00000000000000666134e+00};

// Domain: |v| < 1.4e9 (the shift trick below takes n from the low dword; cf_prep refuses cells that could exceed it).
// exp_core_sat is the same with a saturating conversion, for callers without such a guarantee.
__device__ __forceinline__ void exp_core(double v, double &f, int &n)
{
    // n = rint(v log2 e) by the 1.5 * 2^52 shift: the sum's low mantissa bits ARE the integer (two's complement, |n| < 2^31),
    // so the int comes out of the low dword for free and no v_rndne_f64 / v_cvt_i32_f64 is issued
    const double shift = 6755399441055744.0;   // 0x1.8p52
    double t = __builtin_fma(v, IS3D_LOG2E, shift);
    double dn = t - shift;
    double r = __builtin_fma(-dn, IS3D_LN2_HI, v);
    r = __builtin_fma(-dn, IS3D_LN2_LO, r);
    // The non-inline Taylor coefficients come from constant memory, i.e. they sit in SGPRs: the Horner steps are
    // then 3-address v_fma_f64 with a scalar addend.  (With literal constants hipcc keeps them in VGPRs and emits
    // v_mov_b64 + v_fmac_f64 per step: 9 extra moves per exponential.)
    double p = kExpC[0];
    p = __builtin_fma(p, r, kExpC[1]);
    p = __builtin_fma(p, r, kExpC[2]);
    p = __builtin_fma(p, r, kExpC[3]);
    p = __builtin_fma(p, r, kExpC[4]);
    p = __builtin_fma(p, r, kExpC[5]);
    p = __builtin_fma(p, r, kExpC[6]);
    p = __builtin_fma(p, r, kExpC[7]);
    p = __builtin_fma(p, r, kExpC[8]);
    p = __builtin_fma(p, r, kExpC[9]);
    f = __builtin_fma(p, r, 1.0);
    n = __double2loint(t);
}

__device__ __forceinline__ double ldexp_fast(double f, int n) { return __builtin_amdgcn_ldexp(f, n); }  // v_ldexp_f64

__device__ __forceinline__ double exp_full_sat(double v)
{
    double dn = __builtin_rint(v * IS3D_LOG2E);
    double r = __builtin_fma(-dn, IS3D_LN2_HI, v);
    r = __builtin_fma(-dn, IS3D_LN2_LO, r);
    double p = kExpC[0];
    p = __builtin_fma(p, r, kExpC[1]);
    p = __builtin_fma(p, r, kExpC[2]);
    p = __builtin_fma(p, r, kExpC[3]);
    p = __builtin_fma(p, r, kExpC[4]);
    p = __builtin_fma(p, r, kExpC[5]);
    p = __builtin_fma(p, r, kExpC[6]);
    p = __builtin_fma(p, r, kExpC[7]);
    p = __builtin_fma(p, r, kExpC[8]);
    p = __builtin_fma(p, r, kExpC[9]);
    return ldexp_fast(__builtin_fma(p, r, 1.0), (int)dn);   // v_cvt_i32_f64 saturates: e^-huge = +0, never a wrapped exponent (|v| up to ~1e45; tests/test_gpu_math.py)
}

// e^v; exactly +0 for v < -745.2 (v_ldexp_f64 underflow)
__device__ __forceinline__ double exp_full(double v)
{
    double f; int n;
    exp_core(v, f, n);
    return ldexp_fast(f, n);
}

// e^v for the kernels that pay one exponential PER EVALUATION (modified equilibrium, anisotropic hydro: the square root in the
// exponent does not factorise).  Two instructions shorter than exp_full: the reduction subtracts n ln2 in ONE fma with the double
// nearest ln2 (its 2.3e-17 error times |n| <= 1075 moves the reduced argument by <= 2.5e-14), and e^r is a degree-9 polynomial
// (Chebyshev-node interpolant of (e^r - 1)/r on |r| <= 1.02 ln2/2 computed in long double: max relative error 4.5e-14).  Worst
// case 7e-14 relative -- five orders of magnitude inside the 2e-9 the parity tests assert, seven inside north_star's 1e-6 -- and the
// argument itself (a one-step square root) already carries 3e-15 |v|.  Same domain as exp_core: |v| < 1.4e9.
__attribute__((weak)) __constant__ double kExpD[9] = {2.76278435747019428533e-06, 2.48791812002179739231e-05, 1.98412037179905744030e-04,
    1.38888161319893847084e-03, 8.33333335629010685253e-03, 4.16666669192741895289e-02, 1.66666666666451496193e-01,
    4.99999999997632282867e-01, 1.00000000000000000000e+00};
#define IS3D_LN2 0.693147180559945286
__device__ __forceinline__ double exp_p9_poly(double r)
{
    double p = kExpD[0];
    p = __builtin_fma(p, r, kExpD[1]);
    p = __builtin_fma(p, r, kExpD[2]);
    p = __builtin_fma(p, r, kExpD[3]);
    p = __builtin_fma(p, r, kExpD[4]);
    p = __builtin_fma(p, r, kExpD[5]);
    p = __builtin_fma(p, r, kExpD[6]);
    p = __builtin_fma(p, r, kExpD[7]);
    p = __builtin_fma(p, r, kExpD[8]);
    return __builtin_fma(p, r, 1.0);
}
__device__ __forceinline__ double exp_p9(double v)   // |v| < 1.4e9 (shift trick, see exp_core)
{
    const double shift = 6755399441055744.0;   // 0x1.8p52
    const double t = __builtin_fma(v, IS3D_LOG2E, shift);
    const double dn = t - shift;
    const double r = __builtin_fma(-dn, IS3D_LN2, v);
    return ldexp_fast(exp_p9_poly(r), __double2loint(t));
}
// exp_p9(v) * 2^e with shiftc = kExpShift + e (e a small integer): the shift constant places n + e in the low word, the reduced argument is
// v - n ln2 as before -- a power-of-two factor for free (cf_main_feqmod: the cell's p.dsigma scale comes back here)
constexpr double kExpShift = 6755399441055744.0;   // 0x1.8p52
__device__ __forceinline__ double exp_p9_scaled(double v, double shiftc)
{
    const double t = __builtin_fma(v, IS3D_LOG2E, shiftc);
    const double dn = t - shiftc;
    const double r = __builtin_fma(-dn, IS3D_LN2, v);
    return ldexp_fast(exp_p9_poly(r), __double2loint(t));
}
__device__ __forceinline__ double exp_p9_sat(double v)   // |v| up to ~1e45: v_cvt_i32_f64 saturates, e^-huge = +0 (beyond that the reduced argument's polynomial overflows)
{
    const double dn = __builtin_rint(v * IS3D_LOG2E);
    const double r = __builtin_fma(-dn, IS3D_LN2, v);
    return ldexp_fast(exp_p9_poly(r), (int)dn);
}

// 1/d: v_rcp_f64 seed + two Newton steps (each squares the relative error)
__device__ __forceinline__ double rcp_nr(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    return r;
}

__device__ __forceinline__ double rcp_nr1(double d)
{
    double r = __builtin_amdgcn_rcp(d);      // 4.5e-8 relative (measured, tools/ubench_fp64.hip)
    double e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);           // 2e-15
}

// 1/q[0..RB): one v_rcp_f64 for the whole batch (17 issue cycles against 5 for an FMA, tools/ubench_fp64.hip):
// 1/q_i = (1 / prod q) * prod_{j != i} q_j with 3 (RB - 1) multiplications.  RB = 4, 8: pairwise product tree; other sizes:
// prefix products.  The caller guarantees that prod q neither overflows nor underflows; every quotient carries
// ~log2(RB) + 2 roundings on top of the 2e-15 of rcp_nr1.  A NaN / 0 / inf in one q poisons the whole batch.
template <int RB>
__device__ __forceinline__ void rcp_batch(const double (&q)[RB], double (&inv)[RB])
{
    if constexpr (RB == 1) inv[0] = rcp_nr1(q[0]);
    else if constexpr (RB == 2) {
        const double rp = rcp_nr1(q[0] * q[1]);
        inv[0] = rp * q[1];
        inv[1] = rp * q[0];
    } else if constexpr (RB == 4 || RB == 8) {
        double ph[RB / 2], ih[RB / 2];
#pragma unroll
        for (int i = 0; i < RB / 2; i++) ph[i] = q[2 * i] * q[2 * i + 1];
        if constexpr (RB == 4) {
            const double rp = rcp_nr1(ph[0] * ph[1]);
            ih[0] = rp * ph[1];
            ih[1] = rp * ph[0];
        } else {
            const double pa = ph[0] * ph[1], pb = ph[2] * ph[3];
            const double rp = rcp_nr1(pa * pb);
            const double ia = rp * pb, ib = rp * pa;
            ih[0] = ia * ph[1];
            ih[1] = ia * ph[0];
            ih[2] = ib * ph[3];
            ih[3] = ib * ph[2];
        }
#pragma unroll
        for (int i = 0; i < RB / 2; i++) {
            inv[2 * i] = ih[i] * q[2 * i + 1];
            inv[2 * i + 1] = ih[i] * q[2 * i];
        }
    } else {
        double pf[RB];
        pf[0] = q[0];
#pragma unroll
        for (int i = 1; i < RB; i++) pf[i] = pf[i - 1] * q[i];
        double rp = rcp_nr1(pf[RB - 1]);
#pragma unroll
        for (int i = RB - 1; i > 0; i--) {
            inv[i] = rp * pf[i - 1];
            rp = rp * q[i];
        }
        inv[0] = rp;
    }
}

// fma with the VOP3 clamp modifier: result clamped to [0, 1] (NaN -> 0 under DX10_CLAMP)
__device__ __forceinline__ double fma_clamp01_half(double a, double b)
{
    double u;
    asm("v_fma_f64 %0, %1, %2, 0.5 clamp" : "=v"(u) : "v"(a), "v"(b));
    return u;
}

// a + b and a * b + c with the clamp modifier: result clamped to [0, 1]
__device__ __forceinline__ double add_clamp01(double a, double b)
{
    double u;
    asm("v_add_f64 %0, %1, %2 clamp" : "=v"(u) : "v"(a), "v"(b));
    return u;
}
__device__ __forceinline__ double fma_clamp01(double a, double b, double c)
{
    double u;
    asm("v_fma_f64 %0, %1, %2, %3 clamp" : "=v"(u) : "v"(a), "v"(b), "v"(c));
    return u;
}

// sqrt(x) for normal x > 0: v_rsq_f64 seed, one coupled Newton (Goldschmidt) step for sqrt and 1/(2 sqrt), then one
// residual correction: relative error <= ~1.5e-16 for any seed better than 1e-5.  No range scaling: callers pass
// x in [1e-280, 1e280].
__device__ __forceinline__ double sqrt_nr(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// sqrt(x) with one coupled Newton (Goldschmidt) step only: relative error 1.5 e0^2 for a seed of relative error e0, i.e.
// ~3e-15 with the 4.5e-8 v_rsq_f64 seed of gfx950.  For arguments of exponentials whose result is needed to ~1e-12.
__device__ __forceinline__ double sqrt_g1(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}

// Deltaf_Data::bilinear_interpolation (deltafReader.cpp:412-484) on copies of the full (mu_B, T) grids (device pointers in the
// kernels, host pointers in the host-side evaluations at the surface averages), with the intended [imuB][iT] indexing.
// The reference's calculate_bilinear swaps the indices: it reads f_data[iT][imuB] (:404-407) from arrays allocated
// [points_muB][points_T] (:36-61), i.e. the table value at (mu_B row iT, T column imuB).  b.swap (opts.reference_bilinear_indexing)
// reproduces that read wherever it stays inside the allocation (iTR < points_muB; the column imuBR < points_T always holds) and
// reports the cell as outside the table where the reference reads past its row pointers (undefined behaviour there).
// Returns false outside the table (reference: printf + exit(-1), :423-427).
__host__ __device__ __forceinline__ bool bilinear5(const BilinearDev &b, double T, double muB, double (&v)[5])
{
    const double T_min = b.T[0], B_min = b.muB[0];
    const double dT = fabs(b.T[1] - b.T[0]), dB = fabs(b.muB[1] - b.muB[0]);
    const int iTL = (int)floor((T - T_min) / dT), iTR = iTL + 1;
    const int iBL = (int)floor((muB - B_min) / dB), iBR = iBL + 1;
    if (!(iTL >= 0 && iTR < b.nT) || !(iBL >= 0 && iBR < b.nB)) return false;
    if (b.swap && !(iTR < b.nB && iBR < b.nT)) return false;
    const double TL = b.T[iTL], TR = b.T[iTR], BL = b.muB[iBL], BR = b.muB[iBR];
    for (int k = 0; k < 5; k++) {
        const double *f = b.tab[k];
        double f_LL, f_LR, f_RL, f_RR;
        if (b.swap) {
            f_LL = f[(size_t)iTL * b.nT + iBL]; f_LR = f[(size_t)iTL * b.nT + iBR];
            f_RL = f[(size_t)iTR * b.nT + iBL]; f_RR = f[(size_t)iTR * b.nT + iBR];
        } else {
            f_LL = f[(size_t)iBL * b.nT + iTL]; f_LR = f[(size_t)iBR * b.nT + iTL];
            f_RL = f[(size_t)iBL * b.nT + iTR]; f_RR = f[(size_t)iBR * b.nT + iTR];
        }
        v[k] = ((f_LL * (TR - T) + f_RL * (T - TL)) * (BR - muB) + (f_LR * (TR - T) + f_RR * (T - TL)) * (muB - BL)) / (dT * dB);
    }
    return true;
}

// gsl_interp_cspline evaluation (deltafReader.cpp:339-358 call sites) on LDS-resident tables.
__device__ __forceinline__ double spline_eval_lds(int n, const double *x, const double *y, const double *c, double xq)
{
    int lo = 0, hi = n - 1;
    while (hi > lo + 1) {
        int i = (hi + lo) >> 1;
        if (x[i] > xq) hi = i; else lo = i;
    }
    double x_lo = x[lo], dx = x[lo + 1] - x_lo;
    double y_lo = y[lo], dy = y[lo + 1] - y_lo;
    double delx = xq - x_lo;
    double c_i = c[lo], c_ip1 = c[lo + 1];
    double b_i = (dy / dx) - dx * (c_ip1 + 2.0 * c_i) / 3.0;
    double d_i = (c_ip1 - c_i) / (3.0 * dx);
    return y_lo + delx * (b_i + delx * (c_i + delx * d_i));
}

}  // namespace is3d
