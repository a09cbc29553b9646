// pow4_libm.hpp -- |p|^4 exactly as the reference computes it: p_.magnitude().powf(4.0) (src/baproblem.rs:149).
//
// Rust's f64::powf is llvm.pow.f64 = the platform libm's pow().  glibc's pow (2.28 and later; this image: 2.35) is
// exp(y log x) with a 2^-68-accurate logarithm: log through 128 intervals of z with tabled (1/c, log c hi, log c lo),
// r = z/c - 1 exact by one FMA, a degree-7 tail polynomial; exp through a 128-entry 2^(i/128) table and a degree-5
// polynomial.  Faithful (0.52 ulp), NOT correctly rounded: pow(sqrt(n), 4.0) differs from fl(fl(sqrt n)^4) in ~9e-4 of
// arguments by one ulp (tests/test_pow4.py), and which arguments those are is a property of the tables AND of how the
// library's build fused multiplies with adds.  So this is a restatement of the machine code, not of the C source: the
// operation list below is read off `objdump -d libm.so.6` of Ubuntu GLIBC 2.35-0ubuntu3.11, the variant the ifunc
// resolver of pow() picks on every x86-64 with FMA + AVX2 (`__pow_fma`: e_pow.c compiled -mfma -mavx2, where
// __FP_FAST_FMA selects the FMA forms of log_inline / exp_inline and GCC's default -ffp-contract=fast fused the rest).
// Every fma() below is a v*fmadd*sd of that listing, every separate * and + a vmulsd / vaddsd / vsubsd of it, in the
// data-flow order of the listing.  (aarch64 builds of the same source define __FP_FAST_FMA too; a pre-FMA x86 takes
// `__pow_sse2`, whose split arithmetic rounds differently in rare cases -- not what any current host runs.)
//
// Specialised to y = 4.0 (ehi = 4 hi and elo = 4 lo are exact) but otherwise complete: zero, subnormal, infinite and NaN
// arguments, overflow, underflow and results in the subnormal range take glibc's own paths (checked bit for bit against
// the image's pow over 40 M arguments incl. those ranges: tests/test_pow4.py, tools/probes/pow4_host_harness.cpp).
//
// Host + device: compiles under plain g++ (the CPU half of the test) and under hipcc.  Compile with
// -ffp-contract=off; the tables (csrc/pow_tables.inc, lifted from libm by tools/gen_pow_tables.py) are passed in so a
// kernel can hand over global or LDS copies.
#pragma once
#include <stdint.h>
#include <string.h>
#include "pow_tables.inc"

#if defined(__HIPCC__)
#define C2B_POW_HD __host__ __device__ __forceinline__
#else
#define C2B_POW_HD static inline
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define C2B_POW_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define C2B_POW_FENCE() ((void)0)
#endif

namespace c2b {

struct PowLogRow { uint64_t invc, logc, logctail, pad; };       // 32 bytes
struct PowExpRow { uint64_t tail, sbits; };                     // 16 bytes
#define C2B_POW_LOG_ROWS 128
#define C2B_POW_EXP_ROWS 128

C2B_POW_HD double pow_asdouble(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; memcpy(&d, &u, 8); return d;
#endif
}
C2B_POW_HD uint64_t pow_asuint(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u; memcpy(&u, &d, 8); return u;
#endif
}

// A constant of the algorithm.  On the device it is materialised (two s_mov) where it is used: the opaque asm keeps the
// compiler from hoisting all twenty of them out of a caller's loop, where they would sit in forty SGPRs for the loop's
// whole life (k_cells_visibility spilled scalar registers that way) although only cameras with k2 != 0 ever need them.
C2B_POW_HD double pow_const(uint64_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(bits));
#endif
    return pow_asdouble(bits);
}

// LT / ET: anything indexable that yields PowLogRow / PowExpRow fields (pointers into global memory, LDS, host arrays)
template <typename LT, typename ET>
C2B_POW_HD double pow4_glibc(double x, LT logtab, ET exptab) {
    uint64_t ix = pow_asuint(x);
    uint32_t topx = (uint32_t)(ix >> 52);
    if (topx - 1u > 0x7fdu) {                                   // 0, subnormal, inf, NaN, or the sign bit set
        if (2 * ix - 1 >= 2 * 0x7ff0000000000000ULL - 1) return x * x;        // zeroinfnan(x): x2 (y = 4 > 0, even)
        if (ix >> 63) { ix &= 0x7fffffffffffffffULL; topx &= 0x7ffu; }        // x < 0: y is an even integer, no sign bias
        if (topx == 0) {                                        // subnormal: normalise, exponent goes negative
            ix = pow_asuint(pow_asdouble(ix) * 0x1p52);
            ix &= 0x7fffffffffffffffULL;
            ix -= 52ULL << 52;
        }
    }
    // ---- log_inline: (hi, tail) = log(x) ------------------------------------------------------------------------------
    // (the statements are ordered so that few values are live at once, and on the device C2B_POW_FENCE keeps the
    // scheduler from re-interleaving them: this is a cold path inside kernels that have no registers to spare)
    const uint64_t tmp = ix - 0x3fe6955500000000ULL;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & 0xfff0000000000000ULL);
    const double z = pow_asdouble(iz), kd = (double)k;
    const double invc = pow_asdouble(logtab[i].invc), logc = pow_asdouble(logtab[i].logc), logctail = pow_asdouble(logtab[i].logctail);
    const double r = __builtin_fma(z, invc, -1.0);
    C2B_POW_FENCE();
    const double t1 = __builtin_fma(kd, pow_const(C2B_POW_LN2HI), logc);
    const double lo1 = __builtin_fma(kd, pow_const(C2B_POW_LN2LO), logctail);
    C2B_POW_FENCE();
    const double t2 = r + t1;
    const double lo2 = (t1 - t2) + r;
    const double s12 = lo1 + lo2;
    C2B_POW_FENCE();
    const double ar = r * pow_const(C2B_POW_A0);
    const double ar2 = r * ar;
    const double lo3 = __builtin_fma(ar, r, -ar2);
    const double s123 = s12 + lo3;
    C2B_POW_FENCE();
    const double hi = t2 + ar2;
    const double lo4 = (t2 - hi) + ar2;
    const double s1234 = s123 + lo4;
    C2B_POW_FENCE();
    const double q3 = __builtin_fma(r, pow_const(C2B_POW_A6), pow_const(C2B_POW_A5));
    const double q2 = __builtin_fma(r, pow_const(C2B_POW_A4), pow_const(C2B_POW_A3));
    const double q23 = __builtin_fma(q3, ar2, q2);
    C2B_POW_FENCE();
    const double q1 = __builtin_fma(r, pow_const(C2B_POW_A2), pow_const(C2B_POW_A1));
    const double qq = __builtin_fma(ar2, q23, q1);
    const double ar3 = r * ar2;
    const double lo = __builtin_fma(ar3, qq, s1234);
    C2B_POW_FENCE();
    const double lhi = hi + lo;
    const double ltail = (hi - lhi) + lo;
    // ---- y log x in two pieces (y = 4.0) --------------------------------------------------------------------------------
    const double y = 4.0;
    const double ehi = y * lhi;
    const double elo = __builtin_fma(y, ltail, __builtin_fma(lhi, y, -ehi));
    C2B_POW_FENCE();
    // ---- exp_inline(ehi, elo, sign_bias = 0) ---------------------------------------------------------------------------
    const uint64_t ebits = pow_asuint(ehi);
    uint32_t abstop = (uint32_t)(ebits >> 52) & 0x7ffu;
    if (abstop - 0x3c9u > 0x3eu) {
        if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + ehi;   // |y log x| < 2^-54
        if (abstop > 0x408u) return (ebits >> 63) ? 0.0 : pow_asdouble(0x7ff0000000000000ULL);     // __math_uflow(0) = +0 / __math_oflow(0) = +inf
        abstop = 0;                                             // 512 <= |y log x| < 1024: the scale needs care (specialcase)
    }
    const double shift = pow_const(C2B_EXP_SHIFT);
    double kd2 = __builtin_fma(ehi, pow_const(C2B_EXP_INVLN2N), shift);
    const uint64_t ki = pow_asuint(kd2);
    kd2 = kd2 - shift;
    double rr = __builtin_fma(kd2, pow_const(C2B_EXP_NEGLN2HIN), ehi);
    rr = __builtin_fma(kd2, pow_const(C2B_EXP_NEGLN2LON), rr);
    const int idx = (int)(ki & 127);
    uint64_t sbits = exptab[idx].sbits + (ki << 45);
    rr = elo + rr;
    C2B_POW_FENCE();
    const double p23 = __builtin_fma(rr, pow_const(C2B_EXP_C3), pow_const(C2B_EXP_C2));
    const double tr = rr + pow_asdouble(exptab[idx].tail);
    const double r2 = rr * rr;
    const double s1 = __builtin_fma(p23, r2, tr);
    C2B_POW_FENCE();
    const double p45 = __builtin_fma(rr, pow_const(C2B_EXP_C5), pow_const(C2B_EXP_C4));
    const double r4 = r2 * r2;
    const double tm = __builtin_fma(p45, r4, s1);
    C2B_POW_FENCE();
    if (abstop == 0) {                                          // specialcase(tmp, sbits, ki)
        if ((ki & 0x80000000ULL) == 0) {                        // k > 0: the scale's exponent may have overflowed
            sbits -= 1009ULL << 52;
            const double sc = pow_asdouble(sbits);
            return 0x1p1009 * __builtin_fma(sc, tm, sc);
        }
        sbits += 1022ULL << 52;                                 // k < 0: round once, in the subnormal range's precision
        const double sc = pow_asdouble(sbits);
        const double st = tm * sc;
        double yy = sc + st;
        if (__builtin_fabs(yy) < 1.0) {
            const double one = yy < 0.0 ? -1.0 : 1.0;
            const double l0 = (sc - yy) + st;
            const double h1 = yy + one;
            const double l1 = ((one - h1) + yy) + l0;
            yy = (l1 + h1) - one;
            if (yy == 0.0) yy = pow_asdouble(sbits & 0x8000000000000000ULL);
        }
        return 0x1p-1022 * yy;
    }
    const double scale = pow_asdouble(sbits);
    return __builtin_fma(tm, scale, scale);
}

}  // namespace c2b
