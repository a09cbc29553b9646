// CPU half of the |p|^4 parity claim: csrc/pow4_libm.hpp (the restatement the device runs) against the image's own
// pow(x, 4.0), bit for bit, over the arguments the projection can produce and over the special ranges.
// Built and run by tests/test_pow4.py:  g++ -O2 -mfma -ffp-contract=off -std=c++17 pow4_host_harness.cpp -lm
// Prints one line per range: "<name> <count> <mismatches>"; exit status 1 if any mismatch.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../city2ba_amd/csrc/pow4_libm.hpp"

static const uint64_t kLog[] = {C2B_POW_LOG_TAB};
static const uint64_t kExp[] = {C2B_EXP_TAB};

static uint64_t s[2] = {0x9E3779B97F4A7C15ULL, 0xD1B54A32D192ED03ULL};
static uint64_t next() {                                            // xorshift128+
    uint64_t a = s[0], b = s[1];
    s[0] = b; a ^= a << 23; s[1] = a ^ b ^ (a >> 17) ^ (b >> 26);
    return s[1] + b;
}
static double uni() { return (double)(next() >> 11) * 0x1p-53; }

static long bad_total = 0;
static volatile double four = 4.0;                                   // keep the compiler from folding pow(x, 4.0)
static void check(const char *name, long n, double (*gen)(long)) {
    const c2b::PowLogRow *lt = (const c2b::PowLogRow *)kLog;
    const c2b::PowExpRow *et = (const c2b::PowExpRow *)kExp;
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        const double x = gen(i);
        const double want = pow(x, four), got = c2b::pow4_glibc(x, lt, et);
        uint64_t a, b; memcpy(&a, &want, 8); memcpy(&b, &got, 8);
        if (a != b && !(want != want && got != got)) {
            if (bad < 5) fprintf(stderr, "%s: x=%a libm=%a here=%a\n", name, x, want, got);
            ++bad;
        }
    }
    printf("%s %ld %ld\n", name, n, bad);
    bad_total += bad;
}

static double g_sqrt_u04(long) { return sqrt(4.0 * uni()); }                     // the test domain of tests/test_pow4.py
static double g_sqrt_wide(long) { return sqrt(exp2(-40.0 + 80.0 * uni())); }     // |p| from 1e-6 to 1e6
static double g_any_normal(long) { uint64_t u = next() & 0x7fffffffffffffffULL; double d; memcpy(&d, &u, 8); return d; }   // every exponent, NaN/inf included
static double g_subnormal(long) { uint64_t u = next() & 0x000fffffffffffffULL; double d; memcpy(&d, &u, 8); return d; }
static double g_uflow_edge(long) { return exp2(-272.0 + 20.0 * uni()); }          // x^4 around 2^-1088 .. 2^-1008: subnormal results
static double g_oflow_edge(long) { return exp2(254.0 + 3.0 * uni()); }            // x^4 around 2^1016 .. 2^1028
static double g_near_one(long) { return 1.0 + (uni() - 0.5) * exp2(-10.0 - 50.0 * uni()); }
static double g_special(long i) {
    static const double v[] = {0.0, -0.0, 1.0, -1.0, 2.0, 0.5, INFINITY, -INFINITY, NAN, 0x1p-1074, 0x1p-1022, 0x1.fffffffffffffp1023,
                               -2.5, -0x1p-1050, 0x1p256, 0x1p-256, 0x1p-269, 0x1p-268, 0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bccp0};
    return v[i % (long)(sizeof v / sizeof v[0])];
}
static double g_neg(long) { return -sqrt(4.0 * uni()); }

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 4000000;
    check("sqrt_u04", n, g_sqrt_u04);
    check("sqrt_wide", n, g_sqrt_wide);
    check("any_bits", n, g_any_normal);
    check("subnormal_x", n / 4, g_subnormal);
    check("underflow_edge", n, g_uflow_edge);
    check("overflow_edge", n / 4, g_oflow_edge);
    check("near_one", n, g_near_one);
    check("negative_x", n / 4, g_neg);
    check("special", 20, g_special);
    return bad_total ? 1 : 0;
}
