// Decimal text of the .bal format, shared by the host formatter and the device kernels (text_kernels.hpp).
//
// write_text (src/baproblem.rs:709-733) prints every f64 with Rust's `{}`: the SHORTEST digit string that reads back to
// the same double, laid out without an exponent ("0.0000001", "10000000000000000000000", "-0", "NaN", "inf").  The digit
// search here is Ryu (Adams, "Ryu: fast float-to-string conversion", PLDI 2018), written from the paper: the value's
// rounding interval scaled by a power of ten through one 64 x 128-bit multiplication against a table of 5^i / 5^-i
// (125-bit entries, computed at start-up with exact integer arithmetic below -- no table text in this file), then digits
// removed while the interval still holds a shorter one.  The layout follows core::fmt's digits_to_dec_str: digits,
// then zeros up to the units place when the exponent is positive -- libstdc++'s std::to_chars(fixed) prints the EXACT
// integer value above 2^53 instead ("99999999999999991611392" for 1e23 where Rust prints "1" and 23 zeros), which is
// why the host formatter moved here from to_chars (r04; tests/test_text_device.py holds both against each other).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#if defined(__HIPCC__)
#define C2B_HD __host__ __device__ inline
#else
#define C2B_HD inline
#endif

namespace c2b_dec {

constexpr int kPow5InvCount = 342, kPow5Count = 326, kPow5Bits = 125;
struct Tables {
    uint64_t pow5_inv[kPow5InvCount][2];     // floor(2^(bits(5^i) - 1 + 125) / 5^i) + 1, {low, high}
    uint64_t pow5[kPow5Count][2];            // the top 125 bits of 5^i (shorter powers shifted up), {low, high}
};

// ---- start-up: the two tables by exact arithmetic on little-endian 32-bit limbs ---------------------------------
namespace big {
typedef std::vector<uint32_t> N;
inline void trim(N &a) { while (a.size() > 1 && a.back() == 0) a.pop_back(); }
inline int bits(const N &a) {
    int n = (int)a.size();
    while (n > 1 && a[(size_t)n - 1] == 0) --n;
    const uint32_t top = a[(size_t)n - 1];
    return top == 0 ? 0 : 32 * (n - 1) + (32 - __builtin_clz(top));
}
inline void mul_small(N &a, uint32_t m) {
    uint64_t carry = 0;
    for (auto &l : a) { const uint64_t t = (uint64_t)l * m + carry; l = (uint32_t)t; carry = t >> 32; }
    if (carry) a.push_back((uint32_t)carry);
}
inline N shl(const N &a, int s) {
    N r((size_t)(s / 32), 0u);
    const int b = s % 32;
    uint32_t carry = 0;
    for (uint32_t l : a) { r.push_back(b ? (l << b) | carry : l); carry = b ? l >> (32 - b) : 0; }
    if (carry) r.push_back(carry);
    trim(r);
    return r;
}
inline N shr(const N &a, int s) {
    const size_t w = (size_t)(s / 32);
    const int b = s % 32;
    N r;
    for (size_t i = w; i < a.size(); ++i) {
        const uint32_t lo = a[i] >> b, hi = (b && i + 1 < a.size()) ? a[i + 1] << (32 - b) : 0;
        r.push_back(lo | hi);
    }
    if (r.empty()) r.push_back(0);
    trim(r);
    return r;
}
inline int cmp(const N &a, const N &b) {
    const int ba = bits(a), bb = bits(b);
    if (ba != bb) return ba < bb ? -1 : 1;
    for (size_t i = std::max(a.size(), b.size()); i-- > 0;) {
        const uint32_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}
inline void sub(N &a, const N &b) {                   // a -= b, a >= b
    int64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        int64_t t = (int64_t)a[i] - (i < b.size() ? b[i] : 0) - borrow;
        borrow = t < 0;
        a[i] = (uint32_t)(t + (borrow << 32));
    }
    trim(a);
}
inline void div_small(N &a, uint32_t d) {              // a = floor(a / d)
    uint64_t rem = 0;
    for (size_t i = a.size(); i-- > 0;) {
        const uint64_t cur = (rem << 32) | a[i];
        a[i] = (uint32_t)(cur / d);
        rem = cur % d;
    }
    trim(a);
}
// floor(2^b / 5^k): floor divisions by positive integers compose, so k divisions by 5 -- thirteen at a time, 5^13 < 2^32
inline N pow2_over_pow5(int b, int k) {
    N a = shl(N(1, 1u), b);
    for (; k >= 13; k -= 13) div_small(a, 1220703125u);
    uint32_t d = 1;
    for (int i = 0; i < k; ++i) d *= 5u;
    if (d > 1) div_small(a, d);
    return a;
}
inline void low128(const N &a, uint64_t out[2]) {
    auto limb = [&](size_t i) { return (uint64_t)(i < a.size() ? a[i] : 0); };
    out[0] = limb(0) | (limb(1) << 32);
    out[1] = limb(2) | (limb(3) << 32);
}
}  // namespace big

inline void build_tables(Tables &t) {
    big::N p(1, 1u);                                  // 5^i
    for (int i = 0; i < kPow5InvCount; ++i) {
        const int b = big::bits(p);                   // = Ryu's pow5bits(i): 1 for i = 0
        if (i < kPow5Count) big::low128(b >= kPow5Bits ? big::shr(p, b - kPow5Bits) : big::shl(p, kPow5Bits - b), t.pow5[i]);
        big::N q = big::pow2_over_pow5(b - 1 + kPow5Bits, i);     // floor(2^(b - 1 + 125) / 5^i): at most 126 bits
        q.resize(5, 0u);
        uint64_t carry = 1;                           // + 1
        for (auto &l : q) { const uint64_t s = (uint64_t)l + carry; l = (uint32_t)s; carry = s >> 32; }
        big::low128(q, t.pow5_inv[i]);
        big::mul_small(p, 5u);
    }
}
inline const Tables &host_tables() {
    static const Tables *t = [] { Tables *x = new Tables; build_tables(*x); return x; }();
    return *t;
}

// ---- the digit search ------------------------------------------------------------------------------------------
struct Dec { uint64_t m; int32_t e; };                // value = m * 10^e, 1 <= m < 10^17 (finite non-zero input)

C2B_HD uint32_t pow5bits(int32_t e) { return (uint32_t)(((uint32_t)e * 1217359u) >> 19) + 1u; }     // ceil(log2 5^e), 0 <= e <= 3528
C2B_HD uint32_t log10_pow2(int32_t e) { return ((uint32_t)e * 78913u) >> 18; }                       // floor(e log10 2), 0 <= e <= 1650
C2B_HD uint32_t log10_pow5(int32_t e) { return ((uint32_t)e * 732923u) >> 20; }                      // floor(e log10 5), 0 <= e <= 2620
C2B_HD uint32_t pow5_factor(uint64_t v) {
    uint32_t n = 0;
    while (v != 0 && v % 5u == 0) { v /= 5u; ++n; }
    return n;
}
C2B_HD bool multiple_of_pow5(uint64_t v, uint32_t p) { return pow5_factor(v) >= p; }
C2B_HD bool multiple_of_pow2(uint64_t v, uint32_t p) { return (v & ((1ull << p) - 1ull)) == 0; }
// (m * mul) >> j for a 128-bit mul = {low, high} and 64 < j < 128 + 64: the product's bits [j, j + 64)
C2B_HD uint64_t mul_shift(uint64_t m, const uint64_t *mul, int32_t j) {
    const unsigned __int128 b0 = (unsigned __int128)m * mul[0];
    const unsigned __int128 b2 = (unsigned __int128)m * mul[1];
    return (uint64_t)(((b0 >> 64) + b2) >> (j - 64));
}

C2B_HD uint32_t decimal_length17(uint64_t v) {        // v < 10^17
    uint32_t n = 1;
    uint64_t p = 10;
    while (n < 17 && v >= p) { ++n; p *= 10; }
    return n;
}

// shortest digits of a finite, non-zero double given as its fields (mantissa: 52 bits, exponent: 11 bits)
C2B_HD Dec shortest(uint64_t ieee_mantissa, uint32_t ieee_exponent, const Tables *T) {
    int32_t e2;
    uint64_t m2;
    if (ieee_exponent == 0) { e2 = 1 - 1023 - 52 - 2; m2 = ieee_mantissa; }
    else { e2 = (int32_t)ieee_exponent - 1023 - 52 - 2; m2 = (1ull << 52) | ieee_mantissa; }
    const bool accept_bounds = (m2 & 1) == 0;         // round-half-even reads an end point of the interval back
    // the interval of values that read back: (4 m2 - 1 - shift, 4 m2 + 2) * 2^e2; the lower half is half as wide at a
    // power of two
    const uint64_t mv = 4 * m2;
    const uint32_t mm_shift = (ieee_mantissa != 0 || ieee_exponent <= 1) ? 1u : 0u;
    uint64_t vr, vp, vm;
    int32_t e10;
    bool vm_trailing = false, vr_trailing = false;
    if (e2 >= 0) {
        const uint32_t q = log10_pow2(e2) - (e2 > 3 ? 1u : 0u);
        e10 = (int32_t)q;
        const int32_t k = kPow5Bits + (int32_t)pow5bits((int32_t)q) - 1;
        const int32_t i = -e2 + (int32_t)q + k;
        const uint64_t *mul = T->pow5_inv[q];
        vr = mul_shift(4 * m2, mul, i);
        vp = mul_shift(4 * m2 + 2, mul, i);
        vm = mul_shift(4 * m2 - 1 - mm_shift, mul, i);
        if (q <= 21) {                                // only one of the three can be a multiple of 5
            if (mv % 5u == 0) vr_trailing = multiple_of_pow5(mv, q);
            else if (accept_bounds) vm_trailing = multiple_of_pow5(mv - 1 - mm_shift, q);
            else vp -= multiple_of_pow5(mv + 2, q) ? 1u : 0u;
        }
    } else {
        const uint32_t q = log10_pow5(-e2) - (-e2 > 1 ? 1u : 0u);
        e10 = (int32_t)q + e2;
        const int32_t i = -e2 - (int32_t)q;
        const int32_t k = (int32_t)pow5bits(i) - kPow5Bits;
        const int32_t j = (int32_t)q - k;
        const uint64_t *mul = T->pow5[i];
        vr = mul_shift(4 * m2, mul, j);
        vp = mul_shift(4 * m2 + 2, mul, j);
        vm = mul_shift(4 * m2 - 1 - mm_shift, mul, j);
        if (q <= 1) {
            vr_trailing = true;                       // mv = 4 m2 has two trailing zero bits
            if (accept_bounds) vm_trailing = mm_shift == 1;
            else --vp;
        } else if (q < 63) {
            vr_trailing = multiple_of_pow2(mv, q);
        }
    }
    int32_t removed = 0;
    uint32_t last_removed = 0;
    uint64_t out;
    if (vm_trailing || vr_trailing) {                 // the rare exact cases: track whether what was removed is all zeros
        while (true) {
            const uint64_t vp10 = vp / 10, vm10 = vm / 10;
            if (vp10 <= vm10) break;
            const uint32_t vm_mod = (uint32_t)(vm - 10 * vm10);
            const uint64_t vr10 = vr / 10;
            const uint32_t vr_mod = (uint32_t)(vr - 10 * vr10);
            vm_trailing &= vm_mod == 0;
            vr_trailing &= last_removed == 0;
            last_removed = vr_mod;
            vr = vr10; vp = vp10; vm = vm10;
            ++removed;
        }
        if (vm_trailing) {
            while (true) {
                const uint64_t vm10 = vm / 10;
                const uint32_t vm_mod = (uint32_t)(vm - 10 * vm10);
                if (vm_mod != 0) break;
                const uint64_t vp10 = vp / 10, vr10 = vr / 10;
                const uint32_t vr_mod = (uint32_t)(vr - 10 * vr10);
                vr_trailing &= last_removed == 0;
                last_removed = vr_mod;
                vr = vr10; vp = vp10; vm = vm10;
                ++removed;
            }
        }
        if (vr_trailing && last_removed == 5 && vr % 2 == 0) last_removed = 4;      // exactly ...5000: round to even
        out = vr + (((vr == vm && (!accept_bounds || !vm_trailing)) || last_removed >= 5) ? 1u : 0u);
    } else {
        bool round_up = false;
        while (true) {
            const uint64_t vp10 = vp / 10, vm10 = vm / 10;
            if (vp10 <= vm10) break;
            const uint64_t vr10 = vr / 10;
            round_up = (uint32_t)(vr - 10 * vr10) >= 5;
            vr = vr10; vp = vp10; vm = vm10;
            ++removed;
        }
        out = vr + ((vr == vm || round_up) ? 1u : 0u);
    }
    Dec d;
    d.m = out;
    d.e = e10 + removed;
    return d;
}

// ---- Rust's `{}` layout ------------------------------------------------------------------------------------------
// kinds: 0 finite non-zero, 1 zero, 2 NaN, 3 infinity
struct Text { Dec d; uint32_t n_digits; uint32_t len; uint8_t kind; bool neg; };

C2B_HD Text describe(double v, const Tables *T) {
    uint64_t bits;
    memcpy(&bits, &v, 8);
    Text t;
    t.neg = (bits >> 63) != 0;
    const uint64_t man = bits & ((1ull << 52) - 1);
    const uint32_t ex = (uint32_t)((bits >> 52) & 0x7ffu);
    t.d.m = 0; t.d.e = 0; t.n_digits = 0;
    if (ex == 0x7ffu) {
        t.kind = man ? 2 : 3;
        if (man) t.neg = false;                       // "NaN" carries no sign
        t.len = 3 + (t.neg ? 1u : 0u);
        return t;
    }
    if (ex == 0 && man == 0) { t.kind = 1; t.len = 1 + (t.neg ? 1u : 0u); return t; }
    t.kind = 0;
    t.d = shortest(man, ex, T);
    t.n_digits = decimal_length17(t.d.m);
    const int32_t point = (int32_t)t.n_digits + t.d.e;                  // digits in front of the decimal point
    uint32_t len;
    if (t.d.e >= 0) len = t.n_digits + (uint32_t)t.d.e;                // digits, zeros up to the units place
    else if (point > 0) len = t.n_digits + 1;                           // a point inside the digits
    else len = 2 + (uint32_t)(-point) + t.n_digits;                     // "0." zeros digits
    t.len = len + (t.neg ? 1u : 0u);
    return t;
}

// writes exactly t.len bytes at dst (P: a pointer to char in any address space -- the device kernels write into LDS)
template <typename P>
C2B_HD void emit(const Text &t, P dst) {
    if (t.neg) *dst++ = '-';
    if (t.kind == 1) { *dst = '0'; return; }
    if (t.kind == 2) { dst[0] = 'N'; dst[1] = 'a'; dst[2] = 'N'; return; }
    if (t.kind == 3) { dst[0] = 'i'; dst[1] = 'n'; dst[2] = 'f'; return; }
    const int32_t n = (int32_t)t.n_digits, point = n + t.d.e;
    uint64_t m = t.d.m;
    if (t.d.e >= 0) {
        for (int32_t i = n - 1; i >= 0; --i) { dst[i] = (char)('0' + m % 10); m /= 10; }
        for (int32_t i = 0; i < t.d.e; ++i) dst[n + i] = '0';
    } else if (point > 0) {
        for (int32_t i = n - 1; i >= 0; --i) { dst[i < point ? i : i + 1] = (char)('0' + m % 10); m /= 10; }
        dst[point] = '.';
    } else {
        dst[0] = '0'; dst[1] = '.';
        for (int32_t i = 0; i < -point; ++i) dst[2 + i] = '0';
        P q = dst + 2 - point;
        for (int32_t i = n - 1; i >= 0; --i) { q[i] = (char)('0' + m % 10); m /= 10; }
    }
}

// unsigned decimal integers (indices, counts)
C2B_HD uint32_t uint_len(uint64_t v) {
    uint32_t n = 1;
    while (v >= 10) { v /= 10; ++n; }
    return n;
}
template <typename P>
C2B_HD void uint_emit(uint64_t v, uint32_t n, P dst) {
    for (int32_t i = (int32_t)n - 1; i >= 0; --i) { dst[i] = (char)('0' + v % 10); v /= 10; }
}

}  // namespace c2b_dec

// ================================================================================================================
// Reading: decimal text -> the nearest double (from_file_text, src/baproblem.rs:580-629, reads its numbers with nom's
// `double`, i.e. str::parse::<f64>: correctly rounded).  The significand's digits go into a 64-bit integer w (at most
// 19 significant digits; the writer above never makes more than 17), the value is w * 10^q, and it is rounded by
//   * Clinger's exact case: w <= 2^53 and |q| <= 22 -- both operands are exact doubles, one IEEE multiply or divide;
//   * otherwise Eisel-Lemire (Lemire, "Number parsing at a gigabyte per second", SPE 2021): w times a 128-bit
//     truncation of 5^q gives the leading bits of the product and a bound on what was cut; when that bound cannot
//     decide the rounding the parse reports `unsure` (and the caller hands the file to the host's strtod) instead of
//     guessing.  The table -- 5^q for q in [-342, 308], normalised to 128 bits, reciprocals rounded up -- is computed at
//     start-up by the exact arithmetic above, like the writer's.
// Anything but [+-]digits[.digits][(e|E)[+-]digits] is `irregular` -- the host parser owns every other spelling.
namespace c2b_dec {

constexpr int kPow10Min = -342, kPow10Max = 308;
struct ParseTables {
    uint64_t pow5_128[kPow10Max - kPow10Min + 1][2];      // {high, low}
    double exact10[23];                                     // 10^0 .. 10^22
};

namespace big {
inline void add_one(N &a) {
    for (auto &l : a) { if (++l != 0) return; }
    a.push_back(1u);
}
}  // namespace big

inline void build_parse_tables(ParseTables &t) {
    auto store = [&](int q, big::N c) {                   // c has exactly 128 bits
        uint64_t lohi[2];
        big::low128(c, lohi);
        t.pow5_128[q - kPow10Min][0] = lohi[1];
        t.pow5_128[q - kPow10Min][1] = lohi[0];
    };
    big::N p(1, 1u);                                      // 5^k
    std::vector<big::N> pow5;
    for (int k = 0; k <= 342; ++k) { pow5.push_back(p); big::mul_small(p, 5u); }
    for (int q = 0; q <= kPow10Max; ++q) {                // 5^q with its top bit at bit 127, truncated
        const big::N &v = pow5[(size_t)q];
        const int b = big::bits(v);
        store(q, b <= 128 ? big::shl(v, 128 - b) : big::shr(v, b - 128));
    }
    for (int q = -1; q >= kPow10Min; --q) {               // 2^b / 5^-q, rounded up, 128 bits
        const big::N &v = pow5[(size_t)(-q)];
        int z = 0;                                        // smallest z with 2^z >= 5^-q
        { const int b = big::bits(v); z = b; big::N one = big::shl(big::N(1, 1u), b - 1); if (big::cmp(one, v) >= 0) z = b - 1; }
        const int b = q >= -27 ? z + 127 : 2 * z + 128;
        big::N c = big::pow2_over_pow5(b, -q);
        big::add_one(c);
        const int cb = big::bits(c);
        if (cb > 128) c = big::shr(c, cb - 128);
        store(q, c);
    }
    double e = 1.0;
    for (int k = 0; k < 23; ++k) { t.exact10[k] = e; e *= 10.0; }
}
inline const ParseTables &host_parse_tables() {
    static const ParseTables *t = [] { ParseTables *x = new ParseTables; build_parse_tables(*x); return x; }();
    return *t;
}

enum { PARSE_OK = 0, PARSE_IRREGULAR = 1, PARSE_UNSURE = 2 };

// w * 10^q, w != 0, rounded to nearest even; *status = PARSE_UNSURE when the truncated product cannot decide
C2B_HD double scale10(uint64_t w, int32_t q, const ParseTables *T, int *status) {
    const double inf = __builtin_huge_val();
    if (q < kPow10Min) return 0.0;                        // w < 2^64 < 10^20: below half the smallest subnormal
    if (q > kPow10Max) return inf;
    if (w <= (1ull << 53) && q >= -22 && q <= 22) {       // both exact: one correctly rounded operation
        const double d = (double)w;
        return q < 0 ? d / T->exact10[-q] : d * T->exact10[q];
    }
    const int lz0 = __builtin_clzll(w);
    w <<= lz0;
    const uint64_t *five = T->pow5_128[q - kPow10Min];
    unsigned __int128 first = (unsigned __int128)w * five[0];
    uint64_t upper = (uint64_t)(first >> 64), lower = (uint64_t)first;
    if ((upper & 0x1ffu) == 0x1ffu) {                     // the nine bits below the rounding bit are all ones: look further
        const unsigned __int128 second = (unsigned __int128)w * five[1];
        const uint64_t sh = (uint64_t)(second >> 64);
        lower += sh;
        if (sh > lower) ++upper;
        if (lower == 0xffffffffffffffffull) { *status = PARSE_UNSURE; return 0.0; }     // what was cut could still carry
    }
    const int upperbit = (int)(upper >> 63);
    uint64_t mantissa = upper >> (upperbit + 9);          // 54 bits: the 53 of the result and the rounding bit
    // floor(log2 10^q) + 63 is the binary exponent of bit 63 of 5^q's table entry times 2^q; the bias of 1023 goes in here
    int32_t power2 = (int32_t)((((int64_t)(152170 + 65536) * q) >> 16) + 63) + upperbit - lz0 + 1023;
    if (power2 <= 0) {                                    // subnormal (or zero)
        if (-power2 + 1 >= 64) return 0.0;
        mantissa >>= -power2 + 1;
        mantissa += mantissa & 1;
        mantissa >>= 1;
        const uint64_t bits = mantissa;                   // exponent field 0, or 1 if the rounding carried into it
        double d;
        memcpy(&d, &bits, 8);
        return d;
    }
    // exactly half way between two doubles: only possible when 5^q fits 64 bits; then the product is exact
    if (lower <= 1 && q >= -4 && q <= 23 && (mantissa & 3) == 1 && (mantissa << (upperbit + 9)) == upper) mantissa &= ~1ull;
    mantissa += mantissa & 1;
    mantissa >>= 1;
    if (mantissa >= (2ull << 52)) { mantissa = 1ull << 52; ++power2; }
    mantissa &= ~(1ull << 52);
    if (power2 >= 0x7ff) return inf;
    const uint64_t bits = ((uint64_t)power2 << 52) | mantissa;
    double d;
    memcpy(&d, &bits, 8);
    return d;
}

// one whitespace-delimited token [s, s + n) -> double (P: a pointer to const char in any address space)
template <typename P>
C2B_HD double parse_f64(P s, int32_t n, const ParseTables *T, int *status) {
    int32_t i = 0;
    bool neg = false;
    if (i < n && (s[i] == '-' || s[i] == '+')) { neg = s[i] == '-'; ++i; }
    uint64_t w = 0;
    int32_t sig = 0, q = 0, digits = 0;
    bool lost = false;                                    // a non-zero digit beyond the 19 kept
    for (; i < n && s[i] >= '0' && s[i] <= '9'; ++i, ++digits) {
        const uint32_t d = (uint32_t)(s[i] - '0');
        if (sig < 19) { w = w * 10 + d; sig += (w != 0) ? 1 : 0; }
        else { ++q; lost |= d != 0; }
    }
    if (i < n && s[i] == '.') {
        ++i;
        for (; i < n && s[i] >= '0' && s[i] <= '9'; ++i, ++digits) {
            const uint32_t d = (uint32_t)(s[i] - '0');
            if (sig < 19) { w = w * 10 + d; sig += (w != 0) ? 1 : 0; --q; }
            else lost |= d != 0;
        }
    }
    if (digits == 0) { *status = PARSE_IRREGULAR; return 0.0; }
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        ++i;
        bool eneg = false;
        if (i < n && (s[i] == '-' || s[i] == '+')) { eneg = s[i] == '-'; ++i; }
        int32_t e = 0, ed = 0;
        for (; i < n && s[i] >= '0' && s[i] <= '9'; ++i, ++ed) if (e < 100000) e = e * 10 + (s[i] - '0');
        if (ed == 0) { *status = PARSE_IRREGULAR; return 0.0; }
        q += eneg ? -e : e;
    }
    if (i != n || lost) { *status = i != n ? PARSE_IRREGULAR : PARSE_UNSURE; return 0.0; }
    const double v = w == 0 ? 0.0 : scale10(w, q, T, status);
    return neg ? -v : v;
}

// an index or a count: digits only
template <typename P>
C2B_HD uint64_t parse_u64(P s, int32_t n, int *status) {
    uint64_t v = 0;
    if (n <= 0 || n > 19) { *status = PARSE_IRREGULAR; return 0; }
    for (int32_t i = 0; i < n; ++i) {
        if (s[i] < '0' || s[i] > '9') { *status = PARSE_IRREGULAR; return 0; }
        v = v * 10 + (uint64_t)(s[i] - '0');
    }
    return v;
}

}  // namespace c2b_dec
