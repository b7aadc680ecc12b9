// sinf / cosf bit-identical to the glibc 2.35 x86-64 build the reference links against
// (SURVEY.md §7 hard part 2: caveflyer and bossfight feed std::cos/std::sin of float angles into physics, so the
// last bit matters for reward/done parity).
//
// glibc's sinf/cosf are the Arm "optimized-routines" single-precision kernels: the argument is widened to double,
// reduced by a multiple of π/2 (fast path |x| < 120, otherwise a 192-bit fixed-point reduction with the 4/π bit
// table), and a degree-7/8 polynomial is evaluated in double and rounded once to float.  On an FMA-capable CPU
// the dynamic linker selects the *_fma variants, whose a*b+c steps are single-rounding fused operations; the
// operation order below (which products are separate, which are fused) is transcribed from that build's code, and
// the constants are the ones in its table (__sincosf_table, __inv_pio4, pi63).  Doubles and fused multiply-adds
// are IEEE on gfx950, so the same sequence gives the same bits; tests/cpp/test_primitives.cpp compares the host
// twin of this code against the real sinf/cosf over tens of millions of arguments.
#pragma once

#include "pg_defs.h"

#include <cmath>
#include <cstring>

namespace pg {

struct SinCosPoly {
    double c0, c1, c2, c3, c4, s1, s2, s3;
};

PG_HD double sc_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

PG_HD uint32_t sc_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
#endif
}

// sinf_poly: the sine (n even) or cosine (n odd) polynomial on the reduced argument, one rounding to float.
PG_HD float sc_poly(double x, double x2, bool negated, int n) {
    // __sincosf_table[0] and [1] differ only in the sign of the cosine coefficients.
    const double sg = negated ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5;
    const double c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t = sc_fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = sc_fma(x3, s1, x);
        return static_cast<float>(sc_fma(t, x7, s));
    }
    const double x4 = x2 * x2;
    const double a = sc_fma(x2, c1, c0);
    const double b = sc_fma(x2, c4, c3);
    const double x6 = x4 * x2;
    const double c = sc_fma(x4, c2, a);
    return static_cast<float>(sc_fma(b, x6, c));
}

// reduce_fast: |x| < 120.  n = round(x / (π/2)) via the 2^24-scaled reciprocal, x - n·π/2 fused.
PG_HD double sc_reduce_fast(double x, int& n) {
    const double r = x * 0x1.45f306dc9c883p+23;
    n = (static_cast<int32_t>(r) + 0x800000) >> 24;
    return sc_fma(-static_cast<double>(n), 0x1.921fb54442d18p+0, x);
}

// reduce_large: 120 <= |x| < inf, integer arithmetic on the mantissa and 96 bits of 4/π.
PG_HD double sc_reduce_large(uint32_t xi, int& n) {
    const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44,
                                   0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                   0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62,
                                   0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    uint64_t res0 = static_cast<uint64_t>(xi * arr[0]);  // 32-bit product, as in the reference
    const uint64_t res1 = static_cast<uint64_t>(xi) * arr[4];
    const uint64_t res2 = static_cast<uint64_t>(xi) * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const uint64_t q = (res0 + (1ULL << 61)) >> 62;
    res0 -= q << 62;
    n = static_cast<int>(q);
    return static_cast<double>(static_cast<int64_t>(res0)) * 0x1.921fb54442d18p-62;
}

PG_HD float sc_sinf(float y) {
    const double x = y;
    const uint32_t top = (sc_bits(y) >> 20) & 0x7ff;
    if (top < 0x3f4) {  // |y| < π/4
        if (top < 0x398) return y;  // |y| < 2^-12
        return sc_poly(x, x * x, false, 0);
    }
    int n, q;  // n: parity that picks sine/cosine polynomial; q: quadrant that picks the signs
    double r;
    if (top < 0x42f) {  // |y| < 120
        r = sc_reduce_fast(x, n);
        q = n;
    } else if (top < 0x7f8) {
        const uint32_t xi = sc_bits(y);
        r = sc_reduce_large(xi, n);
        q = n + static_cast<int>(xi >> 31);
    } else {
        return y - y;  // inf / nan → nan (the games never get here)
    }
    const double s = ((q & 3) == 0 || (q & 3) == 3) ? 1.0 : -1.0;  // sign[] = {1, -1, -1, 1}
    return sc_poly(r * s, r * r, (q & 2) != 0, n);
}

PG_HD float sc_cosf(float y) {
    const double x = y;
    const uint32_t top = (sc_bits(y) >> 20) & 0x7ff;
    if (top < 0x3f4) {
        if (top < 0x398) return 1.0f;
        return sc_poly(x, x * x, false, 1);
    }
    int n, q;
    double r;
    if (top < 0x42f) {
        r = sc_reduce_fast(x, n);
        q = n;
    } else if (top < 0x7f8) {
        const uint32_t xi = sc_bits(y);
        r = sc_reduce_large(xi, n);
        q = n + static_cast<int>(xi >> 31);
    } else {
        return y - y;
    }
    const double s = ((q & 3) == 0 || (q & 3) == 3) ? 1.0 : -1.0;
    return sc_poly(r * s, r * r, (q & 2) != 0, n ^ 1);
}

// sc_cosf(y) or sc_sinf(y), picked per call: the same code as above with the one point where they differ (which
// polynomial the parity selects) made an argument — so the lanes of a wavefront can evaluate different functions of
// different angles side by side (pg_gang.h gangs deal the trigonometry of an env out over their lanes).
PG_HD float sc_trig(float y, bool want_cos) {
    const double x = y;
    const uint32_t top = (sc_bits(y) >> 20) & 0x7ff;
    const int flip = want_cos ? 1 : 0;
    if (top < 0x3f4) {
        if (top < 0x398) return want_cos ? 1.0f : y;
        return sc_poly(x, x * x, false, flip);
    }
    int n, q;
    double r;
    if (top < 0x42f) {
        r = sc_reduce_fast(x, n);
        q = n;
    } else if (top < 0x7f8) {
        const uint32_t xi = sc_bits(y);
        r = sc_reduce_large(xi, n);
        q = n + static_cast<int>(xi >> 31);
    } else {
        return y - y;
    }
    const double s = ((q & 3) == 0 || (q & 3) == 3) ? 1.0 : -1.0;
    return sc_poly(r * s, r * r, (q & 2) != 0, n ^ flip);
}

}  // namespace pg
