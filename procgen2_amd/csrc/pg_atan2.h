// atan2f as the reference's process computes it: glibc 2.35's __ieee754_atan2f / __atanf (sysdeps/ieee754/flt-32/
// e_atan2f.c, s_atanf.c — the fdlibm single-precision code; on x86-64 there is exactly one build of each, scalar SSE,
// no FMA: checked against the disassembly of libm.so.6, constants read from its .rodata).  jumper's compass needle
// takes its angle from std::atan2(float, float) (games/jumper/jumper.cpp:480); every operation below is an IEEE
// single-precision add / mul / div in the same order, so the device result is the host's bit for bit
// (tests/cpp/test_primitives.cpp runs the host twin of this code against libm).  Compile with -ffp-contract=off.
#pragma once

#include "pg_defs.h"
#include "pg_sincos.h"  // sc_bits

namespace pg {

PG_HD float at_from_bits(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
#endif
}

PG_HD float at_atanf(float x) {
    const float hi[4] = {at_from_bits(0x3eed6338u), at_from_bits(0x3f490fdau), at_from_bits(0x3f7b985eu),
                         at_from_bits(0x3fc90fdau)};
    const float lo[4] = {at_from_bits(0x31ac3769u), at_from_bits(0x33222168u), at_from_bits(0x33140fb4u),
                         at_from_bits(0x33a22168u)};
    const float aT0 = at_from_bits(0x3eaaaaabu), aT1 = at_from_bits(0xbe4ccccdu), aT2 = at_from_bits(0x3e124925u),
                aT3 = at_from_bits(0xbde38e38u), aT4 = at_from_bits(0x3dba2e6eu), aT5 = at_from_bits(0xbd9d8795u),
                aT6 = at_from_bits(0x3d886b35u), aT7 = at_from_bits(0xbd6ef16bu), aT8 = at_from_bits(0x3d4bda59u),
                aT9 = at_from_bits(0xbd15a221u), aT10 = at_from_bits(0x3c8569d7u);
    const uint32_t hx = sc_bits(x);
    const uint32_t ix = hx & 0x7fffffffu;
    const bool negative = (hx >> 31) != 0;
    if (ix >= 0x4c000000u) {  // |x| >= 2^26 (or NaN)
        if (ix > 0x7f800000u) return x + x;
        return negative ? -hi[3] - lo[3] : hi[3] + lo[3];
    }
    int id;
    if (ix < 0x3ee00000u) {            // |x| < 0.4375
        if (ix < 0x31000000u) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = negative ? -x : x;
        if (ix < 0x3f980000u) {      // |x| < 1.1875
            if (ix < 0x3f300000u) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - 1.0f) / (2.0f + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - 1.0f) / (x + 1.0f);
            }
        } else {
            if (ix < 0x401c0000u) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (1.0f + 1.5f * x);
            } else {  // 2.4375 <= |x| < 2^26
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = hi[id] - ((x * (s1 + s2) - lo[id]) - x);
    return negative ? -r : r;
}

PG_HD float at_atan2f(float y, float x) {
    const float tiny = at_from_bits(0x0da24260u), pi_o_4 = at_from_bits(0x3f490fdbu), pi_o_2 = at_from_bits(0x3fc90fdbu),
                pi = at_from_bits(0x40490fdbu), pi_lo = at_from_bits(0xb3bbbd2eu);
    const uint32_t ux = sc_bits(x), uy = sc_bits(y);
    const int32_t hx = static_cast<int32_t>(ux), hy = static_cast<int32_t>(uy);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
    if (hx == 0x3f800000) return at_atanf(y);              // x = 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);     // 2·sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        }
        switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60)
        z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60)
        z = 0.0f;
    else {
        z = at_atanf(at_from_bits(sc_bits(y / x) & 0x7fffffffu));  // fabsf(y / x)
    }
    switch (m) {
        case 0: return z;
        case 1: return at_from_bits(sc_bits(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

}  // namespace pg
