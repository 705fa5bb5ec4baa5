// sq_math.h -- binary32 cosf / sinf / powf / expf / logf with host-libm-identical results.
//
// Why this exists: the reference's surface sampler (src/super_quadric/learnable_primitives/
// fast_sampler/sampling.cpp:59-125) takes integer decisions -- round(dA/(dA+dB)*(N-1)) and a
// lower_bound in a float CDF -- on values that come out of std::cos / std::sin / std::pow on
// float, i.e. glibc's cosf/sinf/powf.  A device math library that differs by one ulp flips some
// of those decisions and the 200-step Adam trajectory leaves the reference's.  These routines
// evaluate the published algorithms of glibc >= 2.28 (the Arm "optimized-routines" single-
// precision kernels: table + polynomial in binary64, one final rounding to binary32), so that
// the device results equal the host libm's bit for bit.  tests/test_sq_math.py sweeps them
// against the host libm on the CPU build of this very header.
//
// Everything is written with explicit operation order; compile with -ffp-contract=off.
//
// Provenance of the algorithms and constants (polynomial coefficients, table construction, range reduction): the
// single-precision routines of Arm Optimized Routines (sinf, cosf, powf, expf, logf), which glibc adopted in 2.28.
// Upstream notice of that project:
//
//   Copyright (c) 1999-2019, Arm Limited.
//   SPDX-License-Identifier: MIT
//
//   Permission is hereby granted, free of charge, to any person obtaining a copy of this software and associated
//   documentation files (the "Software"), to deal in the Software without restriction, including without limitation
//   the rights to use, copy, modify, merge, publish, distribute, sublicense, and/or sell copies of the Software, and to
//   permit persons to whom the Software is furnished to do so, subject to the following conditions:
//   The above copyright notice and this permission notice shall be included in all copies or substantial portions of
//   the Software.
//   THE SOFTWARE IS PROVIDED "AS IS", WITHOUT WARRANTY OF ANY KIND, EXPRESS OR IMPLIED, INCLUDING BUT NOT LIMITED TO
//   THE WARRANTIES OF MERCHANTABILITY, FITNESS FOR A PARTICULAR PURPOSE AND NONINFRINGEMENT. IN NO EVENT SHALL THE
//   AUTHORS OR COPYRIGHT HOLDERS BE LIABLE FOR ANY CLAIM, DAMAGES OR OTHER LIABILITY, WHETHER IN AN ACTION OF CONTRACT,
//   TORT OR OTHERWISE, ARISING FROM, OUT OF OR IN CONNECTION WITH THE SOFTWARE OR THE USE OR OTHER DEALINGS IN THE
//   SOFTWARE.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ODAM_HD __host__ __device__ __forceinline__
#else
#define ODAM_HD static inline
#endif

namespace odam_math {

ODAM_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
ODAM_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
ODAM_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
ODAM_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
ODAM_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ffu; }

// ---- sin / cos -------------------------------------------------------------------------------
// polynomial set p = 0: (+), p = 1: (-)   [c0,c1,c2,c3,c4 | s1,s2,s3]
struct sincos_poly { double c0, c1, c2, c3, c4, s1, s2, s3; };

ODAM_HD sincos_poly sc_tab(int neg) {
    sincos_poly p;
    const double sg = neg ? -1.0 : 1.0;
    p.c0 = sg * 0x1p0;
    p.c1 = sg * -0x1.ffffffd0c621cp-2;
    p.c2 = sg * 0x1.55553e1068f19p-5;
    p.c3 = sg * -0x1.6c087e89a359dp-10;
    p.c4 = sg * 0x1.99343027bf8c3p-16;
    p.s1 = -0x1.555545995a603p-3;
    p.s2 = 0x1.1107605230bc4p-7;
    p.s3 = -0x1.994eb3774cf24p-13;
    return p;
}

ODAM_HD float sc_poly(double x, double x2, const sincos_poly& p, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = p.s2 + x2 * p.s3;
        double x7 = x3 * x2;
        double s = x + x3 * p.s1;
        return (float)(s + x7 * s1);
    } else {
        double x4 = x2 * x2;
        double c2 = p.c3 + x2 * p.c4;
        double c1 = p.c0 + x2 * p.c1;
        double x6 = x4 * x2;
        double c = c1 + x4 * p.c2;
        return (float)(c + x6 * c2);
    }
}

// valid for |y| < 120 (the sampler only ever passes |y| <= pi); larger arguments are not needed
ODAM_HD float sinf_(float y) {
    double x = (double)y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double s = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sc_poly(x, s, sc_tab(0), 0);
    }
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // {1,-1,-1,1}[n&3]
    return sc_poly(x * sign, x * x, sc_tab((n & 2) != 0), n);
}

ODAM_HD float cosf_(float y) {
    double x = (double)y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        double x2 = x * x;
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sc_poly(x, x2, sc_tab(0), 1);
    }
    double r = x * 0x1.45F306DC9C883p+23;
    int n = ((int32_t)r + 0x800000) >> 24;
    x = x - (double)n * 0x1.921FB54442D18p0;
    const double sign = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return sc_poly(x * sign, x * x, sc_tab((n & 2) != 0), n ^ 1);
}

// ---- shared 2^(i/32) table (exp2f / expf / powf) ----------------------------------------------
ODAM_HD uint64_t exp2_tab(uint32_t i) {
    static constexpr uint64_t T[32] = {
        0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51,
        0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1,
        0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
        0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585,
        0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
        0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
        0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
        0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
    return T[i & 31u];
}

// invc / logc pairs shared by log2 (powf) and ln (logf): same invc, different logc
ODAM_HD double log_invc(uint32_t i) {
    static constexpr double T[16] = {
        0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
        0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
        0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
        0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
    return T[i & 15u];
}
ODAM_HD double log2_logc(uint32_t i) {
    static constexpr double T[16] = {
        -0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2,
        -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3,  -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4,
        -0x1.a6f9db6475fcep-5, 0x0p+0,                0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,
        0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,  0x1.ce0a44eb17bccp-2};
    return T[i & 15u];
}
ODAM_HD double ln_logc(uint32_t i) {
    static constexpr double T[16] = {
        -0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
        -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
        -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
        0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
    return T[i & 15u];
}

// ---- powf for x >= 0, finite y (the sampler's  |cos|^e, e in (0.2, 1.6)) ------------------------
ODAM_HD float powf_pos(float x, float y) {
    uint32_t ix = f2u(x);
    if (ix == 0u) {                       // pow(+0, y): y > 0 -> 0, y < 0 -> inf, y == 0 -> 1
        if (y == 0.0f) return 1.0f;
        return (y > 0.0f) ? 0.0f : __builtin_inff();
    }
    if (y == 0.0f) return 1.0f;
    if (ix < 0x00800000u) {               // subnormal base
        ix = f2u(x * 0x1p23f);
        ix &= 0x7fffffffu;
        ix -= 23u << 23;
    }
    // log2(x)
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> (23 - 4)) % 16u;
    uint32_t top = tmp & 0xff800000u;
    uint32_t iz = ix - top;
    int k = (int32_t)top >> 23;
    double invc = log_invc(i);
    double logc = log2_logc(i);
    double z = (double)u2f(iz);
    double r = z * invc - 1.0;
    double y0 = logc + (double)k;
    double r2 = r * r;
    double yy = 0x1.27616c9496e0bp-2 * r + -0x1.71969a075c67ap-2;
    double p = 0x1.ec70a6ca7baddp-2 * r + -0x1.7154748bef6c8p-1;
    double r4 = r2 * r2;
    double q = 0x1.71547652ab82bp0 * r + y0;
    q = p * r2 + q;
    double logx = yy * r4 + q;
    double ylogx = (double)y * logx;
    if (((d2u(ylogx) >> 47) & 0xffffu) >= (d2u(126.0) >> 47)) {
        if (ylogx > 0x1.fffffffd1d571p+6) return __builtin_inff();
        if (ylogx <= -150.0) return 0.0f;
    }
    // exp2(ylogx)
    double kd = ylogx + 0x1.8p+47;        // shift / 32
    uint64_t ki = d2u(kd);
    kd = kd - 0x1.8p+47;
    double rr = ylogx - kd;
    uint64_t t = exp2_tab((uint32_t)(ki % 32u));
    t += ki << (52 - 5);
    double s = u2d(t);
    double zz = 0x1.c6af84b912394p-5 * rr + 0x1.ebfce50fac4f3p-3;
    double rr2 = rr * rr;
    double e = 0x1.62e42ff0c52d6p-1 * rr + 1.0;
    e = zz * rr2 + e;
    e = e * s;
    return (float)e;
}

// ---- expf, |x| < 88 (sigmoid of the two shape logits) -----------------------------------------
ODAM_HD float expf_(float x) {
    uint32_t abstop = (f2u(x) >> 20) & 0x7ffu;
    if (abstop >= ((f2u(88.0f) >> 20) & 0x7ffu)) {
        if (f2u(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8u) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_inff();
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double xd = (double)x;
    double z = 0x1.71547652b82fep+5 * xd;  // 32/ln2
    double kd = z + 0x1.8p+52;
    uint64_t ki = d2u(kd);
    kd = kd - 0x1.8p+52;
    double r = z - kd;
    uint64_t t = exp2_tab((uint32_t)(ki % 32u));
    t += ki << (52 - 5);
    double s = u2d(t);
    z = 0x1.c6af84b912394p-20 * r + 0x1.ebfce50fac4f3p-13;
    double r2 = r * r;
    double y = 0x1.62e42ff0c52d6p-6 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return (float)y;
}

// ---- logf for x > 0 finite --------------------------------------------------------------------
ODAM_HD float logf_pos(float x) {
    uint32_t ix = f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix == 0u) return -__builtin_inff();
    if (ix < 0x00800000u) {
        ix = f2u(x * 0x1p23f);
        ix -= 23u << 23;
    }
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> (23 - 4)) % 16u;
    int k = (int32_t)tmp >> 23;
    uint32_t iz = ix - (tmp & 0xff800000u);
    double invc = log_invc(i);
    double logc = ln_logc(i);
    double z = (double)u2f(iz);
    double r = z * invc - 1.0;
    double y0 = logc + (double)k * 0x1.62e42fefa39efp-1;
    double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

}  // namespace odam_math
