// sq_core.h -- leaf arithmetic of the super-quadric multi-view fit, shared by the gfx950 kernel
// (sq_fit.hip) and the host-side sampler entry point (odam_sq_sample).
//
// Every function states the reference lines it restates (paths relative to the upstream
// likojack/ODAM tree).  Operation order is part of the contract: binary32, no implicit FMA
// contraction (-ffp-contract=off), fmaf() exactly where written.
#pragma once
#include "sq_math.h"

namespace odam_sq {

constexpr int N_PTS = 1000;  // sq_libs.py:545
constexpr int BUF = 201;     // fast_sampler/_sampler.pyx:423
constexpr float PI_F = 3.14159274101257324f;  // float(std::acos(-1))  sampling.cpp:14

using odam_math::cosf_;
using odam_math::expf_;
using odam_math::logf_pos;
using odam_math::powf_pos;
using odam_math::sinf_;

ODAM_HD float absf(float x) { return __builtin_fabsf(x); }
ODAM_HD float sgnf(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

// sampling.cpp:59   copysign(pow(|x|, p), x)
ODAM_HD float fexp(float x, float p) { return __builtin_copysignf(powf_pos(absf(x), p), x); }

// sampling.cpp:64
ODAM_HD void xy(float th, float a1, float a2, float e, float& cx, float& cy) {
    cx = a1 * fexp(cosf_(th), e);
    cy = a2 * fexp(sinf_(th), e);
}

// sampling.cpp:69
ODAM_HD float dist(float ax, float ay, float bx, float by) {
    float d1 = ax - bx;
    float d2 = ay - by;
    return __builtin_sqrtf(d1 * d1 + d2 * d2);
}

// one pending arc of the equal-arc-length subdivision (sampling.cpp:31-56)
struct Arc {
    float ax, ay, bx, by, ta, tb;
    int n, off;
};

// sampling.cpp:96-124 given the midpoint theta and its curve point C: split count and the two child arcs.
// Returns the buffer slot the midpoint fills.
ODAM_HD int arc_split(const Arc& nd, float th, float cx, float cy, Arc& left, Arc& right) {
    float dA = dist(nd.ax, nd.ay, cx, cy);
    float dB = dist(cx, cy, nd.bx, nd.by);
    int nA = (int)__builtin_roundf((dA / (dA + dB)) * (float)(nd.n - 1));
    int nB = nd.n - nA - 1;
    left.ax = nd.ax; left.ay = nd.ay; left.bx = cx; left.by = cy;
    left.ta = nd.ta; left.tb = th; left.n = nA; left.off = nd.off;
    right.ax = cx; right.ay = cy; right.bx = nd.bx; right.by = nd.by;
    right.ta = th; right.tb = nd.tb; right.n = nB; right.off = nd.off + nA + 1;
    return nA + nd.off;
}

// sampling.cpp:92-124: midpoint, curve point, split.
ODAM_HD int arc_step(const Arc& nd, float a1, float a2, float e, float& th, Arc& left, Arc& right) {
    th = (nd.ta + nd.tb) / 2;
    float cx, cy;
    xy(th, a1, a2, e, cx, cy);
    return arc_split(nd, th, cx, cy, left, right);
}

// derived shape of a parameter vector p = translate[3], angle, scales[3], shapes[2]
struct Shape {
    float c, s;        // cos / sin(angle)             sq_libs.py:566-567
    float a[3];        // scales^2                     sq_libs.py:581
    float e[2];        // 0.2 + 1.4 sigmoid(shapes)    sq_libs.py:26-27, 583
    float sig[2];
};

ODAM_HD Shape make_shape(const float* p) {
    Shape S;
    S.c = cosf_(p[3]);
    S.s = sinf_(p[3]);
    for (int k = 0; k < 3; k++) S.a[k] = p[4 + k] * p[4 + k];
    for (int j = 0; j < 2; j++) {
        S.sig[j] = 1.0f / (1.0f + expf_(-p[7 + j]));
        S.e[j] = S.sig[j] * 1.4f + 0.2f;
    }
    return S;
}

// per buffer entry: cos, sin of the (zero-nudged) angle and their signed powers
// sampling.py:591-592 (theta == 0 -> 1e-6), :605-607 (fexp)
ODAM_HD void entry_factors(float theta, float e, float& ct, float& st, float& fc, float& fs) {
    if (theta == 0.0f) theta += 1e-6f;
    ct = cosf_(theta);
    st = sinf_(theta);
    fc = sgnf(ct) * powf_pos(absf(ct), e);
    fs = sgnf(st) * powf_pos(absf(st), e);
}

ODAM_HD float clamp_away(float x) {  // sampling.py:613-615
    return ((x > 0.0f) ? 1.0f : -1.0f) * __builtin_fmaxf(absf(x), 1e-6f);
}

// sampling.py:605-615 + sq_libs.py:590-592
ODAM_HD void surface_point(const Shape& S, const float* t, float f1, float fz, float f2, float fs2,
                           float raw[3], float po[3], float pw[3]) {
    raw[0] = (S.a[0] * f1) * f2;
    raw[1] = (S.a[1] * f1) * fs2;
    raw[2] = S.a[2] * fz;
    po[0] = clamp_away(raw[0]);
    po[1] = clamp_away(raw[1]);
    po[2] = clamp_away(raw[2]);
    float wx = __builtin_fmaf(po[1], -S.s, po[0] * S.c);
    float wy = __builtin_fmaf(po[1], S.c, po[0] * S.s);
    pw[0] = wx + t[0];
    pw[1] = wy + t[1];
    pw[2] = po[2] + t[2];
}

// one row of  homog(p) @ M^T   (sq_libs.py:398)
ODAM_HD float proj_row(float w0, float w1, float w2, float m0, float m1, float m2, float m3) {
    return __builtin_fmaf(w2, m2, __builtin_fmaf(w1, m1, w0 * m0)) + m3;
}

// gradient of  mask * |ext - tgt| / F  w.r.t. the 9 parameters through one winning point.
// (closed form of what autograd computes through sq_libs.py:395-430, sampling.py:605-615,
//  sq_libs.py:577-595).  Returns the loss term l*m.
struct Winner {
    float pw[3], po[3], raw[3];
    float f1, fz, f2, fs2;
    float ce, se, cw, sw;
};

ODAM_HD float edge_grad(int d, float ext, bool has_winner, float tgt, float m, float invF,
                        const float* M, const Shape& S, const float* p, const Winner& W,
                        bool optimise_shapes, float g9[9]) {
    for (int k = 0; k < 9; k++) g9[k] = 0.0f;
    float diff = ext - tgt;
    float l = absf(diff);
    if (l != l) l = 0.0f;  // NaN -> 0   sq_libs.py:426-427
    float lm = l * m;
    if (!has_winner || m == 0.0f || diff != diff) return lm;
    float g = (sgnf(diff) * m) * invF;
    if (g == 0.0f) return lm;
    const int r = (d < 2) ? 0 : 1;
    float qr = proj_row(W.pw[0], W.pw[1], W.pw[2], M[4 * r], M[4 * r + 1], M[4 * r + 2], M[4 * r + 3]);
    float qz = proj_row(W.pw[0], W.pw[1], W.pw[2], M[8], M[9], M[10], M[11]);
    float den = absf(qz) + 1e-6f;
    float val = qr / den;
    float gq = g / den;
    float gden = -(g * (val / den));
    float gqz = gden * sgnf(qz);
    float gw0 = __builtin_fmaf(gqz, M[8], gq * M[4 * r + 0]);
    float gw1 = __builtin_fmaf(gqz, M[9], gq * M[4 * r + 1]);
    float gw2 = __builtin_fmaf(gqz, M[10], gq * M[4 * r + 2]);
    g9[0] = gw0;
    g9[1] = gw1;
    g9[2] = gw2;
    g9[3] = gw0 * ((-S.s) * W.po[0] - S.c * W.po[1]) + gw1 * (S.c * W.po[0] - S.s * W.po[1]);
    float gx = __builtin_fmaf(gw1, S.s, gw0 * S.c);
    float gy = __builtin_fmaf(gw1, S.c, gw0 * (-S.s));
    float gz = gw2;
    if (!(absf(W.raw[0]) > 1e-6f)) gx = 0.0f;
    if (!(absf(W.raw[1]) > 1e-6f)) gy = 0.0f;
    if (!(absf(W.raw[2]) > 1e-6f)) gz = 0.0f;
    g9[4] = (gx * (W.f1 * W.f2)) * (2.0f * p[4]);
    g9[5] = (gy * (W.f1 * W.fs2)) * (2.0f * p[5]);
    g9[6] = (gz * W.fz) * (2.0f * p[6]);
    if (optimise_shapes) {
        float lce = logf_pos(absf(W.ce)), lse = logf_pos(absf(W.se));
        float lcw = logf_pos(absf(W.cw)), lsw = logf_pos(absf(W.sw));
        float ge1 = (gx * (W.raw[0] * lce) + gy * (W.raw[1] * lce)) + gz * (W.raw[2] * lse);
        float ge2 = gx * (W.raw[0] * lcw) + gy * (W.raw[1] * lsw);
        g9[7] = ge1 * (1.4f * (S.sig[0] * (1.0f - S.sig[0])));
        g9[8] = ge2 * (1.4f * (S.sig[1] * (1.0f - S.sig[1])));
    }
    return lm;
}

// scale prior 20 * d^T S d  (sq_libs.py:463-466): value and gradient w.r.t. scales
ODAM_HD float prior_term(const float* icov, const float* s0, const float* p, float gs[3]) {
    float dv[3] = {s0[0] - p[4], s0[1] - p[5], s0[2] - p[6]};
    float Sd[3], Std[3];
    for (int r = 0; r < 3; r++) {
        Sd[r] = (icov[3 * r] * dv[0] + icov[3 * r + 1] * dv[1]) + icov[3 * r + 2] * dv[2];
        Std[r] = (icov[r] * dv[0] + icov[3 + r] * dv[1]) + icov[6 + r] * dv[2];
    }
    float quad = (dv[0] * Sd[0] + dv[1] * Sd[1]) + dv[2] * Sd[2];
    for (int r = 0; r < 3; r++) gs[r] = -20.0f * (Sd[r] + Std[r]);
    return quad * 20.0f;
}

// torch.optim.Adam single-tensor step on one scalar (sq_libs.py:373-387); neg_step = -lr/bc1 and
// bc2_sqrt are computed on the host in binary64 and rounded once, as torch does.
ODAM_HD void adam_scalar(float& p, float& m, float& v, float g, float neg_step, float bc2_sqrt) {
    const float w1 = (float)(1.0 - 0.9);
    const float w2 = (float)(1.0 - 0.999);
    const float b2 = (float)0.999;
    m = __builtin_fmaf(w1, g - m, m);
    v = v * b2;
    v = __builtin_fmaf(w2 * g, g, v);
    float denom = __builtin_sqrtf(v) / bc2_sqrt + (float)1e-8;
    p = p + (neg_step * m) / denom;
}

}  // namespace odam_sq
