/*
 * oracle/sq_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's multi-view super-quadric fit
 * (likojack/ODAM).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's library; the product path
 * (odam_amd/csrc) never links, includes or calls anything in here.
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - sampler (oracle_sq_sample): bit-exact against the reference's own C++
 *     `sample_on_batch`, compiled from /root/reference into oracle/_ref/ and against
 *     committed golden vectors (tests/golden/sq_sampler.npz).
 *   - points / residual / gradient / Adam step: checked per-iteration against
 *     vectors produced by importing the reference Python (tests/golden/make_golden.py).
 *
 * Reference citations (paths relative to /root/reference):
 *   sampler ........ src/super_quadric/learnable_primitives/fast_sampler/sampling.cpp:59-215
 *                    (seed 0, buffer 201: fast_sampler/_sampler.pyx:413-441)
 *   points ......... src/super_quadric/learnable_primitives/sampling.py:558-622
 *                    src/super_quadric/sq_libs.py:556-595 (rotz, compute_ellipsoid_points), :26-27 (squashing)
 *   residual ....... src/super_quadric/sq_libs.py:395-430 (constraint_2d)
 *   prior .......... src/super_quadric/sq_libs.py:463-466
 *   optimiser ...... src/super_quadric/sq_libs.py:373-387 (torch.optim.Adam, lr .01/.1), :455-472 (loop)
 *
 * All arithmetic is IEEE binary32 with no implicit FMA contraction (build with
 * -ffp-contract=off); fused multiply-adds appear only where written as fmaf(), namely where the
 * reference's CPU kernels were found (bitwise probes against torch 2.10) to fuse them,
 * transcendental calls go to the host libm (glibc), exactly as the reference's C++ sampler
 * does; the torch tensor ops of the reference use a vector math library for most lanes, so
 * points/gradients agree with torch to a few ulp, not bitwise (DESIGN.md section "Numerics").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SQ_N 1000   /* sq_libs.py:545  EqualDistanceSamplerSQ(1000) */
#define SQ_BUF 201  /* _sampler.pyx:423 */

/* ------------------------------------------------------------------------------------------
 * std::mt19937(seed) + std::uniform_real_distribution<float>(0,1)  (sampling.cpp:18-28)
 * libstdc++: generate_canonical<float,24> draws ONE 32-bit word: r = float(u32) / 2^32,
 * clamped to nextafter(1,0) when the float conversion rounds up to 1.
 * ---------------------------------------------------------------------------------------- */
static void mt19937_uniform_floats(uint32_t seed, int n, float *out) {
    uint32_t mt[624];
    int idx = 624;
    mt[0] = seed;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    for (int j = 0; j < n; j++) {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        float r = (float)y / 4294967296.0f;
        if (r >= 1.0f) r = nextafterf(1.0f, 0.0f);
        out[j] = r;
    }
}

void oracle_sq_uniform_draws(int seed, int n, float *out) { mt19937_uniform_floats((uint32_t)seed, n, out); }

/* ------------------------------------------------------------------------------------------
 * Sampler  (sampling.cpp:59-215)
 * ---------------------------------------------------------------------------------------- */
static inline float fexp_(float x, float p) { return copysignf(powf(fabsf(x), p), x); } /* :59 */

static inline void xy_(float th, float a1, float a2, float e, float C[2]) {            /* :64 */
    C[0] = a1 * fexp_(cosf(th), e);
    C[1] = a2 * fexp_(sinf(th), e);
}

static inline float dist_(const float A[2], const float B[2]) {                        /* :69 */
    float d1 = A[0] - B[0];
    float d2 = A[1] - B[1];
    return sqrtf(d1 * d1 + d2 * d2);
}

/* equal-arc-length subdivision of the super-ellipse between theta_a and theta_b; the reference
 * runs it with an explicit LIFO stack (:76-125); every node only depends on its parent, so a
 * recursive walk fills the same buffer slots with the same values. */
static void subdivide_(float a1, float a2, float e, const float A[2], const float B[2],
                       float ta, float tb, int N, int off, float *buf) {
    if (N <= 0) return;
    float C[2];
    float th = (ta + tb) / 2;
    xy_(th, a1, a2, e, C);
    float dA = dist_(A, C);
    float dB = dist_(C, B);
    int nA = (int)roundf((dA / (dA + dB)) * (float)(N - 1));
    int nB = N - nA - 1;
    buf[nA + off] = th;
    subdivide_(a1, a2, e, A, C, ta, th, nA, off, buf);
    subdivide_(a1, a2, e, C, B, th, tb, nB, off + nA + 1, buf);
}

static void superellipse_buffer_(float a1, float a2, float e, float ta, float tb, float *buf) {
    float A[2], B[2];
    xy_(ta, a1, a2, e, A);
    xy_(tb, a1, a2, e, B);
    buf[0] = ta;
    subdivide_(a1, a2, e, A, B, ta, tb, SQ_BUF - 2, 1, buf);
    buf[SQ_BUF - 1] = tb;
}

/* a[3] = (a1,a2,a3), e[2] = (e1,e2); outputs etas[1000], omegas[1000].
 * Optional outputs (may be NULL): the two 201-entry theta buffers and the per-sample buffer
 * indices (the device kernel works on indices; the tests check them too). */
void oracle_sq_sample_ex(const float *a, const float *e, float *etas, float *omegas,
                         float *buf_eta_out, float *buf_omega_out, int *idx_eta, int *idx_omega) {
    static float draws[2 * SQ_N];
    static int have_draws = 0;
    if (!have_draws) { mt19937_uniform_floats(0u, 2 * SQ_N, draws); have_draws = 1; }

    const float pi = (float)acos(-1);  /* sampling.cpp:14 */
    const float pi_2 = pi / 2;
    float buf[SQ_BUF], cdf[SQ_BUF];

    /* etas: curve (a1, a3, e1) from +pi/2 to -pi/2  (:180-196) */
    superellipse_buffer_(a[0], a[2], e[0], pi_2, -pi_2, buf);
    if (buf_eta_out) memcpy(buf_eta_out, buf, sizeof(buf));
    {
        const float smoothing = 0.001f;
        float a1a2 = a[0] + a[1];
        cdf[0] = smoothing;
        for (int i = 1; i < SQ_BUF; i++) cdf[i] = cdf[i - 1] + smoothing + a1a2 * fexp_(cosf(buf[i]), e[0]);
        float s = cdf[SQ_BUF - 1];
        for (int i = 0; i < SQ_BUF; i++) cdf[i] /= s;
        for (int i = 0; i < SQ_N; i++) {
            float r = draws[i];
            int lo = 0, hi = SQ_BUF; /* std::lower_bound: first k with !(cdf[k] < r) */
            while (lo < hi) {
                int mid = lo + (hi - lo) / 2;
                if (cdf[mid] < r) lo = mid + 1; else hi = mid;
            }
            etas[i] = buf[lo];
            if (idx_eta) idx_eta[i] = lo;
        }
    }
    /* omegas: curve (a1, a2, e2) from +pi to -pi  (:199-212) */
    superellipse_buffer_(a[0], a[1], e[1], pi, -pi, buf);
    if (buf_omega_out) memcpy(buf_omega_out, buf, sizeof(buf));
    for (int i = 0; i < SQ_N; i++) {
        int k = (int)(draws[SQ_N + i] * (float)SQ_BUF);
        omegas[i] = buf[k];
        if (idx_omega) idx_omega[i] = k;
    }
}

void oracle_sq_sample(const float *a, const float *e, float *etas, float *omegas) {
    oracle_sq_sample_ex(a, e, etas, omegas, NULL, NULL, NULL, NULL);
}

/* ------------------------------------------------------------------------------------------
 * Parameters: p[9] = translate[3], angle, scales[3], shapes[2]   (sq_libs.py:532-544)
 * ---------------------------------------------------------------------------------------- */
static inline float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float sgn_(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

typedef struct {
    float c, s;          /* cos/sin(angle)                         sq_libs.py:566-567 */
    float a[3], e[2];    /* a = scales^2, e = squashing(shapes)    sq_libs.py:581-584 */
    float sig[2];
    float eta[SQ_N], omega[SQ_N];
    float ce[SQ_N], se[SQ_N], cw[SQ_N], sw[SQ_N];
    float f1[SQ_N], fz[SQ_N], f2[SQ_N], fs2[SQ_N];
    float raw[SQ_N][3];  /* before the 1e-6 clamp */
    float po[SQ_N][3];   /* object frame */
    float pw[SQ_N][3];   /* world frame */
} sq_state_t;

static void sq_forward_points_(const float *p, sq_state_t *S) {
    S->c = cosf(p[3]);
    S->s = sinf(p[3]);
    for (int k = 0; k < 3; k++) S->a[k] = p[4 + k] * p[4 + k];
    for (int j = 0; j < 2; j++) {
        S->sig[j] = sigmoid_(p[7 + j]);
        S->e[j] = S->sig[j] * 1.4f + 0.2f;
    }
    oracle_sq_sample(S->a, S->e, S->eta, S->omega);
    for (int i = 0; i < SQ_N; i++) {
        float eta = S->eta[i], om = S->omega[i];
        if (eta == 0.0f) eta += 1e-6f;               /* sampling.py:591-592 */
        if (om == 0.0f) om += 1e-6f;
        float ce = cosf(eta), se = sinf(eta), cw = cosf(om), sw = sinf(om);
        S->ce[i] = ce; S->se[i] = se; S->cw[i] = cw; S->sw[i] = sw;
        float f1 = sgn_(ce) * powf(fabsf(ce), S->e[0]);  /* sampling.py:605-607 */
        float fz = sgn_(se) * powf(fabsf(se), S->e[0]);
        float f2 = sgn_(cw) * powf(fabsf(cw), S->e[1]);
        float fs2 = sgn_(sw) * powf(fabsf(sw), S->e[1]);
        S->f1[i] = f1; S->fz[i] = fz; S->f2[i] = f2; S->fs2[i] = fs2;
        float x = (S->a[0] * f1) * f2;
        float y = (S->a[1] * f1) * fs2;
        float z = S->a[2] * fz;
        S->raw[i][0] = x; S->raw[i][1] = y; S->raw[i][2] = z;
        /* sampling.py:613-615  sign+ * max(|.|, 1e-6) */
        x = ((x > 0.0f) ? 1.0f : -1.0f) * fmaxf(fabsf(x), 1e-6f);
        y = ((y > 0.0f) ? 1.0f : -1.0f) * fmaxf(fabsf(y), 1e-6f);
        z = ((z > 0.0f) ? 1.0f : -1.0f) * fmaxf(fabsf(z), 1e-6f);
        S->po[i][0] = x; S->po[i][1] = y; S->po[i][2] = z;
        /* pts @ R.T + translate, R = rotz(angle)      sq_libs.py:556-575, 590-592
         * The CPU matmul accumulates k-ordered with fused multiply-add (acc = a0*b0; acc =
         * fma(a_k, b_k, acc)) -- checked bitwise against torch; the z*0 terms change nothing. */
        float wx = fmaf(y, -S->s, x * S->c);
        float wy = fmaf(y, S->c, x * S->s);
        float wz = z;
        S->pw[i][0] = wx + p[0];
        S->pw[i][1] = wy + p[1];
        S->pw[i][2] = wz + p[2];
    }
}

/* world points of a parameter vector: out[1000*3]   (compute_ellipsoid_points, sq_libs.py:577-595) */
void oracle_sq_points(const float *p, float *out) {
    sq_state_t *S = (sq_state_t *)malloc(sizeof(sq_state_t));
    sq_forward_points_(p, S);
    memcpy(out, S->pw, sizeof(S->pw));
    free(S);
}

/* ------------------------------------------------------------------------------------------
 * Residual (constraint_2d, sq_libs.py:395-430) + prior (:463-466) + closed-form gradient.
 *   P    [F][12]  row-major 3x4 projection matrices (float32 of K @ inv(T_wc)[:3,:])
 *   tgt  [F][4]   target pixel per direction d = 0:x_min 1:x_max 2:y_min 3:y_max  ( = -gt )
 *   mask [F][4]   1 where the bbox edge is a constraint, else 0
 *   prior_icov[9] row-major 3x3 inverse covariance of the class (NULL = no prior)
 *   s0[3]         initial scales (prior centre)
 * Outputs: loss2d, loss (incl. prior), grad[9], ext[F][4], win[F][4] (winner point, -1 = none).
 * Direction order in the loss follows names = [x_min, x_max, y_min, y_max] (sq_libs.py:438).
 * ---------------------------------------------------------------------------------------- */
/* Measurement hook (tests/golden/make_sq_divergence.py only): the search's results for every (view, direction) -- winner index,
 * extent, and the winner's projected q_r, q_z -- supplied from outside (the reference's own torch run) instead of computed. */
typedef struct { const int *win; const float *ext, *qr, *qz; } sq_search_t;

static void sq_loss_grad_(const float *p, const sq_state_t *S, int F, const float *P,
                          const float *tgt, const float *mask, const float *prior_icov,
                          const float *s0, int optimise_shapes,
                          float *loss2d_out, float *loss_out, float *grad, float *ext_out, int *win_out,
                          const sq_search_t *given) {
    int Pn = 1;
    while (Pn < F) Pn <<= 1;
    float (*contrib)[9] = (float (*)[9])calloc((size_t)Pn, sizeof(float[9]));   /* one row per view */
    float dirsum[4] = {0, 0, 0, 0};
    const float invF = 1.0f / (float)F;

    for (int f = 0; f < F; f++) {
        const float *M = P + 12 * f;
        float best[4] = {1000000.0f, -1000000.0f, 1000000.0f, -1000000.0f};
        int bi[4] = {-1, -1, -1, -1};
        for (int i = 0; i < (given ? 0 : SQ_N); i++) {
            const float *w = S->pw[i];
            /* pixels = homog(pts) @ M^T, k-ordered   sq_libs.py:398 */
            float qx = fmaf(w[2], M[2], fmaf(w[1], M[1], w[0] * M[0])) + M[3];
            float qy = fmaf(w[2], M[6], fmaf(w[1], M[5], w[0] * M[4])) + M[7];
            float qz = fmaf(w[2], M[10], fmaf(w[1], M[9], w[0] * M[8])) + M[11];
            if (!(qz > 0.5f)) continue;                       /* valid_pts  :399 */
            float den = fabsf(qz) + 1e-6f;                    /* :400 */
            float u = qx / den, v = qy / den;
            /* first index wins ties (torch.min/max(dim) semantics) */
            if (u < best[0]) { best[0] = u; bi[0] = i; }
            if (u > best[1]) { best[1] = u; bi[1] = i; }
            if (v < best[2]) { best[2] = v; bi[2] = i; }
            if (v > best[3]) { best[3] = v; bi[3] = i; }
        }
        float dg[4][9];
        memset(dg, 0, sizeof(dg));
        for (int d = 0; d < 4; d++) {
            float ext = given ? given->ext[4 * f + d] : best[d];
            int wi = given ? given->win[4 * f + d] : bi[d];
            if (ext_out) ext_out[4 * f + d] = ext;
            if (win_out) win_out[4 * f + d] = wi;
            float m = mask[4 * f + d];
            float diff = ext - tgt[4 * f + d];
            float l = fabsf(diff);
            if (isnan(l)) l = 0.0f;                           /* :426-427 */
            dirsum[d] += l * m;
            float *g9 = dg[d];
            if (wi < 0 || m == 0.0f || isnan(diff)) continue;
            float g = (sgn_(diff) * m) * invF;                /* d(mean |.|)/d ext */
            if (g == 0.0f) continue;
            const float *w = S->pw[wi];
            int r = (d < 2) ? 0 : 1;
            float qr = fmaf(w[2], M[4 * r + 2], fmaf(w[1], M[4 * r + 1], w[0] * M[4 * r])) + M[4 * r + 3];
            float qz = fmaf(w[2], M[10], fmaf(w[1], M[9], w[0] * M[8])) + M[11];
            if (given) { qr = given->qr[4 * f + d]; qz = given->qz[4 * f + d]; }
            float den = fabsf(qz) + 1e-6f;
            float val = qr / den;
            float gq = g / den;                               /* d val / d q_r */
            float gden = -(g * (val / den));                  /* d val / d den */
            float gqz = gden * sgn_(qz);
            float gw0 = fmaf(gqz, M[8], gq * M[4 * r + 0]);   /* grad_pixels @ M, k-ordered fma */
            float gw1 = fmaf(gqz, M[9], gq * M[4 * r + 1]);
            float gw2 = fmaf(gqz, M[10], gq * M[4 * r + 2]);
            const float *o = S->po[wi];
            const float *rw = S->raw[wi];
            /* translate */
            g9[0] = gw0; g9[1] = gw1; g9[2] = gw2;
            /* angle: d(R po)/d theta */
            g9[3] = gw0 * ((-S->s) * o[0] - S->c * o[1]) + gw1 * (S->c * o[0] - S->s * o[1]);
            /* back to the object frame */
            float gx = fmaf(gw1, S->s, gw0 * S->c);           /* grad_pw @ R */
            float gy = fmaf(gw1, S->c, gw0 * (-S->s));
            float gz = gw2;
            /* 1e-6 clamp passes the gradient only where |raw| > 1e-6 */
            if (!(fabsf(rw[0]) > 1e-6f)) gx = 0.0f;
            if (!(fabsf(rw[1]) > 1e-6f)) gy = 0.0f;
            if (!(fabsf(rw[2]) > 1e-6f)) gz = 0.0f;
            float f1 = S->f1[wi], f2 = S->f2[wi], fs2 = S->fs2[wi], fz = S->fz[wi];
            /* a = scales^2 */
            g9[4] = (gx * (f1 * f2)) * (2.0f * p[4]);
            g9[5] = (gy * (f1 * fs2)) * (2.0f * p[5]);
            g9[6] = (gz * fz) * (2.0f * p[6]);
            if (optimise_shapes) {
                float lce = logf(fabsf(S->ce[wi])), lse = logf(fabsf(S->se[wi]));
                float lcw = logf(fabsf(S->cw[wi])), lsw = logf(fabsf(S->sw[wi]));
                float ge1 = (gx * (rw[0] * lce) + gy * (rw[1] * lce)) + gz * (rw[2] * lse);
                float ge2 = gx * (rw[0] * lcw) + gy * (rw[1] * lsw);
                /* e = 1.4*sigmoid(h)+0.2 */
                g9[7] = ge1 * (1.4f * (S->sig[0] * (1.0f - S->sig[0])));
                g9[8] = ge2 * (1.4f * (S->sig[1] * (1.0f - S->sig[1])));
            }
        }
        for (int k = 0; k < 9; k++) contrib[f][k] = (dg[0][k] + dg[1][k]) + (dg[2][k] + dg[3][k]);
    }
    /* fixed-shape reduction: per view (d0+d1)+(d2+d3) (done inside the view loop), then a
     * pairwise tree over the views padded to a power of two */
    for (int stride = Pn / 2; stride >= 1; stride >>= 1)
        for (int i = 0; i < stride; i++)
            for (int k = 0; k < 9; k++) contrib[i][k] += contrib[i + stride][k];
    for (int k = 0; k < 9; k++) grad[k] = contrib[0][k];
    free(contrib);

    float loss2d = 0.0f;
    for (int d = 0; d < 4; d++) loss2d += dirsum[d] * invF;
    float loss = loss2d;
    if (prior_icov) {
        /* 20 * d^T S d, d = s0 - s   (sq_libs.py:463-466) */
        float dv[3] = {s0[0] - p[4], s0[1] - p[5], s0[2] - p[6]};
        float Sd[3], Std[3];
        for (int r = 0; r < 3; r++) {
            Sd[r] = (prior_icov[3 * r] * dv[0] + prior_icov[3 * r + 1] * dv[1]) + prior_icov[3 * r + 2] * dv[2];
            Std[r] = (prior_icov[r] * dv[0] + prior_icov[3 + r] * dv[1]) + prior_icov[6 + r] * dv[2];
        }
        float quad = (dv[0] * Sd[0] + dv[1] * Sd[1]) + dv[2] * Sd[2];
        loss += quad * 20.0f;
        for (int r = 0; r < 3; r++) grad[4 + r] += -20.0f * (Sd[r] + Std[r]);
    }
    if (loss2d_out) *loss2d_out = loss2d;
    if (loss_out) *loss_out = loss;
}

void oracle_sq_loss_grad(const float *p, int F, const float *P, const float *tgt, const float *mask,
                         const float *prior_icov, const float *s0, int optimise_shapes,
                         float *loss2d, float *loss, float *grad, float *ext, int *win) {
    sq_state_t *S = (sq_state_t *)malloc(sizeof(sq_state_t));
    sq_forward_points_(p, S);
    sq_loss_grad_(p, S, F, P, tgt, mask, prior_icov, s0, optimise_shapes, loss2d, loss, grad, ext, win, NULL);
    free(S);
}

/* ------------------------------------------------------------------------------------------
 * Measurement hooks (tests/golden/make_sq_divergence.py: WHICH part of a step makes the free-running oracle leave the
 * reference's trajectory).  The forward state of a step -- everything sq_loss_grad_ reads besides the parameters -- as one
 * flat float array, so that it can be produced either here (oracle_sq_state) or by the reference's own torch operations:
 *   [0] cos(angle) [1] sin(angle) [2..4] a [5..6] e [7..8] sigmoid(shapes)
 *   then 8 arrays of 1000: cos(eta) sin(eta) cos(omega) sin(omega) f1 fz f2 fs2
 *   then 3 arrays of 1000 x 3: raw (before the 1e-6 clamp), po (object frame), pw (world frame)
 * oracle_sq_loss_grad_state evaluates the gradient from such a state; `win` non-null: with the search's results given as well.
 * ---------------------------------------------------------------------------------------- */
#define SQ_STATE_FLOATS (9 + 8 * SQ_N + 9 * SQ_N)
int oracle_sq_state_floats(void) { return SQ_STATE_FLOATS; }

static void state_pack_(const sq_state_t *S, float *o) {
    o[0] = S->c; o[1] = S->s; memcpy(o + 2, S->a, 12); memcpy(o + 5, S->e, 8); memcpy(o + 7, S->sig, 8);
    const float *arr[8] = {S->ce, S->se, S->cw, S->sw, S->f1, S->fz, S->f2, S->fs2};
    for (int k = 0; k < 8; k++) memcpy(o + 9 + k * SQ_N, arr[k], SQ_N * 4);
    memcpy(o + 9 + 8 * SQ_N, S->raw, SQ_N * 12);
    memcpy(o + 9 + 11 * SQ_N, S->po, SQ_N * 12);
    memcpy(o + 9 + 14 * SQ_N, S->pw, SQ_N * 12);
}
static void state_unpack_(const float *o, sq_state_t *S) {
    S->c = o[0]; S->s = o[1]; memcpy(S->a, o + 2, 12); memcpy(S->e, o + 5, 8); memcpy(S->sig, o + 7, 8);
    float *arr[8] = {S->ce, S->se, S->cw, S->sw, S->f1, S->fz, S->f2, S->fs2};
    for (int k = 0; k < 8; k++) memcpy(arr[k], o + 9 + k * SQ_N, SQ_N * 4);
    memcpy(S->raw, o + 9 + 8 * SQ_N, SQ_N * 12);
    memcpy(S->po, o + 9 + 11 * SQ_N, SQ_N * 12);
    memcpy(S->pw, o + 9 + 14 * SQ_N, SQ_N * 12);
}
void oracle_sq_state(const float *p, float *state_out) {
    sq_state_t *S = (sq_state_t *)malloc(sizeof(sq_state_t));
    sq_forward_points_(p, S);
    state_pack_(S, state_out);
    free(S);
}
void oracle_sq_loss_grad_state(const float *p, const float *state, int F, const float *P, const float *tgt, const float *mask,
                               const float *prior_icov, const float *s0, int optimise_shapes, const int *win, const float *ext,
                               const float *qr, const float *qz, float *loss2d, float *loss, float *grad, float *ext_out, int *win_out) {
    sq_state_t *S = (sq_state_t *)malloc(sizeof(sq_state_t));
    state_unpack_(state, S);
    sq_search_t g = {win, ext, qr, qz};
    sq_loss_grad_(p, S, F, P, tgt, mask, prior_icov, s0, optimise_shapes, loss2d, loss, grad, ext_out, win_out, win ? &g : NULL);
    free(S);
}

/* ------------------------------------------------------------------------------------------
 * torch.optim.Adam single-tensor step, betas (0.9, 0.999), eps 1e-8   (sq_libs.py:373-387)
 *   m.lerp_(g, 1-b1); v.mul_(b2).addcmul_(g, g, 1-b2);
 *   denom = sqrt(v)/sqrt(bc2) + eps; p.addcdiv_(m, denom, -lr/bc1)
 * bias corrections are Python float64, cast to float32 where they meet a tensor.
 * state[27] = p[9], m[9], v[9]; t = 1-based step; lr_main for p[0..6], lr_shape for p[7..8].
 * ---------------------------------------------------------------------------------------- */
void oracle_adam_step(float *p, float *m, float *v, const float *g, int t, int n_opt, double lr_main,
                      double lr_shape) {
    const double b1 = 0.9, b2 = 0.999, eps = 1e-8;
    double bc1 = 1.0 - pow(b1, (double)t);
    double bc2 = 1.0 - pow(b2, (double)t);
    float bc2_sqrt = (float)sqrt(bc2);
    float w1 = (float)(1.0 - b1);
    float w2 = (float)(1.0 - b2);
    float fb2 = (float)b2;
    for (int k = 0; k < n_opt; k++) {
        double lr = (k < 7) ? lr_main : lr_shape;
        float neg_step = (float)(-(lr / bc1));
        m[k] = fmaf(w1, g[k] - m[k], m[k]);          /* lerp_: contracted in the torch CPU kernel */
        v[k] = v[k] * fb2;
        v[k] = fmaf(w2 * g[k], g[k], v[k]);          /* addcmul_: likewise */
        float denom = sqrtf(v[k]) / bc2_sqrt + (float)eps;
        p[k] = p[k] + (neg_step * m[k]) / denom;
    }
}

/* ------------------------------------------------------------------------------------------
 * SuperQuadricOptimizer.run  (sq_libs.py:432-475)
 *   representation: 0 super_quadric (9 params optimised), 1 cube / 2 quadric (7 params)
 *   traj (nullable): [n_iters][9] parameters after every step; loss_log (nullable): [n_iters] loss_2d
 * ---------------------------------------------------------------------------------------- */
void oracle_sq_fit(const float *p_init, int F, const float *P, const float *tgt, const float *mask,
                   const float *prior_icov, int representation, int n_iters, float *p_out,
                   float *traj, float *loss_log) {
    float p[9], m[9], v[9], g[9], s0[3];
    memcpy(p, p_init, sizeof(p));
    memset(m, 0, sizeof(m));
    memset(v, 0, sizeof(v));
    s0[0] = p[4]; s0[1] = p[5]; s0[2] = p[6];
    int n_opt = (representation == 0) ? 9 : 7;
    sq_state_t *S = (sq_state_t *)malloc(sizeof(sq_state_t));
    for (int it = 0; it < n_iters; it++) {
        float l2d, l;
        sq_forward_points_(p, S);
        sq_loss_grad_(p, S, F, P, tgt, mask, prior_icov, s0, n_opt == 9, &l2d, &l, g, NULL, NULL, NULL);
        oracle_adam_step(p, m, v, g, it + 1, n_opt, 0.01, 0.1);
        if (traj) memcpy(traj + 9 * it, p, sizeof(p));
        if (loss_log) loss_log[it] = l2d;
    }
    memcpy(p_out, p, sizeof(p));
    free(S);
}
