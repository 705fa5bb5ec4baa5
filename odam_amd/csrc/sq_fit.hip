// sq_fit.hip -- persistent per-object super-quadric multi-view fit for gfx950 (MI355X).
//
// One 1024-thread workgroup (16 wavefronts of 64) owns one object for all Adam steps; parameters,
// Adam moments, the 1000 surface points, the sampler's theta buffers / CDF and the per-view
// gradient rows live in LDS for the whole fit, so HBM sees only the view constraints
// (48 + 32 bytes per view) streamed once per step through L2 and one 36-byte result per object.
//
// Replaces (reference, likojack/ODAM): SuperQuadricOptimizer.run  src/super_quadric/sq_libs.py:432-475
// with everything it calls per step: fast_sampler/sampling.cpp:158-215 (surface sampler),
// learnable_primitives/sampling.py:558-622 (points), sq_libs.py:395-430 (projected-extent
// residual), autograd (closed form in sq_core.h::edge_grad), torch.optim.Adam (sq_libs.py:373-387).
//
// Phases of one step (barriers between them):
//   A  shape scalars (cos/sin of yaw, a = s^2, e = squashing(h))            7 lanes
//   B  equal-arc-length subdivision of the two super-ellipses, breadth first:
//      every pending arc is independent of its siblings, so one level = one parallel step
//   C  CDF terms + per-buffer-entry trig/power factors (201 + 201 + 201 lanes), then the
//      strictly sequential float32 prefix sum on one lane, then normalisation
//   D  1000 lanes: inverse-CDF lookup, surface point, rotate + translate -> LDS
//   E  views dealt to wavefronts; per view each lane projects 16 points, keeps the four
//      running extremes (value, first index), 6-step xor butterfly merges the 64 lanes;
//      winners of up to 16 views are parked one (view, direction) per lane and the closed-form
//      gradient runs on all 64 lanes at once; 2 more xor steps add the four directions
//   F  fixed-shape pairwise tree over the per-view gradient rows, prior, Adam on 9 lanes
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (division and sqrt are the
// correctly rounded forms, hipcc's default).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../include/odam_sq.h"
#include "odam_config.h"
#include "odam_err.h"
#include "sq_core.h"

namespace {
using namespace odam_sq;

#ifndef ODAM_SQ_NT
#define ODAM_SQ_NT 1024
#endif
constexpr int NT_MAX = ODAM_SQ_NT;  // threads per workgroup: 1024 (one workgroup per CU), or 512 where two per CU pay (launch_fit)
constexpr int BUFP = 208;       // padded theta-buffer length
constexpr int QCAP = 200;       // pending arcs per curve per level (sum of n over a level <= 199)
constexpr int NONE = 0x7fffffff;
constexpr int TABN = 513;        // dyadic theta table: depth 9 of the parameter bisection

struct QArc {                    // pending arc + its position in the dyadic table (-1: deeper than the table)
    Arc a;
    int qa, qb;
};

struct FitArgs {
    const float* init_params;
    const int* class_id;
    const int* view_offsets;
    const float* P;
    const float* tgt;
    const float* mask;
    const float* prior_icov;
    const float* draws;               // [1000] uniform draws for eta
    const unsigned short* omega_idx;  // [1000] buffer index of omega
    const float* adam_tab;            // [max_iters][4]: -lr/bc1 (main), -lr/bc1 (shapes), sqrt(bc2), pad
    int n_iters;
    int representation;
    int pn_views;                     // rows of the per-view gradient buffer in LDS (largest sub-tree of any workgroup)
    float* out_params;
    float* out_points;
    float* loss_log;
    float* traj;
    // view split (phase E of one object over `split` workgroups, see the kernel): exchange rows + progress flags
    int n_obj;
    int split;                        // workgroups per object (1 = off)
    int split_min_views;              // objects with fewer views stay on one workgroup
    int max_views;                    // row pitch of the exchange buffers
    unsigned long long* xroots;       // [n_iters][n_obj][split][9] sub-tree roots as {step + 1, float bits}, zeroed before launch
    float* xloss;                     // [n_iters][n_obj][max_views][4] (only with loss_log)
    unsigned* done;                   // [n_obj][split]: steps published by each workgroup, zeroed before launch
    int force_lanes_per_view;         // 0 = choose (experiments: ODAM_SQ_S)
    unsigned split_wait_ticks;        // bounded wait for a partner's rows, 100 MHz wall clock (30 us: > one slice of phase E)
    const int* order;                 // launches of more workgroups than CUs (no view split): workgroup b fits object order[b] -- longest first; or null
};

struct Lds {
    float* p;       // [12]
    float* m;       // [12]
    float* v;       // [12]
    float* s0;      // [4]
    float* icov;    // [12]
    float* shape;   // [12]  c, s, a0..2, e0..1, sig0..1
    float* misc;    // [8]   cdf sum, dirsum[4]
    int* cnt;       // [8]   3 x 2 level counters
    int* xready;    // [32]  view split: partner rows arrived this step
    float* xroot;   // [32][9] sub-tree roots of the residue classes
    float* pwx;     // [1024] x3
    float* pwy;
    float* pwz;
    unsigned short* ie;  // [1024]
    unsigned short* io;  // [1024]
    float* buf;     // [2][BUFP]
    float* cdf;     // [BUFP]
    float* term;    // [BUFP]
    float* fac;     // [8][BUFP]  ce, se, f1, fz, cw, sw, f2, fs2
    QArc* q;        // [2 curves][2][QCAP]
    float* tth;     // [2][TABN] theta, curve point x, y of every dyadic parameter position
    float* tcx;
    float* tcy;
    float* tct;     // [2][TABN] the per-entry factors of phase C at every dyadic position (entry_factors: cos, sin, signed powers of the
    float* tst;     //           zero-nudged angle), so that a buffer slot filled from the table needs no transcendental of its own
    float* tfc;
    float* tfs;
    float* tfx;     // [TABN]    curve 0: fexp(cos theta, e1) of the un-nudged angle (the CDF term's factor, sampling.cpp:141-146)
    float* rtab;    // [2][512]  split ratio dA / (dA + dB) of every dyadic node (heap order: node i of depth d, position j = 2^d - 1 + j)
    int* bufq;      // [2][BUFP] dyadic position that filled a buffer slot, -1: an arc deeper than the table did
    float* vgrad;   // [pn][9]
    float* lterm;   // [views][4], only when the loss is logged
};

// LDS regions with disjoint lifetimes share their words (round 6: 97 KB + 36 B per view -> 81 KB for any view count, which is what lets two
// 512-thread workgroups share a CU's 160 KB):
//   * the breadth-first queues of phase B2 (q: 8000 words; dead from the end of B2 to the next step's B2) | the surface points and their
//     buffer indices (4096 words; phase D to phase E);
//   * the dyadic tables of phases B1 / B2 / C (theta, curve points, per-entry factors, split ratios, slot origins: 9240 words; dead after
//     phase C) | the per-view gradient rows and the loss terms (9 words per row + 4 per logged view; phase E to phase F).
constexpr size_t LDS_TABLE_WORDS = 3 * 2 * 520 + 4 * 2 * 520 + 520 + 2 * 512 + 2 * BUFP;      // tth tcx tcy | tct tst tfc tfs | tfx | rtab | bufq
constexpr size_t LDS_QUEUE_WORDS = (size_t)2 * 2 * QCAP * 10;
static_assert(LDS_QUEUE_WORDS >= 3 * 1024 + 1024, "points + indices fit the queue region");
__host__ __device__ inline size_t lds_words(int rows, int loss_rows) {
    const size_t grad = (size_t)rows * 9 + (size_t)loss_rows * 4;
    return 12 * 6 + 8 + 8 + 32 + 288 + 2 * BUFP + 2 * BUFP + 8 * BUFP + (grad > LDS_TABLE_WORDS ? grad : LDS_TABLE_WORDS) + LDS_QUEUE_WORDS;
}

__device__ inline Lds carve(float* base, int pn, int loss_rows) {
    Lds L;
    float* w = base;
    L.p = w; w += 12;
    L.m = w; w += 12;
    L.v = w; w += 12;
    L.s0 = w; w += 12;
    L.icov = w; w += 12;
    L.shape = w; w += 12;
    L.misc = w; w += 8;
    L.cnt = (int*)w; w += 8;
    L.xready = (int*)w; w += 32;
    L.xroot = w; w += 288;
    L.buf = w; w += 2 * BUFP;
    L.cdf = w; w += BUFP;
    L.term = w; w += BUFP;
    L.fac = w; w += 8 * BUFP;
    // tables (phases B1 .. C) | gradient rows + loss terms (phases E, F)
    float* tables = w;
    L.vgrad = w;
    L.lterm = w + (size_t)pn * 9;
    L.tth = w; w += 2 * 520;
    L.tcx = w; w += 2 * 520;
    L.tcy = w; w += 2 * 520;
    L.tct = w; w += 2 * 520;
    L.tst = w; w += 2 * 520;
    L.tfc = w; w += 2 * 520;
    L.tfs = w; w += 2 * 520;
    L.tfx = w; w += 520;
    L.rtab = w; w += 2 * 512;
    L.bufq = (int*)w; w += 2 * BUFP;
    const size_t grad = (size_t)pn * 9 + (size_t)loss_rows * 4;
    w = tables + (grad > LDS_TABLE_WORDS ? grad : LDS_TABLE_WORDS);
    // phase B2's queues | points + indices (phases D, E)
    L.q = (QArc*)w;
    L.pwx = w; w += 1024;
    L.pwy = w; w += 1024;
    L.pwz = w; w += 1024;
    L.ie = (unsigned short*)w; w += 512;
    L.io = (unsigned short*)w; w += 512;
    return L;
}

__device__ inline Shape load_shape(const float* s) {
    Shape S;
    S.c = s[0]; S.s = s[1];
    S.a[0] = s[2]; S.a[1] = s[3]; S.a[2] = s[4];
    S.e[0] = s[5]; S.e[1] = s[6];
    S.sig[0] = s[7]; S.sig[1] = s[8];
    return S;
}

// lexicographic (value, first index) merge of two candidates
__device__ inline void merge_min(float& v, int& i, float ov, int oi) {
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ inline void merge_max(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

template <int NT>      // 1024: sixteen waves, one workgroup per CU; 512: eight waves at <= 128 registers, two workgroups per CU (launch_fit decides)
__global__ __launch_bounds__(NT, 4) void sq_fit_kernel(FitArgs A) {      // (second argument: waves per SIMD -- 128 registers either way)
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // View split: the per-view extent search (phase E) is pure VALU work on one CU, ~0.3 us per view per step.  With
    // few objects and many idle CUs an object is given `split` workgroups: each runs the whole (deterministic) step --
    // sampler, points, tree, Adam -- redundantly and in lock step, but only its own slice of the views in phase E; the
    // per-view gradient rows are exchanged through HBM once per step.  The exchange never blocks: rows of every step
    // have their own slot (no reuse, so a published row is final), a workgroup that does not find a partner's rows
    // after a bounded wait computes that slice itself (same arithmetic, same bits), so the result does not depend on
    // which workgroups are resident or how far apart they run, and nothing can deadlock.
    // Workgroup ids are dealt round-robin over the 8 XCDs: the workgroups of one object share an XCD (one L2).
    int obj = blockIdx.x, wj = 0;
    if (A.split > 1) {
        const int g = blockIdx.x, xcd = g & 7, slot = g >> 3;
        wj = slot % A.split;
        obj = (slot / A.split) * 8 + xcd;
        if (obj >= A.n_obj) return;
    } else if (A.order) {
        obj = A.order[blockIdx.x];
    }
    Lds L = carve(smem, A.pn_views, A.loss_log ? A.max_views : 0);

    int v0 = 0, F = 0;
    if (A.view_offsets) {
        v0 = A.view_offsets[obj];
        F = A.view_offsets[obj + 1] - v0;
    }
    const float* Pm = A.P + (size_t)12 * v0;
    const float* tgt = A.tgt + (size_t)4 * v0;
    const float* msk = A.mask + (size_t)4 * v0;
    const int cls = A.class_id ? A.class_id[obj] : -1;
    const bool use_prior = (cls >= 0) && (A.prior_icov != nullptr);
    const int n_opt = (A.representation == ODAM_SQ_SUPER_QUADRIC) ? 9 : 7;
    const bool want_loss = (A.loss_log != nullptr);
    // (more than 16 workgroups only where they still get eight views each: at 128 views 32 workgroups lose to the exchange what the scan gains,
    //  at 298 they are 3.4 us per step ahead -- profiles/r5_sq8_stamps.txt)
    const int keff = (A.split > 1 && F >= A.split_min_views) ? ((A.split > 16 && F < 8 * A.split) ? 16 : A.split) : 1;
    if (wj >= keff) return;
    const int fs = (F + keff - 1) / keff;           // views per slice
    const bool lead = (wj == 0);                    // writes the object's outputs
    unsigned absent = 0;                              // bit jj: partner jj missed a deadline once -> no more waiting for it

    if (tid < 9) {
        L.p[tid] = A.init_params[(size_t)obj * 9 + tid];
        L.m[tid] = 0.0f;
        L.v[tid] = 0.0f;
        if (use_prior) L.icov[tid] = A.prior_icov[cls * 9 + tid];
    }
    if (tid < 3) L.s0[tid] = A.init_params[(size_t)obj * 9 + 4 + tid];
    __syncthreads();

    int Pn = 1;
    while (Pn < F) Pn <<= 1;
    const float invF = 1.0f / (float)F;
    // Lanes per view in phase E (a power of two; any choice gives the same bits).  Measured on one workgroup
    // (tests/native/sweep_sq_s.py, 10..1024 views x S = 1..64): a round in which all 16 waves are busy takes
    // t = 16 + P "units" for P = ceil(1000 / S) points per lane; a first round with only w waves busy takes
    // (0.45 + 0.55 w/16) t (fewer waves per SIMD run faster, but far from proportionally), and a trailing partial
    // round of w waves adds only (w/16) t because nothing separates the rounds.  The model is within 5 % of the
    // measured step time at every point of the sweep and picks the measured optimum.
    int S = 1;
    {
        float best_cost = 3.0e38f;
        for (int c = 1; c <= 64; c <<= 1) {
            const int lanes = fs * c, full = lanes / NT, w = (lanes - full * NT + 63) / 64;
            const float t = 16.0f + (float)((N_PTS + c - 1) / c);
            const float cost = t * ((float)full + (full == 0 ? 0.45f + 0.55f * (float)w / 16.0f : (float)w / 16.0f));
            if (cost < best_cost) { best_cost = cost; S = c; }
        }
        if (A.force_lanes_per_view > 0) S = A.force_lanes_per_view;
    }

#ifdef SQ_STAMPS   // timing experiments: accumulated 100 MHz ticks per phase, written over out_points of object 0
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = wall_clock64();
#define STAMP(k) do { if (tid == 0) { const unsigned long long now_ = wall_clock64(); st_acc[k] += now_ - st_last; st_last = now_; } } while (0)
#else
#define STAMP(k) do { } while (0)
#endif
    const int tid_k = tid;
    for (int it = 0; it <= A.n_iters; ++it) {
        // Lane constants are re-derived from an opaque copy of the thread id in every step: left to itself the compiler computes every
        // lane-dependent invariant of the step body ONCE, before this loop (per-level lane masks, table addresses, ...), and then keeps
        // them in scratch -- stored once, reloaded every step by 1024 threads of up to 240 workgroups in lock step.
        int tid_i = tid_k;
        asm volatile("" : "+v"(tid_i));
        const int tid = tid_i, lane = tid & 63;
        const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
        // ---------------- A: shape scalars ------------------------------------------------------
        if (tid == 0) L.shape[0] = cosf_(L.p[3]);
        else if (tid == 1) L.shape[1] = sinf_(L.p[3]);
        else if (tid < 5) L.shape[tid] = L.p[2 + tid] * L.p[2 + tid];          // a = scales^2
        else if (tid < 7) {
            float sg = 1.0f / (1.0f + expf_(-L.p[2 + tid]));                      // shapes at p[7], p[8]
            L.shape[2 + tid] = sg;                                                // sig -> [7], [8]
            L.shape[tid] = sg * 1.4f + 0.2f;                                      // e   -> [5], [6]
        }
        __syncthreads();
        STAMP(0);
        // (the derived shape and the parameters are read from LDS where a phase needs them -- broadcast reads -- instead of sitting in 18
        //  registers from here to the end of the step: those were spilled around every phase that inlines a transcendental)

        // ---------------- B: arc subdivision ------------------------------------------------------
        // curve 0: eta  (a1, a3, e1) from +pi/2 to -pi/2;  curve 1: omega (a1, a2, e2) from +pi to -pi.
        // The reference bisects the PARAMETER interval at every node (sampling.cpp:92); only the slot counts
        // depend on the arc lengths.  So the theta of every node down to depth 9 is known up front:
        //   B1  all 2 x 513 dyadic thetas and their curve points (the trig/pow work) in ONE parallel pass,
        //   B2  the breadth-first slot allocation then needs two distances, a division and a round per node;
        //       nodes deeper than the table (rare, strongly unbalanced curves) evaluate their point on the fly.
        // 2 x 512 entries in one pass of the 1024 threads; the far end point of each curve (theta = -ta) is the mirror
        // image of the near one -- cosf_ is even, sinf_ and fexp are odd, exactly -- so thread q = 0 writes both.
        static_assert(TABN == 513 && NT >= 64, "table pass layout");
        for (int w = tid; w < 2 * (TABN - 1); w += NT) {
            const int c = w / (TABN - 1), q = w - c * (TABN - 1);
            const float a1 = L.shape[2], a2 = c ? L.shape[3] : L.shape[4], e = L.shape[5 + c];
            const float ta = c ? PI_F : PI_F / 2;
            float lo = ta, hi = -ta, th;
            if (q == 0) th = lo;
            else {
                int lq = 0, hq = TABN - 1;
                for (;;) {
                    const int mq = (lq + hq) >> 1;
                    th = (lo + hi) / 2;
                    if (q == mq) break;
                    if (q < mq) { hq = mq; hi = th; } else { lq = mq; lo = th; }
                }
            }
            // xy(th, a1, a2, e) with its parts kept: the same cosine, sine and powers are the per-entry factors of phase C for a buffer
            // slot that this position fills (entry_factors: sgn(x) |x|^e of the ZERO-NUDGED angle).  The one position with theta == 0 (the
            // midpoint of either curve) takes its factors from 1e-6 and its curve point in closed form -- cos 0 = 1, sin 0 = 0, and
            // powf_pos(1, e) = 1, powf_pos(0, e) = 0 exactly (glibc's table has the entry {1, 0}) -- so no lane evaluates anything twice
            const bool z0 = th == 0.0f;
            const float thf = z0 ? 1e-6f : th;
            const int t = c * TABN + q;
            // one transcendental at a time, every result stored as soon as it exists (the four inlined routines interleaved keep ~40 more
            // registers alive than the 128 a 1024-thread workgroup has: their spill traffic of 16 partner workgroups meets in one XCD's L2)
            L.tth[t] = th;
            const float ect = cosf_(thf);
            L.tct[t] = ect;
            __builtin_amdgcn_sched_barrier(0);
            const float pc = powf_pos(absf(ect), e);
            const float fxc = z0 ? 1.0f : __builtin_copysignf(pc, ect);      // fexp(cos th)
            L.tfc[t] = sgnf(ect) * pc;
            L.tcx[t] = a1 * fxc;
            if (c == 0) L.tfx[q] = fxc;
            __builtin_amdgcn_sched_barrier(0);
            const float est = sinf_(thf);
            L.tst[t] = est;
            __builtin_amdgcn_sched_barrier(0);
            const float ps = powf_pos(absf(est), e);
            const float fxs = z0 ? 0.0f : __builtin_copysignf(ps, est);      // fexp(sin th)
            L.tfs[t] = sgnf(est) * ps;
            L.tcy[t] = a2 * fxs;
            if (q == 0) {      // the far end (theta = -ta): cosf_ is even, sinf_ and the signed powers are odd, exactly (read back: no register kept for it)
                const int t1 = t + TABN - 1;
                L.tth[t1] = -ta; L.tcx[t1] = L.tcx[t]; L.tcy[t1] = -L.tcy[t];
                L.tct[t1] = L.tct[t]; L.tst[t1] = -L.tst[t]; L.tfc[t1] = L.tfc[t]; L.tfs[t1] = -L.tfs[t];
                if (c == 0) L.tfx[TABN - 1] = L.tfx[0];
            }
        }
        if (tid < 6) L.cnt[tid] = 0;
        if (tid < 2) {   // end points of the two buffers
            const float ta = tid ? PI_F : PI_F / 2;
            L.buf[tid * BUFP] = ta;
            L.buf[tid * BUFP + BUF - 1] = -ta;
            L.bufq[tid * BUFP] = 0;
            L.bufq[tid * BUFP + BUF - 1] = TABN - 1;
        }
        __syncthreads();
        STAMP(1);
        // B2a: the split ratio dA / (dA + dB) of EVERY dyadic node down to depth 8 (2 x 511 of them, one per thread): it depends on the
        // three table points only, not on how many slots reach the node (arc_split: the first three statements).
        for (int w = tid; w < 2 * 511; w += NT) {
            const int c = w / 511, i = w - c * 511;
            const int d = 31 - __builtin_clz(i + 1), j = i + 1 - (1 << d);
            const int qa = j << (9 - d), qb = qa + (1 << (9 - d)), qm = (qa + qb) >> 1;
            const int t0 = c * TABN;
            const float dA = dist(L.tcx[t0 + qa], L.tcy[t0 + qa], L.tcx[t0 + qm], L.tcy[t0 + qm]);
            const float dB = dist(L.tcx[t0 + qm], L.tcy[t0 + qm], L.tcx[t0 + qb], L.tcy[t0 + qb]);
            L.rtab[c * 512 + i] = dA / (dA + dB);
        }
        __syncthreads();
#ifndef SQ_ABLATE_B
        // B2b: slot counts down the dyadic tree.  Round 4 walked the levels with a queue in LDS (a node's record, its table points, an
        // atomic per child: four dependent LDS round trips per level, 6.9 us per step).  The tree has a FIXED shape: node j of depth d
        // covers table positions [j 2^(9-d), (j + 1) 2^(9-d)], so lane j of one wavefront per curve owns node j of every depth d <= 6,
        // hands (n, offset) to its children in lanes 2 j, 2 j + 1 with ONE shuffle per level, and walks depths 7 and 8 -- the four
        // grandchildren of its depth-6 node -- alone.  Per node: nA = round(ratio (n - 1)), the slot nA + offset receives the node's
        // midpoint theta, children (nA, offset) and (n - nA - 1, offset + nA + 1): arc_split's arithmetic on the same values, so the
        // buffers come out bit for bit as before.  Arcs that still hold slots below depth 8 (strongly unbalanced curves) go to the
        // queue and the level loop of round 4 (on-the-fly midpoints), which runs only then.
        if (wid < 2) {
            const int c = wid;
            const int t0 = c * TABN;
            QArc* q0 = &L.q[(c * 2 + 0) * QCAP];
            // every ratio and midpoint theta this lane can need, requested up front: node (d, lane) for d <= 6, its two depth-7 and four
            // depth-8 descendants (one LDS round trip instead of one per level)
            float rr[7], tt[7];
#pragma unroll
            for (int d = 0; d <= 6; d++) {
                const int j = (lane < (1 << d)) ? lane : 0;
                rr[d] = L.rtab[c * 512 + (1 << d) - 1 + j];
                tt[d] = L.tth[t0 + (j << (9 - d)) + (1 << (8 - d))];
            }
            auto visit = [&](float r, float th, int qm, int n, int off, int& nl, int& ol, int& nr, int& orr) {
                // a node holding n slots from offset off: fills its slot, returns the children's (n, offset); n <= 0: nothing
                nl = 0; nr = 0; ol = off; orr = off;
                if (n <= 0) return;
                const int nA = (int)__builtin_roundf(r * (float)(n - 1));
                L.buf[c * BUFP + nA + off] = th;
                L.bufq[c * BUFP + nA + off] = qm;
                nl = nA; ol = off; nr = n - nA - 1; orr = off + nA + 1;
            };
            int n = (lane == 0) ? BUF - 2 : 0, off = 1;
#pragma unroll
            for (int d = 0; d <= 6; d++) {
                int nl, ol, nr, orr;
                visit(rr[d], tt[d], (lane << (9 - d)) + (1 << (8 - d)), (lane < (1 << d)) ? n : 0, off, nl, ol, nr, orr);
                if (d < 6) {      // children 2 j, 2 j + 1 read their parent's results from lane j: one shuffle (n, offset <= 200: a byte each)
                    const int pk = __shfl(nl | (ol << 8) | (nr << 16) | (orr << 24), lane >> 1);
                    n = ((lane & 1) ? (pk >> 16) : pk) & 0xff;
                    off = ((lane & 1) ? (pk >> 24) : (pk >> 8)) & 0xff;
                } else {          // depths 7 and 8 of this lane's depth-6 node, in the lane (their ratios and thetas: a second round trip)
                    float r7[2], t7[2], r8[4], t8[4];
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        r7[h] = L.rtab[c * 512 + 127 + 2 * lane + h];
                        t7[h] = L.tth[t0 + ((2 * lane + h) << 2) + 2];
                    }
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        r8[g] = L.rtab[c * 512 + 255 + 4 * lane + g];
                        t8[g] = L.tth[t0 + ((4 * lane + g) << 1) + 1];
                    }
                    int n7[2] = {nl, nr}, o7[2] = {ol, orr};
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        int n8l, o8l, n8r, o8r;
                        visit(r7[h], t7[h], ((2 * lane + h) << 2) + 2, n7[h], o7[h], n8l, o8l, n8r, o8r);
                        int n8[2] = {n8l, n8r}, o8[2] = {o8l, o8r};
#pragma unroll
                        for (int g = 0; g < 2; g++) {
                            int n9l, o9l, n9r, o9r;
                            const int j8 = 4 * lane + 2 * h + g;
                            visit(r8[2 * h + g], t8[2 * h + g], (j8 << 1) + 1, n8[g], o8[g], n9l, o9l, n9r, o9r);
                            // depth 9: intervals of length 1, no midpoint in the table -- what is left goes to the queue
                            const int nn[2] = {n9l, n9r}, oo[2] = {o9l, o9r};
#pragma unroll
                            for (int k = 0; k < 2; k++) {
                                if (nn[k] > 0) {
                                    const int qa = 2 * j8 + k, qb = qa + 1;
                                    QArc o;
                                    o.a.ax = L.tcx[t0 + qa]; o.a.ay = L.tcy[t0 + qa]; o.a.ta = L.tth[t0 + qa];
                                    o.a.bx = L.tcx[t0 + qb]; o.a.by = L.tcy[t0 + qb]; o.a.tb = L.tth[t0 + qb];
                                    o.a.n = nn[k]; o.a.off = oo[k]; o.qa = -1; o.qb = -1;
                                    q0[atomicAdd(&L.cnt[0 * 2 + c], 1)] = o;
                                }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // the arcs below the table, level by level as in round 4 (no arc: the loop ends at once)
            for (int lvl = 0; lvl < BUF; ++lvl) {
                const int cur = lvl % 3, nxt = (lvl + 1) % 3, clr = (lvl + 2) % 3;
                const int nq = L.cnt[cur * 2 + c];
                if (nq == 0) break;
                if (lane == 0) L.cnt[clr * 2 + c] = 0;
                for (int i = lane; i < nq; i += 64) {
                    QArc nd = L.q[(c * 2 + (lvl & 1)) * QCAP + i];
                    Arc lf, rt;
                    float th;
                    const float a1 = L.shape[2], a2 = c ? L.shape[3] : L.shape[4], e = L.shape[5 + c];
                    const int slot = arc_step(nd.a, a1, a2, e, th, lf, rt);
                    L.buf[c * BUFP + slot] = th;
                    L.bufq[c * BUFP + slot] = -1;
                    QArc* qn = &L.q[(c * 2 + ((lvl + 1) & 1)) * QCAP];
                    if (lf.n > 0) { QArc o; o.a = lf; o.qa = -1; o.qb = -1; qn[atomicAdd(&L.cnt[nxt * 2 + c], 1)] = o; }
                    if (rt.n > 0) { QArc o; o.a = rt; o.qa = -1; o.qb = -1; qn[atomicAdd(&L.cnt[nxt * 2 + c], 1)] = o; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
        }
        __syncthreads();
#endif
        STAMP(2);

        // ---------------- C: CDF and per-entry factors ----------------------------------------
        for (int w = tid; w < 768; w += NT) {
            const int grp = w >> 8, k = w & 255;
            if (k >= BUF) continue;
            if (grp == 0) {
                if (k >= 1) {
                    const int q = L.bufq[k];
                    L.term[k] = (L.shape[2] + L.shape[3]) * (q >= 0 ? L.tfx[q] : fexp(cosf_(L.buf[k]), L.shape[5]));
                }
            } else {
                const int c = grp - 1;
                const int q = L.bufq[c * BUFP + k];
                float ct, st, fc, fs;
                if (q >= 0) { ct = L.tct[c * TABN + q]; st = L.tst[c * TABN + q]; fc = L.tfc[c * TABN + q]; fs = L.tfs[c * TABN + q]; }
                else entry_factors(L.buf[c * BUFP + k], L.shape[5 + c], ct, st, fc, fs);
                L.fac[(4 * c + 0) * BUFP + k] = ct; L.fac[(4 * c + 1) * BUFP + k] = st;
                L.fac[(4 * c + 2) * BUFP + k] = fc; L.fac[(4 * c + 3) * BUFP + k] = fs;
            }
        }
        __syncthreads();
#ifndef SQ_ABLATE_SCAN
        if (wid == 0) {  // sampling.cpp:141-146: strictly ordered float32 running sum
            // The chain of 2 x 200 dependent additions is what it is; what can go is the LDS round trip per element:
            // the terms sit in registers (lane l holds entries l, l + 64, ...), every step broadcasts one of them
            // with v_readlane, all 64 lanes carry the running sum and lane (i & 63) keeps entry i of the CDF.
            float t[4], cd[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = lane + 64 * j;
                t[j] = (i >= 1 && i < BUF) ? L.term[i] : 0.0f;
                cd[j] = 0.0f;
            }
            float c = 0.001f;
            if (lane == 0) cd[0] = c;
#pragma unroll
            for (int i = 1; i < BUF; i++) {
                const float ti = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t[i >> 6]), i & 63));
                c = c + 0.001f + ti;
                if (lane == (i & 63)) cd[i >> 6] = c;
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (lane + 64 * j < BUF) L.cdf[lane + 64 * j] = cd[j];
            if (lane == 0) L.misc[0] = c;
        }
#endif
        __syncthreads();
        if (tid < BUF) L.cdf[tid] = L.cdf[tid] / L.misc[0];
        __syncthreads();

        STAMP(3);
        // ---------------- D: inverse-CDF lookup + surface points --------------------------------
        {
        const Shape S_ = load_shape(L.shape);
        float p[9];
#pragma unroll
        for (int k = 0; k < 9; k++) p[k] = L.p[k];
        for (int i = tid; i < N_PTS; i += NT) {
            const float r = A.draws[i];
            int lo = 0, hi = BUF;  // std::lower_bound
            while (lo < hi) {
                const int mid = lo + ((hi - lo) >> 1);
                if (L.cdf[mid] < r) lo = mid + 1; else hi = mid;
            }
            const int ke = lo, ko = A.omega_idx[i];
            float raw[3], po[3], pw[3];
            surface_point(S_, p, L.fac[2 * BUFP + ke], L.fac[3 * BUFP + ke], L.fac[6 * BUFP + ko],
                          L.fac[7 * BUFP + ko], raw, po, pw);
            L.pwx[i] = pw[0]; L.pwy[i] = pw[1]; L.pwz[i] = pw[2];
            L.ie[i] = (unsigned short)ke;
            L.io[i] = (unsigned short)ko;
        }
        }
        __syncthreads();

        if (it == A.n_iters) {
            if (A.out_points && lead) {
                for (int i = tid; i < N_PTS; i += NT) {
                    float* o = A.out_points + ((size_t)obj * N_PTS + i) * 3;
                    o[0] = L.pwx[i]; o[1] = L.pwy[i]; o[2] = L.pwz[i];
                }
            }
            break;
        }

        STAMP(4);
        // ---------------- E: projected extents + per-view gradient rows -----------------------
        // A view is owned by S consecutive lanes (S = largest power of two <= NT / F, at most one wavefront):
        // each walks every S-th surface point keeping its four running extremes (value, first index), an
        // S-lane xor butterfly merges them, then the closed-form gradient of the view's four edges runs on
        // those same lanes and is added in the order (d0 + d1) + (d2 + d3).
        // views f = res, res + keff, res + 2 keff, ... (cnt of them); row m of L.vgrad receives view res + keff * m
        auto views = [&](int res, int cnt) {
        const Shape S_ = load_shape(L.shape);
        float p[9];
#pragma unroll
        for (int k = 0; k < 9; k++) p[k] = L.p[k];
        for (int v0 = 0; v0 < cnt; v0 += NT / S) {
            const int mrow = v0 + tid / S;
            const int f = res + keff * mrow;
            const int sub = tid & (S - 1);
            const bool vf = mrow < cnt;
            float Mv[12];
#pragma unroll
            for (int k = 0; k < 12; k++) Mv[k] = vf ? Pm[(size_t)12 * f + k] : 0.0f;
            float bv0 = 1000000.0f, bv1 = -1000000.0f, bv2 = 1000000.0f, bv3 = -1000000.0f;
            int bi0 = NONE, bi1 = NONE, bi2 = NONE, bi3 = NONE;
            if (vf) {
                // Phase E is bound by vector ISSUE (16 waves x ceil(1000 / S) points x ~55 instructions on one CU), so the scan is written for few
                // instructions per point: no branch around the projection (a point behind the camera gets NaN coordinates, every comparison below
                // is then false), and the two IEEE divisions by the same denominator share their reciprocal.  div2 is the compiler's own fp32
                // division sequence (v_rcp + one Newton step, quotient + two residual corrections, all fused) WITHOUT v_div_scale / v_div_fixup:
                // those rescale operands whose quotient or residuals would leave the normal range, i.e. never for a depth in [0.5, 2^100) and a
                // numerator that is 0 or >= 2^-100 in magnitude -- there every intermediate of the plain sequence is the scaled sequence's times
                // a power of two, so the result is the correctly rounded quotient, bit for bit.  The smallest |numerator| the lane saw is kept
                // (one v_min3); a view in which it ever falls below 2^-100 (or is exactly zero: never measured) is scanned again with the
                // compiler's division.
                auto scan = [&](auto exact) {
                    float mn = 3.0e38f;
                    bv0 = 1000000.0f; bv1 = -1000000.0f; bv2 = 1000000.0f; bv3 = -1000000.0f;
                    bi0 = NONE; bi1 = NONE; bi2 = NONE; bi3 = NONE;
#pragma unroll 2
                    for (int i = sub; i < N_PTS; i += S) {
                        const float wx = L.pwx[i], wy = L.pwy[i], wz = L.pwz[i];
                        const float qz = proj_row(wx, wy, wz, Mv[8], Mv[9], Mv[10], Mv[11]);
                        const float qx = proj_row(wx, wy, wz, Mv[0], Mv[1], Mv[2], Mv[3]);
                        const float qy = proj_row(wx, wy, wz, Mv[4], Mv[5], Mv[6], Mv[7]);
                        const float den = absf(qz) + 1e-6f;
                        float u, v;
                        if constexpr (decltype(exact)::value) {
                            u = qx / den; v = qy / den;
                        } else {
                            float r = __builtin_amdgcn_rcpf(den);
                            r = __builtin_fmaf(__builtin_fmaf(-den, r, 1.0f), r, r);
                            float a = qx * r, b = qy * r;
                            a = __builtin_fmaf(__builtin_fmaf(-den, a, qx), r, a);
                            b = __builtin_fmaf(__builtin_fmaf(-den, b, qy), r, b);
                            u = __builtin_fmaf(__builtin_fmaf(-den, a, qx), r, a);
                            v = __builtin_fmaf(__builtin_fmaf(-den, b, qy), r, b);
                            mn = __builtin_fminf(mn, __builtin_fminf(absf(qx), absf(qy)));
                        }
                        const bool front = qz > 0.5f;
                        u = front ? u : __builtin_nanf("");
                        v = front ? v : __builtin_nanf("");
                        if (u < bv0) { bv0 = u; bi0 = i; }
                        if (u > bv1) { bv1 = u; bi1 = i; }
                        if (v < bv2) { bv2 = v; bi2 = i; }
                        if (v > bv3) { bv3 = v; bi3 = i; }
                    }
                    return mn;
                };
                const float mn = scan(std::false_type{});
                if (__builtin_expect(mn < 0x1p-100f, 0)) scan(std::true_type{});
            }
            for (int sft = S >> 1; sft >= 1; sft >>= 1) {
                merge_min(bv0, bi0, __shfl_xor(bv0, sft), __shfl_xor(bi0, sft));
                merge_max(bv1, bi1, __shfl_xor(bv1, sft), __shfl_xor(bi1, sft));
                merge_min(bv2, bi2, __shfl_xor(bv2, sft), __shfl_xor(bi2, sft));
                merge_max(bv3, bi3, __shfl_xor(bv3, sft), __shfl_xor(bi3, sft));
            }
            // directions owned by this lane: S >= 4 -> {sub} for sub < 4; S == 2 -> {2 sub, 2 sub + 1}; S == 1 -> all
            const int nd = (S >= 4) ? ((sub < 4) ? 1 : 0) : (4 / S);
            const int d0 = (S >= 4) ? sub : sub * nd;
            float g9[9];
#pragma unroll
            for (int k = 0; k < 9; k++) g9[k] = 0.0f;
            float gp[9];   // partial of the first pair when one lane owns all four directions
#pragma unroll
            for (int k = 0; k < 9; k++) gp[k] = 0.0f;
            for (int dd = 0; dd < nd; dd++) {
                const int d = d0 + dd;
                float e9[9];
                float lm = 0.0f;
                if (vf) {
                    const float ext = (d == 0) ? bv0 : (d == 1) ? bv1 : (d == 2) ? bv2 : bv3;
                    const int win = (d == 0) ? bi0 : (d == 1) ? bi1 : (d == 2) ? bi2 : bi3;
                    const bool has = (win != NONE);
                    Winner W;
                    if (has) {
                        const int ke = L.ie[win], ko = L.io[win];
                        W.ce = L.fac[0 * BUFP + ke]; W.se = L.fac[1 * BUFP + ke];
                        W.f1 = L.fac[2 * BUFP + ke]; W.fz = L.fac[3 * BUFP + ke];
                        W.cw = L.fac[4 * BUFP + ko]; W.sw = L.fac[5 * BUFP + ko];
                        W.f2 = L.fac[6 * BUFP + ko]; W.fs2 = L.fac[7 * BUFP + ko];
                        surface_point(S_, p, W.f1, W.fz, W.f2, W.fs2, W.raw, W.po, W.pw);
                    } else {
                        W = Winner{};
                    }
                    lm = edge_grad(d, ext, has, tgt[4 * f + d], msk[4 * f + d], invF, Mv, S_, p, W, n_opt == 9, e9);
                    if (want_loss) L.lterm[f * 4 + d] = lm;
                } else {
#pragma unroll
                    for (int k = 0; k < 9; k++) e9[k] = 0.0f;
                }
                // within-lane order: (d0 + d1) [+ (d2 + d3)]
                if (dd == 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) g9[k] = e9[k];
                } else if (dd == 1) {
#pragma unroll
                    for (int k = 0; k < 9; k++) g9[k] = g9[k] + e9[k];
                } else if (dd == 2) {
#pragma unroll
                    for (int k = 0; k < 9; k++) gp[k] = e9[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 9; k++) g9[k] = g9[k] + (gp[k] + e9[k]);
                }
            }
            if (S >= 4) {
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    g9[k] = g9[k] + __shfl_xor(g9[k], 1);   // (d0+d1), (d2+d3)
                    g9[k] = g9[k] + __shfl_xor(g9[k], 2);   // (d0+d1)+(d2+d3)
                }
            } else if (S == 2) {
#pragma unroll
                for (int k = 0; k < 9; k++) g9[k] = g9[k] + __shfl_xor(g9[k], 1);
            }
            if (vf && sub == 0) {
#pragma unroll
                for (int k = 0; k < 9; k++) L.vgrad[mrow * 9 + k] = g9[k];
            }
        }
        };
        // The reference sums the per-view rows with a stride-halving tree over the view index (padded to Pn).  After
        // its levels with stride >= keff, position j holds exactly the views f = j (mod keff), combined by the same
        // stride-halving tree over THAT subsequence -- so the workgroup that owns residue class j can finish its
        // sub-tree alone and only the keff sub-tree roots (9 floats each) travel; the last log2(keff) levels are
        // repeated by everyone.  Bit-identical to the single-workgroup tree for every keff.
        const int Pk = Pn / keff;
        auto slice_root = [&](int res) {      // rows + sub-tree of one residue class -> L.xroot[res]
            const int cnt = res < F ? (F - res + keff - 1) / keff : 0;
            views(res, cnt);
            for (int idx = cnt * 9 + tid; idx < Pk * 9; idx += NT) L.vgrad[idx] = 0.0f;
            __syncthreads();
            for (int stride = Pk >> 1; stride >= 1; stride >>= 1) {
                for (int idx = tid; idx < stride * 9; idx += NT) L.vgrad[idx] += L.vgrad[idx + stride * 9];
                __syncthreads();
            }
            if (tid < 9) L.xroot[res * 9 + tid] = L.vgrad[tid];
            __syncthreads();
        };
        slice_root(wj);
        STAMP(7);
        if (keff > 1) {
            // Publish my root (and, for the loss log, my views' loss terms) and collect the others, one wavefront per
            // partner: wavefront wj stores and raises the flag (release), wavefront jj != wj waits for partner jj's
            // flag (acquire, bounded).  Roots that did not arrive are computed here.
            // A root travels as nine 64-bit words {step + 1, float bits}, each written and polled with one relaxed
            // device-scope atomic: the tag arrives with the payload, so there is no separate flag, no cache write-back
            // on the writer's side and no invalidate on the reader's (the buffer is zeroed before the launch and every
            // step has its own slot).  The loss terms (only when the loss is logged) use a fenced flag.
            const size_t slot = (size_t)it * A.n_obj + obj;
            unsigned long long* xq = A.xroots + slot * (size_t)(A.split * 9);
            float* xl = want_loss ? A.xloss + slot * (size_t)A.max_views * 4 : nullptr;
            const unsigned long long tag = (unsigned long long)(it + 1) << 32;
            // wavefront w looks after residue classes w and w + 16 (a split of up to 32): its own class is published, a partner's collected.
            // The own class goes FIRST (pass 0): a wavefront that waited for partner w before publishing class w + 16 would hand a late
            // partner's delay on to everyone waiting for w + 16
            for (int pass = 0; pass < 2; pass++)
            for (int jj = wid; jj < keff; jj += NT / 64) {
            if ((jj == wj) != (pass == 0)) continue;
            if (jj == wj) {
                // (the partners of an object share an XCD -- see the workgroup-id mapping above --, so the root only has to reach that XCD's
                //  L2: a workgroup-scope store, read back below by loads that miss the vector L1 (sc0 sc1); tests/native/xcd_barrier_probe.hip)
                if (lane < 9)
                    __hip_atomic_store(&xq[wj * 9 + lane], tag | (unsigned long long)__float_as_uint(L.xroot[wj * 9 + lane]),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (want_loss) {
                    for (int m = lane; m * keff + wj < F; m += 64) {
                        const int f = m * keff + wj;
#pragma unroll
                        for (int d = 0; d < 4; d++) xl[f * 4 + d] = L.lterm[f * 4 + d];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    if (lane == 0)
                        __hip_atomic_store(&A.done[obj * A.split + wj], (unsigned)(it + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                unsigned long long v = 0;
                bool ok = lane >= 9;
                auto peek = [&](const unsigned long long* q) {
                    unsigned long long r;
                    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(q) : "memory");
                    return r;
                };
                if (!ok) {
                    v = peek(&xq[jj * 9 + lane]);
                    ok = (v >> 32) == (unsigned long long)(it + 1);
                }
                int ready = __all(ok);
                if (!ready && !((absent >> jj) & 1)) {
                    const unsigned long long t0 = wall_clock64();   // 100 MHz
                    while (!ready && wall_clock64() - t0 < A.split_wait_ticks) {
                        __builtin_amdgcn_s_sleep(1);
                        if (lane < 9 && !ok) {
                            v = peek(&xq[jj * 9 + lane]);
                            ok = (v >> 32) == (unsigned long long)(it + 1);
                        }
                        ready = __all(ok);
                    }
                }
                if (ready && want_loss && lead) {   // the partner's loss terms: published before its `done` flag
                    int lr = 0;
                    if (lane == 0) {
                        const unsigned* flag = &A.done[obj * A.split + jj];
                        const unsigned long long t0 = wall_clock64();
                        do {
                            lr = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(it + 1);
                        } while (!lr && wall_clock64() - t0 < A.split_wait_ticks);
                    }
                    ready = __shfl(lr, 0);
                    if (ready) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        for (int m = lane; m * keff + jj < F; m += 64) {
                            const int f = m * keff + jj;
#pragma unroll
                            for (int d = 0; d < 4; d++) L.lterm[f * 4 + d] = xl[f * 4 + d];
                        }
                    }
                }
                if (ready && lane < 9) L.xroot[jj * 9 + lane] = __uint_as_float((unsigned)(v & 0xffffffffull));
                if (lane == 0) L.xready[jj] = ready;
            }
            }
            __syncthreads();
            for (int jj = 0; jj < keff; jj++) {
                if (jj == wj || L.xready[jj]) continue;      // uniform: read from LDS after the barrier
                absent |= 1u << jj;                          // no more waiting for this partner
                slice_root(jj);
            }
        }

        STAMP(5);
        // ---------------- F: last tree levels, prior, Adam ---------------------------------------
        if (want_loss) {
            if (tid >= 64 && tid < 68) {
                const int dd = tid - 64;
                float s = 0.0f;
                for (int f = 0; f < F; f++) s += L.lterm[f * 4 + dd];
                L.misc[1 + dd] = s;
            }
            __syncthreads();   // the sums are read by thread 0 below
        }
        if (tid < 9) {
            for (int stride = keff >> 1; stride >= 1; stride >>= 1)       // 9 lanes of one wavefront: in order
                for (int i = 0; i < stride; i++) L.xroot[i * 9 + tid] += L.xroot[(i + stride) * 9 + tid];
        }
        if (tid < 9) {
            float g = (F > 0) ? L.xroot[tid] : 0.0f;
            if (use_prior && tid >= 4 && tid < 7) {
                float gs[3];
                prior_term(L.icov, L.s0, L.p, gs);
                g += gs[tid - 4];
            }
            if (tid < n_opt) {
                const float* tab = A.adam_tab + 4 * it;
                float pp = L.p[tid], mm = L.m[tid], vv = L.v[tid];
                adam_scalar(pp, mm, vv, g, (tid < 7) ? tab[0] : tab[1], tab[2]);
                L.p[tid] = pp; L.m[tid] = mm; L.v[tid] = vv;
            }
            if (A.traj && lead) A.traj[((size_t)obj * A.n_iters + it) * 9 + tid] = L.p[tid];
        }
        if (want_loss && tid == 0 && lead) {
            float l2d = 0.0f;
            for (int dd = 0; dd < 4; dd++) l2d += L.misc[1 + dd] * invF;
            A.loss_log[(size_t)obj * A.n_iters + it] = l2d;
        }
        __syncthreads();
        STAMP(6);
    }
#ifdef SQ_STAMPS
    if (tid == 0 && obj == 0 && lead && A.out_points)
        for (int k = 0; k < 8; k++) A.out_points[k] = (float)st_acc[k];
#endif
    if (A.out_params && tid < 9 && lead) A.out_params[(size_t)obj * 9 + tid] = L.p[tid];
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct odam_sq_ctx {
    int max_iters;
    float* d_draws;
    unsigned short* d_omega_idx;
    float* d_adam;
    float* d_dummy_params;
    // view-split exchange (grow-only; a context serves one launch at a time, like a stream)
    float* d_xch = nullptr;
    size_t xch_bytes = 0;
    unsigned* d_done = nullptr;
    size_t done_n = 0;
    int n_cu = 256;
    // odam_sq_project_extents: surface points of the tracks of one frame (grow-only)
    float* d_proj_pts = nullptr;
    size_t proj_pts_n = 0;
    double* d_cam = nullptr;      // (unused: the camera travels in the kernel arguments)
    // launches of more objects than CUs: the objects longest first (grow-only)
    int* d_order = nullptr;
    int* h_order = nullptr;       // pinned staging of the same size (the upload is stream-ordered: no NULL-stream copy that would wait for other streams)
    size_t order_n = 0;
};

static std::vector<float> sampler_draws() {
    // sampling.cpp:18-28 with seed 0 (_sampler.pyx:438): identical generator, identical draws
    std::mt19937 gen(0);
    std::uniform_real_distribution<float> dis(0, 1);
    std::vector<float> d(2 * N_PTS);
    for (auto& x : d) x = dis(gen);
    return d;
}

extern "C" int odam_sq_destroy(odam_sq_ctx* c);

extern "C" int odam_sq_create(int max_iters, odam_sq_ctx** out) {
    if (!out || max_iters < 0) return odam_fail(ODAM_E_INVALID, "odam_sq_create: bad arguments");
    std::vector<float> draws = sampler_draws();
    std::vector<unsigned short> oidx(N_PTS);
    for (int i = 0; i < N_PTS; i++) {
        int k = (int)(draws[N_PTS + i] * (float)BUF);  // sampling.cpp:211
        if (k < 0 || k >= BUF) return odam_fail(ODAM_E_INVALID, "sampler draw table out of range");
        oidx[i] = (unsigned short)k;
    }
    std::vector<float> tab((size_t)4 * (max_iters + 1), 0.0f);
    for (int t = 1; t <= max_iters; t++) {  // torch/optim/adam.py _single_tensor_adam, float64 scalars
        double bc1 = 1.0 - std::pow(0.9, (double)t);
        double bc2 = 1.0 - std::pow(0.999, (double)t);
        tab[4 * (t - 1) + 0] = (float)(-(0.01 / bc1));
        tab[4 * (t - 1) + 1] = (float)(-(0.1 / bc1));
        tab[4 * (t - 1) + 2] = (float)std::pow(bc2, 0.5);
    }
    odam_sq_ctx* c = new odam_sq_ctx();
    c->max_iters = max_iters;
    c->d_draws = nullptr; c->d_omega_idx = nullptr; c->d_adam = nullptr; c->d_dummy_params = nullptr;
    const int rc = [&]() -> int {   // any failure below frees what was allocated so far
        ODAM_HIP(hipMalloc(&c->d_draws, sizeof(float) * N_PTS));
        ODAM_HIP(hipMalloc(&c->d_omega_idx, sizeof(unsigned short) * N_PTS));
        ODAM_HIP(hipMalloc(&c->d_adam, sizeof(float) * tab.size()));
        ODAM_HIP(hipMemcpy(c->d_draws, draws.data(), sizeof(float) * N_PTS, hipMemcpyHostToDevice));
        ODAM_HIP(hipMemcpy(c->d_omega_idx, oidx.data(), sizeof(unsigned short) * N_PTS, hipMemcpyHostToDevice));
        ODAM_HIP(hipMemcpy(c->d_adam, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice));
        ODAM_HIP(hipFuncSetAttribute((const void*)sq_fit_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ODAM_HIP(hipFuncSetAttribute((const void*)sq_fit_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return ODAM_OK;
    }();
    if (rc != ODAM_OK) {
        (void)odam_sq_destroy(c);
        return rc;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        c->n_cu = prop.multiProcessorCount;
    *out = c;
    return ODAM_OK;
}

extern "C" int odam_sq_destroy(odam_sq_ctx* c) {
    if (c) {
        if (c->d_proj_pts) (void)hipFree(c->d_proj_pts);
        if (c->d_cam) (void)hipFree(c->d_cam);
        c->d_proj_pts = nullptr; c->d_cam = nullptr;
    }
    if (!c) return ODAM_OK;
    (void)hipFree(c->d_draws);
    (void)hipFree(c->d_omega_idx);
    (void)hipFree(c->d_adam);
    if (c->d_xch) (void)hipFree(c->d_xch);
    if (c->d_done) (void)hipFree(c->d_done);
    if (c->d_order) (void)hipFree(c->d_order);
    if (c->h_order) (void)hipHostFree(c->h_order);
    delete c;
    return ODAM_OK;
}

static int launch_fit(odam_sq_ctx* ctx, int n_obj, FitArgs& A, int max_views, hipStream_t stream) {
    int pn = 1;
    while (pn < max_views) pn <<= 1;
    A.draws = ctx->d_draws;
    A.omega_idx = ctx->d_omega_idx;
    A.adam_tab = ctx->d_adam;
    // View split: only when every workgroup of the grid can be resident at once (one 1024-thread workgroup per CU),
    // the objects are large enough to gain, and the exchange history fits a modest buffer.
    const int split_env = odam_cfg::get(odam_cfg::SQ_SPLIT);
    const int wait_env = odam_cfg::get(odam_cfg::SQ_SPLIT_WAIT);   // 0: never wait (tests)
    const int s_env = odam_cfg::get(odam_cfg::SQ_LANES);
    A.force_lanes_per_view = s_env;
    A.n_obj = n_obj; A.split = 1; A.split_min_views = 64; A.max_views = max_views; A.split_wait_ticks = (unsigned)wait_env;
    int grid = n_obj;
    const int n_pad = (n_obj + 7) / 8 * 8;
    int k = split_env < 1 ? 1 : (split_env > 32 ? 32 : split_env);
    while (k > 1 && n_pad * k > ctx->n_cu) k >>= 1;
    if (k > 1 && A.n_iters > 0 && max_views >= A.split_min_views) {
        const size_t roots = (size_t)A.n_iters * n_obj * k * 9;                 // 64-bit words: one tagged root per workgroup and step
        const size_t rows = A.loss_log ? (size_t)A.n_iters * n_obj * max_views * 4 : 0;   // loss terms, only when logged
        const size_t need = roots * sizeof(unsigned long long) + rows * sizeof(float);
        if (need <= ((size_t)1 << 30)) {
            if (need > ctx->xch_bytes) {
                if (ctx->d_xch) { ODAM_HIP(hipStreamSynchronize(stream)); (void)hipFree(ctx->d_xch); ctx->d_xch = nullptr; ctx->xch_bytes = 0; }
                ODAM_HIP(hipMalloc(&ctx->d_xch, need));
                ctx->xch_bytes = need;
            }
            if ((size_t)n_obj * k > ctx->done_n) {
                if (ctx->d_done) { ODAM_HIP(hipStreamSynchronize(stream)); (void)hipFree(ctx->d_done); ctx->d_done = nullptr; ctx->done_n = 0; }
                ODAM_HIP(hipMalloc(&ctx->d_done, sizeof(unsigned) * n_obj * k));
                ctx->done_n = (size_t)n_obj * k;
            }
            ODAM_HIP(hipMemsetAsync(ctx->d_done, 0, sizeof(unsigned) * n_obj * k, stream));
            ODAM_HIP(hipMemsetAsync(ctx->d_xch, 0, roots * sizeof(unsigned long long), stream));
            A.split = k;
            A.xroots = (unsigned long long*)ctx->d_xch;
            A.xloss = A.loss_log ? (float*)((unsigned long long*)ctx->d_xch + roots) : nullptr;
            A.done = ctx->d_done;
            grid = n_pad * k;
        }
    }
    // LDS: a workgroup reduces at most its own residue class (pn / split rows; objects too small to be split have
    // fewer than split_min_views rows), so the views one object may have grow with the split.
    int rows = pn;
    if (A.split > 1) {
        rows = pn / A.split;
        if (rows < A.split_min_views) rows = A.split_min_views;
    }
    if (rows > ODAM_SQ_MAX_VIEWS)
        return odam_fail(ODAM_E_LIMIT, "odam_sq: more than ODAM_SQ_MAX_VIEWS views per workgroup (too many views per object "
                                       "for this many objects: fit fewer objects per call)");
    A.pn_views = rows;
    const size_t bytes = lds_words(rows, A.loss_log ? max_views : 0) * sizeof(float);
    if (bytes > 160 * 1024) return odam_fail(ODAM_E_LIMIT, "odam_sq: LDS budget exceeded (loss logging with this many views)");
    // More workgroups than CUs (no view split then: one workgroup per object).  Two things decide how long such a launch takes, and both
    // need the view counts, which live on the device: the one small read-back + stream synchronisation below is paid by these launches only
    // (tens of milliseconds each).
    //   * objects of very different view counts (a real scene: 10 ... 300): the launch ends when the last LONG object does -- a 300-view
    //     object that happens to start in the second round finishes 15 ms after everything else.  Workgroup b therefore fits object
    //     order[b], longest first (500 objects of 10 ... 300 views: 29.3 -> see profiles/r6_sq_two_per_cu.txt);
    //   * objects of about equal view counts: eight-wave workgroups, two per CU, so that one's latency-bound phases (sampler chain, row
    //     tree, Adam) run under the other's vector-bound extent scan (500 x 256: 29 -> 24 ms).  With unequal objects this form LOSES (a
    //     long object alone on its CU's half of the lanes is the last to finish: 29.3 -> 32.4 ms), hence the test on max / mean.
    // Same bits either way: nothing of an object's fit depends on its workgroup's id or size.  81 KB per workgroup up to 1026 rows (lds_words).
    A.order = nullptr;
    bool uniform = false;
    if (A.split == 1 && grid > ctx->n_cu && A.view_offsets && A.n_iters > 0 && odam_cfg::get(odam_cfg::SQ_TWO_PER_CU) != 0) {
        std::vector<int> off((size_t)n_obj + 1);
        ODAM_HIP(hipMemcpyAsync(off.data(), A.view_offsets, sizeof(int) * ((size_t)n_obj + 1), hipMemcpyDeviceToHost, stream));
        ODAM_HIP(hipStreamSynchronize(stream));
        long total = 0; int mx = 0;
        for (int i = 0; i < n_obj; i++) { const int v = off[i + 1] - off[i]; total += v; if (v > mx) mx = v; }
        uniform = (double)mx * n_obj <= 1.25 * (double)total;
        if (!uniform) {
            std::vector<int> ord((size_t)n_obj);
            for (int i = 0; i < n_obj; i++) ord[i] = i;
            std::stable_sort(ord.begin(), ord.end(), [&](int a_, int b_) { return off[a_ + 1] - off[a_] > off[b_ + 1] - off[b_]; });
            if ((size_t)n_obj > ctx->order_n) {
                if (ctx->d_order) { (void)hipFree(ctx->d_order); ctx->d_order = nullptr; }      // (the stream was synchronised above: nothing reads them)
                if (ctx->h_order) { (void)hipHostFree(ctx->h_order); ctx->h_order = nullptr; }
                ctx->order_n = 0;
                ODAM_HIP(hipMalloc(&ctx->d_order, sizeof(int) * (size_t)n_obj));
                ODAM_HIP(hipHostMalloc((void**)&ctx->h_order, sizeof(int) * (size_t)n_obj, hipHostMallocDefault));
                ctx->order_n = (size_t)n_obj;
            }
            // (the pinned staging is free: the previous launch that read d_order finished before the synchronisation above)
            std::memcpy(ctx->h_order, ord.data(), sizeof(int) * (size_t)n_obj);
            ODAM_HIP(hipMemcpyAsync(ctx->d_order, ctx->h_order, sizeof(int) * (size_t)n_obj, hipMemcpyHostToDevice, stream));
            A.order = ctx->d_order;
        }
    }
    const bool two = NT_MAX == 1024 && uniform && 2 * bytes <= 160 * 1024;
    if (two) hipLaunchKernelGGL(sq_fit_kernel<512>, dim3(grid), dim3(512), bytes, stream, A);
    else hipLaunchKernelGGL(sq_fit_kernel<NT_MAX>, dim3(grid), dim3(NT_MAX), bytes, stream, A);
    ODAM_HIP(hipGetLastError());
    return ODAM_OK;
}

extern "C" int odam_sq_fit_batch(odam_sq_ctx* ctx, int n_obj, const float* init_params, const int* class_id,
                                 const int* view_offsets, const float* P, const float* tgt,
                                 const float* mask, const float* prior_icov, int n_iters,
                                 int representation, int max_views, float* out_params, float* out_points,
                                 float* loss_log, float* traj, void* stream) {
    if (!ctx || !init_params || !view_offsets || !P || !tgt || !mask || !out_params)
        return odam_fail(ODAM_E_INVALID, "odam_sq_fit_batch: null pointer");
    if (n_obj < 0 || n_iters < 0 || representation < 0 || representation > 2)
        return odam_fail(ODAM_E_INVALID, "odam_sq_fit_batch: bad size / representation");
    if (n_iters > ctx->max_iters) return odam_fail(ODAM_E_LIMIT, "odam_sq_fit_batch: n_iters > max_iters of the context");
    if (max_views < 1 || max_views > 16 * ODAM_SQ_MAX_VIEWS)
        return odam_fail(ODAM_E_LIMIT, "odam_sq_fit_batch: max_views outside 1..16 * ODAM_SQ_MAX_VIEWS");
    if (n_obj == 0) return ODAM_OK;
    FitArgs A{};
    A.init_params = init_params; A.class_id = class_id; A.view_offsets = view_offsets;
    A.P = P; A.tgt = tgt; A.mask = mask; A.prior_icov = prior_icov;
    A.n_iters = n_iters; A.representation = representation;
    A.out_params = out_params; A.out_points = out_points; A.loss_log = loss_log; A.traj = traj;
    return launch_fit(ctx, n_obj, A, max_views, (hipStream_t)stream);
}

extern "C" int odam_sq_points_batch(odam_sq_ctx* ctx, int n, const float* params, float* out_points,
                                    void* stream) {
    if (!ctx || !params || !out_points || n < 0) return odam_fail(ODAM_E_INVALID, "odam_sq_points_batch: bad arguments");
    if (n == 0) return ODAM_OK;
    FitArgs A{};
    A.init_params = params;
    A.n_iters = 0;
    A.representation = ODAM_SQ_SUPER_QUADRIC;
    A.out_params = nullptr;
    A.out_points = out_points;
    return launch_fit(ctx, n, A, 1, (hipStream_t)stream);
}

// ---- projected extent of n super-quadric surfaces in one camera (OdamProcess._prepare_tracks, src/processor.py:181-207) --------
// The reference samples each live track's surface (1000 points, float32), moves them to the camera with T_cw and to the image
// with K in float64 numpy, divides by depth and takes min / max of u and v: the box the track is expected to fill in this
// frame.  One workgroup per track: points from odam_sq_points_batch's kernel, then the same float64 products per point and a
// min / max reduction -- 32 bytes per track come back instead of 12 KB of points and 48,000 host flops.
namespace {
struct Cam21 { double v[21]; };      // T_cw rows 0..2 (12), K (9): travels in the kernel arguments, no upload of its own
__global__ __launch_bounds__(256) void project_extent_kernel(const float* __restrict__ pts, const Cam21 camv, double* __restrict__ out) {
    const double* cam = camv.v;
    __shared__ double red[4][4];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* p = pts + (size_t)t * N_PTS * 3;
    double T[12], K[9];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = cam[i];
#pragma unroll
    for (int i = 0; i < 9; i++) K[i] = cam[12 + i];
    double umin = INFINITY, vmin = INFINITY, umax = -INFINITY, vmax = -INFINITY;
    for (int i = tid; i < N_PTS; i += 256) {
        const double x = (double)p[3 * i], y = (double)p[3 * i + 1], z = (double)p[3 * i + 2];
        const double c0 = x * T[0] + y * T[1] + z * T[2] + T[3];
        const double c1 = x * T[4] + y * T[5] + z * T[6] + T[7];
        const double c2 = x * T[8] + y * T[9] + z * T[10] + T[11];
        const double q0 = c0 * K[0] + c1 * K[1] + c2 * K[2];
        const double q1 = c0 * K[3] + c1 * K[4] + c2 * K[5];
        const double q2 = c0 * K[6] + c1 * K[7] + c2 * K[8];
        const double u = q0 / q2, v = q1 / q2;
        umin = fmin(umin, u); umax = fmax(umax, u); vmin = fmin(vmin, v); vmax = fmax(vmax, v);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        umin = fmin(umin, __shfl_xor(umin, o)); vmin = fmin(vmin, __shfl_xor(vmin, o));
        umax = fmax(umax, __shfl_xor(umax, o)); vmax = fmax(vmax, __shfl_xor(vmax, o));
    }
    if (lane == 0) { red[wave][0] = umin; red[wave][1] = vmin; red[wave][2] = umax; red[wave][3] = vmax; }
    __syncthreads();
    if (tid < 4) {
        double r = red[0][tid];
        for (int w = 1; w < 4; w++) r = tid < 2 ? fmin(r, red[w][tid]) : fmax(r, red[w][tid]);
        out[(size_t)t * 4 + tid] = r;      // x_min, y_min, x_max, y_max
    }
}
}  // namespace

extern "C" int odam_sq_project_extents(odam_sq_ctx* ctx, int n, const float* params, const double* T_cw12_K9, double* out_px,
                                       void* stream) {
    if (!ctx || !params || !T_cw12_K9 || !out_px || n < 0) return odam_fail(ODAM_E_INVALID, "odam_sq_project_extents: bad arguments");
    if (n == 0) return ODAM_OK;
    hipStream_t st = (hipStream_t)stream;
    if ((size_t)n > ctx->proj_pts_n) {
        if (ctx->d_proj_pts) { ODAM_HIP(hipStreamSynchronize(st)); ODAM_HIP(hipFree(ctx->d_proj_pts)); ctx->d_proj_pts = nullptr; }
        const size_t cap = (size_t)n > 64 ? (size_t)n * 2 : 128;
        ODAM_HIP(hipMalloc((void**)&ctx->d_proj_pts, cap * N_PTS * 3 * sizeof(float)));
        ctx->proj_pts_n = cap;
    }
    if (int rc = odam_sq_points_batch(ctx, n, params, ctx->d_proj_pts, stream)) return rc;
    Cam21 cam;
    for (int i = 0; i < 21; i++) cam.v[i] = T_cw12_K9[i];
    hipLaunchKernelGGL(project_extent_kernel, dim3(n), dim3(256), 0, st, ctx->d_proj_pts, cam, out_px);
    ODAM_HIP(hipGetLastError());
    return ODAM_OK;
}

// host sampler: same leaf arithmetic as the device path, serial walk with an explicit work list
// (sampling.cpp:76-125; the order in which pending arcs are visited does not matter -- every arc fills its own slot)
static void host_buffer(float a1, float a2, float e, float ta, float tb, float* buf, int nbuf, std::vector<Arc>& work) {
    Arc root;
    xy(ta, a1, a2, e, root.ax, root.ay);
    xy(tb, a1, a2, e, root.bx, root.by);
    root.ta = ta; root.tb = tb; root.n = nbuf - 2; root.off = 1;
    buf[0] = ta;
    work.clear();
    work.push_back(root);
    while (!work.empty()) {
        const Arc nd = work.back();
        work.pop_back();
        if (nd.n <= 0) continue;
        Arc lf, rt;
        float th;
        const int slot = arc_step(nd, a1, a2, e, th, lf, rt);
        buf[slot] = th;
        work.push_back(lf);
        work.push_back(rt);
    }
    buf[nbuf - 1] = tb;
}

// The reference's one native symbol, with its exact signature (fast_sampler/sampling.hpp:5-15, bound by
// _sampler.pyx:430-439): B x M primitives, N samples each, `buffer_size` equal-arc-length angles per curve, one
// std::mt19937(seed) stream shared by all primitives in (b, m) order -- N draws for the etas, then N for the omegas.
// Host buffers in, host buffers out, no error path (void), re-entrant.
extern "C" void sample_on_batch(float* shapes, float* epsilons, float* etas, float* omegas, int B, int M, int N,
                                int buffer_size, int seed) {
    if (!shapes || !epsilons || !etas || !omegas || B <= 0 || M <= 0 || N <= 0 || buffer_size < 2) return;
    std::mt19937 gen(seed);                                  // sampling.cpp:18-28
    std::uniform_real_distribution<float> dis(0, 1);
    std::vector<float> buf(buffer_size), cdf(buffer_size);
    std::vector<Arc> work;
    for (long p = 0; p < (long)B * M; p++) {
        const float* a = shapes + 3 * p;
        const float* e = epsilons + 2 * p;
        float* eta = etas + (long)N * p;
        float* omega = omegas + (long)N * p;
        host_buffer(a[0], a[2], e[0], PI_F / 2, -(PI_F / 2), buf.data(), buffer_size, work);
        float c = 0.001f;                                    // sampling.cpp:137-147
        cdf[0] = c;
        for (int i = 1; i < buffer_size; i++) {
            c = c + 0.001f + (a[0] + a[1]) * fexp(cosf_(buf[i]), e[0]);
            cdf[i] = c;
        }
        for (int i = 0; i < buffer_size; i++) cdf[i] = cdf[i] / c;
        for (int i = 0; i < N; i++) {                        // first entry not below the draw (sampling.cpp:150-153)
            const float r = dis(gen);
            int lo = 0, hi = buffer_size;
            while (lo < hi) {
                int mid = lo + ((hi - lo) >> 1);
                if (cdf[mid] < r) lo = mid + 1; else hi = mid;
            }
            eta[i] = buf[lo < buffer_size ? lo : buffer_size - 1];
        }
        host_buffer(a[0], a[1], e[1], PI_F, -PI_F, buf.data(), buffer_size, work);
        for (int i = 0; i < N; i++) {                                                        // sampling.cpp:210-212
            // libstdc++'s uniform_real_distribution<float> can return 1.0f: the reference then reads one element past its
            // buffer.  Clamped here -- every draw below 1 indexes exactly as the reference does.
            const int k = (int)(dis(gen) * (float)buffer_size);
            omega[i] = buf[k < buffer_size ? k : buffer_size - 1];
        }
    }
}

// the call the reference's binding makes (_sampler.pyx:415-441: B = M = 1 per optimiser step, buffer 201, seed 0)
extern "C" int odam_sq_sample(const float* a, const float* e, float* etas, float* omegas) {
    if (!a || !e || !etas || !omegas) return odam_fail(ODAM_E_INVALID, "odam_sq_sample: null pointer");
    float aa[3] = {a[0], a[1], a[2]}, ee[2] = {e[0], e[1]};
    sample_on_batch(aa, ee, etas, omegas, 1, 1, N_PTS, BUF, 0);
    return ODAM_OK;
}
