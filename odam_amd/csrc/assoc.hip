// assoc.hip -- the association network's forward pass on gfx950.
//
// Reference (likojack/ODAM src/models/associator.py:163-312): keypoint encoder MLP, sine frame-index encoding,
// 2 self-attention GNN layers over each track's time steps, average pooling over time, 8 alternating self/cross
// GNN layers between tracks and detections (4 heads of 64), final projection, score matrix / 16, 100 Sinkhorn
// iterations in log space.  The reference runs ~300 tiny PyTorch kernels per frame for this; here every Conv1d(k=1)
// is a conv_gemm launch (fp32 MFMA), attention is the fused kernel with head dimension 64, and the whole Sinkhorn
// loop is ONE single-workgroup kernel.  The Hungarian step stays on the host (scipy, as the reference).
//
// Layout: token-major rows.  A token set lives in a [N, 512] buffer "cat": columns 0..255 hold x, columns
// 256..511 receive the attention message, so cat([x, message]) (associator.py:97) needs no copy.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/odam_assoc.h"
#include "conv_gemm.h"
#include "detr_kernels.h"
#include "sk_wave.h"
#include "odam_config.h"
#include "odam_err.h"

using odam_cg::ConvGemmArgs;

namespace {

constexpr int D = 256, NF = 79, NT = 100, ND = 30, FPAD = 128;

struct Lin { float* w = nullptr; float* b = nullptr; int K = 0, N = 0; };
struct Prop { Lin qkv, merge, m0, m2; };   // AttentionalPropagation (associator.py:85-97); q, k, v stacked

struct HostT { std::vector<long long> shape; std::vector<float> data; };

// ---- kernels ------------------------------------------------------------------------------------------------------
// in [n_seq, 79, L] channel-first (row 0 = frame index, rows 1..78 features) ->
// feat [n_seq*L, 128] (features, zero padded) and cat[:, 0:256] = sine encoding of the frame index
// A second input (in2: n_tok2 tokens of sequences of length L2) continues the token rows behind the first: tracks and detections in
// one launch.
__global__ __launch_bounds__(256) void prepare_kernel(const float* __restrict__ in, int L, int n_tok,
                                                      const float* __restrict__ div_term, float* __restrict__ feat,
                                                      float* __restrict__ cat, const float* __restrict__ in2 = nullptr, int L2 = 1, int n_tok2 = 0) {
    const int tok = blockIdx.x;
    if (tok >= n_tok + n_tok2) return;
    if (tok >= n_tok) { in = in2; L = L2; }
    const int t = tok >= n_tok ? tok - n_tok : tok;
    const int s = t / L, l = t - s * L;
    const float* src = in + (size_t)s * NF * L + l;
    const int c = threadIdx.x;
    if (c < FPAD) feat[(size_t)tok * FPAD + c] = (c < NF - 1) ? src[(size_t)(c + 1) * L] : 0.0f;
    const float pos = src[0];
    const float a = pos * div_term[c >> 1];                    // associator.py:325-326
    cat[(size_t)tok * 512 + c] = (c & 1) ? cosf(a) : sinf(a);
}

// F.avg_pool1d over the L time steps of each track: out[s, c] = mean_l cat[(s*L + l), c]   (associator.py:231-232)
// ... and blocks T .. T + n_tail - 1 move the rows that follow the n_seq sequences in `cat` (the encoded detections, which went through
// the encoder launches as 30 more rows of the same matrix) behind the means: out becomes the [T + 30] row block of the matching GNN
__global__ __launch_bounds__(256) void time_mean_kernel(const float* __restrict__ cat, int L, float* __restrict__ out, int n_seq) {
    const int s = blockIdx.x, c = threadIdx.x;
    if (s >= n_seq) {
        out[(size_t)s * 512 + c] = cat[((size_t)n_seq * L + (s - n_seq)) * 512 + c];
        return;
    }
    const float* p = cat + (size_t)s * L * 512 + c;
    float acc = 0.0f;
    int l = 0;
    for (; l + 10 <= L; l += 10) {          // ten rows in flight, added in order (the sum is the sequential one)
        float v[10];
#pragma unroll
        for (int i = 0; i < 10; i++) v[i] = p[(size_t)(l + i) * 512];
#pragma unroll
        for (int i = 0; i < 10; i++) acc += v[i];
    }
    for (; l < L; l++) acc += p[(size_t)l * 512];
    out[(size_t)s * 512 + c] = acc / (float)L;
}

// log_optimal_transport + log_sinkhorn_iterations (associator.py:283-312), one 1024-thread workgroup:
//   couplings Z[(m+1) x (n+1)] = [[scores, alpha], [alpha, alpha]] in LDS; u, v in LDS;
//   iters x { u = log_mu - logsumexp_j(Z + v);  v = log_nu - logsumexp_i(Z + u) };  out = Z + u + v - norm
// A row (or column) is reduced by a group of 8 lanes (xor butterflies inside the group), 128 groups at a time.
constexpr int SK_NT = 1024, SK_G = 8;

__device__ __forceinline__ float group_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4));
    return v;
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    return v;
}

__global__ __launch_bounds__(SK_NT) void sinkhorn_kernel(const float* __restrict__ scores, int lds, int m, int n,
                                                         float alpha, int iters, float* __restrict__ out,
                                                         const int* __restrict__ n_dev, const unsigned* __restrict__ err, unsigned* lost_count) {
    extern __shared__ float sm[];
    if (n_dev) n = *n_dev;      // replayed from a captured graph: the number of detections of THIS frame lives in memory
    const int M1 = m + 1, N1 = n + 1;
    float* Z = sm;                 // [M1][N1]
    float* u = Z + M1 * N1;        // [M1]
    float* v = u + M1;             // [N1]
    const int tid = threadIdx.x;
    for (int i = tid; i < M1 * N1; i += SK_NT) {
        const int r = i / N1, c = i - r * N1;
        Z[i] = (r < m && c < n) ? scores[(size_t)r * lds + c] : alpha;
    }
    for (int i = tid; i < M1; i += SK_NT) u[i] = 0.0f;
    for (int i = tid; i < N1; i += SK_NT) v[i] = 0.0f;
    const float norm = -logf((float)m + (float)n);
    const float log_mu_last = logf((float)n) + norm, log_nu_last = logf((float)m) + norm;
    __syncthreads();
    const int grp = tid / SK_G, gl = tid % SK_G;
    constexpr int NG = SK_NT / SK_G;
    for (int it = 0; it < iters; ++it) {
        for (int r0 = 0; r0 < M1; r0 += NG) {            // u: one group per row
            const int r = r0 + grp;
            float mx = -INFINITY;
            if (r < M1) for (int c = gl; c < N1; c += SK_G) mx = fmaxf(mx, Z[r * N1 + c] + v[c]);
            mx = group_max(mx);
            float sum = 0.0f;
            if (r < M1) for (int c = gl; c < N1; c += SK_G) sum += expf(Z[r * N1 + c] + v[c] - mx);
            sum = group_sum(sum);
            if (r < M1 && gl == 0) u[r] = ((r < m) ? norm : log_mu_last) - (logf(sum) + mx);
        }
        __syncthreads();
        for (int c0 = 0; c0 < N1; c0 += NG) {            // v: one group per column
            const int c = c0 + grp;
            float mx = -INFINITY;
            if (c < N1) for (int r = gl; r < M1; r += SK_G) mx = fmaxf(mx, Z[r * N1 + c] + u[r]);
            mx = group_max(mx);
            float sum = 0.0f;
            if (c < N1) for (int r = gl; r < M1; r += SK_G) sum += expf(Z[r * N1 + c] + u[r] - mx);
            sum = group_sum(sum);
            if (c < N1 && gl == 0) v[c] = ((c < n) ? norm : log_nu_last) - (logf(sum) + mx);
        }
        __syncthreads();
    }
    for (int i = tid; i < M1 * N1; i += SK_NT) {
        const int r = i / N1, c = i - r * N1;
        out[i] = (err && *err) ? NAN : Z[i] + u[r] + v[c] - norm;     // the matching kernel's launch was lost (grid_barrier): fail loudly downstream
    }
    if (tid == 0 && err && *err && lost_count) __hip_atomic_fetch_add(lost_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the matching GNN + final projection as ONE persistent launch ----------------------------------------
// At 40 tracks the GNN is 8 layers x 6 launches on <= 70 rows: every launch is a dispatch and a memory round trip long
// (~10 us) however little it computes.  This kernel keeps PG_WG workgroups resident (one per CU) and walks the same stages
// with a device-wide barrier in between (5 per layer).  A stage is a few hundred independent 16x16 output blocks, ONE per
// single-wave workgroup: the data path of a CU, not the arithmetic, is what a stage costs (a block reads 16 rows of the
// input and 16 rows of the weights, 32-64 KB; with 8 blocks per CU a stage took 11 us), so the blocks are spread over all
// CUs and half a block's operand bytes are in flight before the first v_mfma_f32_16x16x4_f32 issues.
//
// Coherence without cache maintenance: bytes one workgroup writes and another reads later in the launch are stored and
// loaded with sc1 (agent-coherent: written through / fetched past the non-coherent L1 and L2 lines), the stores are
// drained (vmcnt(0)) before the barrier's counter is touched, and the counters are relaxed agent-scope atomics
// (MI355X_MICROARCH.md, inter-workgroup visibility, second valid form).  Release / acquire fences at the barrier instead
// write back and invalidate the whole L2 40 times per launch, weights included.  The weights are read-only: plain loads.
// Barrier: two levels -- 8 group counters (workgroup id % 8; 32 arrivals each on its own line), the last arrival of a group
// bumps the global generation every workgroup polls; the spin is bounded so a lost workgroup cannot hang the device.
constexpr int PG_WG = 256, PG_NT = 256, PG_NW = PG_NT / 64, PG_MAXL = 16, PG_GROUPS = 8;
constexpr int PG_BAR_WORDS = 32 * (2 + 2 * PG_GROUPS);     // generation + flag line, 8 group counters, 8 per-XCD counters (each on its own 128-byte line),
                                                           // one line of 8 placement words (gnn_rowpart_kernel: XCC id + 1 of each group)
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct GnnLayerW { const float *qkv_w, *qkv_b, *mg_w, *mg_b, *m0_w, *m0_b, *m2_w, *m2_b; int cross; };
struct GnnArgs {
    GnnLayerW L[PG_MAXL];
    int n_layers;
    const float *fin_w, *fin_b;
    float *X, *kv, *att, *h, *mT;     // [MX][512], [MX][768], [MX][256], [MX][512], [MX + 2][256]
    float* kv2;                       // second [MX][768] (the row-partitioned kernel alternates between the two by layer)
    int T;
    unsigned* bar;                    // [0] generation, [1] error flag (a barrier timed out), [32 (1 + g)] group counters; zeroed before every launch
    unsigned long long timeout_ticks; // bound of one barrier wait in 100 MHz ticks (0: give up at the first barrier -- tests)
    unsigned long long* stamps;       // diagnostics (odam_assoc_stage_stamps): 100 MHz timer of workgroup 0 after every stage, or null
    int fake_misplaced;               // tests (odam_assoc_debug_misplace): gnn_rowpart_kernel's placement check behaves as if a group straddled XCDs
    int merged;                       // odam_config assoc.merge: m0_w / m0_b hold [W0x | W0m Wm] / b0 + W0m bm, the attention writes into X[:, 256:], no merge stage
};

constexpr int SC1 = 16;      // cache-policy bit of the buffer intrinsics on gfx94x / gfx950
__device__ __forceinline__ __amdgpu_buffer_rsrc_t coh_buf(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float4 coh_ld4(__amdgpu_buffer_rsrc_t r, int float_idx) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, float_idx * 4, 0, SC1));
}
__device__ __forceinline__ float coh_ld(__amdgpu_buffer_rsrc_t r, int float_idx) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, float_idx * 4, 0, SC1));
}
__device__ __forceinline__ void coh_st(__amdgpu_buffer_rsrc_t r, int float_idx, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, float_idx * 4, 0, SC1);
}

// Returns false when the launch is lost: this workgroup waited longer than the bound (not every workgroup became resident:
// a foreign kernel holds CUs, or the device is smaller than the host's gate assumed), or another one said so.  A lost launch
// is abandoned at once -- every workgroup returns at its next barrier instead of spinning through the remaining stages --
// and reported: the flag turns Z into NaN in the Sinkhorn kernel and bumps a host-visible counter, and the host re-runs the
// frame through the launch sequence (odam_assoc_forward_sequence).  The counters are zeroed before every launch, so one
// lost launch does not poison the next.
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned& target, unsigned long long timeout_ticks, int* lost) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's sc1 stores are at their coherence point
    __syncthreads();
    target += PG_GROUPS;
    if (threadIdx.x == 0) {
        unsigned* grp = bar + 32 * (1 + (blockIdx.x & (PG_GROUPS - 1)));
        const unsigned old = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((old + 1u) & (PG_WG / PG_GROUPS - 1)) == 0u) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long* word = reinterpret_cast<unsigned long long*>(bar);      // generation | error flag << 32: one poll sees both
        bool ok = timeout_ticks != 0;
        const unsigned long long t0 = wall_clock64();
        for (int spins = 0; ok; ++spins) {
            const unsigned long long w = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32)) { ok = false; break; }
            if ((int)((unsigned)w - target) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63) == 63 && wall_clock64() - t0 > timeout_ticks) ok = false;
        }
        if (!ok) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lost = ok ? 0 : 1;
    }
    __syncthreads();
    return *lost == 0;
}

// Y[M, N] = act((X[M, K] W[N, K]^T) * scale + b (+ res)); one 16x16 output block per workgroup, wave w takes k in
// [w K / 4, (w + 1) K / 4).  v_mfma_f32_16x16x4_f32: lane l carries A[row l % 16][k l / 16] and B[col l % 16][k l / 16]; a
// lane's 16-byte load holds the k slots of four consecutive instructions (A and B permute k the same way).
// X, res and Y are exchanged between workgroups (sc1); W is read-only (plain, cached).
// (One ascending-k chain per output on a single wave was tried as well: 386 us per launch instead of 250.)
template <int K, bool RELU>
__device__ void stage_gemm(const float* X, int lda, const float* W, const float* b, int M, int N, const float* res, float* Y,
                           int ldc, float* red, int wave, int lane) {
    constexpr int KW = K / PG_NW, NL = KW / 16;            // k per wave, 16-byte loads per lane and operand
    const int RB = (M + 15) >> 4, CB = N >> 4;
    const int li = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t xb = coh_buf(X), yb = coh_buf(Y), rsb = coh_buf(res ? res : Y);
    for (int item = blockIdx.x; item < RB * CB; item += PG_WG) {
        const int rb = item / CB, cb = item - rb * CB;
        const int r = rb * 16 + li;
        const int xo = (r < M ? r : M - 1) * lda + wave * KW + 4 * kq;
        const float* wp = W + (size_t)(cb * 16 + li) * K + wave * KW + 4 * kq;
        float4 a[NL], w[NL];
#pragma unroll
        for (int t = 0; t < NL; t++) {
            a[t] = coh_ld4(xb, xo + 16 * t);
            w[t] = *reinterpret_cast<const float4*>(wp + 16 * t);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NL; t++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, w[t].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, w[t].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, w[t].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, w[t].w, acc1, 0, 0, 0);
        }
        const f32x4 part = acc0 + acc1;
        if (wave) *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = part;
        __syncthreads();
        if (wave == 0) {
            f32x4 tot = part;
#pragma unroll
            for (int w2 = 0; w2 < PG_NW - 1; w2++) tot += *reinterpret_cast<const f32x4*>(red + (w2 * 64 + lane) * 4);
            const int col = cb * 16 + li;
            const float bias = b ? b[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) {      // accumulator register i of lane l: row 4 (l / 16) + i, column l % 16
                const int row = rb * 16 + 4 * kq + i;
                if (row < M) {
                    float v = tot[i] + bias;
                    if (res) v += coh_ld(rsb, row * ldc + col);
                    if (RELU) v = fmaxf(v, 0.0f);
                    coh_st(yb, row * ldc + col, v);
                }
            }
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float wave_max64(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// softmax(Q K^T / 8) V per (query row, head), head dimension 64 (associator.py:75-82); one wavefront per item:
// lane j scores key j, lane d accumulates output channel d, 64 source rows at a time (online softmax across chunks);
// the key rows and the value columns of a chunk are all in flight together.  sc: this wave's LDS strip [64].
template <int ldatt>
__device__ void stage_attn(const float* kv, int T, int cross, float* att, float* sc, int wave_g, int lane) {
    const int MX = T + ND;
    const __amdgpu_buffer_rsrc_t kb = coh_buf(kv), ab = coh_buf(att);
    for (int item = wave_g; item < MX * 4; item += PG_WG * PG_NW) {
        const int q = item >> 2, hd = item & 3;
        const bool is_track = q < T;
        const bool src_tracks = is_track != (cross != 0);
        const int src0 = src_tracks ? 0 : T, nsrc = src_tracks ? T : ND;
        const int qo = q * 768 + hd * 64;
        float4 qq[16];
#pragma unroll
        for (int d = 0; d < 16; d++) qq[d] = coh_ld4(kb, qo + 4 * d);
        float run_max = -INFINITY, run_sum = 0.0f, o = 0.0f;
        for (int j0 = 0; j0 < nsrc; j0 += 64) {
            const int j = j0 + lane, nj = nsrc - j0 < 64 ? nsrc - j0 : 64;
            const int ko = (src0 + (j < nsrc ? j : nsrc - 1)) * 768 + 256 + hd * 64;
            const int vo = (src0 + j0) * 768 + 512 + hd * 64 + lane;
            float4 kk[16];
            float vv[64];
#pragma unroll
            for (int d = 0; d < 16; d++) kk[d] = coh_ld4(kb, ko + 4 * d);
#pragma unroll
            for (int jj = 0; jj < 64; jj++) vv[jj] = coh_ld(kb, vo + (jj < nj ? jj : nj - 1) * 768);
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 16; d++) {
                acc += qq[d].x * kk[d].x; acc += qq[d].y * kk[d].y; acc += qq[d].z * kk[d].z; acc += qq[d].w * kk[d].w;
            }
            const float sdot = j < nsrc ? acc * 0.125f : -INFINITY;
            const float mx = fmaxf(run_max, wave_max64(sdot));
            const float p = j < nsrc ? expf(sdot - mx) : 0.0f;
            const float corr = expf(run_max - mx);            // 0 on the first chunk (run_max = -inf)
            run_sum = run_sum * corr + wave_sum64(p);
            run_max = mx;
            sc[lane] = p;
            __builtin_amdgcn_wave_barrier();
            o *= corr;
#pragma unroll
            for (int jj = 0; jj < 64; jj++) o += (jj < nj ? sc[jj] : 0.0f) * vv[jj];
            __builtin_amdgcn_wave_barrier();
        }
        coh_st(ab, q * ldatt + hd * 64 + lane, o / run_sum);
    }
}

template <bool MERGED>      // odam_config assoc.merge (two instantiations: one kernel holding both forms went from 245 to 248 + 32 registers and 256 B of scratch)
__global__ __launch_bounds__(PG_NT) void gnn_persistent_kernel(GnnArgs a) {
    __shared__ float red[(PG_NW - 1) * 64 * 4];
    __shared__ float scs[PG_NW * 64];
    __shared__ int lost;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_g = wave * PG_WG + blockIdx.x;          // consecutive attention items go to different workgroups
    unsigned target = 0;
    const int T = a.T, MX = T + ND;
    int n_stamp = 0;
    auto stamp = [&] { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[n_stamp++] = wall_clock64(); };
#define PG_BARRIER() do { if (!grid_barrier(a.bar, target, a.timeout_ticks, &lost)) return; stamp(); } while (0)
    stamp();
    float* sc = scs + wave * 64;
    for (int l = 0; l < a.n_layers; l++) {
        const GnnLayerW& P = a.L[l];
        stage_gemm<D, false>(a.X, 512, P.qkv_w, P.qkv_b, MX, 3 * D, nullptr, a.kv, 3 * D, red, wave, lane);
        PG_BARRIER();
        if constexpr (MERGED) {      // the attention's rows ARE the second half of the MLP's input (merge folded into m0_w)
            stage_attn<512>(a.kv, T, P.cross, a.X + D, sc, wave_g, lane);
            PG_BARRIER();
        } else {
            stage_attn<256>(a.kv, T, P.cross, a.att, sc, wave_g, lane);
            PG_BARRIER();
            stage_gemm<D, false>(a.att, D, P.mg_w, P.mg_b, MX, D, nullptr, a.X + D, 512, red, wave, lane);
            PG_BARRIER();
        }
        stage_gemm<2 * D, true>(a.X, 512, P.m0_w, P.m0_b, MX, 2 * D, nullptr, a.h, 2 * D, red, wave, lane);
        PG_BARRIER();
        stage_gemm<2 * D, false>(a.h, 2 * D, P.m2_w, P.m2_b, MX, D, a.X, a.X, 512, red, wave, lane);
        PG_BARRIER();
    }
#undef PG_BARRIER
    stage_gemm<D, false>(a.X, 512, a.fin_w, a.fin_b, MX, D, nullptr, a.mT, D, red, wave, lane);
    stamp();
}

// ---- the same network, ROWS DEALT TO THE XCDs: one device-wide barrier per layer instead of five -------------------------------------
// Of a layer's five stages only the attention needs rows of other workgroups' making in bulk -- keys and values of ALL rows; the
// projections and the MLP are row-wise.  Workgroup b of a plain launch always runs on XCD b mod 8 (tests/native/xcd_probe.hip), and an
// exchange among the 32 workgroups of one XCD -- plain stores, sc1 loads that hit that XCD's L2, a workgroup-scope counter executed
// in that L2 -- costs 1.1 us where the device-wide form costs 4.9 (tests/native/xcd_barrier_probe.hip).  So XCD x owns rows
// [x R, (x + 1) R), R = ceil(rows / 8): its workgroups project q | k | v of those rows (stored sc1: the one thing that crosses XCDs),
// all 256 workgroups meet ONCE, then attention, merge and the two MLP layers of those rows run behind XCD-local barriers.  Every
// output element is computed exactly as in gnn_persistent_kernel (same 16x16 blocks, same K split over the four waves): the two
// kernels agree bit for bit.  k | v alternate between two buffers by layer: a fast XCD may project layer l + 1 while a slow one still
// reads layer l's keys (the device barrier of layer l + 1 is what frees buffer l & 1 again).
__device__ __forceinline__ bool xcd_barrier(unsigned* bar, unsigned* cnt, unsigned& target, unsigned long long timeout_ticks, int* lost) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have reached the L2 the readers will hit
    __syncthreads();
    target += PG_WG / PG_GROUPS;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool ok = timeout_ticks != 0;
        const unsigned long long t0 = wall_clock64();
        for (int spins = 0; ok; ++spins) {
            unsigned g, e;      // sc0 sc1: past the vector L1 (an sc0-only poll can spin on a stale line)
            asm volatile("global_load_dword %0, %2, off sc0 sc1\n\tglobal_load_dword %1, %3, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(g), "=&v"(e) : "v"(cnt), "v"(bar + 1) : "memory");
            if (e) { ok = false; break; }
            if ((int)(g - target) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63) == 63 && wall_clock64() - t0 > timeout_ticks) ok = false;
        }
        if (!ok) __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lost = ok ? 0 : 1;
    }
    __syncthreads();
    return *lost == 0;
}

// stage_gemm over rows [r0, r0 + nR) only, blocks dealt over the nw workgroups of this XCD (wi = 0 .. nw - 1); FAR: the output is read
// by other XCDs (sc1 store), otherwise a plain store that the XCD-local barrier drains
template <int K, bool RELU, bool FAR>
__device__ void stage_gemm_rows(const float* X, int lda, const float* W, const float* b, int r0, int nR, int N, const float* res,
                                float* Y, int ldc, float* red, int wave, int lane, int wi, int nw) {
    constexpr int KW = K / PG_NW, NL = KW / 16;
    const int RB = (nR + 15) >> 4, CB = N >> 4, rlim = r0 + nR;
    const int li = lane & 15, kq = lane >> 4;
    const __amdgpu_buffer_rsrc_t xb = coh_buf(X), yb = coh_buf(Y), rsb = coh_buf(res ? res : Y);
    for (int item = wi; item < RB * CB; item += nw) {
        const int rb = item / CB, cb = item - rb * CB;
        const int r = r0 + rb * 16 + li;
        const int xo = (r < rlim ? r : rlim - 1) * lda + wave * KW + 4 * kq;
        const float* wp = W + (size_t)(cb * 16 + li) * K + wave * KW + 4 * kq;
        float4 a[NL], w[NL];
#pragma unroll
        for (int t = 0; t < NL; t++) {
            a[t] = coh_ld4(xb, xo + 16 * t);
            w[t] = *reinterpret_cast<const float4*>(wp + 16 * t);
        }
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NL; t++) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, w[t].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, w[t].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, w[t].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, w[t].w, acc1, 0, 0, 0);
        }
        const f32x4 part = acc0 + acc1;
        if (wave) *reinterpret_cast<f32x4*>(red + ((wave - 1) * 64 + lane) * 4) = part;
        __syncthreads();
        if (wave == 0) {
            f32x4 tot = part;
#pragma unroll
            for (int w2 = 0; w2 < PG_NW - 1; w2++) tot += *reinterpret_cast<const f32x4*>(red + (w2 * 64 + lane) * 4);
            const int col = cb * 16 + li;
            const float bias = b ? b[col] : 0.0f;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int row = r0 + rb * 16 + 4 * kq + i;
                if (row < rlim) {
                    float v = tot[i] + bias;
                    if (res) v += coh_ld(rsb, row * ldc + col);
                    if (RELU) v = fmaxf(v, 0.0f);
                    if (FAR) coh_st(yb, row * ldc + col, v);
                    else Y[(size_t)row * ldc + col] = v;
                }
            }
        }
        __syncthreads();
    }
}

// stage_attn for the queries [r0, r0 + nR) (items dealt over the n_waves wavefronts of this XCD); keys and values of every row come
// from kv (sc1 loads), the output stays in this XCD (plain store)
template <int ldatt>
__device__ void stage_attn_rows(const float* kv, int T, int cross, float* att, float* sc, int r0, int nR, int wave_x, int n_waves, int lane) {
    const __amdgpu_buffer_rsrc_t kb = coh_buf(kv);
    for (int item = wave_x; item < nR * 4; item += n_waves) {
        const int q = r0 + (item >> 2), hd = item & 3;
        const bool is_track = q < T;
        const bool src_tracks = is_track != (cross != 0);
        const int src0 = src_tracks ? 0 : T, nsrc = src_tracks ? T : ND;
        const int qo = q * 768 + hd * 64;
        float4 qq[16];
#pragma unroll
        for (int d = 0; d < 16; d++) qq[d] = coh_ld4(kb, qo + 4 * d);
        float run_max = -INFINITY, run_sum = 0.0f, o = 0.0f;
        for (int j0 = 0; j0 < nsrc; j0 += 64) {
            const int j = j0 + lane, nj = nsrc - j0 < 64 ? nsrc - j0 : 64;
            const int ko = (src0 + (j < nsrc ? j : nsrc - 1)) * 768 + 256 + hd * 64;
            const int vo = (src0 + j0) * 768 + 512 + hd * 64 + lane;
            float4 kk[16];
            float vv[64];
#pragma unroll
            for (int d = 0; d < 16; d++) kk[d] = coh_ld4(kb, ko + 4 * d);
#pragma unroll
            for (int jj = 0; jj < 64; jj++) vv[jj] = coh_ld(kb, vo + (jj < nj ? jj : nj - 1) * 768);
            float acc = 0.0f;
#pragma unroll
            for (int d = 0; d < 16; d++) {
                acc += qq[d].x * kk[d].x; acc += qq[d].y * kk[d].y; acc += qq[d].z * kk[d].z; acc += qq[d].w * kk[d].w;
            }
            const float sdot = j < nsrc ? acc * 0.125f : -INFINITY;
            const float mx = fmaxf(run_max, wave_max64(sdot));
            const float p = j < nsrc ? expf(sdot - mx) : 0.0f;
            const float corr = expf(run_max - mx);
            run_sum = run_sum * corr + wave_sum64(p);
            run_max = mx;
            sc[lane] = p;
            __builtin_amdgcn_wave_barrier();
            o *= corr;
#pragma unroll
            for (int jj = 0; jj < 64; jj++) o += (jj < nj ? sc[jj] : 0.0f) * vv[jj];
            __builtin_amdgcn_wave_barrier();
        }
        att[(size_t)q * ldatt + hd * 64 + lane] = o / run_sum;
    }
}

template <bool MERGED>
__global__ __launch_bounds__(PG_NT) void gnn_rowpart_kernel(GnnArgs a) {
    __shared__ float red[(PG_NW - 1) * 64 * 4];
    __shared__ float scs[PG_NW * 64];
    __shared__ int lost;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & (PG_GROUPS - 1), wi = blockIdx.x >> 3, nw = PG_WG / PG_GROUPS;
    const int T = a.T, MX = T + ND;
    const int R = (MX + PG_GROUPS - 1) / PG_GROUPS;
    const int r0 = xcd * R, nR = MX - r0 < 0 ? 0 : (MX - r0 < R ? MX - r0 : R);      // the rows of this XCD (the last XCDs may have none)
    const int wave_x = wave * nw + wi, n_waves = nw * PG_NW;
    unsigned target = 0, target_x = 0;
    unsigned* cnt_x = a.bar + 32 * (1 + PG_GROUPS + xcd);
    int n_stamp = 0;
    auto stamp = [&] { if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[n_stamp++] = wall_clock64(); };
#define PG_BARRIER() do { if (!grid_barrier(a.bar, target, a.timeout_ticks, &lost)) return; stamp(); } while (0)
#define PX_BARRIER() do { if (!xcd_barrier(a.bar, cnt_x, target_x, a.timeout_ticks, &lost)) return; stamp(); } while (0)
    // Placement check.  The XCD-local stages below are exchanged with plain stores and a workgroup-scope counter, which is right only
    // if the 32 workgroups of a group (equal blockIdx mod 8) run behind ONE L2.  That is how the dispatcher has always placed a plain
    // launch here (tests/native/xcd_probe.hip), but HIP promises nothing: every workgroup reads the hardware's XCC id and the first of a
    // group to arrive records it; one that finds another id raises the launch's error flag -- every workgroup then returns at its
    // first barrier, the Sinkhorn kernel reports the launch as lost and the host re-runs the frame through the launch sequence
    // (and, after three such losses in a row, stays on it: associator.py).
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) + 1u;       // HW_REG_XCC_ID [3:0], + 1: 0 = nobody yet
        unsigned seen = 0u;
        if (!__hip_atomic_compare_exchange_strong(a.bar + 32 * (1 + 2 * PG_GROUPS) + xcd, &seen, xcc, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) &&
            (seen != xcc || a.fake_misplaced))
            __hip_atomic_store(a.bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp();
    float* sc = scs + wave * 64;
    for (int l = 0; l < a.n_layers; l++) {
        const GnnLayerW& P = a.L[l];
        float* kv = (l & 1) ? a.kv2 : a.kv;
        stage_gemm_rows<D, false, true>(a.X, 512, P.qkv_w, P.qkv_b, r0, nR, 3 * D, nullptr, kv, 3 * D, red, wave, lane, wi, nw);
        PG_BARRIER();
        if constexpr (MERGED) {
            stage_attn_rows<512>(kv, T, P.cross, a.X + D, sc, r0, nR, wave_x, n_waves, lane);
            PX_BARRIER();
        } else {
            stage_attn_rows<256>(kv, T, P.cross, a.att, sc, r0, nR, wave_x, n_waves, lane);
            PX_BARRIER();
            stage_gemm_rows<D, false, false>(a.att, D, P.mg_w, P.mg_b, r0, nR, D, nullptr, a.X + D, 512, red, wave, lane, wi, nw);
            PX_BARRIER();
        }
        stage_gemm_rows<2 * D, true, false>(a.X, 512, P.m0_w, P.m0_b, r0, nR, 2 * D, nullptr, a.h, 2 * D, red, wave, lane, wi, nw);
        PX_BARRIER();
        stage_gemm_rows<2 * D, false, false>(a.h, 2 * D, P.m2_w, P.m2_b, r0, nR, D, a.X, a.X, 512, red, wave, lane, wi, nw);
        PX_BARRIER();
    }
#undef PG_BARRIER
#undef PX_BARRIER
    stage_gemm_rows<D, false, false>(a.X, 512, a.fin_w, a.fin_b, r0, nR, D, nullptr, a.mT, D, red, wave, lane, wi, nw);
    stamp();
}

// ---- Sinkhorn for at most 31 columns (the associator's case: <= 30 detections + dustbin) ----------------------------------
// Same iteration as sinkhorn_kernel; what differs is who reduces what.  A row (<= 32 entries) is one 16-lane DPP row of a
// wavefront, two columns per lane: four rows per wavefront, 64 rows per sweep of the 16 waves, reduced by quad permutes and
// half-row / row mirrors alone (register-file speed).  A wavefront takes two columns at once, lane = row; its 64-lane
// reductions finish through v_readlane.  No ds_bpermute is left on the dependent path of an iteration, and Z is stored with
// an odd row stride so that a column walks all LDS banks.  Measured per 100 iterations: DESIGN.md section 4.
__device__ __forceinline__ float dpp_f(float v, int ctrl_sel) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    switch (ctrl_sel) {
        case 0: r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false); break;     // quad_perm [1,0,3,2]
        case 1: r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false); break;     // quad_perm [2,3,0,1]
        case 2: r = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false); break;    // row_half_mirror
        default: r = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false); break;   // row_mirror
    }
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float max16(float v) {
    v = fmaxf(v, dpp_f(v, 0)); v = fmaxf(v, dpp_f(v, 1)); v = fmaxf(v, dpp_f(v, 2)); v = fmaxf(v, dpp_f(v, 3));
    return v;
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_f(v, 0); v += dpp_f(v, 1); v += dpp_f(v, 2); v += dpp_f(v, 3);
    return v;
}

// exp / log of the iteration: the hardware's v_exp_f32 / v_log_f32 (through exp2 / log2).  The library functions are a
// range reduction and a polynomial each -- four of them sit on the dependent path of every iteration and were most of its
// 1.9 us; arguments here are (x - max) <= 0 and sums in [1, 64], where the hardware forms are good to ~1e-6 relative.
#define SK_EXP(x) __expf(x)
#define SK_LOG(x) __logf(x)
__global__ __launch_bounds__(SK_NT) void sinkhorn32_kernel(const float* __restrict__ scores, int lds, int m, int n,
                                                           float alpha, int iters, float* __restrict__ out,
                                                           const int* __restrict__ n_dev, const unsigned* __restrict__ err, unsigned* lost_count) {
    extern __shared__ float sm[];
    if (n_dev) n = *n_dev;
    const int M1 = m + 1, N1 = n + 1;       // N1 <= 32
    constexpr int ZS = 33;
    float* Z = sm;                          // [M1][33]
    float* u = Z + (size_t)M1 * ZS;         // [M1]
    float* v = u + M1;                      // [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = SK_NT / 64;
    for (int i = tid; i < M1 * N1; i += SK_NT) {
        const int r = i / N1, c = i - r * N1;
        Z[r * ZS + c] = (r < m && c < n) ? scores[(size_t)r * lds + c] : alpha;
    }
    for (int i = tid; i < M1; i += SK_NT) u[i] = 0.0f;
    if (tid < 32) v[tid] = 0.0f;
    const float norm = -logf((float)m + (float)n);
    const float log_mu_last = logf((float)n) + norm, log_nu_last = logf((float)m) + norm;
    __syncthreads();
    const int q4 = lane >> 4, l16 = lane & 15;
    // 64-lane reductions: DPP inside the four 16-lane rows, then the four row results through SGPRs (v_readlane) --
    // no ds_bpermute on the dependent path
    auto max64 = [](float x) {
        x = max16(x);
        const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
        const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
        const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
        const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
        return fmaxf(fmaxf(a0, a1), fmaxf(a2, a3));
    };
    auto sum64 = [](float x) {
        x = sum16(x);
        const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
        const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 16));
        const float a2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 32));
        const float a3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 48));
        return (a0 + a1) + (a2 + a3);
    };
    for (int it = 0; it < iters; ++it) {
        // u: four rows per wavefront, one per 16-lane DPP row; a lane holds columns l16 and l16 + 16 of its row
        for (int r0 = 4 * wave; r0 < M1; r0 += 4 * NWV) {
            const int r = r0 + q4;
            const bool ok0 = r < M1 && l16 < N1, ok1 = r < M1 && l16 + 16 < N1;
            const float x0 = ok0 ? Z[r * ZS + l16] + v[l16] : -INFINITY;
            const float x1 = ok1 ? Z[r * ZS + l16 + 16] + v[l16 + 16] : -INFINITY;
            const float mx = max16(fmaxf(x0, x1));
            const float e = sum16((ok0 ? SK_EXP(x0 - mx) : 0.0f) + (ok1 ? SK_EXP(x1 - mx) : 0.0f));
            if (r < M1 && l16 == 0) u[r] = ((r < m) ? norm : log_mu_last) - (SK_LOG(e) + mx);
        }
        __syncthreads();
        // v: columns wave and wave + 16 on this wavefront (two independent chains), lane = row
        {
            const int c0 = wave, c1 = wave + NWV;
            const bool h0 = c0 < N1, h1 = c1 < N1;
            float m0 = -INFINITY, m1 = -INFINITY;
            for (int r = lane; r < M1; r += 64) {
                const float ur = u[r];
                if (h0) m0 = fmaxf(m0, Z[r * ZS + c0] + ur);
                if (h1) m1 = fmaxf(m1, Z[r * ZS + c1] + ur);
            }
            m0 = max64(m0); m1 = max64(m1);
            float e0 = 0.0f, e1 = 0.0f;
            for (int r = lane; r < M1; r += 64) {
                const float ur = u[r];
                if (h0) e0 += SK_EXP(Z[r * ZS + c0] + ur - m0);
                if (h1) e1 += SK_EXP(Z[r * ZS + c1] + ur - m1);
            }
            e0 = sum64(e0); e1 = sum64(e1);
            if (lane == 0) {
                if (h0) v[c0] = ((c0 < n) ? norm : log_nu_last) - (SK_LOG(e0) + m0);
                if (h1) v[c1] = ((c1 < n) ? norm : log_nu_last) - (SK_LOG(e1) + m1);
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < M1 * N1; i += SK_NT) {
        const int r = i / N1, c = i - r * N1;
        out[i] = (err && *err) ? NAN : Z[r * ZS + c] + u[r] + v[c] - norm;
    }
    // ... and tell the host (pinned counter, read after the stream has been synchronised: odam_assoc_lost_launches)
    if (tid == 0 && err && *err && lost_count) __hip_atomic_fetch_add(lost_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- Sinkhorn on ONE wavefront: <= 128 rows, <= 32 columns -----------------------------------------------------------------------
// The iteration of log_sinkhorn_iterations (associator.py:283-312) on scaling factors instead of potentials.  With
// K_rc = exp(Z_rc + u_r + v_c) (the coupling at the potentials reached so far), a_r, b_c the factors since then, an iteration is
//     a_r = mu_r / sum_c K_rc b_c        b_c = nu_c / sum_r K_rc a_r
// -- multiply-adds and two reciprocals; no exp / log / max, no LDS and no barrier.  Every J iterations the factors are ABSORBED:
// u += log a, v += log b, K recomputed from Z, u, v, a = b = 1.  In exact arithmetic this is the log-space iteration; in float32
// it stays so as long as the factors of J iterations stay far inside the float range (they are checked at every absorption:
// 1e-18 .. 1e18) -- an entry of K that underflows is one whose mass is below e^-87 of a unit AT the current potentials, which the
// log-space sum drops in the same way; and it is recomputed at the next absorption.  The potentials start at u = -(row maximum),
// v = 0, so that no entry of K overflows whatever the size of the scores (the hand-built scene weights reach +-1000).  A failed
// check restarts the whole loop with J = 1 (absorb after every iteration), then in log space -- same launch, wave-uniform decision.
// Layout: lane = row (RB rows per lane), its K row in NC registers.  The column sums are one TRANSPOSING reduction (sk_wave.h): a
// level combines two registers into one -- the lanes whose level bit is clear keep the first column and receive the partner
// lane's share of it, the others the second -- so 32 columns cost 16 + 8 + 4 + 2 + 1 exchanges, not 32 x 6, and end with column
// c's total in lane sk_lane(c): b is ONE register (lane = column), one reciprocal per iteration; the row sums read it back
// through v_readlane.
template <int RB, int NC>
__global__ __launch_bounds__(64) void sinkhorn_wave_kernel(const float* __restrict__ scores, int lds, int m, int n, float alpha, int iters,
                                                           float* __restrict__ out, const int* __restrict__ n_dev,
                                                           const unsigned* __restrict__ err, unsigned* lost_count, int first_mode) {
    constexpr int P = NC <= 8 ? 8 : NC <= 16 ? 16 : 32;
    if (n_dev) n = *n_dev;
    const int lane = threadIdx.x, M1 = m + 1, N1 = n + 1;
    const float inv = 1.0f / ((float)m + (float)n);          // exp(norm)
    const float norm = -logf((float)m + (float)n);
    float z[RB][NC], rho[RB], mu[RB], u[RB], vc[NC];
#pragma unroll
    for (int j = 0; j < RB; j++) {
        const int r = lane + 64 * j;
        const bool valid = r < M1;
        rho[j] = -INFINITY;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            z[j][c] = (valid && c < N1) ? ((r < m && c < n) ? scores[(size_t)r * lds + c] : alpha) : -INFINITY;
            rho[j] = fmaxf(rho[j], z[j][c]);
        }
        mu[j] = valid ? (r < m ? inv : (float)n * inv) : 0.0f;
    }
    const int cl = sk_lane(lane & 31) & (P - 1);             // the column whose total this lane receives
    const float nu = cl < n ? inv : (cl == n ? (float)m * inv : 0.0f);
    auto bcast = [&](float x, float (&o)[NC]) {             // lane = column -> every lane holds all columns
#pragma unroll
        for (int c = 0; c < NC; c++) o[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), sk_lane(c)));
    };
    bool done = false;
    for (int mode = first_mode; mode < 2 && !done; mode++) {
        const int J = mode == 0 ? 10 : 1;
        float K[RB][NC], a[RB], vl = 0.0f, b = nu != 0.0f ? 1.0f : 0.0f;
        bool good = true;
#pragma unroll
        for (int j = 0; j < RB; j++) {
            u[j] = mu[j] != 0.0f ? -rho[j] : 0.0f;
            a[j] = 0.0f;
#pragma unroll
            for (int c = 0; c < NC; c++) K[j][c] = expf(z[j][c] + u[j]);          // exp(-inf) = 0 outside the matrix
        }
        for (int it0 = 0; it0 < iters; it0 += J) {
            const int ne = iters - it0 < J ? iters - it0 : J;
            for (int e = 0; e < ne; e++) {
                float bc[NC];
                bcast(b, bc);
#pragma unroll
                for (int j = 0; j < RB; j++) {
                    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
                    for (int c = 0; c < NC; c += 2) { s0 = fmaf(K[j][c], bc[c], s0); s1 = fmaf(K[j][c + 1], bc[c + 1], s1); }
                    a[j] = mu[j] != 0.0f ? mu[j] * __builtin_amdgcn_rcpf(s0 + s1) : 0.0f;
                }
                float t[32];
#pragma unroll
                for (int c = 0; c < 32; c++) {
                    if (c < NC) {
                        t[c] = K[0][c] * a[0];
#pragma unroll
                        for (int j = 1; j < RB; j++) t[c] = fmaf(K[j][c], a[j], t[c]);
                    } else t[c] = 0.0f;
                }
                const float tot = sk_colsum<P>(t, lane);
                b = nu != 0.0f ? nu * __builtin_amdgcn_rcpf(tot) : 0.0f;
            }
            // absorb the factors into the potentials
            good = good && (nu == 0.0f || (b > 1e-18f && b < 1e18f));
            vl += nu != 0.0f ? logf(b) : 0.0f;
#pragma unroll
            for (int j = 0; j < RB; j++) {
                good = good && (mu[j] == 0.0f || (a[j] > 1e-18f && a[j] < 1e18f));
                u[j] += mu[j] != 0.0f ? logf(a[j]) : 0.0f;
            }
            if (it0 + J < iters) {
                bcast(vl, vc);
#pragma unroll
                for (int j = 0; j < RB; j++)
#pragma unroll
                    for (int c = 0; c < NC; c++) K[j][c] = expf(z[j][c] + u[j] + vc[c]);
                b = nu != 0.0f ? 1.0f : 0.0f;
            }
        }
        bcast(vl, vc);
        done = __all(good);
    }
    if (!done) {
        // log space with the same layout: a row's logsumexp is per lane, a column's a 64-lane reduction
        const float log_mu_last = logf((float)n) + norm, log_nu_last = logf((float)m) + norm;
#pragma unroll
        for (int c = 0; c < NC; c++) vc[c] = 0.0f;
#pragma unroll
        for (int j = 0; j < RB; j++) u[j] = 0.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < RB; j++) {
                const int r = lane + 64 * j;
                float mx = -INFINITY, sum = 0.0f;
#pragma unroll
                for (int c = 0; c < NC; c++) mx = fmaxf(mx, z[j][c] + vc[c]);
#pragma unroll
                for (int c = 0; c < NC; c++) sum += expf(z[j][c] + vc[c] - mx);          // exp(-inf) = 0 outside the matrix
                u[j] = r < M1 ? ((r < m) ? norm : log_mu_last) - (logf(sum) + mx) : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < RB; j++) mx = fmaxf(mx, z[j][c] + u[j]);
                mx = sk_wave_max(mx);
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < RB; j++) sum += expf(z[j][c] + u[j] - mx);
                sum = sk_wave_sum(sum);
                vc[c] = c < N1 ? ((c < n) ? norm : log_nu_last) - (logf(sum) + mx) : 0.0f;
            }
        }
    }
    const bool lost = err && *err;
#pragma unroll
    for (int j = 0; j < RB; j++) {
        const int r = lane + 64 * j;
        if (r < M1) {
#pragma unroll
            for (int c = 0; c < NC; c++)
                if (c < N1) out[(size_t)r * N1 + c] = lost ? NAN : z[j][c] + u[j] + vc[c] - norm;
        }
    }
    if (lane == 0 && lost && lost_count) __hip_atomic_fetch_add(lost_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // err = the persistent matching kernel's flag word (bar + 1): that launch is over (stream order) and its flag has been read
    // above -- leave generation, flag, the group counters and the per-XCD counters at zero for the next launch (saves a memset per frame)
    if (err && lane <= 2 * PG_GROUPS + PG_GROUPS) {
        unsigned* bar = const_cast<unsigned*>(err) - 1;
        if (lane == 0) { bar[0] = 0u; bar[1] = 0u; }
        else if (lane <= 2 * PG_GROUPS) bar[32 * lane] = 0u;
        else bar[32 * (1 + 2 * PG_GROUPS) + (lane - 2 * PG_GROUPS - 1)] = 0u;      // the placement words of gnn_rowpart_kernel
    }
}

// log_optimal_transport on the device: the 31-column kernel where it applies, the general one otherwise
static int launch_sinkhorn(const float* scores, int lds_, int m_, int n_, int n_cap, float alpha, int iters, float* Z_out,
                           const int* n_dev, hipStream_t st, const unsigned* err = nullptr, unsigned* lost_count = nullptr, bool* cleans_bar = nullptr) {
    if (n_cap + 1 <= 32 && m_ + 1 <= 128 && odam_cfg::get(odam_cfg::ASSOC_SK_FAST) != 0) {       // one wavefront, registers only
        const int nc = (n_cap + 1 + 7) >> 3;
#define ODAM_SKW(RB, NC) hipLaunchKernelGGL((sinkhorn_wave_kernel<RB, NC>), dim3(1), dim3(64), 0, st, scores, lds_, m_, n_, alpha, iters, Z_out, n_dev, err, lost_count, odam_cfg::get(odam_cfg::ASSOC_SK_FAST) - 1)
        if (m_ + 1 <= 64) { if (nc == 1) ODAM_SKW(1, 8); else if (nc == 2) ODAM_SKW(1, 16); else if (nc == 3) ODAM_SKW(1, 24); else ODAM_SKW(1, 32); }
        else { if (nc == 1) ODAM_SKW(2, 8); else if (nc == 2) ODAM_SKW(2, 16); else if (nc == 3) ODAM_SKW(2, 24); else ODAM_SKW(2, 32); }
#undef ODAM_SKW
        if (cleans_bar) *cleans_bar = err != nullptr;
    } else if (n_cap + 1 <= 32 && m_ + 1 <= 1100) {        // 1100 rows of 33 floats + u + v fit the 150 KB of dynamic LDS
        const size_t lds = ((size_t)(m_ + 1) * 33 + (m_ + 1) + 32) * sizeof(float);
        hipLaunchKernelGGL(sinkhorn32_kernel, dim3(1), dim3(SK_NT), lds, st, scores, lds_, m_, n_, alpha, iters, Z_out, n_dev, err, lost_count);
    } else {
        const size_t lds = ((size_t)(m_ + 1) * (n_cap + 1) + (m_ + 1) + (n_cap + 1)) * sizeof(float);
        hipLaunchKernelGGL(sinkhorn_kernel, dim3(1), dim3(SK_NT), lds, st, scores, lds_, m_, n_, alpha, iters, Z_out, n_dev, err, lost_count);
    }
    ODAM_HIP(hipGetLastError());
    return 0;
}

int lin(const Lin& L, const float* x, int lda, int M, const float* res, bool relu, float* y, int ldc, const float* scale,
        hipStream_t st) {
    ConvGemmArgs a{};
    a.A = x; a.Wt = L.w; a.scale = scale; a.bias = L.b; a.res = res; a.C = y;
    a.B = 1; a.H = 1; a.W = M; a.Cin = L.K; a.lda = lda;
    int lg = 0; while ((1 << lg) < L.K) lg++;
    a.log2Cin = lg;
    a.Ho = 1; a.Wo = M; a.Cout = L.N; a.KH = a.KW = 1; a.stride = 1; a.pad = 0; a.Kpad = L.K;
    a.relu = relu ? 1 : 0; a.M = M; a.ldc = ldc;
    a.no_pin = 1;      // replicated on every rank with identical rows: nothing to pin, and 4,000 rows want the small tiles
    return odam_cg::launch_conv_gemm(a, st);
}

#define RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

}  // namespace

struct odam_assoc {
    int max_tracks = 0, n_self = 0, n_gnn = 0, iters = 100;
    std::vector<int> gnn_cross;          // per GNN layer: 1 = cross
    std::map<std::string, HostT> host;
    bool finalized = false;
    std::vector<void*> allocs;
    Lin enc0, enc2, final_proj;
    std::vector<Prop> fuser, gnn;
    float bin_score = 1.0f;
    float *div_term = nullptr, *sc16 = nullptr;
    // workspace
    float *feat = nullptr, *h256 = nullptr, *catT = nullptr, *kv = nullptr, *att = nullptr, *h512 = nullptr;
    float *kvX = nullptr, *kvX2 = nullptr, *attX = nullptr, *hX = nullptr;
    float *catTr = nullptr, *featD = nullptr, *mT = nullptr, *scores = nullptr;
    // persistent matching kernel: barrier counters (+ error flag), zeroed on the stream before every launch
    unsigned* bar = nullptr;
    bool persist = true;
    int resident_capacity = 0;               // workgroups of the persistent kernel the device can hold at once (with the guide's margin)
    unsigned long long timeout_ticks = 2000000ull;   // 20 ms per barrier wait (100 MHz ticks); a healthy wait is ~2 us, ~1 ms under a saturated device
    unsigned* lost_count = nullptr;          // pinned host word: launches abandoned at a barrier so far (the Sinkhorn kernel bumps it)
    unsigned long long* stamps = nullptr;    // device, 128 entries; written only while want_stamps
    bool want_stamps = false;
    bool fake_misplaced = false;             // tests: odam_assoc_debug_misplace
    bool bar_clean = false;                  // the counters are zero: the one-wavefront Sinkhorn kernel zeroes them behind the launch that used them
    bool merged = false;                     // odam_config assoc.merge as read by odam_assoc_finalize: every Prop's m0 holds the merge projection too

    // Zeroing goes through a private non-blocking stream that finalize waits for: hipMemset would be ordered on the NULL stream -- it
    // may still be pending when it returns (and then land behind the first results of a caller that works on a stream of its own),
    // and waiting for the NULL stream instead would wait for whatever another thread has queued there (a detector's chunk copies).
    hipStream_t init_stream = nullptr;
    int alloc(float** p, size_t n) {
        ODAM_HIP(hipMalloc((void**)p, n * sizeof(float)));
        if (!init_stream) ODAM_HIP(hipStreamCreateWithFlags(&init_stream, hipStreamNonBlocking));
        ODAM_HIP(hipMemsetAsync(*p, 0, n * sizeof(float), init_stream));
        allocs.push_back(*p);
        return 0;
    }
    int init_done() {
        if (init_stream) { ODAM_HIP(hipStreamSynchronize(init_stream)); ODAM_HIP(hipStreamDestroy(init_stream)); init_stream = nullptr; }
        return 0;
    }
    int upload(float** p, const std::vector<float>& v) {
        if (int rc = alloc(p, v.size())) return rc;
        ODAM_HIP(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice, init_stream));   // behind the memset of the same buffer
        ODAM_HIP(hipStreamSynchronize(init_stream));        // v may be a temporary
        return 0;
    }
};

namespace {

const HostT* findw(odam_assoc* m, const std::string& n) {
    auto it = m->host.find(n);
    return it == m->host.end() ? nullptr : &it->second;
}
#define NEEDW(var, name)                                                                                  \
    const HostT* var = findw(m, name);                                                                     \
    if (!var) {                                                                                            \
        std::snprintf(g_odam_err, sizeof(g_odam_err), "odam_assoc_finalize: missing weight %s", std::string(name).c_str()); \
        return 1;                                                                                          \
    }

// Conv1d(k=1) weight [N, K(,1)] + bias -> Lin with K padded to Kp; optional row/column permutations
int pack(odam_assoc* m, Lin& L, const HostT& w, const HostT* b, int Kp, const std::vector<int>* row_perm,
         const std::vector<int>* col_perm) {
    const int N = (int)w.shape[0], K = (int)w.shape[1];
    std::vector<float> p((size_t)N * Kp, 0.0f), bb(N, 0.0f);
    for (int r = 0; r < N; r++) {
        const int sr = row_perm ? (*row_perm)[r] : r;
        for (int c = 0; c < K; c++) {
            const int sc = col_perm ? (*col_perm)[c] : c;
            p[(size_t)r * Kp + c] = w.data[(size_t)sr * K + sc];
        }
        if (b) bb[r] = b->data[sr];
    }
    L.K = Kp; L.N = N;
    if (int rc = m->upload(&L.w, p)) return rc;
    return m->upload(&L.b, bb);
}

// the reference views the 256 projected channels as (dim 64, head 4): channel c = d*4 + h (associator.py:77-81);
// the attention kernel wants head-major channels c' = h*64 + d: perm[c'] = d*4 + h
std::vector<int> head_perm() {
    std::vector<int> p(D);
    for (int h = 0; h < 4; h++)
        for (int d = 0; d < 64; d++) p[h * 64 + d] = d * 4 + h;
    return p;
}

int pack_prop(odam_assoc* m, Prop& P, const std::string& pre) {
    const std::vector<int> hp = head_perm();
    NEEDW(w0, pre + "attn.proj.0.weight"); NEEDW(b0, pre + "attn.proj.0.bias");
    NEEDW(w1, pre + "attn.proj.1.weight"); NEEDW(b1, pre + "attn.proj.1.bias");
    NEEDW(w2, pre + "attn.proj.2.weight"); NEEDW(b2, pre + "attn.proj.2.bias");
    HostT wqkv, bqkv;        // query, key and value projections of one row block: one [768, 256] layer, rows head-major
    wqkv.shape = {3 * D, D};
    for (const HostT* w : {w0, w1, w2})
        for (int r = 0; r < D; r++)
            wqkv.data.insert(wqkv.data.end(), w->data.begin() + (size_t)hp[r] * D, w->data.begin() + (size_t)(hp[r] + 1) * D);
    for (const HostT* b : {b0, b1, b2})
        for (int r = 0; r < D; r++) bqkv.data.push_back(b->data[hp[r]]);
    bqkv.shape = {3 * D};
    RC(pack(m, P.qkv, wqkv, &bqkv, D, nullptr, nullptr));
    NEEDW(wm, pre + "attn.merge.weight"); NEEDW(bm, pre + "attn.merge.bias");
    RC(pack(m, P.merge, *wm, bm, D, nullptr, &hp));      // its input channels arrive head-major
    NEEDW(m0w, pre + "mlp.0.weight"); NEEDW(m0b, pre + "mlp.0.bias");
    NEEDW(m2w, pre + "mlp.2.weight"); NEEDW(m2b, pre + "mlp.2.bias");
    if (m->merged) {
        // message = Wm att + bm feeds only the first MLP layer, h = relu(W0x x + W0m message + b0) (associator.py:92-97): the same function of
        // [x | att] with W0' = [W0x | W0m Wm] and b0' = b0 + W0m bm -- products in binary64, rounded to float32 once.  att arrives head-major
        // (column c' of the attention's output is the reference's channel hp[c']), so the folded columns are taken in that order.
        HostT w0f, b0f;
        w0f.shape = {2 * D, 2 * D}; b0f.shape = {2 * D};
        w0f.data.resize((size_t)2 * D * 2 * D); b0f.data.resize(2 * D);
        for (int o = 0; o < 2 * D; o++) {
            const float* w0row = m0w->data.data() + (size_t)o * 2 * D;
            for (int c = 0; c < D; c++) w0f.data[(size_t)o * 2 * D + c] = w0row[c];
            for (int c = 0; c < D; c++) {
                double s_ = 0.0;
                for (int r = 0; r < D; r++) s_ += (double)w0row[D + r] * (double)wm->data[(size_t)r * D + hp[c]];
                w0f.data[(size_t)o * 2 * D + D + c] = (float)s_;
            }
            double sb = (double)m0b->data[o];
            for (int r = 0; r < D; r++) sb += (double)w0row[D + r] * (double)bm->data[r];
            b0f.data[o] = (float)sb;
        }
        RC(pack(m, P.m0, w0f, &b0f, 2 * D, nullptr, nullptr));
    } else {
        RC(pack(m, P.m0, *m0w, m0b, 2 * D, nullptr, nullptr));
    }
    RC(pack(m, P.m2, *m2w, m2b, 2 * D, nullptr, nullptr));
    return 0;
}

// self-attention layer over Bt sequences of L rows (the fuser): message into cat_x[:, 256:], delta added in place
int prop_forward_self(const Prop& P, float* cat_x, int Mx, int Bt, int L, float* qkvb, float* attb, float* hb, hipStream_t st, bool merged) {
    RC(lin(P.qkv, cat_x, 512, Mx, nullptr, false, qkvb, 3 * D, nullptr, st));
    if (merged) {      // the attention's rows are the second half of the MLP's input as they stand (merge folded into P.m0: pack_prop)
        RC(odam_dk::launch_attention_d64(qkvb, 3 * D, qkvb + D, 3 * D, qkvb + 2 * D, 3 * D, cat_x + D, 512, Bt, 4, L, L, st));
    } else {
        RC(odam_dk::launch_attention_d64(qkvb, 3 * D, qkvb + D, 3 * D, qkvb + 2 * D, 3 * D, attb, D, Bt, 4, L, L, st));
        RC(lin(P.merge, attb, D, Mx, nullptr, false, cat_x + D, 512, nullptr, st));
    }
    RC(lin(P.m0, cat_x, 512, Mx, nullptr, true, hb, 2 * D, nullptr, st));
    RC(lin(P.m2, hb, 2 * D, Mx, cat_x, false, cat_x, 512, nullptr, st));
    return 0;
}

}  // namespace

extern "C" int odam_assoc_create(int max_tracks, int n_self_layers, const int* gnn_is_cross, int n_gnn_layers,
                                 int sinkhorn_iters, odam_assoc** out) {
    if (!out || max_tracks < 1 || max_tracks > 1024 || n_self_layers < 0 || n_gnn_layers < 0 || !gnn_is_cross)
        return odam_fail(1, "odam_assoc_create: bad argument (1 <= max_tracks <= 1024)");
    odam_assoc* m = new odam_assoc();
    m->max_tracks = max_tracks; m->n_self = n_self_layers; m->n_gnn = n_gnn_layers; m->iters = sinkhorn_iters;
    m->gnn_cross.assign(gnn_is_cross, gnn_is_cross + n_gnn_layers);
    *out = m;
    return 0;
}

extern "C" int odam_assoc_destroy(odam_assoc* m) {
    if (!m) return 0;
    if (m->lost_count) (void)hipHostFree(m->lost_count);
    if (m->init_stream) { (void)hipStreamSynchronize(m->init_stream); (void)hipStreamDestroy(m->init_stream); }
    for (void* p : m->allocs) (void)hipFree(p);
    delete m;
    return 0;
}

extern "C" int odam_assoc_set_weight(odam_assoc* m, const char* name, const float* data, const long long* shape, int ndim) {
    if (!m || !name || !data || ndim < 0 || ndim > 3 || m->finalized) return odam_fail(1, "odam_assoc_set_weight: bad argument");
    HostT t;
    size_t n = 1;
    for (int i = 0; i < ndim; i++) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data, data + n);
    m->host[name] = std::move(t);
    return 0;
}

extern "C" int odam_assoc_finalize(odam_assoc* m) {
    if (!m) return odam_fail(1, "odam_assoc_finalize: null model");
    if (m->finalized) return 0;
    {
        NEEDW(w0, "encoder.0.weight"); NEEDW(b0, "encoder.0.bias");
        NEEDW(w2, "encoder.2.weight"); NEEDW(b2, "encoder.2.bias");
        RC(pack(m, m->enc0, *w0, b0, FPAD, nullptr, nullptr));
        RC(pack(m, m->enc2, *w2, b2, D, nullptr, nullptr));
        NEEDW(fw, "final_proj.weight"); NEEDW(fb, "final_proj.bias");
        RC(pack(m, m->final_proj, *fw, fb, D, nullptr, nullptr));
        NEEDW(bs, "bin_score");
        m->bin_score = bs->data[0];
        NEEDW(dv, "pe_div_term");
        if (dv->data.size() != D / 2) return odam_fail(1, "pe_div_term must have 128 entries");
        RC(m->upload(&m->div_term, dv->data));
        RC(m->upload(&m->sc16, std::vector<float>(ND, 1.0f / 16.0f)));   // scores / descriptor_dim ** 0.5
    }
    m->merged = odam_cfg::get(odam_cfg::ASSOC_MERGE) != 0;
    for (int i = 0; i < m->n_self; i++) {
        Prop P;
        RC(pack_prop(m, P, "fuser.layers." + std::to_string(i) + "."));
        m->fuser.push_back(P);
    }
    for (int i = 0; i < m->n_gnn; i++) {
        Prop P;
        RC(pack_prop(m, P, "gnn.layers." + std::to_string(i) + "."));
        m->gnn.push_back(P);
    }
    const size_t T = m->max_tracks, N = T * NT;
    RC(m->alloc(&m->feat, (N + ND) * FPAD)); RC(m->alloc(&m->h256, (N + ND) * D)); RC(m->alloc(&m->catT, (N + ND) * 512));      // + the 30 detection rows behind the tracks'
    RC(m->alloc(&m->kv, N * 768)); RC(m->alloc(&m->att, N * D)); RC(m->alloc(&m->h512, N * 512));
    // fused tracks [T] and the 30 detection slots share one row block (detections start at row T of the frame) so the
    // shared-weight GNN layers see both sets as ONE matrix
    RC(m->alloc(&m->catTr, (T + ND) * 512)); RC(m->alloc(&m->featD, ND * FPAD));      // (featD: unused since the detections ride in feat's last 30 rows)
    RC(m->alloc(&m->kvX, (T + ND) * 768)); RC(m->alloc(&m->kvX2, (T + ND) * 768)); RC(m->alloc(&m->attX, (T + ND) * D));
    RC(m->alloc(&m->hX, (T + ND) * 512));
    RC(m->alloc(&m->mT, (T + ND + 2) * D)); RC(m->alloc(&m->scores, T * 32));   // the score block reads 32 detection rows
    ODAM_HIP(hipFuncSetAttribute((const void*)sinkhorn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    ODAM_HIP(hipFuncSetAttribute((const void*)sinkhorn32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    {
        float* b = nullptr;
        RC(m->alloc(&b, PG_BAR_WORDS));
        m->bar = reinterpret_cast<unsigned*>(b);
    }
    {
        float* b = nullptr;
        RC(m->alloc(&b, 256));
        m->stamps = reinterpret_cast<unsigned long long*>(b);
    }
    ODAM_HIP(hipHostMalloc((void**)&m->lost_count, 64, hipHostMallocDefault));
    *m->lost_count = 0u;
    // odam_config assoc.persist = 0 / odam_assoc_set_persistent(m, 0): the matching GNN as one launch per layer op (the
    // round-1 sequence; tests compare the two)
    m->persist = m->n_gnn <= PG_MAXL && odam_cfg::get(odam_cfg::ASSOC_PERSIST) != 0;
    {   // The persistent launch is a plain launch whose PG_WG workgroups wait for each other: all of them must be resident at
        // once.  How many fit is asked of the runtime for THIS kernel (246 VGPRs: two workgroups per CU), less one per CU
        // where it says more than one (MI355X_MICROARCH.md: the API reads one high for some SGPR counts), times the CUs;
        // a CU mask hides CUs from the launch without changing multiProcessorCount, so it turns the path off.
        int dev = 0, n_cu = 0, per_cu = 0;
        ODAM_HIP(hipGetDevice(&dev));
        ODAM_HIP(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        ODAM_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gnn_persistent_kernel<false>, PG_NT, 0));
        m->resident_capacity = (per_cu > 1 ? per_cu - 1 : per_cu) * n_cu;
        if (getenv("HSA_CU_MASK") != nullptr || getenv("ROC_GLOBAL_CU_MASK") != nullptr) m->resident_capacity = 0;
        if (m->resident_capacity < PG_WG) m->persist = false;
    }
    m->host.clear();
    // every buffer alloc() zeroed is zero from here on (see alloc)
    RC(m->init_done());
    m->finalized = true;
    return 0;
}

// the launch sequence of one forward: ~65 kernels, every one a memory round trip long at these sizes (<= 70 rows)
static int enqueue_forward(odam_assoc* m, const float* tracks, int T, const float* detections, int n_det, const int* n_dev,
                           float* Z_out, hipStream_t st, bool allow_persist = true) {
    const int N = T * NT;
    // encoder + frame-index encoding (associator.py:222-229)
    float* X = m->catTr;                      // [T + 30][512]: fused tracks, then the detection slots (rows T ..)
    const int MX = T + ND;
    // The detections go through the encoder as 30 more rows of the tracks' matrix (rows N .. N + 29 of feat / catT: same weights, rows are
    // independent, and the same tile shape serves 30 and 100 T + 30 rows -- one launch per layer instead of two); time_mean_kernel moves
    // their encodings behind the track means afterwards.
    // (the same three launches on a second stream beside the tracks' branch, forked and joined by events: measured, no change; not kept)
    hipLaunchKernelGGL(prepare_kernel, dim3(N + ND), dim3(256), 0, st, tracks, NT, N, m->div_term, m->feat, m->catT, detections, ND, ND);
    ODAM_HIP(hipGetLastError());
    RC(lin(m->enc0, m->feat, FPAD, N + ND, nullptr, true, m->h256, D, nullptr, st));
    RC(lin(m->enc2, m->h256, D, N + ND, m->catT, false, m->catT, 512, nullptr, st));           // + pe, in place
    // fuser: self-attention over each track's 100 time steps (associator.py:143-160, 230)
    for (const Prop& P : m->fuser)
        RC(prop_forward_self(P, m->catT, N, T, NT, m->kv, m->att, m->h512, st, m->merged));
    hipLaunchKernelGGL(time_mean_kernel, dim3(T + ND), dim3(256), 0, st, m->catT, NT, X, T);
    ODAM_HIP(hipGetLastError());
    // matching GNN between the fused tracks [T] and all 30 detection slots (associator.py:111-139, 240).  Both sets
    // go through the same weights, so every projection / MLP runs once on the [T + 30] row block; only the attention
    // differs per side (self: own set, cross: the other set).  All deltas come from the layer's inputs: the query,
    // key and value projections are taken before the residual update of either set.
    if (m->persist && allow_persist && !n_dev) {
        GnnArgs g{};
        for (size_t i = 0; i < m->gnn.size(); i++) {
            const Prop& P = m->gnn[i];
            g.L[i] = GnnLayerW{P.qkv.w, P.qkv.b, P.merge.w, P.merge.b, P.m0.w, P.m0.b, P.m2.w, P.m2.b, m->gnn_cross[i] != 0 ? 1 : 0};
        }
        g.n_layers = (int)m->gnn.size();
        g.fin_w = m->final_proj.w; g.fin_b = m->final_proj.b;
        g.X = X; g.kv = m->kvX; g.kv2 = m->kvX2; g.att = m->attX; g.h = m->hX; g.mT = m->mT;
        g.T = T;
        g.bar = m->bar; g.timeout_ticks = m->timeout_ticks;
        g.stamps = m->want_stamps ? m->stamps : nullptr;
        g.fake_misplaced = m->fake_misplaced ? 1 : 0;
        g.merged = m->merged ? 1 : 0;
        // counters and the error flag start from zero every launch: zeroed here, unless the Sinkhorn kernel behind the previous launch did it
        if (!m->bar_clean) ODAM_HIP(hipMemsetAsync(m->bar, 0, sizeof(unsigned) * PG_BAR_WORDS, st));
        m->bar_clean = false;
        const bool rowpart = odam_cfg::get(odam_cfg::ASSOC_PERSIST) == 2;
        if (m->merged) {
            if (rowpart) hipLaunchKernelGGL(gnn_rowpart_kernel<true>, dim3(PG_WG), dim3(PG_NT), 0, st, g);
            else hipLaunchKernelGGL(gnn_persistent_kernel<true>, dim3(PG_WG), dim3(PG_NT), 0, st, g);
        } else {
            if (rowpart) hipLaunchKernelGGL(gnn_rowpart_kernel<false>, dim3(PG_WG), dim3(PG_NT), 0, st, g);
            else hipLaunchKernelGGL(gnn_persistent_kernel<false>, dim3(PG_WG), dim3(PG_NT), 0, st, g);
        }
        ODAM_HIP(hipGetLastError());
        // The score matrix keeps its own launch on the tiles of conv_gemm.hip: with saturated scores (the hand-built scene
        // weights reach +-1000) the Sinkhorn loop shares a detection's mass equally among several tracks and which of them
        // the Hungarian step then picks hangs on the last bit of the scores -- the reference-run fixtures (tests/test_e2e.py)
        // hold for the summation order of that kernel, and a different order moved 20 of 40 frames' tie-breaks.
        const float* mD = m->mT + (size_t)T * D;
        Lin sc; sc.w = const_cast<float*>(mD); sc.b = nullptr; sc.K = D; sc.N = ND;
        RC(lin(sc, m->mT, D, T, nullptr, false, m->scores, 32, m->sc16, st));
        bool cleaned = false;
        const int rc = launch_sinkhorn(m->scores, 32, T, n_det, n_det, m->bin_score, m->iters, Z_out, nullptr, st, m->bar + 1, m->lost_count, &cleaned);
        m->bar_clean = rc == 0 && cleaned;
        return rc;
    }
    for (size_t i = 0; i < m->gnn.size(); i++) {
        const Prop& P = m->gnn[i];
        const bool cross = m->gnn_cross[i] != 0;
        RC(lin(P.qkv, X, 512, MX, nullptr, false, m->kvX, 3 * D, nullptr, st));     // rows: q | k | v
        const float* srcT = cross ? m->kvX + (size_t)T * 3 * D : m->kvX;     // source rows of the track queries
        const float* srcD = cross ? m->kvX : m->kvX + (size_t)T * 3 * D;     // ... of the detection queries
        const int nT = cross ? ND : T, nD = cross ? T : ND;
        float* attO = m->merged ? X + D : m->attX;
        const int ldO = m->merged ? 512 : D;
        RC(odam_dk::launch_attention_d64(m->kvX, 3 * D, srcT + D, 3 * D, srcT + 2 * D, 3 * D, attO, ldO, 1, 4, T, nT, st));
        RC(odam_dk::launch_attention_d64(m->kvX + (size_t)T * 3 * D, 3 * D, srcD + D, 3 * D, srcD + 2 * D, 3 * D,
                                         attO + (size_t)T * ldO, ldO, 1, 4, ND, nD, st));
        if (!m->merged) RC(lin(P.merge, m->attX, D, MX, nullptr, false, X + D, 512, nullptr, st));
        RC(lin(P.m0, X, 512, MX, nullptr, true, m->hX, 2 * D, nullptr, st));
        RC(lin(P.m2, m->hX, 2 * D, MX, X, false, X, 512, nullptr, st));
    }
    // descriptors, scores, optimal transport (associator.py:242-254)
    RC(lin(m->final_proj, X, 512, MX, nullptr, false, m->mT, D, nullptr, st));
    const float* mD = m->mT + (size_t)T * D;
    Lin sc; sc.w = const_cast<float*>(mD); sc.b = nullptr; sc.K = D; sc.N = ND;
    RC(lin(sc, m->mT, D, T, nullptr, false, m->scores, 32, m->sc16, st));
    return launch_sinkhorn(m->scores, 32, T, n_det, n_det, m->bin_score, m->iters, Z_out, n_dev, st);   // n_dev (graph replay): n_det is the capacity, 30
}

// One frame (stream-ordered).  (Replaying the launch sequence from a hipGraph per track count was built in round 2 and measured
// no gain -- the sequence is bound by ~65 dependent kernels of ~10 us on the device, not by host launches -- and is gone.)
extern "C" int odam_assoc_forward(odam_assoc* m, const float* tracks, int T, const float* detections, int n_det,
                                  float* Z_out, void* stream) {
    if (!m || !tracks || !detections || !Z_out) return odam_fail(1, "odam_assoc_forward: null pointer");
    if (!m->finalized) return odam_fail(1, "odam_assoc_forward: call odam_assoc_finalize first");
    if (T < 1 || T > m->max_tracks || n_det < 1 || n_det > ND) return odam_fail(3, "odam_assoc_forward: T / n_det out of range");
    return enqueue_forward(m, tracks, T, detections, n_det, nullptr, Z_out, (hipStream_t)stream);
}

// The same forward with the matching layers as separate launches (no device-wide barrier, no residency assumption): what
// the host re-runs a frame through when odam_assoc_lost_launches has moved.
extern "C" int odam_assoc_forward_sequence(odam_assoc* m, const float* tracks, int T, const float* detections, int n_det,
                                           float* Z_out, void* stream) {
    if (!m || !tracks || !detections || !Z_out) return odam_fail(1, "odam_assoc_forward_sequence: null pointer");
    if (!m->finalized) return odam_fail(1, "odam_assoc_forward_sequence: call odam_assoc_finalize first");
    if (T < 1 || T > m->max_tracks || n_det < 1 || n_det > ND) return odam_fail(3, "odam_assoc_forward_sequence: T / n_det out of range");
    return enqueue_forward(m, tracks, T, detections, n_det, nullptr, Z_out, (hipStream_t)stream, false);
}

extern "C" int odam_assoc_lost_launches(odam_assoc* m, unsigned* count) {
    if (!m || !m->finalized || !count) return odam_fail(1, "odam_assoc_lost_launches: bad argument");
    *count = __atomic_load_n(m->lost_count, __ATOMIC_ACQUIRE);
    return 0;
}

extern "C" int odam_assoc_set_persistent(odam_assoc* m, int on) {
    if (!m || !m->finalized) return odam_fail(1, "odam_assoc_set_persistent: bad argument");
    m->persist = on != 0 && m->n_gnn <= PG_MAXL && m->resident_capacity >= PG_WG;
    return 0;
}

extern "C" int odam_assoc_info(odam_assoc* m, int* persistent, int* resident_capacity, int* workgroups) {
    if (!m || !m->finalized) return odam_fail(1, "odam_assoc_info: bad argument");
    if (persistent) *persistent = m->persist ? 1 : 0;
    if (resident_capacity) *resident_capacity = m->resident_capacity;
    if (workgroups) *workgroups = PG_WG;
    return 0;
}

extern "C" int odam_assoc_debug_misplace(odam_assoc* m, int on) {
    if (!m) return odam_fail(1, "odam_assoc_debug_misplace: null handle");
    m->fake_misplaced = on != 0;
    return 0;
}

extern "C" int odam_assoc_set_barrier_timeout_us(odam_assoc* m, long long us) {
    if (!m || us < 0) return odam_fail(1, "odam_assoc_set_barrier_timeout_us: bad argument");
    m->timeout_ticks = (unsigned long long)us * 100ull;
    return 0;
}

// ---- device-resident track windows (OdamProcess._preprocess_tracks, src/processor.py:133-170) --------------------------------
// Every frame the reference rebuilds, on the host, the associator's track input [T, 79, 100]: the last 100 observations of every
// live track moved into the CURRENT camera frame (centre through inv(T_wc), azimuth relative to the camera's, the box replaced
// by the projected extent of the fitted surface).  At 40 tracks that is a 1.3 MB tensor built in numpy and uploaded per frame.
// Here the observations stay on the device in the WORLD frame (append-only ring per track: 14 float64 per observation), and one
// launch builds the tensor from them: the same float64 arithmetic per value, rounded to float32 once.
struct odam_trackwin {
    int max_tracks = 0, window = 0;
    double* rows = nullptr;     // [max_tracks][window][14]: frame id, class, bbox px x4, dims x3, t_wo x3, az_wo, score
    int* count = nullptr;       // [max_tracks] observations appended so far
    double* stage = nullptr;    // pinned [8][32][14 + 1]: rows + track id of one append
    double* d_stage = nullptr;  // (unused since the append kernel reads the pinned slot itself; still allocated)
    double* d_cam = nullptr;    // [16] (unused since the camera travels in the kernel arguments; kept so the handle layout of create / destroy stays as tested)
    double* h_load = nullptr;   // pinned [max_tracks][window][14]: what odam_trackwin_load's kernel reads (mapped host memory; allocated once -- the
    int* h_load_meta = nullptr; // load path has no allocation, no hipFree (it waits for every stream of the device) and no copy command)
    unsigned slot = 0;
    hipEvent_t copied[8] = {};  // recorded behind the upload from pinned slot i: the host rewrites a slot only after its copy has run
    bool copied_armed[8] = {};
    // host side of OdamProcess._prepare_tracks (src/processor.py:172-180): what the surface of a track is evaluated from -- the
    // means of all its observations' centre, azimuth and dimensions -- kept as running sums, extended by every append
    struct Sums { long n = 0; double st[3] = {0, 0, 0}, sd[3] = {0, 0, 0}; std::vector<double> az; };
    std::vector<Sums> sums;
    bool sums_ok = true;        // false once an append skipped ids or a load came without the full columns
    float* h_params = nullptr;  // pinned [8][max_tracks][9]
    float* d_params = nullptr;  // [max_tracks][9] (unused since the surface kernel reads the pinned rows itself; still allocated)
    double* d_proj = nullptr;   // [max_tracks][4]
    hipEvent_t params_copied[8] = {};
    bool params_armed[8] = {};
    unsigned params_slot = 0;
};

namespace {
constexpr int TW_COLS = 14;
__global__ void trackwin_append_kernel(const double* __restrict__ st, int n, double* __restrict__ rows, int* __restrict__ count, int window) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const int tid_ = (int)st[i * (TW_COLS + 1) + TW_COLS];
    const int c = count[tid_];      // one block per observation; observations of one append go to different tracks
    if (threadIdx.x < TW_COLS) rows[((size_t)tid_ * window + (c % window)) * TW_COLS + threadIdx.x] = st[i * (TW_COLS + 1) + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) count[tid_] = c + 1;
}

// track t: its last k = min(c, window) observations to the ring positions an append sequence of c rows would have left them in
__global__ __launch_bounds__(128) void trackwin_load_kernel(const double* __restrict__ src, const int* __restrict__ meta, int T,
                                                            double* __restrict__ rows, int* __restrict__ count, int window) {
    const int t = blockIdx.x, l = threadIdx.x;
    const int first = meta[t], c = meta[T + t], k = c < window ? c : window;
    if (l < k) {
        const double* r = src + (size_t)(first + l) * TW_COLS;
        double* o = rows + ((size_t)t * window + ((c - k + l) % window)) * TW_COLS;
#pragma unroll
        for (int j = 0; j < TW_COLS; j++) o[j] = r[j];
    }
    if (l == 0) count[t] = c;
}

// out [T][79][window] float32; cam: T_cw rows 0..2 (12), cam_azi, img_w, img_h
struct Cam15 { double v[15]; };      // travels in the kernel arguments
__global__ __launch_bounds__(128) void trackwin_build_kernel(const double* __restrict__ rows, const int* __restrict__ count, int window,
                                                             const double* __restrict__ proj_px, const Cam15 camv,
                                                             float* __restrict__ out) {
    const double* cam = camv.v;
    const int t = blockIdx.x, l = threadIdx.x;
    if (l >= window) return;
    const int c = count[t], k = c < window ? c : window;
    float* o = out + (size_t)t * 79 * window + l;
    if (l >= k) {
#pragma unroll 1
        for (int ch = 0; ch < 79; ch++) o[(size_t)ch * window] = -1.0f;
        return;
    }
    const double* r = rows + ((size_t)t * window + ((c - k + l) % window)) * TW_COLS;
    const double iw = cam[13], ih = cam[14];
    auto clip = [](double x) { return x < -1.0 ? -1.0 : (x > 2.0 ? 2.0 : x); };
    double v[15];
    v[0] = r[0]; v[1] = r[1];
    v[2] = clip(proj_px[t * 4 + 0] / iw); v[3] = clip(proj_px[t * 4 + 1] / ih);
    v[4] = clip(proj_px[t * 4 + 2] / iw); v[5] = clip(proj_px[t * 4 + 3] / ih);
    v[6] = r[6]; v[7] = r[7]; v[8] = r[8];
    const double x = r[9], y = r[10], z = r[11];
    v[9] = x * cam[0] + y * cam[1] + z * cam[2] + cam[3];
    v[10] = x * cam[4] + y * cam[5] + z * cam[6] + cam[7];
    v[11] = x * cam[8] + y * cam[9] + z * cam[10] + cam[11];
    const double rel = r[12] - cam[12];
    v[12] = sin(rel); v[13] = cos(rel);
    v[14] = r[13];
#pragma unroll
    for (int ch = 0; ch < 15; ch++) o[(size_t)ch * window] = (float)v[ch];
#pragma unroll 1
    for (int ch = 15; ch < 79; ch++) o[(size_t)ch * window] = -1.0f;
}
}  // namespace

extern "C" int odam_trackwin_create(int max_tracks, int window, odam_trackwin** out) {
    if (!out || max_tracks < 1 || max_tracks > 4096 || window < 1 || window > 128) return odam_fail(1, "odam_trackwin_create: bad argument");
    odam_trackwin* w = new odam_trackwin();
    w->max_tracks = max_tracks; w->window = window;
    if (hipMalloc((void**)&w->rows, sizeof(double) * (size_t)max_tracks * window * TW_COLS) != hipSuccess ||
        hipMalloc((void**)&w->count, sizeof(int) * (size_t)max_tracks) != hipSuccess ||
        hipMalloc((void**)&w->d_stage, sizeof(double) * 32 * (TW_COLS + 1)) != hipSuccess ||
        hipMalloc((void**)&w->d_cam, sizeof(double) * 16) != hipSuccess ||
        hipHostMalloc((void**)&w->h_load, sizeof(double) * (size_t)max_tracks * window * TW_COLS, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&w->h_load_meta, sizeof(int) * 2 * (size_t)max_tracks, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&w->stage, sizeof(double) * 8 * 32 * (TW_COLS + 1), hipHostMallocDefault) != hipSuccess) {
        if (w->rows) (void)hipFree(w->rows);
        if (w->count) (void)hipFree(w->count);
        if (w->d_stage) (void)hipFree(w->d_stage);
        if (w->d_cam) (void)hipFree(w->d_cam);
        if (w->h_load) (void)hipHostFree(w->h_load);
        if (w->h_load_meta) (void)hipHostFree(w->h_load_meta);
        if (w->stage) (void)hipHostFree(w->stage);
        delete w;
        return odam_fail(2, "odam_trackwin_create: allocation failed");
    }
    // counts start at zero, and are zero when this call returns: through a private non-blocking stream -- hipMemset would be
    // ordered on the NULL stream, may still be pending when it returns and then land behind a load / append the caller issues on
    // a stream of its own (it did: the counts of a freshly loaded store went back to zero -- GPU test
    // test_stores_created_while_the_default_stream_is_busy); waiting for the NULL stream instead would wait for whatever another
    // thread has queued there
    {
        hipStream_t init = nullptr;
        const bool ok = hipStreamCreateWithFlags(&init, hipStreamNonBlocking) == hipSuccess &&
                        hipMemsetAsync(w->count, 0, sizeof(int) * (size_t)max_tracks, init) == hipSuccess &&
                        hipStreamSynchronize(init) == hipSuccess;
        if (init) (void)hipStreamDestroy(init);
        if (!ok) {
            (void)hipFree(w->rows); (void)hipFree(w->count); (void)hipFree(w->d_stage); (void)hipFree(w->d_cam); (void)hipHostFree(w->h_load); (void)hipHostFree(w->h_load_meta); (void)hipHostFree(w->stage);
            delete w;
            return odam_fail(2, "odam_trackwin_create: initialisation failed");
        }
    }
    for (int i = 0; i < 8; i++)
        if (hipEventCreateWithFlags(&w->copied[i], hipEventDisableTiming) != hipSuccess) {
            for (int j = 0; j < i; j++) (void)hipEventDestroy(w->copied[j]);
            (void)hipFree(w->rows); (void)hipFree(w->count); (void)hipFree(w->d_stage); (void)hipFree(w->d_cam); (void)hipHostFree(w->h_load); (void)hipHostFree(w->h_load_meta); (void)hipHostFree(w->stage);
            delete w;
            return odam_fail(2, "odam_trackwin_create: event creation failed");
        }
    *out = w;
    return 0;
}

extern "C" int odam_trackwin_destroy(odam_trackwin* w) {
    if (!w) return 0;
    for (int i = 0; i < 8; i++) {
        if (w->params_armed[i]) (void)hipEventSynchronize(w->params_copied[i]);
        if (w->params_copied[i]) (void)hipEventDestroy(w->params_copied[i]);
    }
    if (w->h_params) (void)hipHostFree(w->h_params);
    if (w->d_params) (void)hipFree(w->d_params);
    if (w->d_proj) (void)hipFree(w->d_proj);
    for (int i = 0; i < 8; i++) {
        if (w->copied_armed[i]) (void)hipEventSynchronize(w->copied[i]);      // no upload may still be reading the pinned ring
        (void)hipEventDestroy(w->copied[i]);
    }
    (void)hipFree(w->rows); (void)hipFree(w->count); (void)hipFree(w->d_stage); (void)hipFree(w->d_cam); (void)hipHostFree(w->h_load); (void)hipHostFree(w->h_load_meta); (void)hipHostFree(w->stage);
    delete w;
    return 0;
}

extern "C" int odam_trackwin_reset(odam_trackwin* w, void* stream) {
    if (!w) return odam_fail(1, "odam_trackwin_reset: null handle");
    ODAM_HIP(hipMemsetAsync(w->count, 0, sizeof(int) * (size_t)w->max_tracks, (hipStream_t)stream));
    w->sums.clear(); w->sums_ok = true;
    return 0;
}

extern "C" int odam_trackwin_append(odam_trackwin* w, int n, const int* track_ids, const double* rows14, void* stream) {
    if (!w || n < 0 || n > 32 || (n && (!track_ids || !rows14))) return odam_fail(1, "odam_trackwin_append: bad argument (at most 32 observations per call)");
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; i++)
        if (track_ids[i] < 0 || track_ids[i] >= w->max_tracks) return odam_fail(3, "odam_trackwin_append: track id outside the handle's capacity");
    // The ring slot may be rewritten only once the kernel that read it eight appends ago has run -- appends can queue behind long
    // kernels on the stream (a detector forward, a rebuild).
    const int si = (int)(w->slot++ & 7);
    if (w->copied_armed[si]) ODAM_HIP(hipEventSynchronize(w->copied[si]));
    double* slot = w->stage + (size_t)si * 32 * (TW_COLS + 1);
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < TW_COLS; c++) slot[i * (TW_COLS + 1) + c] = rows14[i * TW_COLS + c];
        slot[i * (TW_COLS + 1) + TW_COLS] = (double)track_ids[i];
    }
    // the kernel reads the pinned slot itself (mapped host memory; <= 3.8 KB): no copy command in the frame's chain.  The slot is the
    // host's again once that kernel has run -- the event behind it is what the next use of the slot waits for.
    hipLaunchKernelGGL(trackwin_append_kernel, dim3(n), dim3(64), 0, st, slot, n, w->rows, w->count, w->window);
    ODAM_HIP(hipGetLastError());
    ODAM_HIP(hipEventRecord(w->copied[si], st));
    w->copied_armed[si] = true;
    for (int i = 0; i < n; i++) {
        const size_t t = (size_t)track_ids[i];
        if (t > w->sums.size()) w->sums_ok = false;        // a gap in the ids: no sums for the skipped tracks
        if (t >= w->sums.size()) w->sums.resize(t + 1);
        odam_trackwin::Sums& q = w->sums[t];
        const double* r = rows14 + (size_t)i * TW_COLS;
        q.n++;
        for (int c = 0; c < 3; c++) { q.sd[c] += r[6 + c]; q.st[c] += r[9 + c]; }      // row after row: numpy's axis-0 reduce of an [n, 3] block
        q.az.push_back(r[12]);
    }
    return 0;
}

// Bulk (re)build of the mirror: track t gets the last min(lengths[t], window) of its observations -- rows14 holds exactly those,
// track after track -- and the count lengths[t].  One upload and one launch; synchronises the stream (a rebuild is rare: first
// use, or the host edited its track list).
extern "C" int odam_trackwin_load(odam_trackwin* w, int T, const int* lengths, const double* rows14, void* stream) {
    if (!w || T < 0 || T > w->max_tracks || (T && (!lengths || !rows14))) return odam_fail(1, "odam_trackwin_load: bad argument");
    hipStream_t st = (hipStream_t)stream;
    ODAM_HIP(hipMemsetAsync(w->count, 0, sizeof(int) * (size_t)w->max_tracks, st));
    w->sums.clear(); w->sums_ok = T == 0;
    if (T == 0) return 0;
    std::vector<int> meta(2 * (size_t)T);      // [t] = first row of track t in rows14, [T + t] = its length
    long total = 0;
    for (int t = 0; t < T; t++) {
        if (lengths[t] < 0) return odam_fail(1, "odam_trackwin_load: negative track length");
        meta[t] = (int)total; meta[T + t] = lengths[t];
        total += lengths[t] < w->window ? lengths[t] : w->window;
    }
    // The kernel reads the handle's pinned staging itself (mapped host memory, sized for max_tracks full windows at creation): no
    // allocation here, no hipFree (it waits for every stream of the device -- for a detector running beside this caller) and no
    // copy command (free: every load ends with a stream synchronisation, so the staging is the caller's again when this returns).
    int rc = 0;
    std::memcpy(w->h_load, rows14, sizeof(double) * (size_t)total * TW_COLS);
    std::memcpy(w->h_load_meta, meta.data(), sizeof(int) * 2 * (size_t)T);
    {
        hipLaunchKernelGGL(trackwin_load_kernel, dim3(T), dim3(128), 0, st, w->h_load, w->h_load_meta, T, w->rows, w->count, w->window);
        if (hipGetLastError() != hipSuccess) rc = odam_fail(2, "odam_trackwin_load: launch failed");
    }
    if (hipStreamSynchronize(st) != hipSuccess && !rc) rc = odam_fail(2, "odam_trackwin_load: stream synchronisation failed");
    return rc;
}

// np.add.reduce of a 1-D float64 array (numpy/core/src/umath/loops_utils.h.src, pairwise sum, started from the identity):
// fewer than 8 values one after the other; up to 128 in eight interleaved partial sums combined as a tree, the tail after; longer
// arrays halved (the first half a multiple of 8) -- tests/test_assoc_gpu.py checks it against numpy bit for bit
static double np_pairwise(const double* a, size_t n) {
    if (n < 8) {
        double r = 0.0;
        for (size_t i = 0; i < n; i++) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        size_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise(a, n2) + np_pairwise(a + n2, n - n2);
}

// the sums of a host track list in one go (after odam_trackwin_load): cols7 [host][sum_t lengths[t]][7] = columns 6..12 (dims x3,
// t_wo x3, az_wo) of EVERY observation of track 0, then of track 1, ...
extern "C" int odam_trackwin_load_sums(odam_trackwin* w, int T, const int* lengths, const double* cols7) {
    if (!w || T < 0 || T > w->max_tracks || (T && (!lengths || !cols7))) return odam_fail(1, "odam_trackwin_load_sums: bad argument");
    w->sums.assign((size_t)T, odam_trackwin::Sums());
    const double* r = cols7;
    for (int t = 0; t < T; t++) {
        odam_trackwin::Sums& q = w->sums[t];
        if (lengths[t] < 1) return odam_fail(1, "odam_trackwin_load_sums: a track without observations");
        q.n = lengths[t];
        q.az.resize((size_t)lengths[t]);
        for (int i = 0; i < lengths[t]; i++, r += 7) {
            for (int c = 0; c < 3; c++) { q.sd[c] += r[c]; q.st[c] += r[3 + c]; }
            q.az[i] = r[6];
        }
    }
    w->sums_ok = true;
    return 0;
}

// parameter rows the surfaces of the tracks are evaluated from, as sq.init_params / _prepare_tracks build them (processor.py:172-180):
// mean centre, mean azimuth, sqrt(max(mean dims, 0.05) / 2), shape exponents -0;  out [host][T][9] float32
extern "C" int odam_trackwin_params(odam_trackwin* w, int T, float* out) {
    if (!w || T < 0 || (T && !out)) return odam_fail(1, "odam_trackwin_params: bad argument");
    if (!w->sums_ok || (size_t)T != w->sums.size()) return odam_fail(4, "odam_trackwin_params: the running sums do not cover these tracks (load them: odam_trackwin_load_sums)");
    for (int t = 0; t < T; t++) {
        const odam_trackwin::Sums& q = w->sums[t];
        const double n = (double)q.n;
        float* o = out + (size_t)t * 9;
        for (int c = 0; c < 3; c++) o[c] = (float)(q.st[c] / n);
        o[3] = (float)((0.0 + np_pairwise(q.az.data(), q.az.size())) / n);
        for (int c = 0; c < 3; c++) {
            const double d = q.sd[c] / n;
            o[4 + c] = (float)std::sqrt((d > 0.05 ? d : 0.05) / 2);       // np.maximum(., 0.05); NaN stays NaN in numpy -- not reachable, dims come from a sigmoid
        }
        o[7] = o[8] = -0.0f;
    }
    return 0;
}

// One call for OdamProcess._prepare_tracks: parameter rows (above) -> upload -> odam_sq_project_extents on `sq` -> the window tensor
// out [dev][T][79][window] for the camera T_cw12_K9 = rows 0..2 of inv(T_wc) (12) + K (9), azimuth cam_azi, image size.  Stream-ordered.
extern "C" int odam_sq_project_extents(struct odam_sq_ctx* ctx, int n, const float* params, const double* T_cw12_K9, double* out_px, void* stream);
extern "C" int odam_trackwin_build_tracks(odam_trackwin* w, struct odam_sq_ctx* sq, int T, const double* T_cw12_K9, double cam_azi,
                                          double img_w, double img_h, float* out, void* stream) {
    if (!w || !sq || T < 0 || T > w->max_tracks || (T && (!T_cw12_K9 || !out))) return odam_fail(1, "odam_trackwin_build_tracks: bad argument");
    if (T == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!w->h_params) {
        if (hipHostMalloc((void**)&w->h_params, sizeof(float) * 8 * (size_t)w->max_tracks * 9, hipHostMallocDefault) != hipSuccess ||
            hipMalloc((void**)&w->d_params, sizeof(float) * (size_t)w->max_tracks * 9) != hipSuccess ||
            hipMalloc((void**)&w->d_proj, sizeof(double) * (size_t)w->max_tracks * 4) != hipSuccess) {
            if (w->h_params) { (void)hipHostFree(w->h_params); w->h_params = nullptr; }
            if (w->d_params) { (void)hipFree(w->d_params); w->d_params = nullptr; }
            return odam_fail(2, "odam_trackwin_build_tracks: allocation failed");
        }
        for (int i = 0; i < 8; i++) ODAM_HIP(hipEventCreateWithFlags(&w->params_copied[i], hipEventDisableTiming));
    }
    const int si = (int)(w->params_slot++ & 7);
    if (w->params_armed[si]) ODAM_HIP(hipEventSynchronize(w->params_copied[si]));
    float* hp = w->h_params + (size_t)si * w->max_tracks * 9;
    if (int rc = odam_trackwin_params(w, T, hp)) return rc;
    // the surface kernel reads the pinned rows itself (mapped host memory, 36 bytes per track): no copy command; the slot is free
    // again when the launches that read it have run
    if (int rc = odam_sq_project_extents(sq, T, hp, T_cw12_K9, w->d_proj, stream)) return rc;
    ODAM_HIP(hipEventRecord(w->params_copied[si], st));
    w->params_armed[si] = true;
    Cam15 cam;
    for (int i = 0; i < 12; i++) cam.v[i] = T_cw12_K9[i];
    cam.v[12] = cam_azi; cam.v[13] = img_w; cam.v[14] = img_h;
    hipLaunchKernelGGL(trackwin_build_kernel, dim3(T), dim3(128), 0, st, w->rows, w->count, w->window, w->d_proj, cam, out);
    ODAM_HIP(hipGetLastError());
    return 0;
}

// One frame of OdamProcess.process_frame's device work behind ONE call (src/processor.py:320-337): the observations the previous frame
// attached (n_app <= 30 rows: odam_trackwin_append), this frame's track input for its camera (odam_trackwin_build_tracks) and the
// association forward on it (odam_assoc_forward) -- the same three entry points in the same order, so the results are theirs bit for
// bit; what goes is the host time between them (three binding calls, their argument marshalling and the interpreter in between, with the
// device idle).  tracks_out [dev][T][79][window] is the caller's buffer (it needs it again if the frame has to be re-run through
// odam_assoc_forward_sequence); detections / Z_out as in odam_assoc_forward (device or mapped pinned host memory).
extern "C" int odam_assoc_step(odam_assoc* m, odam_trackwin* w, struct odam_sq_ctx* sq, int n_app, const int* app_ids, const double* app_rows14,
                               int T, const double* T_cw12_K9, double cam_azi, double img_w, double img_h, const float* detections, int n_det,
                               float* tracks_out, float* Z_out, void* stream) {
    if (!m || !w || !sq || !tracks_out || !detections || !Z_out) return odam_fail(1, "odam_assoc_step: null pointer");
    if (n_app < 0 || n_app > 32) return odam_fail(1, "odam_assoc_step: at most 32 observations per frame");
    if (n_app) { if (int rc = odam_trackwin_append(w, n_app, app_ids, app_rows14, stream)) return rc; }
    if (int rc = odam_trackwin_build_tracks(w, sq, T, T_cw12_K9, cam_azi, img_w, img_h, tracks_out, stream)) return rc;
    return odam_assoc_forward(m, tracks_out, T, detections, n_det, Z_out, stream);
}

extern "C" int odam_trackwin_build(odam_trackwin* w, int T, const double* proj_px, const double* cam15, float* out, void* stream) {
    if (!w || T < 0 || T > w->max_tracks || (T && (!proj_px || !cam15 || !out))) return odam_fail(1, "odam_trackwin_build: bad argument");
    if (T == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    Cam15 cam;
    for (int i = 0; i < 15; i++) cam.v[i] = cam15[i];
    hipLaunchKernelGGL(trackwin_build_kernel, dim3(T), dim3(128), 0, st, w->rows, w->count, w->window, proj_px, cam, out);
    ODAM_HIP(hipGetLastError());
    return 0;
}

// diagnostics: run the next forwards with stage stamps (enable != 0), or read the stamps of the last one: out[0..n) =
// 100 MHz timer of workgroup 0 at kernel start and after every stage of the persistent matching kernel
extern "C" int odam_assoc_stage_stamps(odam_assoc* m, int enable, unsigned long long* out, int n) {
    if (!m || !m->finalized || n < 0 || n > 128) return odam_fail(1, "odam_assoc_stage_stamps: bad argument");
    m->want_stamps = enable != 0;
    if (out && n) {
        ODAM_HIP(hipDeviceSynchronize());
        ODAM_HIP(hipMemcpy(out, m->stamps, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost));
    }
    return 0;
}

// diagnostics: a workspace buffer of the last forward ([host] out[n]): 0 = row block X [(T + 30), 512], 1 = descriptors mT
// [(T + 30), 256], 2 = scores [T, 32]
extern "C" int odam_assoc_debug_read(odam_assoc* m, int which, float* out, long long n) {
    if (!m || !m->finalized || !out || n <= 0 || which < 0 || which > 2) return odam_fail(1, "odam_assoc_debug_read: bad argument");
    const float* src = which == 0 ? m->catTr : (which == 1 ? m->mT : m->scores);
    ODAM_HIP(hipDeviceSynchronize());
    ODAM_HIP(hipMemcpy(out, src, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int odam_assoc_sinkhorn(const float* scores, int lds_, int m_, int n_, float alpha, int iters, float* Z_out,
                                   void* stream) {
    if (!scores || !Z_out || m_ < 1 || n_ < 1 || (size_t)(m_ + 1) * (n_ + 1) > 36000)
        return odam_fail(1, "odam_assoc_sinkhorn: bad argument");
    static bool attr = false;
    if (!attr) {
        ODAM_HIP(hipFuncSetAttribute((const void*)sinkhorn32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        ODAM_HIP(hipFuncSetAttribute((const void*)sinkhorn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr = true;
    }
    return launch_sinkhorn(scores, lds_, m_, n_, n_, alpha, iters, Z_out, nullptr, (hipStream_t)stream);
}

// ---- the Hungarian step (src/models/associator.py:19-35) on the device ------------------------------------------------------------
// hungarian_matching() hands 1 - exp(Z[:-1, :-1]) to scipy.optimize.linear_sum_assignment and keeps the pairs whose score exceeds the
// threshold.  scipy's solver is the shortest-augmenting-path algorithm for the rectangular problem (Crouse 2016; scipy/optimize/
// rectangular_lsap, version 1.6 on -- a published algorithm with a fixed, sequential tie order: columns are scanned in the order of a
// "remaining" list that starts reversed and shrinks by swap-removal, among equal reduced costs the LAST unassigned column of that scan wins,
// else the first one).  One wavefront restates it: lanes own columns (two per lane: up to 128), the scan over the remaining columns is one
// lexicographic reduction on (reduced cost, assigned?, position in the list), duals and costs in binary64 in scipy's order of operations.
// The result is the same matching whenever the scores are the same floats; the scores themselves are exp() of the device here and torch's
// CPU exp on the host path (one ulp apart at most: a pair changes only on an exact tie or a score within an ulp of the threshold).
namespace {
constexpr int HG_MAXR = 32, HG_MAXC = 128;
// what lane (l ^ (1 << BIT)) holds, on DPP / v_permlane*_swap (sk_wave.h): bit moves, any payload
template <int BIT> __device__ __forceinline__ unsigned hg_xor(unsigned v) { return __builtin_bit_cast(unsigned, sk_xor<BIT>(__builtin_bit_cast(float, v))); }
struct HgKey { double c; unsigned t; };      // reduced cost; tail = not-free << 19 | tie key << 12 | column << 5 | row of the column (if assigned)
template <int BIT> __device__ __forceinline__ void hg_step(HgKey& b) {
    const unsigned long long cb = __builtin_bit_cast(unsigned long long, b.c);
    const unsigned lo = hg_xor<BIT>((unsigned)cb), hi = hg_xor<BIT>((unsigned)(cb >> 32)), ot = hg_xor<BIT>(b.t);
    const double oc = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
    if (oc < b.c || (oc == b.c && ot < b.t)) { b.c = oc; b.t = ot; }
}
__global__ __launch_bounds__(64) void hungarian_wave_kernel(const float* __restrict__ Z, int T, int n_det, int ldz, double thr, int log_domain,
                                                            int* __restrict__ match_out, int* __restrict__ status) {
    __shared__ double cost[HG_MAXR * HG_MAXC];
    __shared__ float sc_[HG_MAXR * HG_MAXC];
    __shared__ double u[HG_MAXR], spc_l[HG_MAXC];
    __shared__ int col4row[HG_MAXR], row4col[HG_MAXC], path_l[HG_MAXC], remaining[HG_MAXC];
    __shared__ unsigned char SR[HG_MAXR];
    const int lane = threadIdx.x;
    // ONE wavefront: its LDS operations execute in program order, so a wave-level fence (no s_barrier) orders a lane's write before the others' reads
    auto sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const bool tr = n_det < T;                       // scipy transposes a tall matrix
    const int nr = tr ? n_det : T, nc = tr ? T : n_det;
    // Z may live in pinned HOST memory (the Sinkhorn kernel writes it there): every element is requested once, the requests of a batch of
    // eight in flight together, and kept in LDS as the score (the threshold test at the end reads it again)
    const int total = T * n_det;
    for (int base = 0; base < total; base += 64 * 8) {
        float z[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int idx = base + k * 64 + lane;
            const int r = idx / n_det, c = idx - r * n_det;
            z[k] = idx < total ? Z[(size_t)r * ldz + c] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int idx = base + k * 64 + lane;
            if (idx < total) {
                const int r = idx / n_det, c = idx - r * n_det;
                const float s = log_domain ? expf(z[k]) : z[k];
                const int i = tr ? c : r, j = tr ? r : c;
                sc_[i * HG_MAXC + j] = s;
                cost[i * HG_MAXC + j] = (double)(1.0f - s);      // numpy: float32 (1 - scores), widened by the solver
            }
        }
    }
    // a lane keeps the state of its two columns (lane, lane + 64) in registers; rows, the scan list and what lane 0's augmentation walks are in LDS
    const int jc[2] = {lane, lane + 64};
    const bool has[2] = {jc[0] < nc, jc[1] < nc};
    double vv[2] = {0.0, 0.0}, spc[2];
    int r4c[2] = {-1, -1}, pth[2] = {-1, -1}, pos[2];
    bool SC[2];
    for (int j = lane; j < nc; j += 64) row4col[j] = -1;
    if (lane < nr) { u[lane] = 0.0; col4row[lane] = -1; }
    sync();
    const double INF = __builtin_huge_val();
    for (int cur = 0; cur < nr; cur++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            if (has[q]) { remaining[nc - 1 - jc[q]] = jc[q]; r4c[q] = row4col[jc[q]]; }      // list position it holds column nc - it - 1
            pos[q] = nc - 1 - jc[q]; spc[q] = INF; SC[q] = !has[q];
        }
        if (lane < nr) SR[lane] = 0;
        sync();
        double minVal = 0.0;
        int i = cur, num_remaining = nc, sink = -1;
        while (sink == -1) {
            if (lane == 0) SR[i] = 1;
            const double ui = u[i];
            HgKey best{INF, 0xffffffffu};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (SC[q]) continue;
                const double r = minVal + cost[i * HG_MAXC + jc[q]] - ui - vv[q];
                if (r < spc[q]) { pth[q] = i; spc[q] = r; }
                const bool free_ = r4c[q] == -1;
                // among equal reduced costs: the LAST free column of the scan (largest list position), else the first assigned one
                const unsigned t = (free_ ? 0u : 1u << 19) | (unsigned)(free_ ? 127 - pos[q] : pos[q]) << 12 | (unsigned)jc[q] << 5 | (unsigned)(free_ ? 0 : r4c[q]);
                if (spc[q] < best.c || (spc[q] == best.c && t < best.t)) { best.c = spc[q]; best.t = t; }
            }
            hg_step<0>(best); hg_step<1>(best); hg_step<2>(best); hg_step<3>(best); hg_step<4>(best); hg_step<5>(best);
            minVal = best.c;
            if (!(minVal < INF)) { if (lane == 0) *status = 1; return; }      // infeasible, or NaN scores (scipy raises)
            const int j = (best.t >> 5) & 127, kk = (best.t >> 12) & 127;
            const bool jfree = !(best.t >> 19);
            const int idx = jfree ? 127 - kk : kk;                             // the winner's position in the list
            if (jfree) sink = j; else i = best.t & 31;
            const int last = remaining[num_remaining - 1];
            sync();
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (jc[q] == j) { SC[q] = true; remaining[idx] = last; }
                if (jc[q] == last) pos[q] = idx;
            }
            num_remaining--;
            sync();
        }
        // what the dual update of the rows and lane 0's augmentation read by column index
#pragma unroll
        for (int q = 0; q < 2; q++)
            if (has[q]) { spc_l[jc[q]] = spc[q]; path_l[jc[q]] = pth[q]; if (SC[q]) vv[q] -= minVal - spc[q]; }
        sync();
        if (lane < nr && SR[lane] && lane != cur) u[lane] += minVal - spc_l[col4row[lane]];
        if (lane == 0) u[cur] += minVal;
        sync();
        if (lane == 0) {
            int j = sink;
            while (true) {
                const int ii = path_l[j];
                row4col[j] = ii;
                const int t = col4row[ii]; col4row[ii] = j; j = t;
                if (ii == cur) break;
            }
        }
        sync();
    }
    for (int c = lane; c < n_det; c += 64) match_out[c] = -1;
    sync();
    if (lane < nr) {
        const int jcol = col4row[lane];
        if (jcol >= 0) {
            const int r = tr ? jcol : lane, c = tr ? lane : jcol;
            if ((double)sc_[lane * HG_MAXC + jcol] > thr) match_out[c] = r;
        }
    }
    if (lane == 0) *status = 0;
}
}  // namespace

// ---- the same step on the HOST, with the tests that follow it (OdamProcess's fast path: one call between a frame's result and the next launch) ----
// scipy.optimize.linear_sum_assignment(1 - scores) as hungarian_matching calls it (associator.py:19-35) -- the sequential form of the solver
// restated above: rows in order, the scan over a "remaining" list that starts reversed and shrinks by swap-removal, among equal reduced costs a
// free column wins and the LAST such column of the scan, binary64 duals in scipy's order of operations -- then `score > match_threshold` for
// the matched pairs and `!(score[pair or dustbin row] < score_threshold)` of _attach_to_tracks (processor.py:228-231), both as numpy / torch
// compare a float32 array with a Python float: in float32.  No device work.
extern "C" int odam_assoc_attach(const float* score, int n_tracks, int n_det, int lds, double match_threshold, double score_threshold,
                                 int* match_out, unsigned char* keep_out) {
    if (!score || !match_out || !keep_out || n_tracks < 0 || n_det < 0 || lds < n_det + 1) return odam_fail(1, "odam_assoc_attach: bad argument");
    for (int c = 0; c < n_det; c++) match_out[c] = -1;
    const bool tr = n_det < n_tracks;                       // scipy transposes a tall matrix
    const int nr = tr ? n_det : n_tracks, nc = tr ? n_tracks : n_det;
    if (nr > 0) {
        std::vector<double> cost((size_t)nr * nc), u(nr, 0.0), v(nc, 0.0), spc(nc);
        std::vector<int> col4row(nr, -1), row4col(nc, -1), path(nc, -1), remaining(nc);
        std::vector<unsigned char> SR(nr), SC(nc);
        for (int r = 0; r < n_tracks; r++)
            for (int c = 0; c < n_det; c++) {
                const double x = (double)(1.0f - score[(size_t)r * lds + c]);      // numpy: float32 (1 - scores), widened by the solver
                if (x != x || x == -__builtin_huge_val()) return odam_fail(4, "odam_assoc_attach: matrix contains invalid numeric entries");
                cost[tr ? (size_t)c * nc + r : (size_t)r * nc + c] = x;
            }
        const double INF = __builtin_huge_val();
        for (int cur = 0; cur < nr; cur++) {
            double minVal = 0.0;
            int num_remaining = nc, sink = -1, i = cur;
            for (int it = 0; it < nc; it++) remaining[it] = nc - it - 1;
            std::fill(SR.begin(), SR.end(), 0); std::fill(SC.begin(), SC.end(), 0); std::fill(spc.begin(), spc.end(), INF);
            while (sink == -1) {
                int index = -1;
                double lowest = INF;
                SR[i] = 1;
                for (int it = 0; it < num_remaining; it++) {
                    const int j = remaining[it];
                    const double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                    if (r < spc[j]) { path[j] = i; spc[j] = r; }
                    if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
                }
                minVal = lowest;
                if (!(minVal < INF)) return odam_fail(5, "odam_assoc_attach: cost matrix is infeasible");
                const int j = remaining[index];
                if (row4col[j] == -1) sink = j; else i = row4col[j];
                SC[j] = 1;
                remaining[index] = remaining[--num_remaining];
            }
            u[cur] += minVal;
            for (int r = 0; r < nr; r++) if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
            for (int j = 0; j < nc; j++) if (SC[j]) v[j] -= minVal - spc[j];
            int j = sink;
            while (true) {
                const int ii = path[j];
                row4col[j] = ii;
                const int t = col4row[ii]; col4row[ii] = j; j = t;
                if (ii == cur) break;
            }
        }
        const float mt = (float)match_threshold;
        for (int r0 = 0; r0 < nr; r0++) {
            const int r = tr ? col4row[r0] : r0, c = tr ? r0 : col4row[r0];
            if (score[(size_t)r * lds + c] > mt) match_out[c] = r;
        }
    }
    const float st = (float)score_threshold;
    for (int c = 0; c < n_det; c++) {
        const int r = match_out[c] < 0 ? n_tracks : match_out[c];      // index -1 reads the dustbin row, as in the reference
        keep_out[c] = !(score[(size_t)r * lds + c] < st);
    }
    return 0;
}

extern "C" int odam_assoc_hungarian(const float* Z, int n_tracks, int n_det, int ldz, double threshold, int log_domain, int* match_out,
                                    int* status, void* stream) {
    if (!Z || !match_out || !status || n_tracks < 0 || n_det < 0 || ldz < n_det)
        return odam_fail(1, "odam_assoc_hungarian: bad argument");
    const int nr = n_det < n_tracks ? n_det : n_tracks, nc = n_det < n_tracks ? n_tracks : n_det;
    if (nr > HG_MAXR || nc > HG_MAXC) return odam_fail(3, "odam_assoc_hungarian: more than 32 x 128 (the caller solves it on the host)");      // (3 = ODAM_E_LIMIT of odam_sq.h)
    hipLaunchKernelGGL(hungarian_wave_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, Z, n_tracks, n_det, ldz, threshold, log_domain, match_out, status);
    ODAM_HIP(hipGetLastError());
    return 0;
}
