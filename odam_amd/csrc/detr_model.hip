// detr_model.hip -- host side of the detector: weight intake under the reference's state_dict names,
// packing, workspace, and the forward schedule over the gfx950 kernels (conv_gemm / detr_kernels).
//
// Reference (likojack/ODAM): src/models/detr.py:18-94 (DETR), src/models/backbone.py:21-110 (FrozenBN,
// ResNet body = torchvision ResNet v1.5 bottlenecks, stride on the 3x3), src/models/transformer.py
// (post-norm encoder/decoder), src/models/position_encoding.py (sine embedding, passed in precomputed).
// Layout: activations NHWC / [B*L, C] batch-major rows; every contraction runs on conv_gemm.
#include <hip/hip_runtime.h>
#include <atomic>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/odam_detr.h"
#include "conv_gemm.h"
#include "detr_kernels.h"
#include "odam_config.h"
#include "odam_err.h"

using odam_cg::ConvGemmArgs;

namespace {

struct HostTensor {
    std::vector<long long> shape;
    std::vector<float> data;
};

struct Conv {          // packed convolution / linear layer on device
    void* w = nullptr;       // [Cout][Kpad], fp32 or bf16
    void* w3 = nullptr;      // fp32 models: the same filters split into three bf16 planes (conv_gemm.h Wt3), or null
    float* scale = nullptr;  // [Cout] or null
    float* bias = nullptr;   // [Cout] or null
    int Cin = 0, Cout = 0, KH = 1, KW = 1, stride = 1, pad = 0, Kpad = 0, dil = 1;
    int k_order = 0;         // conv_gemm.h: 1 = channel-chunk-major K (multi-tap filters with Cin % k-tile == 0)
};

struct Bottleneck {
    Conv c1, c2, c3, ds;
    bool has_ds = false;
};

struct LN { float* g = nullptr; float* b = nullptr; };

struct EncLayer { Conv qk, v, out, l1, l2; LN n1, n2; };
struct DecLayer { Conv qk, v, out, cq, cout, l1, l2; LN n1, n2, n3; };

int ilog2(int x) { int l = 0; while ((1 << l) < x) l++; return l; }
int conv_out(int n, int k, int s, int p) { return (n + 2 * p - k) / s + 1; }

}  // namespace

struct odam_detr {
    odam_detr_cfg cfg;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    std::vector<void*> allocs;

    Conv stem;
    std::vector<Bottleneck> blocks;
    std::vector<int> block_stride;
    Conv input_proj;
    std::vector<EncLayer> enc;
    std::vector<DecLayer> dec;
    Conv cross_k_all, cross_v_all;
    LN dec_norm;
    LN enc_norm;                 // pre_norm only: the encoder's final LayerNorm (transformer.py:27-28)
    const void* memory_out = nullptr;      // where the last forward left the encoder's output (odam_detr_debug_read)
    Conv class_embed;
    Conv mlp[5][3];  // bbox, offset, angle, size, depth
    Conv stem_rows;              // fp32: conv1 as a 7x1 convolution over 32-float rows of the framed NHWC4 image (see forward_impl)
    float* pos = nullptr;        // [L, 256]
    float* query_pos = nullptr;  // [Q, 256]

    // geometry
    int H1, W1, H2, W2, fh, fw, L;

    // workspace (activations: fp32 or bf16 according to cfg.dtype; sizes in bytes = elements * es)
    char *x4 = nullptr, *stem_out = nullptr, *bufA = nullptr, *bufB = nullptr, *t1 = nullptr, *t2 = nullptr,
         *dsb = nullptr;
    char *src = nullptr, *srcpos = nullptr, *qk = nullptr, *v = nullptr, *att = nullptr, *tmp = nullptr, *ffn = nullptr;
    char *kc = nullptr, *vc = nullptr;
    char *tgt = nullptr, *tgtpos = nullptr, *dqk = nullptr, *dv = nullptr, *datt = nullptr, *dq = nullptr,
         *dtmp = nullptr, *dffn = nullptr, *hs = nullptr, *h1 = nullptr, *h2 = nullptr;
    const char* layer4 = nullptr;  // where the last forward left the layer4 map (NHWC)
    int es = 4;                    // bytes per activation / weight element
    int dt = 0;                    // 0 fp32, 1 bf16

    // optional per-launch timing of the contraction kernel (bench.py roofline): event pairs around every
    // conv_gemm launch of a forward, read back by odam_detr_profile_read
    bool profile = false;
    std::vector<hipEvent_t> ev;
    std::vector<double> ev_flops;
    size_t ev_used = 0;
    // input-transform resampling tables per (source length, output length): device [xmin | cnt | K]
    struct Resample { int* xmin; int* cnt; int* K; int ksize; };
    std::map<std::pair<int, int>, Resample> resample;

    std::vector<hipEvent_t> aev;      // same for the fused attention kernel
    std::vector<double> aev_flops;
    size_t aev_used = 0;

    template <typename T>
    int dev_alloc(T** p, size_t n) {
        ODAM_HIP(hipMalloc((void**)p, n * sizeof(T)));
        allocs.push_back((void*)*p);
        return 0;
    }
    int upload(float** p, const std::vector<float>& v) {
        if (int rc = dev_alloc(p, v.size())) return rc;
        ODAM_HIP(hipMemcpy(*p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
        return 0;
    }
    // weights in the model's element type: fp32 as is, bf16 rounded to nearest even on the host
    // fp32 models: the exact three-way bf16 split of a packed filter [Cout][Kpad] for the split contraction mode
    int upload_w3(void** p, const std::vector<float>& v, int Cout, int Kpad) {
        *p = nullptr;
        if (dt != 0 || Kpad % 16 != 0 || (long)Cout * Kpad * 6 >= 0x7fffffffL) return 0;
        std::vector<unsigned short> h((size_t)Cout * Kpad * 3);
        odam_cg::split3_filters(v.data(), Cout, Kpad, h.data());
        unsigned short* d = nullptr;
        if (int rc = dev_alloc(&d, h.size())) return rc;
        ODAM_HIP(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        *p = d;
        return 0;
    }
    int upload_w(void** p, const std::vector<float>& v) {
        if (dt == 0) return upload((float**)p, v);
        std::vector<unsigned short> h(v.size());
        for (size_t i = 0; i < v.size(); i++) {
            unsigned u;
            std::memcpy(&u, &v[i], 4);
            if ((u & 0x7fffffffu) > 0x7f800000u) h[i] = (unsigned short)((u >> 16) | 0x40);  // NaN stays NaN
            else h[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
        }
        unsigned short* d = nullptr;
        if (int rc = dev_alloc(&d, h.size())) return rc;
        ODAM_HIP(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        *p = d;
        return 0;
    }
};

namespace {

const HostTensor* find(odam_detr* m, const std::string& name) {
    auto it = m->host.find(name);
    return it == m->host.end() ? nullptr : &it->second;
}

#define NEED(var, name)                                                                    \
    const HostTensor* var = find(m, name);                                                 \
    if (!var) {                                                                            \
        std::snprintf(g_odam_err, sizeof(g_odam_err), "odam_detr_finalize: missing weight %s", std::string(name).c_str()); \
        return 1;                                                                          \
    }

// [Cout, Cin, KH, KW] (PyTorch) -> [Cout][Kpad], k = (ky*KW + kx)*CinP + ci, CinP = Cin rounded up to 4
int pack_conv(odam_detr* m, Conv& c, const HostTensor& w, int stride, int pad, int dil = 1) {
    const int Cout = (int)w.shape[0], Cin = (int)w.shape[1];
    const int KH = w.shape.size() > 2 ? (int)w.shape[2] : 1, KW = w.shape.size() > 3 ? (int)w.shape[3] : 1;
    const int epc = m->dt ? 8 : 4;             // elements per 16-byte chunk
    const int CinP = (Cin + epc - 1) / epc * epc;
    const int K = KH * KW * CinP;
    const int Kpad = (K + 8 * epc - 1) / (8 * epc) * (8 * epc);
    std::vector<float> p((size_t)Cout * Kpad, 0.0f);
    const int kt = 8 * epc, ntaps = KH * KW;
    const int k_order = (ntaps > 1 && ntaps <= 32 && CinP % kt == 0) ? 1 : 0;
    for (int o = 0; o < Cout; o++)
        for (int ci = 0; ci < Cin; ci++)
            for (int ky = 0; ky < KH; ky++)
                for (int kx = 0; kx < KW; kx++) {
                    const int tap = ky * KW + kx;
                    const int k = k_order ? ((ci / kt) * ntaps + tap) * kt + ci % kt : tap * CinP + ci;
                    p[(size_t)o * Kpad + k] = w.data[(((size_t)o * Cin + ci) * KH + ky) * KW + kx];
                }
    c.k_order = k_order;
    c.Cin = CinP; c.Cout = Cout; c.KH = KH; c.KW = KW; c.stride = stride; c.pad = pad; c.Kpad = Kpad; c.dil = dil;
    if (int rc = m->upload_w3(&c.w3, p, Cout, Kpad)) return rc;
    return m->upload_w(&c.w, p);
}

// rows [r0, r1) of a [N, K] Linear weight (+ bias) as a 1x1 layer
int pack_linear(odam_detr* m, Conv& c, const HostTensor& w, const HostTensor* b, int r0, int r1) {
    const int K = (int)w.shape[1];
    std::vector<float> p(w.data.begin() + (size_t)r0 * K, w.data.begin() + (size_t)r1 * K);
    c.Cin = K; c.Cout = r1 - r0; c.KH = c.KW = 1; c.stride = 1; c.pad = 0; c.Kpad = K;
    if (int rc = m->upload_w3(&c.w3, p, r1 - r0, K)) return rc;
    if (int rc = m->upload_w(&c.w, p)) return rc;
    if (b) {
        std::vector<float> bb(b->data.begin() + r0, b->data.begin() + r1);
        if (int rc = m->upload(&c.bias, bb)) return rc;
    }
    return 0;
}

// FrozenBatchNorm2d.forward (backbone.py:46-56): scale = w * rsqrt(rv + 1e-5); bias = b - rm * scale, in float32
int fold_bn(odam_detr* m, Conv& c, const std::string& prefix) {
    NEED(w, prefix + ".weight"); NEED(b, prefix + ".bias");
    NEED(rm, prefix + ".running_mean"); NEED(rv, prefix + ".running_var");
    const size_t n = w->data.size();
    std::vector<float> sc(n), bi(n);
    for (size_t i = 0; i < n; i++) {
        const float s = w->data[i] * (1.0f / std::sqrt(rv->data[i] + 1e-5f));
        sc[i] = s;
        bi[i] = b->data[i] - rm->data[i] * s;
    }
    if (int rc = m->upload(&c.scale, sc)) return rc;
    return m->upload(&c.bias, bi);
}

int pack_ln(odam_detr* m, LN& ln, const std::string& prefix) {
    NEED(w, prefix + ".weight"); NEED(b, prefix + ".bias");
    if (int rc = m->upload(&ln.g, w->data)) return rc;
    return m->upload(&ln.b, b->data);
}

int run_conv(const Conv& c, const void* x, int B, int H, int W, const void* res, bool relu, void* y, int ldc,
             hipStream_t st, int dtype = 0, int out_f32 = 0) {
    ConvGemmArgs a{};
    a.dtype = dtype; a.out_f32 = out_f32;
    a.A = x; a.Wt = c.w; a.Wt3 = c.w3; a.scale = c.scale; a.bias = c.bias; a.res = res; a.C = y;
    a.B = B; a.H = H; a.W = W; a.Cin = c.Cin; a.log2Cin = ilog2(c.Cin);
    a.Ho = conv_out(H, (c.KH - 1) * c.dil + 1, c.stride, c.pad); a.Wo = conv_out(W, (c.KW - 1) * c.dil + 1, c.stride, c.pad);
    a.Cout = c.Cout; a.KH = c.KH; a.KW = c.KW; a.stride = c.stride; a.pad = c.pad; a.Kpad = c.Kpad; a.dil = c.dil;
    a.relu = relu ? 1 : 0; a.M = B * a.Ho * a.Wo; a.ldc = ldc > 0 ? ldc : c.Cout;
    a.k_order = c.k_order;
    return odam_cg::launch_conv_gemm(a, st);
}

// timed variants used by the forward: identical launches, bracketed by events when profiling is on
int conv_t(odam_detr* m, const Conv& c, const void* x, int B, int H, int W, const void* res, bool relu, void* y,
           hipStream_t st, int out_f32 = 0) {
    if (!m->profile) return run_conv(c, x, B, H, W, res, relu, y, 0, st, m->dt, out_f32);
    if (m->ev_used + 2 > m->ev.size()) {
        for (int k = 0; k < 2; k++) {
            hipEvent_t e;
            ODAM_HIP(hipEventCreate(&e));
            m->ev.push_back(e);
        }
        m->ev_flops.resize(m->ev.size() / 2);
    }
    const int Ho = conv_out(H, (c.KH - 1) * c.dil + 1, c.stride, c.pad), Wo = conv_out(W, (c.KW - 1) * c.dil + 1, c.stride, c.pad);
    const int Cin_true = (c.KH == 7) ? 3 : c.Cin;  // the stem's 4th input channel is zero padding
    m->ev_flops[m->ev_used / 2] = 2.0 * B * Ho * Wo * (double)c.Cout * c.KH * c.KW * Cin_true;
    ODAM_HIP(hipEventRecord(m->ev[m->ev_used], st));
    int rc = run_conv(c, x, B, H, W, res, relu, y, 0, st, m->dt, out_f32);
    ODAM_HIP(hipEventRecord(m->ev[m->ev_used + 1], st));
    m->ev_used += 2;
    return rc;
}
// A bottleneck's 3x3 (c2: 64 channels) and expand + residual + ReLU (c3: 256 channels) as ONE launch where the fused kernel
// of conv_gemm.hip applies (layer1 in fp32 split mode); returns -1 where it does not, and the caller runs the two layers
// next_c1 (nullable) + t_next: also the following bottleneck's 1x1 reduce on the tile, its output (the next 3x3's input) to
// t_next; *chained tells the caller whether that happened
int fused_c2c3_t(odam_detr* m, const Conv& c2, const Conv& c3, const void* x, int B, int H, int W, const void* res, void* y,
                 hipStream_t st, const Conv* next_c1 = nullptr, void* t_next = nullptr, bool* chained = nullptr) {
    if (chained) *chained = false;
    if (c2.dil != 1) return -1;      // dilated 3x3 (the DC5 backbone's layer4): the separate launches
    if (c3.KH != 1 || c3.stride != 1 || c3.Kpad != c2.Cout) return -1;
    if (!m->dt && (!c2.w3 || !c3.w3)) return -1;
    ConvGemmArgs a{};
    a.dtype = m->dt; a.out_f32 = 0;
    a.A = x; a.Wt = c2.w; a.Wt3 = c2.w3; a.scale = c2.scale; a.bias = c2.bias; a.res = nullptr; a.C = nullptr;
    a.B = B; a.H = H; a.W = W; a.Cin = c2.Cin; a.log2Cin = ilog2(c2.Cin);
    a.Ho = conv_out(H, c2.KH, c2.stride, c2.pad); a.Wo = conv_out(W, c2.KW, c2.stride, c2.pad);
    a.Cout = c2.Cout; a.KH = c2.KH; a.KW = c2.KW; a.stride = c2.stride; a.pad = c2.pad; a.Kpad = c2.Kpad;
    a.relu = 1; a.M = B * a.Ho * a.Wo; a.ldc = c2.Cout; a.k_order = c2.k_order;
    a.F_scale = c3.scale; a.F_bias = c3.bias; a.F_res = (const float*)res; a.F_C = (float*)y;
    a.F_ldc = c3.Cout; a.F_relu = 1;
    double flops3 = 0.0;
    const bool next_ok = next_c1 && t_next && next_c1->KH == 1 && next_c1->stride == 1 && next_c1->Kpad == c3.Cout;
    if (m->dt) {
        // bf16: 3x3 + expand + residual (+ the next block's reduce where the channel combination is built) as one launch
        a.F_Wt = c3.w;
        if (next_ok) {
            a.G_Wt = next_c1->w; a.G_scale = next_c1->scale; a.G_bias = next_c1->bias; a.G_C = (float*)t_next; a.G_N = next_c1->Cout;
            if (odam_cg::fused_bf16_ok(a)) {
                if (chained) *chained = true;
                flops3 = 2.0 * a.M * (double)next_c1->Cout * c3.Cout;
            } else {
                a.G_Wt = nullptr; a.G_scale = a.G_bias = nullptr; a.G_C = nullptr; a.G_N = 64;
            }
        }
        if (!odam_cg::fused_bf16_ok(a)) return -1;
    } else {
    a.F_Wt3 = c3.w3;
    if (next_ok && next_c1->w3 && (next_c1->Cout == 64 || next_c1->Cout == 128) && c3.Cout == 256) {
        a.G_Wt3 = next_c1->w3; a.G_scale = next_c1->scale; a.G_bias = next_c1->bias; a.G_C = (float*)t_next; a.G_N = next_c1->Cout;
        if (odam_cg::fused_second_ok(a)) {
            if (chained) *chained = true;
            flops3 = 2.0 * a.M * (double)next_c1->Cout * c3.Cout;
        } else {
            a.G_Wt3 = nullptr; a.G_scale = a.G_bias = nullptr; a.G_C = nullptr; a.G_N = 64;
        }
    }
    if (!odam_cg::fused_second_ok(a)) return -1;
    }
    if (!m->profile) return odam_cg::launch_conv_gemm(a, st);
    if (m->ev_used + 2 > m->ev.size()) {
        for (int k = 0; k < 2; k++) {
            hipEvent_t e;
            ODAM_HIP(hipEventCreate(&e));
            m->ev.push_back(e);
        }
        m->ev_flops.resize(m->ev.size() / 2);
    }
    m->ev_flops[m->ev_used / 2] = 2.0 * a.M * ((double)c2.Cout * c2.KH * c2.KW * c2.Cin + (double)c3.Cout * c2.Cout) + flops3;
    ODAM_HIP(hipEventRecord(m->ev[m->ev_used], st));
    int rc = odam_cg::launch_conv_gemm(a, st);
    ODAM_HIP(hipEventRecord(m->ev[m->ev_used + 1], st));
    m->ev_used += 2;
    return rc;
}

// conv1 over the framed image (forward_impl): the arguments run_conv cannot derive -- output width from the true image,
// pixel stride smaller than the row length
static std::atomic<long long> g_pooled_stem_launches{0};
// *pooled: in: try the max-pool on the tile (ConvGemmArgs::pool; output then goes to `pooled_out` [B, H2, W2, 64]); out: whether it ran so
int stem_rows_t(odam_detr* m, int B, hipStream_t st, void* pooled_out = nullptr, bool* pooled = nullptr) {
    const Conv& c = m->stem_rows;
    const odam_detr_cfg& g = m->cfg;
    ConvGemmArgs a{};
    a.dtype = m->dt; a.out_f32 = 0;
    a.A = m->x4; a.Wt = c.w; a.Wt3 = c.w3; a.scale = c.scale; a.bias = c.bias; a.res = nullptr; a.C = m->stem_out;
    a.B = B; a.H = g.img_h + 6; a.W = g.img_w + 8; a.Cin = 32; a.log2Cin = 5; a.lda = 4;
    a.Ho = m->H1; a.Wo = m->W1;
    a.Cout = c.Cout; a.KH = 7; a.KW = 1; a.stride = 2; a.pad = 0; a.Kpad = c.Kpad;
    a.relu = 1; a.M = B * a.Ho * a.Wo; a.ldc = c.Cout; a.k_order = c.k_order;
    if (pooled) {
        // patches of 17 x 29 conv outputs = 8 x 14 pooled pixels (493 of the tile's 512 rows; 14 % of the conv recomputed at the seams)
        a.pool = 1; a.pool_ph = odam_cg::POOL_PH; a.pool_pw = odam_cg::POOL_PW; a.Hp = m->H2; a.Wp = m->W2; a.C = pooled_out;
        *pooled = odam_cg::pooled_stem_ok(a);
        if (!*pooled) { a.pool = 0; a.C = m->stem_out; }
        else g_pooled_stem_launches.fetch_add(1);
    }
    if (!m->profile) return odam_cg::launch_conv_gemm(a, st);
    if (m->ev_used + 2 > m->ev.size()) {
        for (int k = 0; k < 2; k++) {
            hipEvent_t e;
            ODAM_HIP(hipEventCreate(&e));
            m->ev.push_back(e);
        }
        m->ev_flops.resize(m->ev.size() / 2);
    }
    m->ev_flops[m->ev_used / 2] = 2.0 * a.M * (double)c.Cout * 147;      // the algorithmic products, not the 224 issued
    ODAM_HIP(hipEventRecord(m->ev[m->ev_used], st));
    int rc = odam_cg::launch_conv_gemm(a, st);
    ODAM_HIP(hipEventRecord(m->ev[m->ev_used + 1], st));
    m->ev_used += 2;
    return rc;
}
int lin_t(odam_detr* m, const Conv& c, const void* x, int M, const void* res, bool relu, void* y, hipStream_t st,
          int out_f32 = 0) {
    return conv_t(m, c, x, 1, 1, M, res, relu, y, st, out_f32);
}
int att_t(odam_detr* m, const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
          int B, int H, int Lq, int Lk, hipStream_t st, const unsigned char* key_mask = nullptr) {
    if (!m->profile) return odam_dk::launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Lq, Lk, m->dt, st, key_mask);
    if (m->aev_used + 2 > m->aev.size()) {
        for (int k = 0; k < 2; k++) {
            hipEvent_t e;
            ODAM_HIP(hipEventCreate(&e));
            m->aev.push_back(e);
        }
        m->aev_flops.resize(m->aev.size() / 2);
    }
    m->aev_flops[m->aev_used / 2] = 4.0 * B * H * (double)Lq * Lk * 32;   // QK^T + PV, 2 flop per MAC
    ODAM_HIP(hipEventRecord(m->aev[m->aev_used], st));
    int rc = odam_dk::launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Lq, Lk, m->dt, st, key_mask);
    ODAM_HIP(hipEventRecord(m->aev[m->aev_used + 1], st));
    m->aev_used += 2;
    return rc;
}

#define RC(call)                 \
    do {                         \
        int rc_ = (call);        \
        if (rc_) return rc_;     \
    } while (0)

}  // namespace

extern "C" int odam_detr_create(const odam_detr_cfg* cfg, odam_detr** out) {
    if (!cfg || !out) return odam_fail(1, "odam_detr_create: null argument");
    if (cfg->hidden_dim != 256 || cfg->nheads != 8)
        return odam_fail(3, "odam_detr_create: kernels are built for hidden_dim 256, 8 heads (head dim 32)");
    if (cfg->max_batch < 1 || cfg->img_h < 32 || cfg->img_w < 32) return odam_fail(1, "odam_detr_create: bad sizes");
    if (cfg->dtype != 0 && cfg->dtype != 1) return odam_fail(1, "odam_detr_create: dtype must be 0 (fp32) or 1 (bf16)");
    odam_detr* m = new odam_detr();
    m->cfg = *cfg;
    m->dt = cfg->dtype;
    m->es = cfg->dtype ? 2 : 4;
    m->H1 = conv_out(cfg->img_h, 7, 2, 3); m->W1 = conv_out(cfg->img_w, 7, 2, 3);
    m->H2 = conv_out(m->H1, 3, 2, 1); m->W2 = conv_out(m->W1, 3, 2, 1);
    int h = m->H2, w = m->W2;
    for (int l = 1; l < (cfg->dilation ? 3 : 4); l++) { h = conv_out(h, 3, 2, 1); w = conv_out(w, 3, 2, 1); }      // dilation: layer4 keeps layer3's resolution
    m->fh = h; m->fw = w; m->L = h * w;
    *out = m;
    return 0;
}

extern "C" int odam_detr_destroy(odam_detr* m) {
    if (!m) return 0;
    for (void* p : m->allocs) (void)hipFree(p);
    for (hipEvent_t e : m->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : m->aev) (void)hipEventDestroy(e);
    delete m;
    return 0;
}

extern "C" int odam_detr_feature_hw(const odam_detr* m, int* h, int* w) {
    if (!m || !h || !w) return odam_fail(1, "odam_detr_feature_hw: null argument");
    *h = m->fh; *w = m->fw;
    return 0;
}

extern "C" int odam_detr_set_weight(odam_detr* m, const char* name, const float* data, const long long* shape,
                                    int ndim) {
    if (!m || !name || !data || (!shape && ndim > 0) || ndim < 0 || ndim > 4)
        return odam_fail(1, "odam_detr_set_weight: bad argument");
    if (m->finalized) return odam_fail(1, "odam_detr_set_weight: model already finalized");
    HostTensor t;
    size_t n = 1;
    for (int i = 0; i < ndim; i++) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    t.data.assign(data, data + n);
    m->host[name] = std::move(t);
    return 0;
}

extern "C" int odam_detr_finalize(odam_detr* m) {
    if (!m) return odam_fail(1, "odam_detr_finalize: null model");
    if (m->finalized) return 0;
    const odam_detr_cfg& c = m->cfg;
    const std::string bb = "backbone.0.body.";
    {   // stem
        NEED(w, bb + "conv1.weight");
        RC(pack_conv(m, m->stem, *w, 2, 3));
        RC(fold_bn(m, m->stem, bb + "bn1"));
        {
            // conv1 as KH = 7, KW = 1, Cin = 32: filter row ky of output channel o is the 28 values w[o][c][ky][kx] at
            // position 4 kx + c (what 7 consecutive NHWC4 pixels hold), then 4 zeros; K = 7 * 32 = 224.  fp32: k_order 1 (one
            // 32-channel chunk); bf16: the k-tile is 32 elements = one filter row, tap-major (k_order 0)
            const int Cout = (int)w->shape[0];
            const int Kp = m->dt ? 256 : 224;          // bf16: the dispatcher's k-tile is 64 elements (an eighth, all-zero filter row)
            std::vector<float> pr((size_t)Cout * Kp, 0.0f);
            for (int o = 0; o < Cout; o++)
                for (int ci = 0; ci < 3; ci++)
                    for (int ky = 0; ky < 7; ky++)
                        for (int kx = 0; kx < 7; kx++)
                            pr[(size_t)o * Kp + ky * 32 + kx * 4 + ci] = w->data[(((size_t)o * 3 + ci) * 7 + ky) * 7 + kx];
            Conv& r = m->stem_rows;
            r.Cin = 32; r.Cout = Cout; r.KH = 7; r.KW = 1; r.stride = 2; r.pad = 0; r.Kpad = Kp; r.k_order = m->dt ? 0 : 1;
            RC(m->upload_w3(&r.w3, pr, Cout, Kp));
            RC(m->upload_w(&r.w, pr));
            r.scale = m->stem.scale; r.bias = m->stem.bias;
        }
    }
    for (int l = 0; l < 4; l++) {
        for (int i = 0; i < c.resnet_blocks[l]; i++) {
            const std::string p = bb + "layer" + std::to_string(l + 1) + "." + std::to_string(i) + ".";
            // `dilation` (backbone.py:89-91 -> torchvision replace_stride_with_dilation = [False, False, True]): layer4's stride
            // becomes a dilation -- its first block runs with stride 1 and the PREVIOUS dilation (1), the others with dilation 2
            // and padding 2 (torchvision ResNet._make_layer: previous_dilation for block 0, self.dilation *= stride after it)
            const bool dc5 = c.dilation && l == 3;
            const int stride = (i == 0 && l > 0 && !dc5) ? 2 : 1;
            const int dil = (dc5 && i > 0) ? 2 : 1;
            Bottleneck b;
            NEED(w1, p + "conv1.weight"); NEED(w2, p + "conv2.weight"); NEED(w3, p + "conv3.weight");
            RC(pack_conv(m, b.c1, *w1, 1, 0)); RC(fold_bn(m, b.c1, p + "bn1"));
            RC(pack_conv(m, b.c2, *w2, stride, dil, dil)); RC(fold_bn(m, b.c2, p + "bn2"));   // v1.5: stride on the 3x3
            RC(pack_conv(m, b.c3, *w3, 1, 0)); RC(fold_bn(m, b.c3, p + "bn3"));
            if (i == 0) {
                NEED(wd, p + "downsample.0.weight");
                RC(pack_conv(m, b.ds, *wd, stride, 0)); RC(fold_bn(m, b.ds, p + "downsample.1"));
                b.has_ds = true;
            }
            m->blocks.push_back(b);
            m->block_stride.push_back(stride);
        }
    }
    {
        NEED(w, "input_proj.weight"); NEED(b, "input_proj.bias");
        RC(pack_conv(m, m->input_proj, *w, 1, 0));
        RC(m->upload(&m->input_proj.bias, b->data));
    }
    const int E = c.hidden_dim;
    for (int i = 0; i < c.enc_layers; i++) {
        const std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
        EncLayer e;
        NEED(ipw, p + "self_attn.in_proj_weight"); NEED(ipb, p + "self_attn.in_proj_bias");
        RC(pack_linear(m, e.qk, *ipw, ipb, 0, 2 * E));
        RC(pack_linear(m, e.v, *ipw, ipb, 2 * E, 3 * E));
        NEED(ow, p + "self_attn.out_proj.weight"); NEED(ob, p + "self_attn.out_proj.bias");
        RC(pack_linear(m, e.out, *ow, ob, 0, E));
        NEED(w1, p + "linear1.weight"); NEED(b1, p + "linear1.bias");
        NEED(w2, p + "linear2.weight"); NEED(b2, p + "linear2.bias");
        RC(pack_linear(m, e.l1, *w1, b1, 0, c.dim_feedforward));
        RC(pack_linear(m, e.l2, *w2, b2, 0, E));
        RC(pack_ln(m, e.n1, p + "norm1")); RC(pack_ln(m, e.n2, p + "norm2"));
        m->enc.push_back(e);
    }
    // decoder; the cross-attention key/value projections of all layers read the same (memory + pos) / memory,
    // so their weights are stacked into two [dec_layers*E, E] layers evaluated once per forward
    HostTensor kall, vall, kball, vball;
    kall.shape = {(long long)c.dec_layers * E, E}; vall.shape = kall.shape;
    for (int i = 0; i < c.dec_layers; i++) {
        const std::string p = "transformer.decoder.layers." + std::to_string(i) + ".";
        DecLayer d;
        NEED(ipw, p + "self_attn.in_proj_weight"); NEED(ipb, p + "self_attn.in_proj_bias");
        RC(pack_linear(m, d.qk, *ipw, ipb, 0, 2 * E));
        RC(pack_linear(m, d.v, *ipw, ipb, 2 * E, 3 * E));
        NEED(ow, p + "self_attn.out_proj.weight"); NEED(ob, p + "self_attn.out_proj.bias");
        RC(pack_linear(m, d.out, *ow, ob, 0, E));
        NEED(cw, p + "multihead_attn.in_proj_weight"); NEED(cb, p + "multihead_attn.in_proj_bias");
        RC(pack_linear(m, d.cq, *cw, cb, 0, E));
        kall.data.insert(kall.data.end(), cw->data.begin() + (size_t)E * E, cw->data.begin() + (size_t)2 * E * E);
        vall.data.insert(vall.data.end(), cw->data.begin() + (size_t)2 * E * E, cw->data.begin() + (size_t)3 * E * E);
        kball.data.insert(kball.data.end(), cb->data.begin() + E, cb->data.begin() + 2 * E);
        vball.data.insert(vball.data.end(), cb->data.begin() + 2 * E, cb->data.begin() + 3 * E);
        NEED(cow, p + "multihead_attn.out_proj.weight"); NEED(cob, p + "multihead_attn.out_proj.bias");
        RC(pack_linear(m, d.cout, *cow, cob, 0, E));
        NEED(w1, p + "linear1.weight"); NEED(b1, p + "linear1.bias");
        NEED(w2, p + "linear2.weight"); NEED(b2, p + "linear2.bias");
        RC(pack_linear(m, d.l1, *w1, b1, 0, c.dim_feedforward));
        RC(pack_linear(m, d.l2, *w2, b2, 0, E));
        RC(pack_ln(m, d.n1, p + "norm1")); RC(pack_ln(m, d.n2, p + "norm2")); RC(pack_ln(m, d.n3, p + "norm3"));
        m->dec.push_back(d);
    }
    kball.shape = {(long long)c.dec_layers * E}; vball.shape = kball.shape;
    RC(pack_linear(m, m->cross_k_all, kall, &kball, 0, c.dec_layers * E));
    RC(pack_linear(m, m->cross_v_all, vall, &vball, 0, c.dec_layers * E));
    RC(pack_ln(m, m->dec_norm, "transformer.decoder.norm"));
    if (c.pre_norm) RC(pack_ln(m, m->enc_norm, "transformer.encoder.norm"));
    {
        NEED(w, "class_embed.weight"); NEED(b, "class_embed.bias");
        RC(pack_linear(m, m->class_embed, *w, b, 0, (int)w->shape[0]));
        const char* names[5] = {"bbox_embed", "offset_embed", "angle_embed", "size_embed", "depth_embed"};
        for (int k = 0; k < 5; k++)
            for (int j = 0; j < 3; j++) {
                const std::string p = std::string(names[k]) + ".layers." + std::to_string(j);
                NEED(lw, p + ".weight"); NEED(lb, p + ".bias");
                RC(pack_linear(m, m->mlp[k][j], *lw, lb, 0, (int)lw->shape[0]));
            }
        NEED(q, "query_embed.weight");
        RC(m->upload(&m->query_pos, q->data));
        NEED(pe, "pos_embed");
        if ((int)pe->shape[0] != m->L || (int)pe->shape[1] != E) return odam_fail(1, "pos_embed must be [h*w, hidden_dim]");
        RC(m->upload(&m->pos, pe->data));
    }
    // workspace (bytes = elements * es)
    const size_t B = c.max_batch, es = m->es;
    const size_t big = B * m->H2 * m->W2 * 256 * es;
    RC(m->dev_alloc(&m->x4, B * (c.img_h + 6) * (c.img_w + 8) * (m->dt ? 8 : 4) * es));     // room for the framed fp32 image
    RC(m->dev_alloc(&m->stem_out, B * m->H1 * m->W1 * 64 * es));
    RC(m->dev_alloc(&m->bufA, big)); RC(m->dev_alloc(&m->bufB, big)); RC(m->dev_alloc(&m->dsb, big));
    RC(m->dev_alloc(&m->t1, big / 2)); RC(m->dev_alloc(&m->t2, big / 2));
    const size_t M = B * m->L, Mq = B * c.num_queries, F = c.dim_feedforward;
    RC(m->dev_alloc(&m->src, M * E * es)); RC(m->dev_alloc(&m->srcpos, M * E * es)); RC(m->dev_alloc(&m->qk, M * 2 * E * es));
    RC(m->dev_alloc(&m->v, M * E * es)); RC(m->dev_alloc(&m->att, M * E * es)); RC(m->dev_alloc(&m->tmp, M * E * es));
    RC(m->dev_alloc(&m->ffn, M * F * es));
    RC(m->dev_alloc(&m->kc, M * c.dec_layers * E * es)); RC(m->dev_alloc(&m->vc, M * c.dec_layers * E * es));
    RC(m->dev_alloc(&m->tgt, Mq * E * es)); RC(m->dev_alloc(&m->tgtpos, Mq * E * es)); RC(m->dev_alloc(&m->dqk, Mq * 2 * E * es));
    RC(m->dev_alloc(&m->dv, Mq * E * es)); RC(m->dev_alloc(&m->datt, Mq * E * es)); RC(m->dev_alloc(&m->dq, Mq * E * es));
    RC(m->dev_alloc(&m->dtmp, Mq * E * es)); RC(m->dev_alloc(&m->dffn, Mq * F * es)); RC(m->dev_alloc(&m->hs, Mq * E * es));
    RC(m->dev_alloc(&m->h1, Mq * E * es)); RC(m->dev_alloc(&m->h2, Mq * E * es));
    m->host.clear();
    m->finalized = true;
    return 0;
}

// key_mask / pos_b: null for same-size batches (no padding: the mask is all false and one position table serves every
// image).  Otherwise key_mask [B][L] marks padded tokens (backbone.py:79) and pos_b [B][L][E] is each image's own sine
// embedding, which depends on its mask (position_encoding.py:26-46).
static int forward_impl(odam_detr* m, const float* img, int B, const unsigned char* key_mask, const float* pos_b,
                        float* logits, float* boxes, float* angle, float* offset, float* size, float* depth,
                        float* obj_features, void* stream) {
    if (!m || !img || !logits || !boxes || !angle || !offset || !size || !depth)
        return odam_fail(1, "odam_detr_forward: null pointer");
    if (!m->finalized) return odam_fail(1, "odam_detr_forward: call odam_detr_finalize first");
    if (B < 1 || B > m->cfg.max_batch) return odam_fail(3, "odam_detr_forward: batch outside 1..max_batch");
    hipStream_t st = (hipStream_t)stream;
    m->ev_used = 0;
    m->aev_used = 0;
    const odam_detr_cfg& c = m->cfg;
    const int E = c.hidden_dim, Hh = c.nheads, Q = c.num_queries, dt = m->dt;
    const size_t es = m->es;
    auto off = [&](char* p, size_t n) { return p + n * es; };   // element offset in the activation type

    // ---- backbone (backbone.py:59-94) ----------------------------------------------------------
    {
        // fp32, split contraction mode, enough rows to fill the device: conv1 on the ring kernel.  With Cin = 4 the 7x7
        // filter cannot use its gather (a k-tile must sit inside one tap), but seven consecutive NHWC4 pixels ARE 28
        // contiguous floats: over an image framed with zeros (3 rows above / below, 3 + 5 columns) conv1 is a 7x1
        // convolution with "Cin" = 32 floats per row, pixel stride lda = 4, no padding, K = 224 instead of 147 + pad --
        // a third of the products are zeros, on a path that is 2-3x faster than the 128x64 tiles of the fp32 instruction.
        const bool rows_ok = odam_cfg::get(odam_cfg::STEM_ROWS) != 0;
        bool pooled = false;
        const long Mst = (long)B * m->H1 * m->W1;
        const bool rows_mode = dt ? (m->stem_rows.w != nullptr && odam_cfg::get(odam_cfg::CG_RING) != 0)
                                  : (m->stem_rows.w3 != nullptr && odam_cg::f32_mode() == 2);
        if (rows_ok && rows_mode && (Mst >= 192L * 256 || odam_cfg::get(odam_cfg::CG_PIN))) {
            RC(odam_dk::launch_nchw_to_nhwc4_framed(img, m->x4, B, c.img_h, c.img_w, dt, st));
            RC(stem_rows_t(m, B, st, m->bufA, &pooled));
        } else {
            RC(odam_dk::launch_nchw_to_nhwc4(img, m->x4, B, c.img_h, c.img_w, dt, st));
            RC(conv_t(m, m->stem, m->x4, B, c.img_h, c.img_w, nullptr, true, m->stem_out, st));
        }
        if (!pooled) RC(odam_dk::launch_maxpool3x3s2(m->stem_out, m->bufA, B, m->H1, m->W1, 64, m->H2, m->W2, dt, st));
    }
    char* cur = m->bufA;
    char* nxt = m->bufB;
    int H = m->H2, W = m->W2;
    char* tin = m->t1;          // where this block's 1x1 reduce output lives (the 3x3's input)
    char* tout = m->t2;         // the other small buffer: the 3x3's output, or -- chained -- the NEXT block's reduce output
    bool have_c1 = false;       // the previous block's launch already computed this block's reduce
    for (size_t i = 0; i < m->blocks.size(); i++) {
        const Bottleneck& b = m->blocks[i];
        const int s = m->block_stride[i];
        const int Ho = conv_out(H, 3, s, 1), Wo = conv_out(W, 3, s, 1);
        if (!have_c1) RC(conv_t(m, b.c1, cur, B, H, W, nullptr, true, tin, st));
        have_c1 = false;
        const char* res = cur;
        if (b.has_ds) {
            RC(conv_t(m, b.ds, cur, B, H, W, nullptr, false, m->dsb, st));
            res = m->dsb;
        }
        // the next block's reduce rides along where that block keeps the resolution and the channel counts (layer1's 2nd, 3rd)
        // (layer2's first reduce after layer1's last block included: a bottleneck's stride sits on its 3x3)
        const Conv* next_c1 = (i + 1 < m->blocks.size()) ? &m->blocks[i + 1].c1 : nullptr;
        bool chained = false;
        const int frc = fused_c2c3_t(m, b.c2, b.c3, tin, B, H, W, res, nxt, st, next_c1, tout, &chained);
        if (frc > 0) return frc;
        if (frc < 0) {
            RC(conv_t(m, b.c2, tin, B, H, W, nullptr, true, tout, st));
            RC(conv_t(m, b.c3, tout, B, Ho, Wo, res, true, nxt, st));
        } else if (chained) {
            char* t = tin; tin = tout; tout = t;      // the next block reads its 3x3 input from where this launch put it
            have_c1 = true;
        }
        char* t = cur; cur = nxt; nxt = t;
        H = Ho; W = Wo;
    }
    m->layer4 = cur;
    const int L = m->L, M = B * L, Mq = B * Q;

    // ---- input_proj + encoder (detr.py:70, transformer.py:154-167) -----------------------------
    RC(conv_t(m, m->input_proj, cur, B, H, W, nullptr, false, m->src, st));
    const float* pos = pos_b ? pos_b : m->pos;     // row r of the batch reads pos[r % Lp]
    const int Lp = pos_b ? M : L;
    RC(odam_dk::launch_add_pos(m->src, pos, Lp, m->srcpos, M, dt, st));
    const void* memory_p = m->src;      // the encoder's output (pre_norm: after its final LayerNorm, in tmp)
    for (size_t i = 0; i < m->enc.size(); i++) {
        const EncLayer& e = m->enc[i];
        if (c.pre_norm) {
            // forward_pre (transformer.py:169-183): src2 = norm1(src); q = k = src2 + pos; src += attn(q, k, src2);
            //                                       src2 = norm2(src); src += linear2(relu(linear1(src2)))
            RC(odam_dk::launch_add_layernorm(m->src, nullptr, e.n1.g, e.n1.b, m->tmp, pos, Lp, m->srcpos, M, dt, st));
            RC(lin_t(m, e.qk, m->srcpos, M, nullptr, false, m->qk, st));
            RC(lin_t(m, e.v, m->tmp, M, nullptr, false, m->v, st));
            RC(att_t(m, m->qk, 2 * E, off(m->qk, E), 2 * E, m->v, E, m->att, E, B, Hh, L, L, st, key_mask));
            RC(lin_t(m, e.out, m->att, M, m->src, false, m->src, st));                  // residual added in place
            RC(odam_dk::launch_add_layernorm(m->src, nullptr, e.n2.g, e.n2.b, m->tmp, nullptr, L, nullptr, M, dt, st));
            RC(lin_t(m, e.l1, m->tmp, M, nullptr, true, m->ffn, st));
            RC(lin_t(m, e.l2, m->ffn, M, m->src, false, m->src, st));
            continue;
        }
        RC(lin_t(m, e.qk, m->srcpos, M, nullptr, false, m->qk, st));
        RC(lin_t(m, e.v, m->src, M, nullptr, false, m->v, st));
        RC(att_t(m, m->qk, 2 * E, off(m->qk, E), 2 * E, m->v, E, m->att, E, B, Hh, L, L, st, key_mask));
        RC(lin_t(m, e.out, m->att, M, m->src, false, m->tmp, st));
        RC(odam_dk::launch_add_layernorm(m->tmp, nullptr, e.n1.g, e.n1.b, m->src, nullptr, L, nullptr, M, dt, st));
        RC(lin_t(m, e.l1, m->src, M, nullptr, true, m->ffn, st));
        RC(lin_t(m, e.l2, m->ffn, M, m->src, false, m->tmp, st));
        RC(odam_dk::launch_add_layernorm(m->tmp, nullptr, e.n2.g, e.n2.b, m->src, pos, Lp, m->srcpos, M, dt, st));
    }
    if (c.pre_norm) {      // memory = encoder.norm(src) (transformer.py:82-83), memory + pos for the cross-attention keys
        RC(odam_dk::launch_add_layernorm(m->src, nullptr, m->enc_norm.g, m->enc_norm.b, m->tmp, pos, Lp, m->srcpos, M, dt, st));
        memory_p = m->tmp;
    }
    m->memory_out = memory_p;
    // memory = src, memory + pos = srcpos
    RC(lin_t(m, m->cross_k_all, m->srcpos, M, nullptr, false, m->kc, st));
    RC(lin_t(m, m->cross_v_all, memory_p, M, nullptr, false, m->vc, st));

    // ---- decoder (transformer.py:217-238) --------------------------------------------------------
    ODAM_HIP(hipMemsetAsync(m->tgt, 0, es * (size_t)Mq * E, st));
    RC(odam_dk::launch_add_pos(nullptr, m->query_pos, Q, m->tgtpos, Mq, dt, st));
    const int ldkv = c.dec_layers * E;
    for (size_t i = 0; i < m->dec.size(); i++) {
        const DecLayer& d = m->dec[i];
        if (c.pre_norm) {
            // forward_pre (transformer.py:240-262): every sub-block normalises its INPUT and adds its output to tgt
            RC(odam_dk::launch_add_layernorm(m->tgt, nullptr, d.n1.g, d.n1.b, m->dtmp, m->query_pos, Q, m->tgtpos, Mq, dt, st));
            RC(lin_t(m, d.qk, m->tgtpos, Mq, nullptr, false, m->dqk, st));
            RC(lin_t(m, d.v, m->dtmp, Mq, nullptr, false, m->dv, st));
            RC(att_t(m, m->dqk, 2 * E, off(m->dqk, E), 2 * E, m->dv, E, m->datt, E, B, Hh, Q, Q, st));
            RC(lin_t(m, d.out, m->datt, Mq, m->tgt, false, m->tgt, st));
            RC(odam_dk::launch_add_layernorm(m->tgt, nullptr, d.n2.g, d.n2.b, m->dtmp, m->query_pos, Q, m->tgtpos, Mq, dt, st));
            RC(lin_t(m, d.cq, m->tgtpos, Mq, nullptr, false, m->dq, st));
            RC(att_t(m, m->dq, E, off(m->kc, i * E), ldkv, off(m->vc, i * E), ldkv, m->datt, E, B, Hh, Q, L, st, key_mask));
            RC(lin_t(m, d.cout, m->datt, Mq, m->tgt, false, m->tgt, st));
            RC(odam_dk::launch_add_layernorm(m->tgt, nullptr, d.n3.g, d.n3.b, m->dtmp, nullptr, Q, nullptr, Mq, dt, st));
            RC(lin_t(m, d.l1, m->dtmp, Mq, nullptr, true, m->dffn, st));
            RC(lin_t(m, d.l2, m->dffn, Mq, m->tgt, false, m->tgt, st));
            continue;
        }
        RC(lin_t(m, d.qk, m->tgtpos, Mq, nullptr, false, m->dqk, st));
        RC(lin_t(m, d.v, m->tgt, Mq, nullptr, false, m->dv, st));
        RC(att_t(m, m->dqk, 2 * E, off(m->dqk, E), 2 * E, m->dv, E, m->datt, E, B, Hh, Q, Q, st));
        RC(lin_t(m, d.out, m->datt, Mq, m->tgt, false, m->dtmp, st));
        RC(odam_dk::launch_add_layernorm(m->dtmp, nullptr, d.n1.g, d.n1.b, m->tgt, m->query_pos, Q, m->tgtpos, Mq, dt, st));
        RC(lin_t(m, d.cq, m->tgtpos, Mq, nullptr, false, m->dq, st));
        RC(att_t(m, m->dq, E, off(m->kc, i * E), ldkv, off(m->vc, i * E), ldkv, m->datt, E, B, Hh, Q, L, st, key_mask));
        RC(lin_t(m, d.cout, m->datt, Mq, m->tgt, false, m->dtmp, st));
        RC(odam_dk::launch_add_layernorm(m->dtmp, nullptr, d.n2.g, d.n2.b, m->tgt, nullptr, Q, nullptr, Mq, dt, st));
        RC(lin_t(m, d.l1, m->tgt, Mq, nullptr, true, m->dffn, st));
        RC(lin_t(m, d.l2, m->dffn, Mq, m->tgt, false, m->dtmp, st));
        RC(odam_dk::launch_add_layernorm(m->dtmp, nullptr, d.n3.g, d.n3.b, m->tgt, m->query_pos, Q, m->tgtpos, Mq, dt, st));
    }
    // fp32 mode: the final norm can write straight into the caller's obj_features buffer
    void* hs = (obj_features && dt == 0) ? (void*)obj_features : (void*)m->hs;
    RC(odam_dk::launch_add_layernorm(m->tgt, nullptr, m->dec_norm.g, m->dec_norm.b, hs, nullptr, Q, nullptr, Mq, dt, st));
    if (obj_features && dt != 0) RC(odam_dk::launch_to_f32(m->hs, obj_features, (size_t)Mq * E, dt, st));

    // ---- heads on the last decoder layer (detr.py:73-88); their outputs are fp32 in every mode ----
    RC(lin_t(m, m->class_embed, hs, Mq, nullptr, false, logits, st, 1));
    float* outs[5] = {boxes, offset, angle, size, depth};
    for (int k = 0; k < 5; k++) {
        RC(lin_t(m, m->mlp[k][0], hs, Mq, nullptr, true, m->h1, st));
        RC(lin_t(m, m->mlp[k][1], m->h1, Mq, nullptr, true, m->h2, st));
        RC(lin_t(m, m->mlp[k][2], m->h2, Mq, nullptr, false, outs[k], st, 1));
    }
    RC(odam_dk::launch_sigmoid(boxes, Mq * 4, st));
    return 0;
}

extern "C" int odam_detr_forward(odam_detr* m, const float* img, int B, float* logits, float* boxes, float* angle,
                                 float* offset, float* size, float* depth, float* obj_features, void* stream) {
    return forward_impl(m, img, B, nullptr, nullptr, logits, boxes, angle, offset, size, depth, obj_features, stream);
}

extern "C" int odam_detr_forward_masked(odam_detr* m, const float* img, int B, const unsigned char* key_mask,
                                        const float* pos, float* logits, float* boxes, float* angle, float* offset,
                                        float* size, float* depth, float* obj_features, void* stream) {
    if (!key_mask || !pos) return odam_fail(1, "odam_detr_forward_masked: null mask / position table");
    return forward_impl(m, img, B, key_mask, pos, logits, boxes, angle, offset, size, depth, obj_features, stream);
}

extern "C" int odam_detr_debug_read(odam_detr* m, int B, float* layer4_nchw, float* memory, void* stream) {
    if (!m || !m->finalized || !m->layer4) return odam_fail(1, "odam_detr_debug_read: no forward has run");
    hipStream_t st = (hipStream_t)stream;
    if (layer4_nchw) RC(odam_dk::launch_nhwc_to_nchw(m->layer4, layer4_nchw, B, m->fh, m->fw, 2048, m->dt, st));
    if (memory) RC(odam_dk::launch_to_f32(m->memory_out ? m->memory_out : m->src, memory, (size_t)B * m->L * m->cfg.hidden_dim, m->dt, st));
    return 0;
}

extern "C" int odam_detr_postprocess(odam_detr* m, const float* logits, const float* boxes, const float* angle,
                                     const float* offset, const float* size, const float* depth, int B,
                                     const float* K9, float img_w, float img_h, float* rows, void* stream) {
    if (!m || !logits || !boxes || !angle || !offset || !size || !depth || !K9 || !rows)
        return odam_fail(1, "odam_detr_postprocess: null pointer");
    return odam_dk::launch_postprocess(logits, boxes, angle, offset, size, depth, B, m->cfg.num_queries,
                                       m->cfg.num_classes1, m->cfg.angle_bins, img_w, img_h, K9[0], K9[4], K9[2],
                                       K9[5], rows, (hipStream_t)stream);
}

// ---- single-operator entry points -------------------------------------------------------------------
extern "C" int odam_op_conv2d_nhwc(const float* x, const float* w_packed, const float* scale, const float* bias,
                                   const float* residual, float* y, int B, int H, int W, int Cin, int Cout, int KH,
                                   int KW, int stride, int pad, int Kpad, int relu, int k_order, void* stream) {
    if (!x || !w_packed || !y) return odam_fail(1, "odam_op_conv2d_nhwc: null pointer");
    Conv c;
    c.w = const_cast<float*>(w_packed); c.scale = const_cast<float*>(scale); c.bias = const_cast<float*>(bias);
    c.Cin = Cin; c.Cout = Cout; c.KH = KH; c.KW = KW; c.stride = stride; c.pad = pad; c.Kpad = Kpad;
    c.k_order = k_order;
    // Split contraction mode: a model keeps its filters pre-split into three bf16 planes (upload_w3); this op-level entry builds them for the
    // one call (download, split on the host, upload), so that the op tests drive the SAME kernels a forward does -- the 16x16x32 loop with its
    // ragged last row tile and ragged last columns -- and not only the layer shapes a DETR happens to have.
    unsigned short* d3 = nullptr;
    if (odam_cg::f32_mode() == 2 && Kpad % 16 == 0 && (long)Cout * Kpad * 6 < 0x7fffffffL) {
        std::vector<float> hw((size_t)Cout * Kpad);
        ODAM_HIP(hipStreamSynchronize((hipStream_t)stream));
        ODAM_HIP(hipMemcpy(hw.data(), w_packed, hw.size() * sizeof(float), hipMemcpyDeviceToHost));
        std::vector<unsigned short> h3(hw.size() * 3);
        odam_cg::split3_filters(hw.data(), Cout, Kpad, h3.data());
        ODAM_HIP(hipMalloc(&d3, h3.size() * 2));
        if (hipMemcpy(d3, h3.data(), h3.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d3); return odam_fail(1, "odam_op_conv2d_nhwc: upload of the split filters failed"); }
        c.w3 = d3;
    }
    const int rc = run_conv(c, x, B, H, W, residual, relu != 0, y, 0, (hipStream_t)stream);
    if (d3) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(d3); }
    return rc;
}

extern "C" int odam_op_conv2d_nhwc_bf16(const void* x, const void* w_packed, const float* scale, const float* bias,
                                        const void* residual, void* y, int B, int H, int W, int Cin, int Cout, int KH,
                                        int KW, int stride, int pad, int Kpad, int relu, int out_f32, int k_order, void* stream) {
    if (!x || !w_packed || !y) return odam_fail(1, "odam_op_conv2d_nhwc_bf16: null pointer");
    Conv c;
    c.w = const_cast<void*>(w_packed); c.scale = const_cast<float*>(scale); c.bias = const_cast<float*>(bias);
    c.Cin = Cin; c.Cout = Cout; c.KH = KH; c.KW = KW; c.stride = stride; c.pad = pad; c.Kpad = Kpad;
    c.k_order = k_order;
    return run_conv(c, x, B, H, W, residual, relu != 0, y, 0, (hipStream_t)stream, 1, out_f32);
}

// A whole bf16 bottleneck tail as ONE launch (conv_gemm.hip fused_bf16): 3x3 (P -> P channels, stride 1 or 2, pad 1; filters
// packed k_order 1) + scale / bias / ReLU, 1x1 expand to 4 P channels + scale / bias + residual + ReLU -> y, and optionally the
// next block's 1x1 reduce (4 P -> PN channels, scale / bias / ReLU) of y -> y_next.  Returns 4 where the fused kernel does
// not apply to the shape (the caller then runs the layers one by one with odam_op_conv2d_nhwc_bf16; results are bit-identical).
extern "C" int odam_op_bottleneck_bf16(const void* x, const void* w2, const float* s2, const float* b2, const void* w3,
                                       const float* s3, const float* b3, const void* residual, void* y, const void* w1n,
                                       const float* s1n, const float* b1n, void* y_next, int B, int H, int W, int P, int stride,
                                       int PN, void* stream) {
    if (!x || !w2 || !w3 || !y) return odam_fail(1, "odam_op_bottleneck_bf16: null pointer");
    ConvGemmArgs a{};
    a.dtype = 1;
    a.A = x; a.Wt = w2; a.scale = s2; a.bias = b2;
    a.B = B; a.H = H; a.W = W; a.Cin = P; a.log2Cin = ilog2(P);
    a.Ho = conv_out(H, 3, stride, 1); a.Wo = conv_out(W, 3, stride, 1);
    a.Cout = P; a.KH = 3; a.KW = 3; a.stride = stride; a.pad = 1; a.Kpad = 9 * P;
    a.relu = 1; a.M = B * a.Ho * a.Wo; a.ldc = P; a.k_order = 1;
    a.F_Wt = w3; a.F_scale = s3; a.F_bias = b3; a.F_res = (const float*)residual; a.F_C = (float*)y; a.F_ldc = 4 * P; a.F_relu = 1;
    if (w1n && PN > 0) { a.G_Wt = w1n; a.G_scale = s1n; a.G_bias = b1n; a.G_C = (float*)y_next; a.G_N = PN; }
    if (!odam_cg::fused_bf16_ok(a)) return odam_fail(4, "odam_op_bottleneck_bf16: the fused kernel does not apply to this shape");
    return odam_cg::launch_conv_gemm(a, (hipStream_t)stream);
}

// experiment switch of the bf16-native contraction kernel (conv_gemm.h set_big_mode): 0 off, 1 auto, 2 whenever eligible
extern "C" int odam_op_conv_bf16_mode(int mode) {
    if (mode < 0 || mode > 2) return odam_fail(1, "odam_op_conv_bf16_mode: mode must be 0, 1 or 2");
    odam_cg::set_big_mode(mode);      // = odam_config_set("cg.ring", mode)
    return 0;
}

// fp32 contraction mode of the 256-row kernel (conv_gemm.h set_f32_mode): 0 fp32 matrix instruction on 128x128 tiles,
// 1 the same instruction in the ring kernel, 2 bf16 matrix instruction through the exact three-way split
extern "C" int odam_op_conv_f32_mode(int mode) {
    if (mode != 0 && mode != 2) return odam_fail(1, "odam_op_conv_f32_mode: mode must be 0 or 2");
    odam_cg::set_f32_mode(mode);
    return 0;
}

extern "C" long long odam_op_pooled_stem_launches(void) { return g_pooled_stem_launches.load(); }

extern "C" int odam_op_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O,
                                 int ldo, int B, int H, int Lq, int Lk, void* stream) {
    if (!Q || !K || !V || !O) return odam_fail(1, "odam_op_attention: null pointer");
    return odam_dk::launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Lq, Lk, 0, (hipStream_t)stream);
}

extern "C" int odam_op_attention_bf16(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                                      int ldo, int B, int H, int Lq, int Lk, void* stream) {
    if (!Q || !K || !V || !O) return odam_fail(1, "odam_op_attention_bf16: null pointer");
    if (ldq % 8 || ldk % 8 || ldv % 8 || ldo % 4) return odam_fail(1, "odam_op_attention_bf16: row pitches must be multiples of 8 / 8 / 8 / 4 elements");
    return odam_dk::launch_attention(Q, ldq, K, ldk, V, ldv, O, ldo, B, H, Lq, Lk, 1, (hipStream_t)stream);
}

extern "C" int odam_op_add_layernorm(const float* x, const float* r, const float* gamma, const float* beta, float* y,
                                     int M, void* stream) {
    if (!x || !gamma || !beta || !y) return odam_fail(1, "odam_op_add_layernorm: null pointer");
    return odam_dk::launch_add_layernorm(x, r, gamma, beta, y, nullptr, 1, nullptr, M, 0, (hipStream_t)stream);
}

extern "C" int odam_op_maxpool3x3s2_nhwc(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    if (!x || !y || C % 4) return odam_fail(1, "odam_op_maxpool3x3s2_nhwc: bad argument");
    return odam_dk::launch_maxpool3x3s2(x, y, B, H, W, C, conv_out(H, 3, 2, 1), conv_out(W, 3, 2, 1), 0,
                                        (hipStream_t)stream);
}

// ---- per-launch timing of the contraction kernel (roofline measurement) ---------------------------------
extern "C" int odam_detr_profile_enable(odam_detr* m, int on) {
    if (!m) return odam_fail(1, "odam_detr_profile_enable: null model");
    m->profile = on != 0;
    m->ev_used = 0;
    return 0;
}

extern "C" int odam_detr_profile_read(odam_detr* m, int* n_launches, double* total_ms, double* total_flops) {
    if (!m || !n_launches || !total_ms || !total_flops) return odam_fail(1, "odam_detr_profile_read: null argument");
    double ms = 0.0, fl = 0.0;
    for (size_t i = 0; i + 1 < m->ev_used; i += 2) {
        ODAM_HIP(hipEventSynchronize(m->ev[i + 1]));
        float t = 0.0f;
        ODAM_HIP(hipEventElapsedTime(&t, m->ev[i], m->ev[i + 1]));
        ms += t;
        fl += m->ev_flops[i / 2];
    }
    *n_launches = (int)(m->ev_used / 2);
    *total_ms = ms;
    *total_flops = fl;
    return 0;
}

extern "C" int odam_detr_profile_read_attention(odam_detr* m, int* n_launches, double* total_ms, double* total_flops) {
    if (!m || !n_launches || !total_ms || !total_flops) return odam_fail(1, "odam_detr_profile_read_attention: null argument");
    double ms = 0.0, fl = 0.0;
    for (size_t i = 0; i + 1 < m->aev_used; i += 2) {
        ODAM_HIP(hipEventSynchronize(m->aev[i + 1]));
        float t = 0.0f;
        ODAM_HIP(hipEventElapsedTime(&t, m->aev[i], m->aev[i + 1]));
        ms += t;
        fl += m->aev_flops[i / 2];
    }
    *n_launches = (int)(m->aev_used / 2);
    *total_ms = ms;
    *total_flops = fl;
    return 0;
}

// ---- threshold + greedy 3D/2D NMS on one frame's post-processed rows (host, float32 as the reference's numpy) --
// Pillow src/libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1)
// filter over the whole image: what PIL's Image.resize(BILINEAR) -- torchvision F.resize on a PIL image, the
// reference's transforms.py:105 -- builds before its two integer passes.  Double arithmetic, C truncating casts.
static int resample_table(odam_detr* m, int in_size, int out_size, odam_detr::Resample* r) {
    auto it = m->resample.find({in_size, out_size});
    if (it != m->resample.end()) { *r = it->second; return 0; }
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;
    const int ksize = (int)std::ceil(support) * 2 + 1;
    std::vector<int> xmin(out_size), cnt(out_size), K((size_t)out_size * ksize, 0);
    std::vector<double> w(ksize);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; xx++) {
        const double center = (xx + 0.5) * scale;
        int lo = (int)(center - support + 0.5);
        if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5);
        if (hi > in_size) hi = in_size;
        const int n = hi - lo;
        double ww = 0.0;
        for (int x = 0; x < ksize; x++) w[x] = 0.0;
        for (int x = 0; x < n; x++) {
            double t = (x + lo - center + 0.5) * ss;
            if (t < 0.0) t = -t;
            w[x] = t < 1.0 ? 1.0 - t : 0.0;
            ww += w[x];
        }
        for (int x = 0; x < n; x++)
            if (ww != 0.0) w[x] /= ww;
        for (int x = 0; x < ksize; x++)
            K[(size_t)xx * ksize + x] = w[x] < 0 ? (int)(-0.5 + w[x] * (1 << 22)) : (int)(0.5 + w[x] * (1 << 22));
        xmin[xx] = lo; cnt[xx] = n;
    }
    odam_detr::Resample t{};
    t.ksize = ksize;
    if (int rc = m->dev_alloc(&t.xmin, xmin.size())) return rc;
    if (int rc = m->dev_alloc(&t.cnt, cnt.size())) return rc;
    if (int rc = m->dev_alloc(&t.K, K.size())) return rc;
    ODAM_HIP(hipMemcpy(t.xmin, xmin.data(), xmin.size() * sizeof(int), hipMemcpyHostToDevice));
    ODAM_HIP(hipMemcpy(t.cnt, cnt.data(), cnt.size() * sizeof(int), hipMemcpyHostToDevice));
    ODAM_HIP(hipMemcpy(t.K, K.data(), K.size() * sizeof(int), hipMemcpyHostToDevice));
    m->resample[{in_size, out_size}] = t;
    *r = t;
    return 0;
}

// transforms.py:281-290 on the device: resize to the model's (img_h, img_w) + ToTensor + Normalize
extern "C" int odam_detr_preprocess_u8(odam_detr* m, const unsigned char* rgb, int B, int h, int w, const float* mean,
                                       const float* stdv, float* out, void* stream) {
    if (!m || !rgb || !mean || !stdv || !out) return odam_fail(1, "odam_detr_preprocess_u8: null pointer");
    if (B < 0 || h < 1 || w < 1) return odam_fail(1, "odam_detr_preprocess_u8: bad size");
    odam_detr::Resample tx{}, ty{};
    RC(resample_table(m, w, m->cfg.img_w, &tx));
    RC(resample_table(m, h, m->cfg.img_h, &ty));
    return odam_dk::launch_preprocess_u8(rgb, B, h, w, tx.xmin, tx.cnt, tx.K, tx.ksize, ty.xmin, ty.cnt, ty.K, ty.ksize,
                                         out, m->cfg.img_h, m->cfg.img_w, mean, stdv, (hipStream_t)stream);
}

// detr.py:124-125 (keep = score > threshold, in query order) and :161-205 (nms_3d):
// candidates sorted by descending score; a later candidate is suppressed by a kept one when
// (same class and 3D-AABB IoU > 0.25) or (nms_2d and 2D IoU > 0.5).  keep_idx receives query indices in kept order.
extern "C" int odam_detr_select(const float* rows, int Q, float threshold, int nms_2d, int* keep_idx, int* n_keep) {
    if (!rows || !keep_idx || !n_keep || Q < 0) return odam_fail(1, "odam_detr_select: bad argument");
    std::vector<int> cand;
    for (int q = 0; q < Q; q++)
        if (rows[(size_t)q * 16] > threshold) cand.push_back(q);
    std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return rows[(size_t)a * 16] < rows[(size_t)b * 16]; });
    std::reverse(cand.begin(), cand.end());
    const int n = (int)cand.size();
    std::vector<char> sup(n, 0);
    auto box3 = [&](int q, float lo[3], float hi[3]) {
        const float* r = rows + (size_t)q * 16;
        for (int k = 0; k < 3; k++) {
            lo[k] = (-r[10 + k]) / 2.0f + r[6 + k];
            hi[k] = r[10 + k] / 2.0f + r[6 + k];
        }
    };
    int nk = 0;
    for (int i = 0; i < n; i++) {
        if (sup[i]) continue;
        const int s = cand[i];
        keep_idx[nk++] = s;
        float slo[3], shi[3];
        box3(s, slo, shi);
        const float* rs = rows + (size_t)s * 16;
        for (int j = i + 1; j < n; j++) {
            if (sup[j]) continue;
            const int t = cand[j];
            const float* rt = rows + (size_t)t * 16;
            float tlo[3], thi[3];
            box3(t, tlo, thi);
            float inter = 1.0f, va = 1.0f, vb = 1.0f;
            {
                const float dx = std::max(0.0f, std::min(shi[0], thi[0]) - std::max(slo[0], tlo[0]));
                const float dy = std::max(0.0f, std::min(shi[1], thi[1]) - std::max(slo[1], tlo[1]));
                const float dz = std::max(0.0f, std::min(shi[2], thi[2]) - std::max(slo[2], tlo[2]));
                inter = dx * dy * dz;
                va = (shi[0] - slo[0]) * (shi[1] - slo[1]) * (shi[2] - slo[2]);
                vb = (thi[0] - tlo[0]) * (thi[1] - tlo[1]) * (thi[2] - tlo[2]);
            }
            const float iou3 = inter / (va + vb - inter);
            if (rt[1] == rs[1] && iou3 > 0.25f) { sup[j] = 1; continue; }
            if (nms_2d) {
                const float ix = std::max(0.0f, std::min(rs[4], rt[4]) - std::max(rs[2], rt[2]));
                const float iy = std::max(0.0f, std::min(rs[5], rt[5]) - std::max(rs[3], rt[3]));
                const float ia = ix * iy;
                const float aa = (rs[4] - rs[2]) * (rs[5] - rs[3]);
                const float ab = (rt[4] - rt[2]) * (rt[5] - rt[3]);
                if (ia / (aa + ab - ia) > 0.5f) sup[j] = 1;
            }
        }
    }
    *n_keep = nk;
    return 0;
}
