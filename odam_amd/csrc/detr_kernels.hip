// detr_kernels.hip -- see detr_kernels.h
#include "detr_kernels.h"

#include <cstdlib>

#include "odam_config.h"
#include "odam_err.h"

namespace odam_dk {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned short bf16_t;   // raw bfloat16 bits (bf16 mode: activations in HBM are bf16, arithmetic stays fp32)

__device__ __forceinline__ float bf2f(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// two floats -> one register of two bf16 (lo in the low half), round to nearest even: ONE v_cvt_pk_bf16_f32 (the scalar casts of
// f2bf(lo) | f2bf(hi) << 16 compile to two conversions, a shift and an or)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
    const ushort4 u = *reinterpret_cast<const ushort4*>(p);
    return float4{bf2f(u.x), bf2f(u.y), bf2f(u.z), bf2f(u.w)};
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
    *reinterpret_cast<uint2*>(p) = uint2{pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)};
}
#define ODAM_DISPATCH(dtype, KERNEL, ...)                                                  \
    do {                                                                                  \
        if ((dtype) == 1) { KERNEL(bf16_t, __VA_ARGS__); } else { KERNEL(float, __VA_ARGS__); } \
    } while (0)

// =================================================================================================
// Fused attention, d_head = 32, fp32 on v_mfma_f32_32x32x2_f32.
//
// Workgroup = 4 wavefronts, each owns 32 query rows of one (batch, head); the workgroup walks the
// keys in tiles of 32 staged once in LDS for all four wavefronts (double buffered).
//   S^T tile (keys x queries) = K_tile . Q^T        16 MFMA (K-dim = 32 head channels, 2 per MFMA)
//   online softmax over keys: the key index lives in the accumulator REGISTER (plus the lane
//     half), the query on the lane -> row max / row sum are 15 in-register ops + one lane^32 swap
//   O^T (channels x queries) += V_tile^T . P^T       16 MFMA; the S^T accumulator registers ARE the
//     B operand of this product (same lane <-> query, same register <-> key), so P never leaves
//     the register file and stays fp32.
// Reference semantics: torch.nn.functional.multi_head_attention_forward as called from
// src/models/transformer.py:154-167, 217-238 (q scaled by sqrt(1/32) before the product, softmax
// over keys, no dropout at inference, padding mask all-false for same-size frames).
// =================================================================================================
constexpr int ATT_KT = 32;          // keys per tile

__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// D = head dimension: 32 (detector, 8 heads) or 64 (associator GNN, 4 heads)
template <typename T, int D>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ Q, int ldq,
                                                        const T* __restrict__ K, int ldk,
                                                        const T* __restrict__ V, int ldv,
                                                        T* __restrict__ O, int ldo, int Lq, int Lk, float scale,
                                                        const unsigned char* __restrict__ key_mask) {
    // key_mask (nullable): [B][Lk], non-zero = padded key, excluded from the softmax (key_padding_mask of
    // nn.MultiheadAttention as src/models/transformer.py:157-160, 224-228 pass it)
    constexpr int LD = D + 4;        // LDS row stride (floats)
    constexpr int HD = D / 2;        // channels of the QK^T reduction held by one lane half
    constexpr int NO = D / 32;       // 32-channel output tiles
    constexpr int CPR = D / 4;       // 16-byte... 4-channel chunks per K/V row
    constexpr int RPP = 256 / CPR;   // K/V rows staged per pass
    __shared__ __align__(16) float Ks[2][ATT_KT][LD];
    __shared__ __align__(16) float Vs[2][ATT_KT][LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const T* Qb = Q + (size_t)b * Lq * ldq + h * D;
    const T* Kb = K + (size_t)b * Lk * ldk + h * D;
    const T* Vb = V + (size_t)b * Lk * ldv + h * D;

    // this lane's HD query channels d = HD*half + s, pre-scaled (a power-of-two scale commutes with the product)
    float qreg[HD];
    {
        const int q = q0 + col;
#pragma unroll
        for (int g = 0; g < HD / 4; g++) {
            float4 v = (q < Lq) ? ld4(Qb + (size_t)q * ldq + HD * half + 4 * g) : float4{0.f, 0.f, 0.f, 0.f};
            qreg[4 * g + 0] = v.x * scale; qreg[4 * g + 1] = v.y * scale;
            qreg[4 * g + 2] = v.z * scale; qreg[4 * g + 3] = v.w * scale;
        }
    }

    const int lr = tid / CPR, lc = (tid % CPR) * 4;  // tile loader: row, 4-channel column
    constexpr int NLD = ATT_KT / RPP;                // passes to stage one 32-key tile (1 for D = 32, 2 for D = 64)
    auto load_kv = [&](int kt, float4* kv, float4* vv) {
#pragma unroll
        for (int pss = 0; pss < NLD; pss++) {
            const int key = kt * ATT_KT + lr + pss * RPP;
            if (key < Lk) {
                kv[pss] = ld4(Kb + (size_t)key * ldk + lc);
                vv[pss] = ld4(Vb + (size_t)key * ldv + lc);
            } else {
                kv[pss] = float4{0.f, 0.f, 0.f, 0.f};
                vv[pss] = float4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    auto store_kv = [&](int buf, const float4* kv, const float4* vv) {
#pragma unroll
        for (int pss = 0; pss < NLD; pss++) {
            *reinterpret_cast<float4*>(&Ks[buf][lr + pss * RPP][lc]) = kv[pss];
            *reinterpret_cast<float4*>(&Vs[buf][lr + pss * RPP][lc]) = vv[pss];
        }
    };

    floatx16 oacc[NO];
#pragma unroll
    for (int o = 0; o < NO; o++)
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[o][r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;

    const int ntiles = (Lk + ATT_KT - 1) / ATT_KT;
    float4 kv[NLD], vv[NLD];
    load_kv(0, kv, vv);
    store_kv(0, kv, vv);
    __syncthreads();

    for (int kt = 0; kt < ntiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ntiles) load_kv(kt + 1, kv, vv);

        // S^T = K . Q^T   (rows = keys, cols = queries)
        floatx16 s;
#pragma unroll
        for (int r = 0; r < 16; r++) s[r] = 0.0f;
#pragma unroll
        for (int g = 0; g < HD / 4; g++) {
            const float4 kf = *reinterpret_cast<const float4*>(&Ks[cur][col][HD * half + 4 * g]);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qreg[4 * g + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qreg[4 * g + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qreg[4 * g + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qreg[4 * g + 3], s, 0, 0, 0);
        }
        // mask keys beyond Lk (last tile only) and padded keys
        const int kbase = kt * ATT_KT;
        if (kbase + ATT_KT > Lk) {
#pragma unroll
            for (int r = 0; r < 16; r++)
                if (kbase + acc_row(r, half) >= Lk) s[r] = -1e30f;
        }
        if (key_mask) {
            const unsigned char* mk = key_mask + (size_t)b * Lk + kbase;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int k = acc_row(r, half);
                if (kbase + k < Lk && mk[k]) s[r] = -1e30f;
            }
        }
        // online softmax over the key axis (registers + the other lane half)
        float mx = s[0];
#pragma unroll
        for (int r = 1; r < 16; r++) mx = fmaxf(mx, s[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            s[r] = __expf(s[r] - m_new);
            psum += s[r];
        }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int o = 0; o < NO; o++)
#pragma unroll
            for (int r = 0; r < 16; r++) oacc[o][r] *= alpha;
        // O^T += V^T . P^T : k-slot (step r, lane half) <-> key acc_row(r, half) on both operands
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int o = 0; o < NO; o++) {
                const float vf = Vs[cur][acc_row(r, half)][o * 32 + col];
                oacc[o] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, s[r], oacc[o], 0, 0, 0);
            }
        }
        if (kt + 1 < ntiles) store_kv(cur ^ 1, kv, vv);
        __syncthreads();
    }

    const int q = q0 + col;
    if (q < Lq) {
        const float inv = 1.0f / l_run;
        T* Ob = O + ((size_t)b * Lq + q) * ldo + h * D;
#pragma unroll
        for (int o = 0; o < NO; o++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                float4 v = {oacc[o][4 * g + 0] * inv, oacc[o][4 * g + 1] * inv, oacc[o][4 * g + 2] * inv,
                            oacc[o][4 * g + 3] * inv};
                st4(Ob + o * 32 + 8 * g + 4 * half, v);  // channels o*32 + acc_row(4g..4g+3, half)
            }
    }
}

// =================================================================================================
// Fused attention on the bf16 matrix instruction (bf16 mode, BASELINE config 4), d_head = 32.
// Same dataflow as above with v_mfma_f32_32x32x16_bf16: per 64-key tile 4 MFMAs give S^T = K . Q^T (two 32-key blocks x
// two 16-channel k-steps) and 4 more O^T += V^T . P^T -- 8 x 32 cycles instead of 64 x 64.  P stays in registers: the
// S^T accumulator (query on the lane, key in the register) converted pairwise to bf16 IS the B operand of the PV
// product; its k order inside a 16-key step is 16 s + 8 (j >> 2) + 4 half + (j & 3), so the A operand (V^T) is read
// from an LDS image [channel][key] as two 4-key groups per step.  Softmax in fp32 with exp2 (scale * log2 e folded in).
// LDS: K tile [64 keys][64 B] with the 16-byte chunk c of key r at c ^ ((r >> 2) & 3) (conflict-free ds_read_b128, as
// in conv_gemm's bf16 kernel); V tile [64 keys][64 B] row-major, read transposed by ds_read_b64_tr_b16 (see the kernel).
// =================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void attention_bf16_kernel(const bf16_t* __restrict__ Q, int ldq,
                                                             const bf16_t* __restrict__ K, int ldk,
                                                             const bf16_t* __restrict__ V, int ldv,
                                                             bf16_t* __restrict__ O, int ldo, int Lq, int Lk, float c_log2,
                                                             const unsigned char* __restrict__ key_mask) {
    constexpr int KT = 64, D = 32;
    __shared__ __align__(16) uint4 Ks[2][KT * 4];
    // V stays row-major in LDS ([key][32 channels], 64-byte rows, as it comes from HBM: one 16-byte store per thread) and reaches the PV
    // product's A operand (V^T: channel on the lane, four consecutive keys in the register pair) through ds_read_b64_tr_b16 -- the hardware
    // transpose: a 16-lane group reads a block of 4 keys x 16 channels and lane i receives channel i of the four keys.  (Round 6; until
    // then every thread wrote its 8 values into a transposed image with 8 two-byte stores per tile.  Same values in the same operand
    // lanes: bit-identical.)  The 32 lanes of a half read 4 keys x 64 bytes = 256 contiguous bytes: conflict-free without a swizzle.
    __shared__ __align__(16) uint4 Vs[2][KT * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const bf16_t* Qb = Q + (size_t)b * Lq * ldq + h * D;
    const bf16_t* Kb = K + (size_t)b * Lk * ldk + h * D;
    const bf16_t* Vb = V + (size_t)b * Lk * ldv + h * D;

    uint4 qf[2];     // B operand of S^T = K . Q^T: lane (query col, half) holds Q[q][16 s + 8 half .. + 7]
    {
        const int q = q0 + col;
#pragma unroll
        for (int st = 0; st < 2; st++)
            qf[st] = (q < Lq) ? *reinterpret_cast<const uint4*>(Qb + (size_t)q * ldq + 16 * st + 8 * half) : uint4{0u, 0u, 0u, 0u};
    }
    const int lk = tid >> 2, lc = tid & 3;          // loader: key row of the tile, 8-channel chunk
    auto load_kv = [&](int kt, uint4& kr, uint4& vr) {
        const int key = kt * KT + lk;
        if (key < Lk) {
            kr = *reinterpret_cast<const uint4*>(Kb + (size_t)key * ldk + lc * 8);
            vr = *reinterpret_cast<const uint4*>(Vb + (size_t)key * ldv + lc * 8);
        } else {
            kr = uint4{0u, 0u, 0u, 0u}; vr = uint4{0u, 0u, 0u, 0u};
        }
    };
    auto store_kv = [&](int buf, const uint4& kr, const uint4& vr) {
        Ks[buf][lk * 4 + (lc ^ ((lk >> 2) & 3))] = kr;
        Vs[buf][lk * 4 + lc] = vr;
    };
    // transposed read of V: lane 4 q + p of 16-lane group g supplies the address of key row q, channels 4 p .. 4 p + 3 of the group's
    // channel block (g & 1); the group's half (g >> 1) selects keys 4 half .. of an 8-key step, as the P operand's k order wants them
    typedef short tr4_t __attribute__((ext_vector_type(4)));
    const int tr_off = (((lane >> 5) * 4 + ((lane & 15) >> 2)) * 64) + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;      // bytes inside a tile image

    floatx16 oacc;
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;
    const int ntiles = (Lk + KT - 1) / KT;
    uint4 kr, vr;
    load_kv(0, kr, vr);
    store_kv(0, kr, vr);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ntiles) load_kv(kt + 1, kr, vr);
        floatx16 s[2];
#pragma unroll
        for (int jh = 0; jh < 2; jh++) {
#pragma unroll
            for (int r = 0; r < 16; r++) s[jh][r] = 0.0f;
            const int key = jh * 32 + col;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const uint4 kf = Ks[cur][key * 4 + ((2 * st + half) ^ ((key >> 2) & 3))];
                s[jh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf), __builtin_bit_cast(bf16x8_t, qf[st]),
                                                                s[jh], 0, 0, 0);
            }
        }
        const int kbase = kt * KT;
        if (kbase + KT > Lk || key_mask) {
            const unsigned char* mk = key_mask ? key_mask + (size_t)b * Lk + kbase : nullptr;
#pragma unroll
            for (int jh = 0; jh < 2; jh++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = jh * 32 + acc_row(r, half);
                    if (kbase + k >= Lk || (mk && mk[k])) s[jh][r] = -1e30f;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int jh = 0; jh < 2; jh++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[jh][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_log2);
        const float mc = m_new * c_log2;
        float psum = 0.0f;
#pragma unroll
        for (int jh = 0; jh < 2; jh++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                s[jh][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[jh][r], c_log2, -mc));
                psum += s[jh][r];
            }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[r] *= alpha;
#pragma unroll
        for (int jh = 0; jh < 2; jh++)
#pragma unroll
            for (int st = 0; st < 2; st++) {
                unsigned pw[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    pw[e] = pack_bf16x2(s[jh][8 * st + 2 * e], s[jh][8 * st + 2 * e + 1]);
                typedef tr4_t __attribute__((address_space(3))) * lds_tr_ptr;
                const char* vb = reinterpret_cast<const char*>(&Vs[cur][0]) + (jh * 32 + 16 * st) * 64 + tr_off;
                const tr4_t lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vb));               // keys 4 half .. + 3 of the step
                const tr4_t hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vb + 8 * 64));      // keys 8 + 4 half .. + 3
                const uint2 lo = __builtin_bit_cast(uint2, lo4), hi = __builtin_bit_cast(uint2, hi4);
                const uint4 vf = uint4{lo.x, lo.y, hi.x, hi.y};
                oacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf),
                                                               __builtin_bit_cast(bf16x8_t, uint4{pw[0], pw[1], pw[2], pw[3]}), oacc, 0, 0, 0);
            }
        if (kt + 1 < ntiles) store_kv(cur ^ 1, kr, vr);
        __syncthreads();
    }
    const int q = q0 + col;
    if (q < Lq) {
        const float inv = 1.0f / l_run;
        bf16_t* Ob = O + ((size_t)b * Lq + q) * ldo + h * D;
#pragma unroll
        for (int g = 0; g < 4; g++)
            st4(Ob + 8 * g + 4 * half, float4{oacc[4 * g + 0] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv});
    }
}

// =================================================================================================
// Fused attention for fp32 operands on the bf16 matrix instruction, d_head = 32: the dataflow of attention_bf16_kernel
// with every product taken through the exact three-way bf16 split of conv_gemm.hip (a = hi + mid + lo by truncation, six
// bf16 x bf16 products per fp32 product, fp32 accumulate; max error vs float64 0.8-1.4e-7 of sum |a b|, the fp32 matrix
// instruction's own chain 1.1-1.9e-7).  K and V tiles are split once when they are staged (three bf16 planes each in
// LDS, K key-major with the swizzle of the bf16 kernel, V transposed), the pre-scaled query fragment is split once per
// workgroup, the probabilities -- fp32 in the S^T accumulators -- are split in registers and are the B operand of the PV
// product as they stand.  Per 64-key tile and wave 48 matrix instructions of 32 cycles against 64 of 64 on
// v_mfma_f32_32x32x2_f32.  Softmax as in the fp32 kernel (q pre-scaled, __expf).
// =================================================================================================
__device__ __forceinline__ void att_split8(const float4& c0, const float4& c1, uint4& hi, uint4& mid, uint4& lo) {
    const float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    unsigned uh[8], um[8], ul[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        uh[e] = __builtin_bit_cast(unsigned, v[e]);
        const float r1 = v[e] - __builtin_bit_cast(float, uh[e] & 0xffff0000u);
        um[e] = __builtin_bit_cast(unsigned, r1);
        ul[e] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, um[e] & 0xffff0000u));
    }
    auto pk = [](unsigned lo_, unsigned hi_) { return __builtin_amdgcn_perm(hi_, lo_, 0x07060302u); };   // two top halves
    hi = uint4{pk(uh[0], uh[1]), pk(uh[2], uh[3]), pk(uh[4], uh[5]), pk(uh[6], uh[7])};
    mid = uint4{pk(um[0], um[1]), pk(um[2], um[3]), pk(um[4], um[5]), pk(um[6], um[7])};
    lo = uint4{pk(ul[0], ul[1]), pk(ul[2], ul[3]), pk(ul[4], ul[5]), pk(ul[6], ul[7])};
}

__global__ __launch_bounds__(256) void attention_x3_kernel(const float* __restrict__ Q, int ldq,
                                                           const float* __restrict__ K, int ldk,
                                                           const float* __restrict__ V, int ldv,
                                                           float* __restrict__ O, int ldo, int Lq, int Lk, float scale,
                                                           const unsigned char* __restrict__ key_mask) {
    constexpr int KT = 64, D = 32;
    __shared__ __align__(16) uint4 Ks[2][3][KT * 4];
    __shared__ __align__(16) uint4 Vs[2][3][KT * 4];      // V's three planes row-major, read transposed (ds_read_b64_tr_b16) as in attention_bf16_kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const float* Qb = Q + (size_t)b * Lq * ldq + h * D;
    const float* Kb = K + (size_t)b * Lk * ldk + h * D;
    const float* Vb = V + (size_t)b * Lk * ldv + h * D;

    uint4 qh[2], qm[2], ql[2];     // B operand of S^T = K . Q^T: lane (query col, half) holds Q[q][16 s + 8 half .. + 7], pre-scaled
    {
        const int q = q0 + col;
#pragma unroll
        for (int st = 0; st < 2; st++) {
            float4 a = float4{0.f, 0.f, 0.f, 0.f}, c = a;
            if (q < Lq) {
                a = ld4(Qb + (size_t)q * ldq + 16 * st + 8 * half);
                c = ld4(Qb + (size_t)q * ldq + 16 * st + 8 * half + 4);
            }
            a.x *= scale; a.y *= scale; a.z *= scale; a.w *= scale;
            c.x *= scale; c.y *= scale; c.z *= scale; c.w *= scale;
            att_split8(a, c, qh[st], qm[st], ql[st]);
        }
    }
    const int lk = tid >> 2, lc = tid & 3;          // loader: key row of the tile, 8-channel chunk
    auto load_kv = [&](int kt, float4 (&kr)[2], float4 (&vr)[2]) {
        const int key = kt * KT + lk;
        if (key < Lk) {
            kr[0] = ld4(Kb + (size_t)key * ldk + lc * 8); kr[1] = ld4(Kb + (size_t)key * ldk + lc * 8 + 4);
            vr[0] = ld4(Vb + (size_t)key * ldv + lc * 8); vr[1] = ld4(Vb + (size_t)key * ldv + lc * 8 + 4);
        } else {
            kr[0] = kr[1] = vr[0] = vr[1] = float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_kv = [&](int buf, const float4 (&kr)[2], const float4 (&vr)[2]) {
        uint4 p[3];
        att_split8(kr[0], kr[1], p[0], p[1], p[2]);
#pragma unroll
        for (int pl = 0; pl < 3; pl++) Ks[buf][pl][lk * 4 + (lc ^ ((lk >> 2) & 3))] = p[pl];
        att_split8(vr[0], vr[1], p[0], p[1], p[2]);
#pragma unroll
        for (int pl = 0; pl < 3; pl++) Vs[buf][pl][lk * 4 + lc] = p[pl];
    };
    typedef short tr4_t __attribute__((ext_vector_type(4)));
    typedef tr4_t __attribute__((address_space(3))) * lds_tr_ptr;
    const int tr_off = (((lane >> 5) * 4 + ((lane & 15) >> 2)) * 64) + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;      // see attention_bf16_kernel
#define ODAM_MMA(acc, a, bq) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bq), acc, 0, 0, 0)

    floatx16 oacc;
#pragma unroll
    for (int r = 0; r < 16; r++) oacc[r] = 0.0f;
    float m_run = -1e30f, l_run = 0.0f;
    const int ntiles = (Lk + KT - 1) / KT;
    float4 kr[2], vr[2];
    load_kv(0, kr, vr);
    store_kv(0, kr, vr);
    __syncthreads();
    for (int kt = 0; kt < ntiles; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < ntiles) load_kv(kt + 1, kr, vr);
        floatx16 s[2];
#pragma unroll
        for (int jh = 0; jh < 2; jh++) {
#pragma unroll
            for (int r = 0; r < 16; r++) s[jh][r] = 0.0f;
            const int key = jh * 32 + col;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const int idx = key * 4 + ((2 * st + half) ^ ((key >> 2) & 3));
                const uint4 kh = Ks[cur][0][idx], km = Ks[cur][1][idx], kl = Ks[cur][2][idx];
                ODAM_MMA(s[jh], kl, qh[st]); ODAM_MMA(s[jh], km, qm[st]); ODAM_MMA(s[jh], kh, ql[st]);     // smallest products first
                ODAM_MMA(s[jh], km, qh[st]); ODAM_MMA(s[jh], kh, qm[st]);
                ODAM_MMA(s[jh], kh, qh[st]);
            }
        }
        const int kbase = kt * KT;
        if (kbase + KT > Lk || key_mask) {
            const unsigned char* mk = key_mask ? key_mask + (size_t)b * Lk + kbase : nullptr;
#pragma unroll
            for (int jh = 0; jh < 2; jh++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int k = jh * 32 + acc_row(r, half);
                    if (kbase + k >= Lk || (mk && mk[k])) s[jh][r] = -1e30f;
                }
        }
        float mx = s[0][0];
#pragma unroll
        for (int jh = 0; jh < 2; jh++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[jh][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.0f;
#pragma unroll
        for (int jh = 0; jh < 2; jh++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                s[jh][r] = __expf(s[jh][r] - m_new);
                psum += s[jh][r];
            }
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; r++) oacc[r] *= alpha;
        // O^T += V^T . P^T: the accumulator registers 8 st .. 8 st + 7 of s[jh], split, are the B operand of k-step (jh, st)
#pragma unroll
        for (int jh = 0; jh < 2; jh++)
#pragma unroll
            for (int st = 0; st < 2; st++) {
                uint4 ph, pm, pl_;
                att_split8(float4{s[jh][8 * st + 0], s[jh][8 * st + 1], s[jh][8 * st + 2], s[jh][8 * st + 3]},
                           float4{s[jh][8 * st + 4], s[jh][8 * st + 5], s[jh][8 * st + 6], s[jh][8 * st + 7]}, ph, pm, pl_);
                uint4 vf[3];
#pragma unroll
                for (int pl = 0; pl < 3; pl++) {
                    const char* vb = reinterpret_cast<const char*>(&Vs[cur][pl][0]) + (jh * 32 + 16 * st) * 64 + tr_off;
                    const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vb)));
                    const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)(vb + 8 * 64)));
                    vf[pl] = uint4{lo.x, lo.y, hi.x, hi.y};
                }
                ODAM_MMA(oacc, vf[2], ph); ODAM_MMA(oacc, vf[1], pm); ODAM_MMA(oacc, vf[0], pl_);
                ODAM_MMA(oacc, vf[1], ph); ODAM_MMA(oacc, vf[0], pm);
                ODAM_MMA(oacc, vf[0], ph);
            }
        if (kt + 1 < ntiles) store_kv(cur ^ 1, kr, vr);
        __syncthreads();
    }
#undef ODAM_MMA
    const int q = q0 + col;
    if (q < Lq) {
        const float inv = 1.0f / l_run;
        float* Ob = O + ((size_t)b * Lq + q) * ldo + h * D;
#pragma unroll
        for (int g = 0; g < 4; g++)
            st4(Ob + 8 * g + 4 * half, float4{oacc[4 * g + 0] * inv, oacc[4 * g + 1] * inv, oacc[4 * g + 2] * inv, oacc[4 * g + 3] * inv});
    }
}

int launch_attention(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O, int ldo,
                     int B, int H, int Lq, int Lk, int dtype, hipStream_t stream, const unsigned char* key_mask) {
    if (Lq <= 0 || Lk <= 0 || B <= 0) return 0;
    dim3 grid((Lq + 127) / 128, H, B);
    const float scale = (float)0.1767766952966369;   // sqrt(1/32) as torch computes it in float64, then float32
    const bool bf16_mfma = odam_cfg::get(odam_cfg::ATT_BF16_MFMA) != 0;   // 0: keep bf16 mode on the fp32-instruction kernel
    if (dtype == 1 && bf16_mfma && (ldq % 8) == 0 && (ldk % 8) == 0 && (ldv % 8) == 0 && (ldo % 4) == 0) {
        hipLaunchKernelGGL(attention_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)Q, ldq, (const bf16_t*)K, ldk,
                           (const bf16_t*)V, ldv, (bf16_t*)O, ldo, Lq, Lk, scale * 1.44269504088896341f, key_mask);
        ODAM_HIP(hipGetLastError());
        return 0;
    }
    // fp32: products through the exact three-way bf16 split (ODAM_ATT_X3=0 keeps the fp32 matrix instruction)
    const bool x3 = odam_cfg::get(odam_cfg::ATT_X3) != 0;
    if (dtype == 0 && x3 && (ldq % 4) == 0 && (ldk % 4) == 0 && (ldv % 4) == 0 && (ldo % 4) == 0) {
        hipLaunchKernelGGL(attention_x3_kernel, grid, dim3(256), 0, stream, (const float*)Q, ldq, (const float*)K, ldk,
                           (const float*)V, ldv, (float*)O, ldo, Lq, Lk, scale, key_mask);
        ODAM_HIP(hipGetLastError());
        return 0;
    }
#define ODAM_K(T, ...) hipLaunchKernelGGL((attention_kernel<T, 32>), grid, dim3(256), 0, stream, (const T*)Q, ldq, (const T*)K, ldk, (const T*)V, ldv, (T*)O, ldo, Lq, Lk, scale, key_mask)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}

// head dimension 64, fp32 (associator GNN: scores / 64**0.5, src/models/associator.py:47-56)
int launch_attention_d64(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo,
                         int B, int H, int Lq, int Lk, hipStream_t stream) {
    if (Lq <= 0 || Lk <= 0 || B <= 0) return 0;
    dim3 grid((Lq + 127) / 128, H, B);
    hipLaunchKernelGGL((attention_kernel<float, 64>), grid, dim3(256), 0, stream, Q, ldq, K, ldk, V, ldv, O, ldo, Lq, Lk,
                       0.125f, (const unsigned char*)nullptr);
    ODAM_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
// residual add + LayerNorm over 256 channels: one wavefront per row, 4 channels per lane
// (nn.LayerNorm(256), eps 1e-5, biased variance; transformer.py:147-148, 160-166)
// =================================================================================================
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const T* __restrict__ x, const T* __restrict__ r,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, T* __restrict__ y,
                                                            const float* __restrict__ pos, int L,
                                                            T* __restrict__ y_pos, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    float4 v = ld4(x + (size_t)row * 256 + lane * 4);
    if (r) {
        const float4 t = ld4(r + (size_t)row * 256 + lane * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
    const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
    const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4 g = *reinterpret_cast<const float4*>(gamma + lane * 4);
    const float4 be = *reinterpret_cast<const float4*>(beta + lane * 4);
    float4 o = {dx * rstd * g.x + be.x, dy * rstd * g.y + be.y, dz * rstd * g.z + be.z, dw * rstd * g.w + be.w};
    st4(y + (size_t)row * 256 + lane * 4, o);
    if (y_pos) {
        const float4 p = *reinterpret_cast<const float4*>(pos + (size_t)(row % L) * 256 + lane * 4);
        float4 o2 = {o.x + p.x, o.y + p.y, o.z + p.z, o.w + p.w};
        st4(y_pos + (size_t)row * 256 + lane * 4, o2);
    }
}

int launch_add_layernorm(const void* x, const void* r, const float* gamma, const float* beta, void* y,
                         const float* pos, int L, void* y_pos, int M, int dtype, hipStream_t stream) {
    if (M <= 0) return 0;
#define ODAM_K(T, ...) hipLaunchKernelGGL(add_layernorm_kernel<T>, dim3((M + 3) / 4), dim3(256), 0, stream, (const T*)x, (const T*)r, gamma, beta, (T*)y, pos, L, (T*)y_pos, M)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void add_pos_kernel(const T* __restrict__ x, const float* __restrict__ pos,
                                                      int L, T* __restrict__ out, int M) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // float4 index
    if (i >= (size_t)M * 64) return;
    const int row = (int)(i >> 6), c4 = (int)(i & 63);
    float4 p = *reinterpret_cast<const float4*>(pos + (size_t)(row % L) * 256 + c4 * 4);
    if (x) {
        const float4 v = ld4(x + i * 4);
        p.x += v.x; p.y += v.y; p.z += v.z; p.w += v.w;
    }
    st4(out + i * 4, p);
}

int launch_add_pos(const void* x, const float* pos, int L, void* out, int M, int dtype, hipStream_t stream) {
    if (M <= 0) return 0;
    const size_t n4 = (size_t)M * 64;
#define ODAM_K(T, ...) hipLaunchKernelGGL(add_pos_kernel<T>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const T*)x, pos, L, (T*)out, M)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
// layout / pooling
// =================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                            int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // pixel index over B*H*W
    if (i >= total) return;
    const size_t b = i / HW, p = i - b * HW;
    const float* s = in + b * 3 * (size_t)HW + p;
    float4 v = {s[0], s[HW], s[2 * (size_t)HW], 0.0f};
    if constexpr (sizeof(T) == 2) {      // bf16: one 16-byte chunk = 8 channels per pixel (3 real + 5 zero)
        st4(out + i * 8, v);
        st4(out + i * 8 + 4, float4{0.f, 0.f, 0.f, 0.f});
    } else {
        st4(out + i * 4, v);
    }
}

// fp32 NCHW -> NHWC4 with a zero frame: out [B][H + 6][W + 8][4], the image at rows 3 .. H + 2, columns 3 .. W + 2.
// What the stem-as-rows convolution reads (detr_model.hip, stem_rows): 8 consecutive pixels = 32 floats per filter row.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc4_framed_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                                   int H, int W, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // pixel index over B*(H+6)*(W+8)
    if (i >= total) return;
    const int Wp = W + 8, Hp = H + 6;
    const int xp = (int)(i % Wp);
    const size_t r = i / Wp;
    const int yp = (int)(r % Hp);
    const size_t b = r / Hp;
    const int x = xp - 3, y = yp - 3;
    float4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
        const size_t HW = (size_t)H * W;
        const float* sp = in + b * 3 * HW + (size_t)y * W + x;
        v = float4{sp[0], sp[HW], sp[2 * HW], 0.0f};
    }
    st4(out + i * 4, v);
}

int launch_nchw_to_nhwc4_framed(const float* in, void* out, int B, int H, int W, int dtype, hipStream_t stream) {
    const size_t total = (size_t)B * (H + 6) * (W + 8);
    if (!total) return 0;
#define ODAM_K(T, ...) hipLaunchKernelGGL(nchw_to_nhwc4_framed_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, (T*)out, H, W, total)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}

int launch_nchw_to_nhwc4(const float* in, void* out, int B, int H, int W, int dtype, hipStream_t stream) {
    const size_t total = (size_t)B * H * W;
    if (!total) return 0;
#define ODAM_K(T, ...) hipLaunchKernelGGL(nchw_to_nhwc4_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, in, (T*)out, H * W, total)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ in, T* __restrict__ out, int H,
                                                      int W, int C4, int Ho, int Wo, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // over B*Ho*Wo*C4
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t p = i / C4;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const size_t b = p / Ho;
    float4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
        const int iy = oy * 2 - 1 + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            const int ix = ox * 2 - 1 + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const float4 v = ld4(in + (((b * H + iy) * W + ix) * (size_t)C4 + c4) * 4);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
    }
    st4(out + i * 4, m);
}

int launch_maxpool3x3s2(const void* in, void* out, int B, int H, int W, int C, int Ho, int Wo, int dtype,
                        hipStream_t stream) {
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    if (!total) return 0;
#define ODAM_K(T, ...) hipLaunchKernelGGL(maxpool_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const T*)in, (T*)out, H, W, C / 4, Ho, Wo, total)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}


template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out,
                                                           int HW, int C, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // output index over B*C*HW
    if (i >= total) return;
    const size_t p = i % HW;
    const size_t c = (i / HW) % C;
    const size_t b = i / ((size_t)HW * C);
    if constexpr (sizeof(T) == 2) out[i] = bf2f(in[(b * HW + p) * C + c]);
    else out[i] = in[(b * HW + p) * C + c];
}

int launch_nhwc_to_nchw(const void* in, float* out, int B, int H, int W, int C, int dtype, hipStream_t stream) {
    const size_t total = (size_t)B * H * W * C;
    if (!total) return 0;
#define ODAM_K(T, ...) hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const T*)in, out, H * W, C, total)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ __launch_bounds__(256) void to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) *reinterpret_cast<float4*>(out + i * 4) = ld4(in + i * 4);
}

int launch_to_f32(const void* in, float* out, size_t n, int dtype, hipStream_t stream) {
    const size_t n4 = n / 4;
    if (!n4) return 0;
#define ODAM_K(T, ...) hipLaunchKernelGGL(to_f32_kernel<T>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, (const T*)in, out, n4)
    ODAM_DISPATCH(dtype, ODAM_K, 0);
#undef ODAM_K
    ODAM_HIP(hipGetLastError());
    return 0;
}

__global__ void sigmoid_kernel(float* x, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = 1.0f / (1.0f + expf(-x[i]));
}

int launch_sigmoid(float* x, int n, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(sigmoid_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, x, n);
    ODAM_HIP(hipGetLastError());
    return 0;
}

// =================================================================================================
// DETR.postprocess arithmetic per query (detr.py:111-140); thresholding + greedy NMS stay on the host
// =================================================================================================
__global__ void postprocess_kernel(const float* __restrict__ logits, const float* __restrict__ boxes,
                                   const float* __restrict__ angle, const float* __restrict__ offset,
                                   const float* __restrict__ size, const float* __restrict__ depth, int n, int n_cls1,
                                   int n_bins, float img_w, float img_h, float fx, float fy, float cx, float cy,
                                   float* __restrict__ rows) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* lg = logits + (size_t)i * n_cls1;
    float mx = lg[0];
    for (int c = 1; c < n_cls1; c++) mx = fmaxf(mx, lg[c]);
    float den = 0.0f;
    for (int c = 0; c < n_cls1; c++) den += expf(lg[c] - mx);
    float best = -1.0f;
    int cls = 0;
    for (int c = 0; c < n_cls1 - 1; c++) {  // softmax(-1)[..., :-1].max(-1): first maximum wins
        const float p = expf(lg[c] - mx) / den;
        if (p > best) { best = p; cls = c; }
    }
    const float* bx = boxes + (size_t)i * 4;
    const float x0 = (bx[0] - 0.5f * bx[2]) * img_w, y0 = (bx[1] - 0.5f * bx[3]) * img_h;
    const float x1 = (bx[0] + 0.5f * bx[2]) * img_w, y1 = (bx[1] + 0.5f * bx[3]) * img_h;
    const float d = depth[i];
    const float scx = offset[(size_t)i * 2 + 0] * img_w + (x0 + x1) / 2.0f;
    const float scy = offset[(size_t)i * 2 + 1] * img_h + (y0 + y1) / 2.0f;
    const float c3x = ((scx - cx) / fx) * d;
    const float c3y = ((scy - cy) / fy) * d;
    const float* an = angle + (size_t)i * n_bins;
    int ab = 0;
    float am = an[0];
    for (int k = 1; k < n_bins; k++)
        if (an[k] > am) { am = an[k]; ab = k; }
    float* o = rows + (size_t)i * 16;
    o[0] = best; o[1] = (float)cls;
    o[2] = x0; o[3] = y0; o[4] = x1; o[5] = y1;
    o[6] = c3x; o[7] = c3y; o[8] = d;
    o[9] = (float)ab;
    o[10] = size[(size_t)i * 3 + 0]; o[11] = size[(size_t)i * 3 + 1]; o[12] = size[(size_t)i * 3 + 2];
    o[13] = 0.0f; o[14] = 0.0f; o[15] = 0.0f;
}

int launch_postprocess(const float* logits, const float* boxes, const float* angle, const float* offset,
                       const float* size, const float* depth, int B, int Q, int n_cls1, int n_bins, float img_w,
                       float img_h, float fx, float fy, float cx, float cy, float* rows, hipStream_t stream) {
    const int n = B * Q;
    if (n <= 0) return 0;
    hipLaunchKernelGGL(postprocess_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, logits, boxes, angle, offset,
                       size, depth, n, n_cls1, n_bins, img_w, img_h, fx, fy, cx, cy, rows);
    ODAM_HIP(hipGetLastError());
    return 0;
}

// ---- input transform: PIL bilinear resize of the uint8 frame + ToTensor + Normalize ------------
// Pillow's 8-bit resampling is two passes of integer arithmetic (src/libImaging/Resample.c): each pass rounds to a
// byte, clip8((2^21 + sum_k px[k] * K[k]) >> 22).  One thread owns one output pixel: it recomputes the (<= ksize)
// horizontally resampled bytes of the rows its vertical taps need, so no intermediate image exists.
struct ResampleTable {   // device pointers
    const int* xmin;     // [out] first source index
    const int* cnt;      // [out] taps
    const int* K;        // [out][ksize] fixed-point weights (22 fractional bits)
    int ksize;
};

__device__ __forceinline__ int clip8_22(int v) {
    v >>= 22;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void preprocess_u8_kernel(const unsigned char* __restrict__ rgb, int h, int w,
                                                            ResampleTable tx, ResampleTable ty, int need_x, int need_y,
                                                            float* __restrict__ out, int H, int W, float3 mean,
                                                            float3 stdv) {
    const int X = blockIdx.x * 256 + threadIdx.x;
    const int Y = blockIdx.y;
    const size_t b = blockIdx.z;
    if (X >= W) return;
    const unsigned char* img = rgb + b * (size_t)h * w * 3;
    const int x0 = need_x ? tx.xmin[X] : X, nx = need_x ? tx.cnt[X] : 1;
    const int y0 = need_y ? ty.xmin[Y] : Y, ny = need_y ? ty.cnt[Y] : 1;
    const int* kx = tx.K + (size_t)X * tx.ksize;
    const int* ky = ty.K + (size_t)Y * ty.ksize;
    int acc[3] = {1 << 21, 1 << 21, 1 << 21};
    int last[3] = {0, 0, 0};
    for (int j = 0; j < ny; j++) {
        const unsigned char* row = img + ((size_t)(y0 + j) * w + x0) * 3;
        int hv[3];
        if (need_x) {
            int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
            for (int k = 0; k < nx; k++) {
                const int kk = kx[k];
                a0 += row[k * 3 + 0] * kk; a1 += row[k * 3 + 1] * kk; a2 += row[k * 3 + 2] * kk;
            }
            hv[0] = clip8_22(a0); hv[1] = clip8_22(a1); hv[2] = clip8_22(a2);
        } else {
            hv[0] = row[0]; hv[1] = row[1]; hv[2] = row[2];
        }
        if (need_y) {
            const int kk = ky[j];
            acc[0] += hv[0] * kk; acc[1] += hv[1] * kk; acc[2] += hv[2] * kk;
        }
        last[0] = hv[0]; last[1] = hv[1]; last[2] = hv[2];
    }
    const int u0 = need_y ? clip8_22(acc[0]) : last[0];
    const int u1 = need_y ? clip8_22(acc[1]) : last[1];
    const int u2 = need_y ? clip8_22(acc[2]) : last[2];
    // ToTensor: float32(u8) / 255; Normalize: (x - mean) / std, float32 (transforms.py:222-243)
    const size_t plane = (size_t)H * W, o = (b * 3 * H + Y) * (size_t)W + X;
    out[o] = ((float)u0 / 255.0f - mean.x) / stdv.x;
    out[o + plane] = ((float)u1 / 255.0f - mean.y) / stdv.y;
    out[o + 2 * plane] = ((float)u2 / 255.0f - mean.z) / stdv.z;
}

int launch_preprocess_u8(const unsigned char* rgb, int B, int h, int w, const int* tx_xmin, const int* tx_cnt,
                         const int* tx_K, int tx_ksize, const int* ty_xmin, const int* ty_cnt, const int* ty_K,
                         int ty_ksize, float* out, int H, int W, const float mean[3], const float stdv[3],
                         hipStream_t stream) {
    if (B <= 0) return 0;
    ResampleTable tx{tx_xmin, tx_cnt, tx_K, tx_ksize}, ty{ty_xmin, ty_cnt, ty_K, ty_ksize};
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((W + 255) / 256, H, B), dim3(256), 0, stream, rgb, h, w, tx, ty,
                       (int)(W != w), (int)(H != h), out, H, W, float3{mean[0], mean[1], mean[2]},
                       float3{stdv[0], stdv[1], stdv[2]});
    ODAM_HIP(hipGetLastError());
    return 0;
}

}  // namespace odam_dk
