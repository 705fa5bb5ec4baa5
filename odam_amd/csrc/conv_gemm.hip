// conv_gemm.hip -- see conv_gemm.h
#include "conv_gemm.h"

#include "cg_common.hpp"
#include "cg_launch.h"

namespace odam_cg {

// NST: LDS stages.  2 = one tile in flight under the tile being multiplied (two workgroups per CU hide the rest).
// 4 (UT only; small, latency-bound problems that leave most CUs with one workgroup): three tiles in flight, retired
// with a counted s_waitcnt so the DMA queue stays full across the per-tile barrier.
template <int BM, int BN, int WM, int WN, bool BF16, bool UT, int NST = 2, bool X3 = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_kernel(ConvGemmArgs a) {
    static_assert(NST == 2 || (UT && (NST & (NST - 1)) == 0), "deep pipelines are LDS-DMA only");
    static_assert(!X3 || (UT && !BF16), "split products: fp32 operands on the LDS-DMA path");
    using T = typename std::conditional<BF16, unsigned short, float>::type;
    constexpr int EPC = BF16 ? 8 : 4;   // elements per 16-byte chunk
    constexpr int BKE = 8 * EPC;        // elements per k-tile (32 fp32 / 64 bf16)
    constexpr int NTH = 64 * WM * WN;   // threads per workgroup
    constexpr int RPP = NTH / 8;        // tile rows staged per pass (8 chunks per row)
    constexpr int TM = BM / WM / 32;    // 32x32 MFMA tiles per wave along M
    constexpr int TN = BN / WN / 32;
    constexpr int A_LD = BM / RPP;      // 16-byte loads per thread for the A tile
    constexpr int B_LD = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows divide over the loader threads");

    // LDS tile image.  Register-staged path: rows padded to 36 floats.  UT path: the tiles are written by LDS-DMA
    // (buffer_load ... lds: one wave-instruction fills 8 rows x 128 B, lane-linear), so rows are unpadded and the
    // 16-byte chunk c of row r sits at position c ^ ((r >> 1) & 7) -- the DMA's *source* address carries the
    // permutation, ds_read_b128 applies it again; every 16-lane read group then covers 16 distinct slots.
    constexpr int LDS_LD = UT ? 32 : LDS_STRIDE;
    __shared__ __align__(16) float smem[NST * (BM + BN) * LDS_LD];   // static: 32 / 48 / 64 KiB (UT)
    float* As = smem;                        // [NST][BM][LDS_LD]
    float* Bs = smem + NST * BM * LDS_LD;    // [NST][BN][LDS_LD]

    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(a.A);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(a.Wt);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): give every XCD a contiguous run
    // of tile ids, so the n-tiles that re-read one A row block (and neighbouring m-tiles that share filter
    // taps) meet in one L2.  Bijective for any grid size; placement only changes speed, never results.
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n;
    const int tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread gather bookkeeping (rows are fixed over the K loop) ---------------------
    const int c4 = tid & 7;    // which 16-byte chunk of the k slice
    const int r0 = tid >> 3;   // 0..RPP-1
    int a_base[A_LD], a_iy0[A_LD], a_ix0[A_LD];
    bool a_ok[A_LD];
    const int hw = a.Ho * a.Wo;
    const int lda = a.lda > 0 ? a.lda : a.Cin;   // elements between consecutive input pixels
    const bool plain = a.KH * a.KW == 1 && a.stride == 1 && a.pad == 0;
#pragma unroll
    for (int i = 0; i < A_LD; i++) {
        const int m = m0 + r0 + RPP * i;
        a_ok[i] = m < a.M;
        const int mm = a_ok[i] ? m : 0;
        if (plain) {   // 1x1, stride 1, no padding (a Linear layer, half of the detector's convolutions): pixel = row
            a_iy0[i] = 0; a_ix0[i] = 0;
            a_base[i] = mm * lda;
            continue;
        }
        const int b = mm / hw;
        const int rem = mm - b * hw;
        const int oy = rem / a.Wo;
        const int ox = rem - oy * a.Wo;
        a_iy0[i] = oy * a.stride - a.pad;
        a_ix0[i] = ox * a.stride - a.pad;
        a_base[i] = ((b * a.H + a_iy0[i]) * a.W + a_ix0[i]) * lda;
    }
    const T* w_ptr[B_LD];
    bool b_ok[B_LD];
#pragma unroll
    for (int i = 0; i < B_LD; i++) {
        const int n = n0 + r0 + RPP * i;
        b_ok[i] = n < a.Cout;
        w_ptr[i] = Wg + (size_t)(b_ok[i] ? n : 0) * a.Kpad + c4 * EPC;
    }
    const int ntaps = a.KH * a.KW;
    const int kw_inv = (65536 + a.KW - 1) / a.KW;  // tap / KW for tap < 64, KW <= 7
    const int nt_inv = (65536 + ntaps - 1) / ntaps; // kt / ntaps for kt < 2048, ntaps <= 32 (checked at launch)

    // UT bookkeeping: descriptors over the input (rebased to the first image this tile touches, minus the padding
    // margin so every row offset is non-negative) and the filters; per row a byte offset and a bit per filter tap.
    __amdgpu_buffer_rsrc_t a_rs, w_rs;
    u32x4 a_rs4, w_rs4;   // the same descriptors as four SGPR words, for the hand-written DMA
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    unsigned a_voff[A_LD], a_mask[A_LD], w_voff[B_LD];
    if constexpr (UT) {
        const int b0 = m0 / hw;
        const int csrc = c4 ^ ((r0 >> 1) & 7);   // RPP % 16 == 0: the permutation key is the same for every i
        const int margin = (a.pad * a.W + a.pad) * lda;
        a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(Ag + (size_t)b0 * a.H * a.W * lda - margin), 0, (int)OOB_VOFF, 0x00020000);
        w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wg, 0, (int)OOB_VOFF, 0x00020000);
        const unsigned long long pa = (unsigned long long)(Ag + (size_t)b0 * a.H * a.W * lda - margin), pw = (unsigned long long)Wg;
        a_rs4 = u32x4{(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
        w_rs4 = u32x4{(unsigned)pw, (unsigned)(pw >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
#pragma unroll
        for (int i = 0; i < A_LD; i++) {
            a_voff[i] = (unsigned)(a_base[i] - b0 * a.H * a.W * lda + margin + csrc * EPC) * (unsigned)sizeof(T);
            unsigned mk = plain ? 1u : 0u;
            for (int t = 0; t < (plain ? 0 : ntaps); t++) {
                const int ky = (t * kw_inv) >> 16, kx = t - ky * a.KW;
                if ((unsigned)(a_iy0[i] + ky * a.dil) < (unsigned)a.H && (unsigned)(a_ix0[i] + kx * a.dil) < (unsigned)a.W) mk |= 1u << t;
            }
            a_mask[i] = a_ok[i] ? mk : 0u;
        }
#pragma unroll
        for (int i = 0; i < B_LD; i++)
            w_voff[i] = b_ok[i] ? (unsigned)((n0 + r0 + RPP * i) * a.Kpad + csrc * EPC) * (unsigned)sizeof(T) : OOB_VOFF;
    }

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    float4 ra[A_LD], rb[B_LD];
    auto load_tile = [&](int kt, int buf) {
        if constexpr (UT) {
            const int kb = kt * BKE;
            int tap, cbase;
            if (a.k_order) {   // chunk-major: kt = chunk * ntaps + tap
                const int chunk = (kt * nt_inv) >> 16;
                tap = kt - chunk * ntaps;
                cbase = chunk * BKE;
            } else {
                tap = kb >> a.log2Cin;
                cbase = kb & (a.Cin - 1);
            }
            const int ky = (tap * kw_inv) >> 16;
            const int kx = tap - ky * a.KW;
            const int soff = ((ky * a.W + kx) * a.dil * lda + cbase) * (int)sizeof(T);
#pragma unroll
            for (int i = 0; i < A_LD; i++) {
                const unsigned vo = ((a_mask[i] >> tap) & 1u) ? a_voff[i] : OOB_VOFF;
                if constexpr (NST > 2)
                    lds_dma16(a_rs4, lds_base + (unsigned)((buf * BM + wave_u * 8 + RPP * i) * LDS_LD * 4), vo, soff);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (lds_ptr_t)(As + (buf * BM + wave_u * 8 + RPP * i) * LDS_LD),
                                                             16, (int)vo, soff, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_LD; i++) {
                if constexpr (NST > 2)
                    lds_dma16(w_rs4, lds_base + (unsigned)((NST * BM + buf * BN + wave_u * 8 + RPP * i) * LDS_LD * 4), w_voff[i],
                              kb * (int)sizeof(T));
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rs, (lds_ptr_t)(Bs + (buf * BN + wave_u * 8 + RPP * i) * LDS_LD),
                                                             16, (int)w_voff[i], kb * (int)sizeof(T), 0, 0);
            }
            return;
        }
        const int k0 = kt * BKE + c4 * EPC;
        int tap = k0 >> a.log2Cin;
        int ci = k0 & (a.Cin - 1);
        if (a.k_order) {
            const int chunk = (kt * nt_inv) >> 16;
            tap = kt - chunk * ntaps;
            ci = chunk * BKE + c4 * EPC;
        }
        const int ky = (tap * kw_inv) >> 16;
        const int kx = tap - ky * a.KW;
        const bool tap_ok = tap < ntaps;
        const int koff = (ky * a.W + kx) * a.dil * lda + ci;
#pragma unroll
        for (int i = 0; i < A_LD; i++) {
            const int iy = a_iy0[i] + ky * a.dil, ix = a_ix0[i] + kx * a.dil;
            const bool ok = a_ok[i] && tap_ok && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            ra[i] = ok ? *reinterpret_cast<const float4*>(Ag + (size_t)(a_base[i] + koff)) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < B_LD; i++)
            rb[i] = b_ok[i] ? *reinterpret_cast<const float4*>(w_ptr[i] + kt * BKE) : float4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_tile = [&](int buf) {
        if constexpr (UT) return;
#pragma unroll
        for (int i = 0; i < A_LD; i++)
            *reinterpret_cast<float4*>(&As[(buf * BM + r0 + RPP * i) * LDS_STRIDE + c4 * 4]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_LD; i++)
            *reinterpret_cast<float4*>(&Bs[(buf * BN + r0 + RPP * i) * LDS_STRIDE + c4 * 4]) = rb[i];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int nk = a.Kpad / BKE;

    // Operand fragments: lane (row = lane & 31, half = lane >> 5) reads the 16-byte chunk 2 g + half of its row for
    // k-group g = 0..3.  fp32: 4 values = the k slots of four v_mfma_f32_32x32x2_f32 steps (A and B use the same
    // (step, half) -> k map, so the k order inside a group is free).  bf16: 8 values = exactly the operand of one
    // v_mfma_f32_32x32x16_bf16 (lane half h holds k = 8 h + j).
    const int frag_row = lane & 31;
    const int frag_k = UT ? 0 : (lane >> 5) * 4;
    int frag_q[4];   // float offset of k-group q's chunk inside the row
#pragma unroll
    for (int q = 0; q < 4; q++) frag_q[q] = UT ? ((2 * q + (lane >> 5)) ^ ((frag_row >> 1) & 7)) * 4 : q * 8;

    // one k-tile of MFMAs out of LDS stage `cur`; `issue_next` (the next tile's global loads) runs right after the
    // first fragment reads, under their LDS latency; every k-group's fragments are fetched one group ahead
    auto mma_tile = [&](int cur, auto&& issue_next) {
        const float* Ab = &As[(cur * BM + wm * (BM / WM) + frag_row) * LDS_LD + frag_k];
        const float* Bb = &Bs[(cur * BN + wn * (BN / WN) + frag_row) * LDS_LD + frag_k];
        if constexpr (X3) {
            // the exact three-way bf16 split of both operands (cg_common.hpp cg_split8), six products per 16 k on v_mfma_f32_32x32x16_bf16:
            // lane (row, half) takes its 8 consecutive k of step s from chunks 4 s + 2 half, + 1 of the row
            const int sw = (frag_row >> 1) & 7, hf = lane >> 5;
            float4 a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                a0[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + (((2 * hf) ^ sw) << 2));
                a1[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + (((2 * hf + 1) ^ sw) << 2));
            }
#pragma unroll
            for (int j = 0; j < TN; j++) {
                b0[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + (((2 * hf) ^ sw) << 2));
                b1[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + (((2 * hf + 1) ^ sw) << 2));
            }
            issue_next();
#pragma unroll
            for (int st = 0; st < 2; st++) {
                uint4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; i++) cg_split8(a0[i], a1[i], ah[i], am[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; j++) cg_split8(b0[j], b1[j], bh[j], bm[j], bl[j]);
                if (st == 0) {
#pragma unroll
                    for (int i = 0; i < TM; i++) {
                        a0[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + (((4 + 2 * hf) ^ sw) << 2));
                        a1[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + (((4 + 2 * hf + 1) ^ sw) << 2));
                    }
#pragma unroll
                    for (int j = 0; j < TN; j++) {
                        b0[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + (((4 + 2 * hf) ^ sw) << 2));
                        b1[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + (((4 + 2 * hf + 1) ^ sw) << 2));
                    }
                }
#define ODAM_MMS(x, y) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[i][j], 0, 0, 0)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++) {      // smallest products first
                        ODAM_MMS(al[i], bh[j]); ODAM_MMS(am[i], bm[j]); ODAM_MMS(ah[i], bl[j]);
                        ODAM_MMS(am[i], bh[j]); ODAM_MMS(ah[i], bm[j]);
                        ODAM_MMS(ah[i], bh[j]);
                    }
#undef ODAM_MMS
            }
            return;
        }
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) af[0][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + frag_q[0]);
#pragma unroll
        for (int j = 0; j < TN; j++) bf[0][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + frag_q[0]);
        issue_next();
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int pb = q & 1;
            if (q + 1 < 4) {
#pragma unroll
                for (int i = 0; i < TM; i++)
                    af[pb ^ 1][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_LD + frag_q[(q + 1) & 3]);
#pragma unroll
                for (int j = 0; j < TN; j++)
                    bf[pb ^ 1][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_LD + frag_q[(q + 1) & 3]);
            }
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    if constexpr (BF16) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[pb][i]),
                                                                            __builtin_bit_cast(bf16x8, bf[pb][j]),
                                                                            acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pb][i].x, bf[pb][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pb][i].y, bf[pb][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pb][i].z, bf[pb][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pb][i].w, bf[pb][j].w, acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    if constexpr (NST == 2) {
        if (!(CG_ABLATE & 16)) load_tile(0, 0);   // 16: pretend the first tile is already there (cross-tile prefetch bound)
        store_tile(0);
        if constexpr (UT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            mma_tile(cur, [&] { if (!(CG_ABLATE & 1) && kt + 1 < nk) load_tile(kt + 1, cur ^ 1); });
            if (!(CG_ABLATE & 2) && kt + 1 < nk) store_tile(cur ^ 1);
            if constexpr (UT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA of tile kt + 1 has landed
            if (!(CG_ABLATE & 4)) __syncthreads();
        }
    } else {
        // Tiles kt .. kt + NST - 2 are in flight when tile kt is awaited: each thread's DMAs retire in order, so
        // "at most (NST - 2) tiles' worth of my loads outstanding" means my part of tile kt has landed; the barrier
        // extends that to every wave and also says stage (kt - 1) % NST has been read by all, so it can be refilled.
        constexpr int LPT = A_LD + B_LD;   // DMA instructions per thread per tile
#pragma unroll
        for (int s0 = 0; s0 < NST - 1; s0++)
            if (s0 < nk) load_tile(s0, s0);
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + NST - 1 <= nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            mma_tile(kt & (NST - 1), [&] { if (kt + NST - 1 < nk) load_tile(kt + NST - 1, (kt + NST - 1) & (NST - 1)); });
        }
        __syncthreads();   // the last stage is fully read before the epilogue reuses the tile buffers
    }

    static_assert(WM * WN * 32 * (BN / WN) <= NST * (BM + BN) * LDS_LD, "epilogue staging fits the tile buffers");
    if (epilogue2_ok<BM, BN, WM, WN, BF16>(a)) epilogue2<BM, BN, WM, WN, TM, TN, BF16>(a, acc, smem, m0, n0, wm, wn, wave, lane);      // (its own barrier: one more)
    else epilogue<BM, BN, WM, WN, TM, TN, BF16>(a, acc, smem, (int)sizeof(smem) / 4, m0, n0, wm, wn, wave, lane);
}

#define g_big_mode odam_cfg::get(odam_cfg::CG_RING)      // 0 off, 1 auto, 2 whenever eligible
void set_big_mode(int mode) { odam_cfg::set(odam_cfg::CG_RING, mode); }

void split3_filters(const float* w, int Cout, int Kpad, unsigned short* out) {
    const int nk = Kpad / 16;
    for (int n = 0; n < Cout; n++)
        for (int kt = 0; kt < nk; kt++) {
            unsigned short* o = out + ((size_t)n * nk + kt) * 48;
            for (int e = 0; e < 16; e++) {
                const float v = w[(size_t)n * Kpad + kt * 16 + e];
                unsigned u;
                memcpy(&u, &v, 4);
                float fh; const unsigned uh = u & 0xffff0000u; memcpy(&fh, &uh, 4);
                const float r1 = v - fh;
                unsigned u1; memcpy(&u1, &r1, 4);
                float fm; const unsigned um = u1 & 0xffff0000u; memcpy(&fm, &um, 4);
                const float r2 = r1 - fm;
                unsigned u2; memcpy(&u2, &r2, 4);
                o[e] = (unsigned short)(u >> 16); o[16 + e] = (unsigned short)(u1 >> 16); o[32 + e] = (unsigned short)(u2 >> 16);
            }
        }
}

// the bf16-native kernel takes a layer when its gather is uniform-tap at k-tile 32 and the problem is large enough
// to give every CU a 256-row tile with a K loop worth pipelining
// fp32 layers on the 256-row kernel: 0 = no (128x128 tiles on the fp32 matrix instruction), 2 = products on the bf16 matrix
// instruction through the exact three-way split (the default: same accuracy class as the fp32 instruction, every parity test
// unchanged, 1.26x the R50 forward).  (1, the fp32 instruction inside the ring schedule, was measured slower than the 128x128
// tiles and is gone.)
#define g_big_f32 odam_cfg::get(odam_cfg::CG_F32)
void set_f32_mode(int mode) { odam_cfg::set(odam_cfg::CG_F32, mode == 1 ? 0 : mode); }
int f32_mode() { return g_big_mode ? g_big_f32 : 0; }
// cg.pin: the choice a device-filling problem gets, whatever M is -- for callers whose row count depends on how a scene is sharded
// (the detector); a caller whose rows are the same on every rank (the replicated association network) opts out with no_pin
static inline bool pinned(const ConvGemmArgs& a) { return odam_cfg::get(odam_cfg::CG_PIN) != 0 && !a.no_pin; }
static bool big_eligible(const ConvGemmArgs& a, int bn) {
    const bool bf = a.dtype == ODAM_CG_BF16;
    if (!bf && !g_big_f32) return false;
    const int bke = bf ? 32 : 16, esz = bf ? 2 : 4;
    if (a.Cin % bke != 0 || a.KH * a.KW > 32 || a.Kpad % bke != 0) return false;
    if (a.k_order && a.Cin % (2 * bke) != 0) return false;
    const long lda = a.lda > 0 ? a.lda : a.Cin;
    const long span = (256L / ((long)a.Ho * a.Wo) + 2) * a.H * a.W * lda + (long)(a.pad * a.W + a.pad) * lda + a.Cin;
    if (span * esz >= 0x7fffffffL || (long)a.Cout * a.Kpad * esz >= 0x7fffffffL) return false;
    if (g_big_mode >= 2) return true;
    if (pinned(a)) return a.Cout >= bn;      // the choice a device-filling problem gets, whatever M is
    const long tiles = (long)((a.M + 255) / 256) * ((a.Cout + bn - 1) / bn);
    return tiles >= 192 && a.Cout >= bn;     // short-K expand layers too: with 16-byte stores its one block per CU streams faster than two 128x128 blocks
}

template <int BM, int BN, int WM, int WN, int NST_UT = 2>
static int launch_cfg(const ConvGemmArgs& a, hipStream_t stream) {
    const int tiles = ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
    const bool bf = a.dtype == ODAM_CG_BF16;
    // uniform-tap gather: whole k-tiles inside one tap, a 32-bit tap mask, and 31-bit byte offsets from the
    // first image a tile touches (a tile spans at most BM / (Ho*Wo) + 2 images)
    const long esz = bf ? 2 : 4, lda = a.lda > 0 ? a.lda : a.Cin;
    const long span = ((long)BM / ((long)a.Ho * a.Wo) + 2) * a.H * a.W * lda + (long)(a.pad * a.W + a.pad) * lda + a.Cin;
    const bool ut = odam_cfg::get(odam_cfg::CG_UT) != 0 && a.Cin % (bf ? 64 : 32) == 0 && a.KH * a.KW <= 32 && span * esz < 0x7fffffffL &&
                    (long)a.Cout * a.Kpad * esz < 0x7fffffffL;
    const dim3 g(tiles), b(64 * WM * WN);
    if (bf) {
        if (ut) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, true, true, NST_UT>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, true, false>), g, b, 0, stream, a);
    } else {
        // fp32 split mode reaches the layers too small for the ring kernel as well (cg.small_x3): the same six exact products on
        // v_mfma_f32_32x32x16_bf16, both operands split in registers.  Forward of 2 / 8 / 16 frames 4.89 -> 4.59 / 10.24 -> 9.66 /
        // 17.36 -> 16.84 ms (same box, A/B by config); a batch of 32 has few such layers left (30.45 -> 30.32)
        if (ut && g_big_f32 == 2 && odam_cfg::get(odam_cfg::CG_SMALL_X3) != 0)
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, true, NST_UT, true>), g, b, 0, stream, a);
        else if (ut) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, true, NST_UT>), g, b, 0, stream, a);
        else hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, false>), g, b, 0, stream, a);
    }
    ODAM_HIP(hipGetLastError());
    return 0;
}

// the fused kernel applies: fp32 split mode with pre-split filters on both layers, 64 -> 256 channels, a full device
bool fused_second_ok(const ConvGemmArgs& a) {
    const int fuse = odam_cfg::get(odam_cfg::CG_FUSE);
    if (!fuse || !g_big_mode || g_big_f32 != 2 || a.dtype != ODAM_CG_F32) return false;
    if (!a.F_Wt3 || !a.F_C || !a.Wt3 || a.Kpad % 16 != 0) return false;
    const bool l1 = a.Cout == 64 && a.F_ldc == 256, l2 = a.Cout == 128 && a.F_ldc == 512 && !a.G_Wt3;     // the two shapes built
    if ((!l1 && !l2) || a.dil != 1) return false;
    if ((long)a.Cout * a.Kpad * 6 >= 0x7fffffffL || (long)a.M * a.F_ldc * 4 >= 0x7fffffffL) return false;      // buffer descriptors over the output / residual
    if (a.G_Wt3 && (fuse < 2 || !a.G_C || (a.G_N != 64 && a.G_N != 128))) return false;      // fuse 1: second layer only
    return big_eligible(a, a.Cout);
}

// bf16: the 3x3 + expand (+ next reduce) kernel applies -- ring kernel eligible for the 3x3, the channel combination built,
// residual / output tensors inside 31-bit byte offsets
#define g_fuse_bf16 odam_cfg::get(odam_cfg::CG_FUSE_BF16)      // 0 off, 1 expand only, 2 with the chained reduce
bool fused_bf16_ok(const ConvGemmArgs& a) {
    if (!g_fuse_bf16 || !g_big_mode || a.dtype != ODAM_CG_BF16 || !a.F_Wt || !a.F_C) return false;
    if (a.Cout != 64 && a.Cout != 128 && a.Cout != 256) return false;
    if (a.F_ldc != 4 * a.Cout || a.KH * a.KW != 9 || a.Cin != a.Cout || !a.k_order || a.dil != 1) return false;
    if ((long)a.M * a.F_ldc * 2 >= 0x7fffffffL) return false;
    if (a.G_Wt) {
        if (g_fuse_bf16 < 2 || !a.G_C) return false;
        const bool built = (a.Cout == 64 && (a.G_N == 64 || a.G_N == 128)) || (a.Cout == 128 && a.G_N == 128);
        if (!built) return false;
    }
    return big_eligible(a, a.Cout);
}

// the stem with the max-pool on the tile: what launch_conv_gemm needs to run `a` (pool fields set) on the sixteen-wave 512 x 64
// tiles of the 16x16x32 split loop -- the kernel that has the pooling epilogue
bool pooled_stem_ok(const ConvGemmArgs& a) {
    const bool bf = a.dtype == ODAM_CG_BF16;
    if (!g_big_mode || !(odam_cfg::get(odam_cfg::CG_TILES) & 8) || odam_cfg::get(odam_cfg::STEM_POOL) == 0) return false;
    if (bf) {       // the sixteen-wave bf16 tiles take conv1 as a row convolution (pixel stride < row length), not a 3x3 stride 1
        if (!(a.lda > 0 && a.lda < a.Cin) && g_big_mode < 2) return false;
        if (a.KH * a.KW == 9 && a.stride == 1) return false;
        if (a.out_f32) return false;
    } else {
        if (g_big_f32 != 2 || !a.Wt3 || odam_cfg::get(odam_cfg::CG_PRESPLIT) == 0 || odam_cfg::get(odam_cfg::CG_MFMA16) == 0) return false;
        if (a.Kpad % 32 != 0 || (long)a.Cout * a.Kpad * 6 >= 0x7fffffffL) return false;
    }
    if (a.Cout != 64 || a.ldc != 64 || a.res || !a.relu) return false;
    if (a.pool_ph != POOL_PH || a.pool_pw != POOL_PW) return false;
    if (a.Hp != (a.Ho + 2 - 3) / 2 + 1 || a.Wp != (a.Wo + 2 - 3) / 2 + 1) return false;
    return big_eligible(a, 64) && ((a.M + 511) / 512 >= 256 || pinned(a));
}

int launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream) {
    const int epc = (a.dtype == ODAM_CG_BF16) ? 8 : 4;
    if (a.Kpad % (8 * epc) != 0 || a.Cin < epc || (a.Cin & (a.Cin - 1)) != 0 || a.KW > 7 || a.KH * a.KW > 64)
        return odam_fail(1, "conv_gemm: unsupported shape (Kpad % k-tile, Cin a power of two >= one 16-byte chunk, KW <= 7)");
    if (a.k_order && (a.Cin % (8 * epc) != 0 || a.KH * a.KW > 32 || a.Kpad / (8 * epc) >= 2048))
        return odam_fail(1, "conv_gemm: k_order 1 needs Cin % k-tile == 0, at most 32 taps, fewer than 2048 k-tiles");
    if (a.M <= 0 || a.Cout <= 0) return 0;
    if (a.pool) {
        if (!pooled_stem_ok(a)) return odam_fail(1, "conv_gemm: pooled stem asked for where it does not apply (check pooled_stem_ok)");
        return a.dtype == ODAM_CG_BF16 ? launch_big_bf16(64, 1024, a, stream) : launch_big_f32(4, 64, 1024, a, stream);
    }
    if (a.F_Wt3) {
        if (!fused_second_ok(a)) return odam_fail(1, "conv_gemm: fused second layer asked for where it does not apply (check fused_second_ok)");
        return launch_big_fused(a, stream);
    }
    if (a.F_Wt) {
        if (!fused_bf16_ok(a)) return odam_fail(1, "conv_gemm: fused bf16 bottleneck asked for where it does not apply (check fused_bf16_ok)");
        return launch_fused_bf16(a, stream);
    }
    if (g_big_mode) {
        const bool bf = a.dtype == ODAM_CG_BF16;
        const bool x3 = !bf && g_big_f32 == 2;
        const bool pre = x3 && a.Wt3 != nullptr && odam_cfg::get(odam_cfg::CG_PRESPLIT) != 0 && a.Kpad % 16 == 0 && (long)a.Cout * a.Kpad * 6 < 0x7fffffffL;
        const bool x16 = pre && odam_cfg::get(odam_cfg::CG_MFMA16) != 0 && (a.Cout & 3) == 0 && (a.ldc & 3) == 0 && a.Kpad % 32 == 0;   // 16x16x32 schedule
        if (a.Cout % 256 == 0 || a.Cout >= 384) {
            if (big_eligible(a, 256))
                // bf16 plain layers (no window loop): sixteen waves of 64 x 64 (four per SIMD; 128 registers, the epilogue spills):
                // R101 forward 17.19 -> 16.95 ms per 32 frames, the reduce layers of layer3 90 -> 87 us (same box, cg.tiles 15 vs 31)
                return bf ? ((odam_cfg::get(odam_cfg::CG_TILES) & 16) && !(a.KH * a.KW == 9 && a.stride == 1) ? launch_big_bf16(256, 1024, a, stream) : launch_big_bf16(256, 512, a, stream))
                          : (pre ? (x16 ? launch_big_f32(4, 256, 512, a, stream)
                                        : launch_big_f32(3, 256, 512, a, stream)) : launch_big_f32(2, 256, 512, a, stream));
#ifndef CG_NO_BN128_FALLBACK
            // too few 256-wide tiles to fill the device (the encoder's N = 256 layers at M = 27,200: 107) but enough 128-wide
            // ones: the ring kernel on 256 x 128 tiles instead of the 128 x 128 tiles of the fp32 matrix instruction
            if (pre && big_eligible(a, 128)) return x16 ? launch_big_f32(4, 128, 512, a, stream) : launch_big_f32(3, 128, 512, a, stream);
#endif
        } else if (a.Cout > 64 && (g_big_mode >= 2 || x3) && big_eligible(a, 128)) {   // bf16 / fp32: measured slower than the 128x128 tiles (tests only)
            return bf ? launch_big_bf16(128, 512, a, stream)
                      : (pre ? (x16 ? launch_big_f32(4, 128, 512, a, stream) : launch_big_f32(3, 128, 512, a, stream)) : launch_big_f32(2, 128, 512, a, stream));
        } else if (x3 && a.Cout > 32 && a.Cout <= 64 && big_eligible(a, 64)) {
            // Sixteen waves on 512 x 64 tiles (cg.tiles bit 3): the 64-column loop uses ~105 registers, so FOUR waves fit a SIMD, and a
            // wave of this loop spends more time issuing (5 DMAs of ~150 cycles, two 44-instruction splits, 28 LDS reads per 768
            // cycles of matrix work) than the pipe needs -- with four per SIMD the pipe finds a ready wave more often, and the
            // filter tile is fetched once per 512 rows.  conv1: 1.44 -> 1.15 ms per 32 frames (same box, A/B by config).
            // (128-row tiles with two 4-wave workgroups per CU = the same two waves per SIMD: measured no change, not kept.)
            if (x16 && (odam_cfg::get(odam_cfg::CG_TILES) & 8) && ((a.M + 511) / 512 >= 256 || pinned(a))) return launch_big_f32(4, 64, 1024, a, stream);      // (fewer 512-row tiles than CUs: the 256-row ones)
            return pre ? (x16 ? launch_big_f32(4, 64, 512, a, stream) : launch_big_f32(3, 64, 512, a, stream)) : launch_big_f32(2, 64, 512, a, stream);
        } else if (bf && (g_big_mode >= 2 || (a.lda > 0 && a.lda < a.Cin)) && a.Cout > 32 && a.Cout <= 64 && big_eligible(a, 64)) {
            if ((odam_cfg::get(odam_cfg::CG_TILES) & 8) && !(a.KH * a.KW == 9 && a.stride == 1) && ((a.M + 511) / 512 >= 256 || pinned(a))) return launch_big_bf16(64, 1024, a, stream);
            return launch_big_bf16(64, 512, a, stream);     // conv1 as a row convolution (pixel stride < row length); tests: layer1's 3x3 alone
        }
    }
    const int variant = odam_cfg::get(odam_cfg::CG_TILES);      // 8-wave tiles (+2-3 %)
    const int force = odam_cfg::get(odam_cfg::CG_FORCE);        // tests: pin one tile shape
    if (force == 1) return launch_cfg<128, 64, 4, 2>(a, stream);
    if (force == 2) return launch_cfg<64, 64, 2, 2, 4>(a, stream);
    if (force == 3) return launch_cfg<128, 128, 2, 2>(a, stream);
    if (a.Cout <= 64) return (variant & 2) ? launch_cfg<128, 64, 4, 2>(a, stream) : launch_cfg<128, 64, 4, 1>(a, stream);
    const long tiles128 = (long)((a.M + 127) / 128) * ((a.Cout + 127) / 128);
    if (tiles128 < 128) return (variant & 4) ? launch_cfg<64, 64, 2, 2, 4>(a, stream) : launch_cfg<64, 64, 2, 2>(a, stream);
    if (variant & 1) return launch_cfg<128, 128, 4, 2>(a, stream);
    return launch_cfg<128, 128, 2, 2>(a, stream);
}

}  // namespace odam_cg
