// cg_big.hpp -- the 256-row LDS-DMA ring kernel (conv_gemm_big_kernel) and its launcher template.  Instantiated by
// cg_big_f32.hip, cg_big_bf16.hip (plain layers) and cg_fused_f32.hip, cg_fused_bf16.hip (bottlenecks on the tile).
#pragma once
#include "cg_common.hpp"
#include "cg_tails_f32.hpp"
#include "cg_tail_bf16.hpp"
#include "cg_launch.h"

namespace odam_cg {

// ---- bf16-native schedule (BASELINE config 4) ---------------------------------------------------------------------
// The kernel above was laid out for the fp32 matrix instruction (64 cycles per 32x32x2 step): on v_mfma_f32_32x32x16_bf16
// the same tile spends 8x less time in the matrix pipe per k-tile while its LDS fragment reads (12 ds_read_b128 per 8
// MFMAs per wave), DMA issue and barrier stay -- it ends LDS-bound at ~0.14 of the bf16 peak.  This kernel is sized for
// the bf16 rate instead:
//   * 256 x BN output tile (BN = 256 or 128), eight waves, ONE workgroup per CU; a wave owns 128x64 (BN = 256) or 64x64
//     (BN = 128) of it, i.e. 0.75 / 1.0 fragment reads per MFMA instead of 1.5, and a quarter / half of the DMA
//     instructions per MFMA;
//   * k-tile of 32 bf16 (64-byte LDS rows) in an NST-deep ring (4 x 32 KiB for BN = 256): NST - 1 tiles are in flight
//     by LDS-DMA at any time, retired with a counted s_waitcnt -- vmcnt never drains inside the loop -- and one raw
//     s_barrier per k-tile (16 / 8 MFMAs per wave between barriers);
//   * 64-byte rows: chunk c of row r sits at c ^ ((r >> 2) & 3) -- the 16-lane groups of ds_read_b128
//     ({0-3,12-15,20-27}, ...) then touch rows whose (r >> 2) & 3 are all different inside each set of rows that share
//     bank range (r & 3), i.e. 16 distinct 16-byte slots: conflict-free; the DMA writes lane-linearly, so the
//     permutation is applied to its source address (lane l -> row l >> 2, chunk (l & 3) ^ ((l >> 4) & 3)).
// Gather, epilogue, XCD-aware tile order and numerics (fp32 accumulate, k-ordered inside a k-step) are the ones above.
// MODE 0: bf16 operands (32-element k-tiles on v_mfma_f32_32x32x16_bf16).
// MODE 1: fp32 operands on v_mfma_f32_32x32x2_f32 (16-element k-tiles: the same 64-byte rows, ring and schedule).
// MODE 2: fp32 operands, products on the bf16 matrix instruction through an EXACT three-way split: every fp32 value is
//   a = hi + mid + lo with hi, mid, lo bf16 (8 + 8 + 8 significant bits, by truncation: no rounding anywhere), and
//   a * b is taken as the six products hi hi + (hi mid + mid hi) + (hi lo + mid mid + lo hi); each bf16 x bf16 product is
//   exact in fp32 and the matrix instruction accumulates in fp32.  The three dropped products are below 2^-24 |a b|.
//   Measured against float64 (tests/native/x3_probe.hip, K = 64 ... 4608): max error 0.8-1.4e-7 of sum |a b|, the
//   fp32 matrix instruction's own k-ordered chain 1.1-1.9e-7 -- the same accuracy class, at 6 x 32 instead of 8 x 64
//   matrix-pipe cycles per 16 k.  Results differ from MODE 1 in the last bits (different summation tree), so this mode
//   is a switch (set_f32_mode), validated by the same tests as MODE 1.
// NTH = 512: one workgroup per CU (two waves per SIMD run the same tile in lock step); NTH = 1024: sixteen waves where a wave's
// loop fits ~128 registers (big_bm / big_wm).  (Four-wave workgroups, two per CU with independent barriers, were measured
// slower on every layer shape in round 2 and again in round 3, and are gone.)
// Tile rows and wave grid of the ring kernel.  Eight waves on 256 rows is the general shape; the 64-column loops need ~100
// registers, so sixteen waves on 512 x 64 put four waves on a SIMD (conv1: 1.44 -> 1.15 ms fp32, 0.95 -> 0.74 bf16).
// (The 256-column split loop with 32-row wave tiles was built both as twelve waves on 192 x 256 -- 140 registers, three per
// SIMD -- and as sixteen on 256 x 256 -- 126 registers, four per SIMD: the first loses to tile rounds what it gains (layer4's
// 3x3: 284 tiles = two rounds, 583 -> 822 us; expand layers +3-8 %), the second measured the same as eight waves (forward
// 29.76 vs 29.83 ms): that loop is paced by the matrix pipe and its barrier, not by waves waiting to issue.  Not kept.)
constexpr int big_bm(int BN, int NTH) { return (NTH == 1024 && BN == 64) ? 512 : 256; }
constexpr int big_wm(int MODE, int BN, int NTH) {
    return NTH == 1024 ? (BN == 64 ? 16 : 4)      // bf16, 256 columns: sixteen waves of 64 x 64
         : MODE >= 3 ? (BN == 256 ? ((MODE == 4 && CG_X16_WM8) ? 8 : 4) : 8) : (BN == 256 ? 2 : (BN == 128 ? 4 : 8));
}
// bf16, sixteen waves on 256 x 256 (the plain 1x1 layers of config 4: reduce / expand / downsample / the transformer's wide
// linears): activation tiles two PAIRS ahead in a ring of their own (see "deep A ring" in the main loop).  CG_DEEP_A = 0: off.
#ifndef CG_DEEP_A
#define CG_DEEP_A 1
#endif

constexpr bool big_deep_a(int MODE, int BN, int NST, int NTH) { return CG_DEEP_A != 0 && MODE == 0 && BN == 256 && NST == 4 && NTH == 1024; }
// dynamic LDS of the ring kernel: the ring, or the epilogue's 32-row slab per wave if that is larger
template <int MODE, int BN, int NST, int NTH>
constexpr int big_lds_bytes() {
    const int bm = big_bm(BN, NTH);
    const int ring = NST * (MODE >= 3 ? bm * 64 + 96 * BN : (bm + BN) * 64);
    const int wm = big_wm(MODE, BN, NTH);
    const int wn = NTH / 64 / wm;
    const int slab = MODE == 4 ? (NTH / 64) * 16 * ((BN / wn > 128 ? 128 : BN / wn) + 4) * 4 : (NTH / 64) * 32 * (BN / wn) * 4;      // the 16x16x32 loop's epilogue: 16-row slabs, rows padded by 4
    const int s1 = (MODE == 0 && NTH == 512) ? (BN == 256 ? 2 : (BN == 128 ? 3 : 4)) * (384 * 64 + 3 * BN * 64) : 0;   // the 3x3 stride-1 window ring (bf16)
    const int m = ring > slab ? ring : slab;
    const int deep = big_deep_a(MODE, BN, NST, NTH) ? (6 + 4) * 256 * 64 : 0;      // six activation + four filter slots (see the main loop)
    const int m2 = m > s1 ? m : s1;
    return m2 > deep ? m2 : deep;
}

template <int MODE, int BN, int NST, int NTH = 512, int FUSE = 0>     // FUSE: 1 = second layer on the tile (F_*), 2 = and a third (G_*)
__global__ __launch_bounds__(NTH, (NTH == 1024 ? 4 : 2)) void conv_gemm_big_kernel(ConvGemmArgs a) {
    static_assert(!FUSE || (FUSE <= 4 && (MODE == 3 || MODE == 4) && NTH == 512 && ((FUSE != 3 && BN == 64) || (FUSE == 3 && BN == 128))) ||
                      (FUSE >= 5 && FUSE <= 8 && MODE == 0 && NTH == 512 && NST == 4),
                  "fused layers: pre-split filters with 64-column tiles (FUSE 1, 2, 4) or 128-column tiles (FUSE 3); bf16 (FUSE 5 + "
                  "log2(PN / 32), 5 = no chained reduce)");
    static_assert(MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4,
                  "bf16 operands, fp32 split, fp32 split with pre-split filters (3: 32x32x16 matrix instruction, 4: 16x16x32)");
    static_assert(MODE != 4 || FUSE <= 4, "the 16x16x32 schedule: its own epilogue, or the fp32 bottleneck tails");
    constexpr bool BF16 = MODE == 0;
    // MODE 3 = MODE 2 with the FILTERS split ahead of time (a.Wt3: [Cout][Kpad / 16][3][16] bf16 = hi | mid | lo of every
    // 16-k group): they arrive in LDS as three bf16 planes and go to the matrix pipe as they are, only the activations are
    // split in registers; the wave grid is 4 x 2 (64 x 128 per wave), so a wave splits 2 row tiles instead of 4 + 2
    // -- a third of the vector work of MODE 2.
    constexpr bool PRE = MODE == 3 || MODE == 4;
    constexpr bool X16 = MODE == 4;          // pre-split filters, v_mfma_f32_16x16x32_bf16: see the main loop
    static_assert(!PRE || NTH == 512 || (X16 && NTH == 1024), "pre-split filters: eight-wave workgroups, or sixteen waves (16x16x32 loop)");
    static_assert(NTH != 1024 || (FUSE == 0 && ((BN == 64 && (X16 || BF16)) || (BN == 256 && BF16))), "sixteen waves: plain layers (512 x 64: bf16 or the 16x16x32 split loop; 256 x 256: bf16)");
    using T = typename std::conditional<BF16, unsigned short, float>::type;
    constexpr int EPC = BF16 ? 8 : 4;       // elements per 16-byte chunk
    constexpr int BM = big_bm(BN, NTH), BKE = 4 * EPC;
    constexpr int KU = 2 * BKE;             // channels per unit of the chunk-major K order (conv_gemm.h k_order)
    constexpr int NW = NTH / 64;
    constexpr int WM = big_wm(MODE, BN, NTH), WN = NW / WM;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int RPP = NTH / 4;            // 128 tile rows per DMA pass (4 chunks of 16 bytes per row)
    constexpr int A_LD = BM / RPP, B_LD = (BN + RPP - 1) / RPP;   // BN < RPP: only the first waves carry filter rows
    constexpr int ROWB = 64;                // bytes per LDS row
    constexpr int A_BYTES = BM * ROWB;
    constexpr int STAGE = PRE ? A_BYTES + 3 * BN * 32 : (BM + BN) * ROWB;   // bytes per ring stage (PRE: three 32-byte-row planes)
    constexpr int NB3 = 3 * BN / 32;        // PRE: 1 KiB DMA instructions per filter tile (32 rows x 32 bytes each)
    constexpr int B3_HI = (NB3 + NW - 1) / NW;   // ... per wave: B3_HI for waves < NB3 % NW (or all), B3_HI - 1 for the rest
    constexpr int LPT = A_LD + (PRE ? B3_HI : B_LD);   // DMA instructions per thread per tile (waves that carry the most)
    extern __shared__ __align__(16) float smem[];   // NST * STAGE bytes (dynamic: 96 / 128 KiB)
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const T* __restrict__ Ag = reinterpret_cast<const T*>(a.A);
    const T* __restrict__ Wg = reinterpret_cast<const T*>(a.Wt);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (a.Cout + BN - 1) / BN;
    int bid = blockIdx.x;
#if CG_STAMP
    const unsigned long long st_entry = __builtin_readcyclecounter();      // g_stamps[6]: entry -> main loop, [7]: main loop's end -> last store issued
#endif
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // stem with the max-pool on the tile (ConvGemmArgs::pool): tile = a 2-D patch of conv outputs, (b, first row, first column)
    constexpr bool POOLABLE = NTH == 1024 && BN == 64 && FUSE == 0 && (MODE == 4 || MODE == 0);
    constexpr int pl_tw = 2 * POOL_PW + 1, pl_rows = (2 * POOL_PH + 1) * pl_tw;      // patch: 17 x 29 conv outputs
    static_assert(pl_rows <= 512, "a patch fits a 512-row tile");
    int pl_b = 0, pl_y0 = 0, pl_x0 = 0;
    if constexpr (POOLABLE) {
        if (a.pool) {
            const int txs = (a.Wp + POOL_PW - 1) / POOL_PW, tys = (a.Hp + POOL_PH - 1) / POOL_PH;
            pl_b = bid / (tys * txs);
            const int rem = bid - pl_b * tys * txs, ty = rem / txs, tx = rem - ty * txs;
            pl_y0 = 2 * ty * POOL_PH - 1; pl_x0 = 2 * tx * POOL_PW - 1;
        }
    }
    // second half of the pooled epilogue: S = the patch after scale / bias / ReLU, [512 rows][BN + 4] fp32 in LDS (0 where the conv
    // output does not exist: behind a ReLU, zero padding pools like max_pool2d's -inf padding); every thread takes (pooled pixel, 4
    // channels) items: nine 16-byte reads, one store
    auto pool_store = [&](const float* S) {
        constexpr int SLD = BN + 4;
        const int py0 = (pl_y0 + 1) >> 1, px0 = (pl_x0 + 1) >> 1;
        constexpr int items = POOL_PH * POOL_PW * (BN / 4);
        for (int it = tid; it < items; it += NTH) {
            const int c4p = it & (BN / 4 - 1), pp = it / (BN / 4), ppy = pp / POOL_PW, ppx = pp - ppy * POOL_PW;
            const int py = py0 + ppy, px = px0 + ppx;
            if (py >= a.Hp || px >= a.Wp) continue;
            const float* w0 = S + ((2 * ppy) * pl_tw + 2 * ppx) * SLD + c4p * 4;
            float4 m4 = *reinterpret_cast<const float4*>(w0);
#pragma unroll
            for (int dy = 0; dy < 3; dy++)
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    if (dy == 0 && dx == 0) continue;
                    const float4 v = *reinterpret_cast<const float4*>(w0 + (dy * pl_tw + dx) * SLD);
                    m4.x = fmaxf(m4.x, v.x); m4.y = fmaxf(m4.y, v.y); m4.z = fmaxf(m4.z, v.z); m4.w = fmaxf(m4.w, v.w);
                }
            const size_t o = (((size_t)pl_b * a.Hp + py) * a.Wp + px) * a.ldc + c4p * 4;
            if constexpr (BF16) {
                const unsigned lo = pack_bf16x2(m4.x, m4.y), hi = pack_bf16x2(m4.z, m4.w);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.C) + o) = uint2{lo, hi};
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + o) = m4;
            }
        }
    };
    auto pool_row_ok = [&](int R) {      // does row R of the patch exist as a conv output?
        const int dy = R / pl_tw, oy = pl_y0 + dy, ox = pl_x0 + R - dy * pl_tw;
        return R < pl_rows && (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
    };

    // ---- gather bookkeeping (uniform-tap path of the kernel above) -------------------------------------------------
    const int c4 = tid & 3, r0 = tid >> 2;
    const int hw = a.Ho * a.Wo;
    const int lda = a.lda > 0 ? a.lda : a.Cin;
    const bool plain = a.KH * a.KW == 1 && a.stride == 1 && a.pad == 0;
    const int ntaps = a.KH * a.KW;
    const int kw_inv = (65536 + a.KW - 1) / a.KW;
    const int nt_inv = (65536 + ntaps - 1) / ntaps;
    const int b0 = (POOLABLE && a.pool) ? pl_b : m0 / hw;
    const int csrc = X16 ? (c4 ^ ((lane >> 4) & 1)) : (c4 ^ ((lane >> 4) & 3));     // 16-row fragments: rows r, r + 4 differ in the key (see X16)
    const int margin = (a.pad * a.W + a.pad) * lda;
    const unsigned long long pa = (unsigned long long)(Ag + (size_t)b0 * a.H * a.W * lda - margin), pw = (unsigned long long)Wg;
    const u32x4 a_rs4 = u32x4{(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const u32x4 w_rs4 = u32x4{(unsigned)pw, (unsigned)(pw >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    unsigned a_voff[A_LD], a_mask[A_LD], w_voff[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; i++) {
        const int m = m0 + r0 + RPP * i;
        bool ok = m < a.M;
        const int mm = ok ? m : 0;
        int base, iy0 = 0, ix0 = 0;
        if (POOLABLE && a.pool) {      // row r of the patch = conv output (y0 + r / tw, x0 + r % tw) of image pl_b, where that exists
            const int r = r0 + RPP * i, dy = r / pl_tw, oy = pl_y0 + dy, ox = pl_x0 + r - dy * pl_tw;
            ok = r < pl_rows && (unsigned)oy < (unsigned)a.Ho && (unsigned)ox < (unsigned)a.Wo;
            iy0 = (ok ? oy : 0) * a.stride - a.pad; ix0 = (ok ? ox : 0) * a.stride - a.pad;
            base = ((pl_b * a.H + iy0) * a.W + ix0) * lda;
        } else if (plain) {
            base = mm * lda;
        } else {
            const int b = mm / hw, rem = mm - b * hw, oy = rem / a.Wo, ox = rem - oy * a.Wo;
            iy0 = oy * a.stride - a.pad; ix0 = ox * a.stride - a.pad;
            base = ((b * a.H + iy0) * a.W + ix0) * lda;
        }
        a_voff[i] = (unsigned)(base - b0 * a.H * a.W * lda + margin + csrc * EPC) * (unsigned)sizeof(T);
        unsigned mk = plain ? 1u : 0u;
        for (int t = 0; t < (plain ? 0 : ntaps); t++) {
            const int ky = (t * kw_inv) >> 16, kx = t - ky * a.KW;
            if ((unsigned)(iy0 + ky * a.dil) < (unsigned)a.H && (unsigned)(ix0 + kx * a.dil) < (unsigned)a.W) mk |= 1u << t;
        }
        a_mask[i] = ok ? mk : 0u;
    }
#pragma unroll
    for (int i = 0; i < B_LD; i++) {
        const int n = n0 + r0 + RPP * i;
        w_voff[i] = (n < a.Cout && r0 + RPP * i < BN) ? (unsigned)(n * a.Kpad + csrc * EPC) * (unsigned)sizeof(T) : OOB_VOFF;
    }
    const unsigned lds_base = (unsigned)(size_t)(lds_ptr_t)smem;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool b_wave = PRE ? (NB3 % NW == 0 || wave_u < NB3 % NW)     // PRE: this wave issues B3_HI (else B3_HI - 1) filter DMAs
                            : (BN >= RPP || wave_u * 16 < BN);           // BN = 64 with eight waves: waves 0-3 stage the filter rows
    // PRE: DMA instruction q = wave + 8 j of a tile covers plane q / (BN / 32), rows 32 (q % (BN / 32)) .. + 31; lane l
    // writes row l >> 1, 16-byte half l & 1, and fetches half (l & 1) ^ ((row >> 3) & 1): with 32-byte rows the 16-lane
    // groups of ds_read_b128 then see 16 distinct slots
    unsigned w3_voff[B3_HI > 0 ? B3_HI : 1];
    u32x4 w3_rs4 = w_rs4;
    if constexpr (PRE) {
        const unsigned long long p3 = (unsigned long long)a.Wt3;
        w3_rs4 = u32x4{(unsigned)p3, (unsigned)(p3 >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
#pragma unroll
        for (int j = 0; j < B3_HI; j++) {
            const int q = wave_u + NW * j;
            const int pl = q / (BN / 32), rb = q - pl * (BN / 32);
            const int row = rb * 32 + (lane >> 1);
            const int n = n0 + row;
            const int cs = X16 ? (lane & 1) : ((lane & 1) ^ ((lane >> 4) & 1));
            w3_voff[j] = (q < NB3 && n < a.Cout) ? (unsigned)(n * (a.Kpad / 16) * 96 + pl * 32 + cs * 16) : OOB_VOFF;
        }
    }

    // One tile's DMAs are LPT instructions per thread; tile_prep computes what is wave-uniform about them, dma_part issues
    // the p-th (all of them right after the first fragment reads; spreading them between the matrix instructions was
    // measured at no gain in the split mode and -3 % in bf16, and is gone).
    struct TilePrep { int tap, soff_a, soff_b; };
    auto tile_prep = [&](int kt) {
        int tap, cbase;
        if (a.k_order) {   // chunk-major K in units of KU channels (conv_gemm.h): kt / 2 = chunk * ntaps + tap
            const int ku = kt >> 1, chunk = (ku * nt_inv) >> 16;
            tap = ku - chunk * ntaps;
            cbase = chunk * KU + (kt & 1) * BKE;
        } else {
            const int kb = kt * BKE;
            tap = kb >> a.log2Cin;
            cbase = kb & (a.Cin - 1);
        }
        const int ky = (tap * kw_inv) >> 16, kx = tap - ky * a.KW;
        return TilePrep{tap, ((ky * a.W + kx) * a.dil * lda + cbase) * (int)sizeof(T), PRE ? kt * 96 : kt * BKE * (int)sizeof(T)};
    };
    auto dma_part = [&](const TilePrep& tp, int st, int part) {      // part is a compile-time constant at every call site
        const unsigned sb = lds_base + (unsigned)(st * STAGE + wave_u * 16 * ROWB);
        if (part < A_LD) {
            const unsigned vo = ((a_mask[part] >> tp.tap) & 1u) ? a_voff[part] : OOB_VOFF;
            lds_dma16(a_rs4, sb + (unsigned)(RPP * part * ROWB), vo, tp.soff_a);
        } else if constexpr (PRE) {
            const int j = part - A_LD;
            if (j + 1 < B3_HI || b_wave)      // the last one only on the waves that carry B3_HI
                lds_dma16(w3_rs4, lds_base + (unsigned)(st * STAGE + A_BYTES + (wave_u + NW * j) * 1024), w3_voff[j < B3_HI ? j : 0], tp.soff_b);
        } else if (b_wave) {
            const int i = part - A_LD;
            lds_dma16(w_rs4, sb + (unsigned)((BM + RPP * i) * ROWB), w_voff[i < B_LD ? i : 0], tp.soff_b);
        }
    };
    auto load_tile = [&](int kt, int st) {
        const TilePrep tp = tile_prep(kt);
#pragma unroll
        for (int part = 0; part < LPT; part++) dma_part(tp, st, part);
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    // fragment addresses: lane (row = lane & 31, half = lane >> 5) reads chunk 2 s + half of its row for k-step s
    const int frow = lane & 31;
    const int fsw = (frow >> 2) & 3;
    const char* lds_c = reinterpret_cast<const char*>(smem);
    const int a_off = (wm * (BM / WM) + frow) * ROWB;
    const int b_off = (BM + wn * (BN / WN) + frow) * ROWB;
    int fq[2];
#pragma unroll
    for (int q = 0; q < 2; q++)     // MODE 2: the lane's 8 consecutive k (chunks 2 half, 2 half + 1); else chunk 2 q + half
        fq[q] = ((MODE >= 2 ? 2 * (lane >> 5) + q : 2 * q + (lane >> 5)) ^ fsw) * 16;

    // exact split of 8 fp32 values (two 16-byte chunks) into three bf16x8 operands: hi = top 16 bits of a, mid = top 16
    // bits of a - hi, lo = a - hi - mid (at most 8 significant bits left, so its top 16 bits hold it exactly)
    auto split8 = [](const float4& c0, const float4& c1, uint4& hi, uint4& mid, uint4& lo) {
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
    };

    // PRE: filter fragment of row R = wn (BN / WN) + 32 j + (lane & 31) in plane p: 16 bytes at p BN 32 + R 32 + half'
    const int b3_off = A_BYTES + (wn * (BN / WN) + frow) * 32 + (((lane >> 5) ^ ((frow >> 3) & 1)) * 16);

#if CG_STAMP
    unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = 0;
#endif
    // one k-tile (two k-steps of 16) of bf16 operands: Ab / Bb = this lane's fragment rows of the activation / filter slot
    auto mma_ab = [&](const char* Ab, const char* Bb, auto&& issue_next) {
        float4 af[2][TM], bf[2][TN];
#pragma unroll
        for (int i = 0; i < TM; i++) af[0][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * ROWB + fq[0]);
#pragma unroll
        for (int j = 0; j < TN; j++) bf[0][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * ROWB + fq[0]);
        int part = 0;
        issue_next(part++);
#pragma unroll
        for (; part < LPT; part++) issue_next(part);
#pragma unroll
        for (int i = 0; i < TM; i++) af[1][i] = *reinterpret_cast<const float4*>(Ab + i * 32 * ROWB + fq[1]);
#pragma unroll
        for (int j = 0; j < TN; j++) bf[1][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * ROWB + fq[1]);
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[q][i]),
                                                                        __builtin_bit_cast(bf16x8, bf[q][j]), acc[i][j], 0, 0, 0);
                }
            }
    };
    auto mma_tile = [&](int st, auto&& issue_next) {
        const char* Ab = lds_c + st * STAGE + a_off;
        const char* Bb = lds_c + st * STAGE + b_off;
        static_assert(!PRE || NST == 4, "pre-split filters run the register-pipelined loop below");
        if constexpr (MODE == 2) {
            uint4 bh[TN], bm[TN], bl[TN];
            float4 b0[TN], b1[TN], a0, a1;
#pragma unroll
            for (int j = 0; j < TN; j++) {
                b0[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * ROWB + fq[0]);
                b1[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * ROWB + fq[1]);
            }
            a0 = *reinterpret_cast<const float4*>(Ab + fq[0]);
            a1 = *reinterpret_cast<const float4*>(Ab + fq[1]);
            int part = 0;
            issue_next(part++);
#pragma unroll
            for (; part < LPT; part++) issue_next(part);
#pragma unroll
            for (int j = 0; j < TN; j++) split8(b0[j], b1[j], bh[j], bm[j], bl[j]);
#pragma unroll
            for (int i = 0; i < TM; i++) {
                uint4 ah, am, al;
                split8(a0, a1, ah, am, al);
                if (i + 1 < TM) {      // next row tile's fragments fly under this tile's products
                    a0 = *reinterpret_cast<const float4*>(Ab + (i + 1) * 32 * ROWB + fq[0]);
                    a1 = *reinterpret_cast<const float4*>(Ab + (i + 1) * 32 * ROWB + fq[1]);
                }
#define ODAM_MM(x, y) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[i][j], 0, 0, 0)
#pragma unroll
                for (int j = 0; j < TN; j++) {     // smallest products first
                    ODAM_MM(al, bh[j]); ODAM_MM(am, bm[j]); ODAM_MM(ah, bl[j]);
                    ODAM_MM(am, bh[j]); ODAM_MM(ah, bm[j]);
                    ODAM_MM(ah, bh[j]);
                }
#undef ODAM_MM
            }
            return;
        }
        mma_ab(Ab, Bb, issue_next);
    };

    const int nk = a.Kpad / BKE;
    if constexpr (X16) {
        // ---- pre-split filters on v_mfma_f32_16x16x32_bf16 -------------------------------------------------------------------
        // Same products as MODE 3, issued as 16x16x32 instructions: this loop is paced by the matrix pipe at the clock the chip
        // holds under load, and it holds a higher clock on this shape (tests/native/mfma_shape_probe.hip, the split loop out of
        // LDS on random data: 1.97 vs 1.72 GHz at equal cycles per product = 1.15-1.19x; MI355X_MICROARCH.md, DVFS give-back 7).
        // Measured (interleaved same-box A/B, tests/native/ab_config.py cg.mfma16 0 3): R50 forward 31.5 vs 32.5 ms (-3 to -4 %):
        // -2.7 % from the plain layers (layer3 / layer4, encoder, downsample / first-reduce layers), -1.3 % more with the fused
        // layer1 / layer2 launches on this loop (their tails take either accumulator layout through acc_cols64).  CG_STAMP: the
        // blocks run at 3,700 cycles per wave and 16 k against the pipe's 3,072 (MODE 3: 3,150) -- a 16x16x32 instruction leaves
        // 8 issue cycles where 32x32x16 leaves 24, and ten ~150-cycle DMA issues per pair sit in them -- so the clock gain
        // (1.15x in the bare loop) is only partly realised.  Default: all layers (odam_config cg.mfma16 = 3).
        // One instruction spans 32 k = TWO 16-k ring stages: lane (row l & 15, k-group l >> 4) takes its 8 consecutive k from
        // stage (l >> 5) of the pair, so the ring is two pairs -- the pair being multiplied and the pair in flight -- with one
        // barrier per pair.  A fragments: two 16-byte chunks 2 (kq & 1), + 1 of the lane's row, key (row >> 2) & 1 (the read
        // groups of ds_read_b128 then see rows 0-3 / 12-15 with chunk c and rows 4-11 with chunk c + 2 on 16 distinct slots);
        // filter planes: 16 bytes at half kq & 1 of the 32-byte row, no key needed.  Per pair and wave: the column tiles in
        // groups of four (their 12 fragment registers held), every row tile read and split once per group.
        constexpr int MI = BM / WM / 16, NJ = BN / WN / 16;
        static_assert(NJ % 4 == 0, "column tiles in groups of four");
        floatx4 c16[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NJ; j++) c16[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        const int r16 = lane & 15, kq = lane >> 4;
        const int akey = (r16 >> 2) & 1;
        const int a_lane = (kq >> 1) * STAGE + (wm * (BM / WM) + r16) * ROWB;
        const int a_c0 = ((2 * (kq & 1)) ^ akey) << 4, a_c1 = ((2 * (kq & 1) + 1) ^ akey) << 4;
        const int b_lane = (kq >> 1) * STAGE + A_BYTES + (wn * (BN / WN) + r16) * 32 + (kq & 1) * 16;
        auto dma_tile = [&](int kt, int st) {
            if (kt >= nk) return;
            const TilePrep tp = tile_prep(kt);
#pragma unroll
            for (int part = 0; part < A_LD; part++) {
                const unsigned vo = ((a_mask[part] >> tp.tap) & 1u) ? a_voff[part] : OOB_VOFF;
                lds_dma16(a_rs4, lds_base + (unsigned)(st * STAGE + wave_u * 16 * ROWB + RPP * part * ROWB), vo, tp.soff_a);
            }
#pragma unroll
            for (int j = 0; j < B3_HI; j++)
                if (j + 1 < B3_HI || b_wave)
                    lds_dma16(w3_rs4, lds_base + (unsigned)(st * STAGE + A_BYTES + (wave_u + NW * j) * 1024), w3_voff[j], tp.soff_b);
        };
        // One block = one row tile x ONE column tile = six matrix instructions.  A 16x16x32 instruction holds the SIMD's vector
        // issue for 8 of its 16 cycles (32x32x16: 8 of 32), so per product there is half the room for everything else -- and
        // the split of the activations is 44 vector instructions per fragment.  (First version: four column tiles per block,
        // every row fragment split once per group of four = 8 splits per pair: 3,729 cycles per wave and 16 k against a pipe
        // floor of 3,072, CG_STAMP; MODE 3 runs its blocks at 3,123.)  Here a row fragment is split ONCE per pair (4 splits)
        // and the filter fragments of the column tiles are re-read from LDS instead (3 ds_read_b128 per block, one block
        // ahead, two register sets): LDS issue replaces vector issue.  The next row tile's raw fragment is read in block 1 of
        // the current one and split in two halves in blocks 3 and 5; the next pair's DMAs are spread over the blocks.
        constexpr int NBLK = MI * NJ;
        constexpr int NDMA = 2 * (A_LD + B3_HI);             // DMA instructions per thread for a pair
        auto dma_part16 = [&](int kt, int st, int part) {    // the part-th DMA of tile kt (A rows first, then filter pieces)
            if (kt >= nk) return;
            const TilePrep tp = tile_prep(kt);
            if (part < A_LD) {
                const unsigned vo = ((a_mask[part] >> tp.tap) & 1u) ? a_voff[part] : OOB_VOFF;
                lds_dma16(a_rs4, lds_base + (unsigned)(st * STAGE + wave_u * 16 * ROWB + RPP * part * ROWB), vo, tp.soff_a);
            } else {
                const int j = part - A_LD;
                if (j + 1 < B3_HI || b_wave)
                    lds_dma16(w3_rs4, lds_base + (unsigned)(st * STAGE + A_BYTES + (wave_u + NW * j) * 1024), w3_voff[j < B3_HI ? j : 0], tp.soff_b);
            }
        };
        dma_tile(0, 0); dma_tile(1, 1);
#if CG_HACK & 3  // timing experiments only (tests/native/ring_stamps.hip -DCG_HACK=1 / 3): results are WRONG.  1: no epilogue at all (the main loop's own
                 // time); 3: no epilogue, but its memory traffic -- one 16-byte residual load and one 16-byte store per lane and fourth block, the
                 // tile's own rows -- issued INSIDE the main loop: what a tile costs if the previous tile's epilogue rode under its products
        const size_t hk_off = ((size_t)m0 * a.ldc + n0) * sizeof(float);
        const __amdgpu_buffer_rsrc_t hk_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.res ? a.res : a.C)) + hk_off, 0, a.res ? (int)OOB_VOFF : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t hk_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.C) + hk_off, 0, (int)OOB_VOFF, 0x00020000);
        u32x4 hk_l[8] = {};
        constexpr int HK_LPR = (BN / WN) / 4, HK_RPI = 64 / HK_LPR;
        const int hk_row0 = wm * (BM / WM) + lane / HK_LPR;
        const unsigned hk_col = (unsigned)(wn * (BN / WN) + (lane % HK_LPR) * 4) * 4u;
#endif
#define ODAM_MM16(accv, x, y) accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accv, 0, 0, 0)
#if CG_STAMP
        unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
        const unsigned long long st_loop0 = st_prev;
#endif
        for (int kp = 0; 2 * kp < nk; ++kp) {
            CG_STAMP_AT(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this pair has landed (the next one is issued below)
            CG_STAMP_AT(1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            CG_STAMP_AT(0);
            const int s0 = (2 * kp) & 3;
            const char* Sa = lds_c + s0 * STAGE + a_lane;
            const char* Sb = lds_c + s0 * STAGE + b_lane;
            struct Ops { uint4 h, m, l; };
            constexpr int BD = 2;      // filter fragments BD blocks ahead (one block = 96 pipe cycles: less than an LDS round trip under load; 3 and 4 measured the same)
            Ops A2[2], B2[BD + 1];
            float4 ar0, ar1;
            auto read_b = [&](int nj, Ops& o) {
                const char* bp = Sb + nj * 16 * 32;
                o.h = *reinterpret_cast<const uint4*>(bp);
                o.m = *reinterpret_cast<const uint4*>(bp + BN * 32);
                o.l = *reinterpret_cast<const uint4*>(bp + 2 * BN * 32);
            };
            auto read_a = [&](int mi) {
                ar0 = *reinterpret_cast<const float4*>(Sa + mi * 16 * ROWB + a_c0);
                ar1 = *reinterpret_cast<const float4*>(Sa + mi * 16 * ROWB + a_c1);
            };
            auto half_split = [](const float4& c, Ops& o, int hf) {      // the four values of one 16-byte chunk -> elements (2 hf, 2 hf + 1)
                const float v[4] = {c.x, c.y, c.z, c.w};
                unsigned uh[4], um[4], ul[4];
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    uh[e] = __builtin_bit_cast(unsigned, v[e]);
                    const float r1 = v[e] - __builtin_bit_cast(float, uh[e] & 0xffff0000u);
                    um[e] = __builtin_bit_cast(unsigned, r1);
                    ul[e] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, um[e] & 0xffff0000u));
                }
                auto pk = [](unsigned lo_, unsigned hi_) { return __builtin_amdgcn_perm(hi_, lo_, 0x07060302u); };
                if (hf == 0) {
                    o.h.x = pk(uh[0], uh[1]); o.h.y = pk(uh[2], uh[3]); o.m.x = pk(um[0], um[1]); o.m.y = pk(um[2], um[3]);
                    o.l.x = pk(ul[0], ul[1]); o.l.y = pk(ul[2], ul[3]);
                } else {
                    o.h.z = pk(uh[0], uh[1]); o.h.w = pk(uh[2], uh[3]); o.m.z = pk(um[0], um[1]); o.m.w = pk(um[2], um[3]);
                    o.l.z = pk(ul[0], ul[1]); o.l.w = pk(ul[2], ul[3]);
                }
            };
            // The filter fragments of a GROUP of four column tiles stay in registers while the wave's row tiles pass under them (column group
            // outer, row tile inner; every split row fragment is kept for the later groups): 2 MI + 3 NJ LDS fragment reads per pair and
            // wave instead of 2 MI + 3 MI NJ (the 256-column tiles: 32 instead of 104).  Round 4 re-read the filter fragments of every block
            // from LDS to save vector issue; the light stamps of round 5 say the loop is short of ISSUE slots altogether (two waves x (176 split
            // instructions + 104 LDS reads + 10 DMA issues) per pair against the 50 % of the cycles the matrix instruction leaves), and LDS
            // reads are the part that can go: same box, layer3 3x3 659 -> 616 us, reduce 355 -> 332, expand + residual 422 -> 405, forward of
            // 32 frames 31.5 -> 30.9 ms (profiles/r5_ring_breg.txt).  The six products of a block and their order on every accumulator are
            // unchanged: bit-identical outputs.  (CG_X16_BREG = 0 builds round 4's order: tests/native/ring_stamps.hip A/B.)
            // With the filter reads that cheap, the 256-column tile runs as eight waves of 32 x 256 (big_wm, CG_X16_WM8) instead of 4 x 2
            // waves of 64 x 128: no activation row is split by two waves (88 instead of 176 split instructions per pair and wave) and every
            // wave reads all sixteen column tiles once per pair (52 fragment reads instead of 32): 3x3 layers 604 -> 588 us, reduce 330 ->
            // 321, expand + residual 401 -> 392, forward of 32 frames 30.26 -> 29.92 ms, bit-identical (profiles/r5_ring_wm8.txt).
            constexpr bool BREG = CG_X16_BREG != 0;
            constexpr int NG = NJ / 4;
            Ops A4[BREG ? MI : 1], Bg[BREG ? 4 : 1];
            read_a(0);
            if constexpr (BREG) {
#pragma unroll
                for (int j = 0; j < 4; j++) read_b(j, Bg[j]);
                half_split(ar0, A4[0], 0); half_split(ar1, A4[0], 1);
            } else {
#pragma unroll
                for (int j = 0; j < BD; j++) read_b(j % NJ, B2[j]);
                half_split(ar0, A2[0], 0); half_split(ar1, A2[0], 1);
            }
            int dpart = 0;
#pragma unroll
            for (int b = 0; b < NBLK; b++) {
                const int grp = b / (4 * MI);      // (BREG) column group, row tile, column tile of block b
                const int mi = BREG ? (b / 4) % MI : b / NJ, nj = BREG ? 4 * grp + b % 4 : b % NJ;
                const Ops& oa = BREG ? A4[BREG ? mi : 0] : A2[mi & 1];
                const Ops& ob = BREG ? Bg[BREG ? b % 4 : 0] : B2[b % (BD + 1)];
                // smallest products first, as MODE 3
                ODAM_MM16(c16[mi][nj], oa.l, ob.h); ODAM_MM16(c16[mi][nj], oa.m, ob.m); ODAM_MM16(c16[mi][nj], oa.h, ob.l);
                ODAM_MM16(c16[mi][nj], oa.m, ob.h); ODAM_MM16(c16[mi][nj], oa.h, ob.m);
                ODAM_MM16(c16[mi][nj], oa.h, ob.h);
                if constexpr (BREG) {
                    const int j = b % 4;
                    if (grp == 0 && mi + 1 < MI) {      // first column group: the next row tile is read and split under this one's four blocks
                        if (j == 0) read_a(mi + 1);
                        if (j == 2) half_split(ar0, A4[BREG ? mi + 1 : 0], 0);
                        if (j == 3) half_split(ar1, A4[BREG ? mi + 1 : 0], 1);
                    }
                    if (mi == MI - 1 && grp + 1 < NG)   // a group's last row tile: column tile j's registers are free, the next group's tile j moves in
                        read_b(4 * (grp + 1) + j, Bg[BREG ? j : 0]);
                } else {
                if (b + BD < NBLK) read_b((b + BD) % NJ, B2[(b + BD) % (BD + 1)]);
                if (mi + 1 < MI) {
                    if (nj == 1) read_a(mi + 1);
                    if (nj == 3) half_split(ar0, A2[(mi + 1) & 1], 0);
                    if (nj == (NJ > 5 ? 5 : 3)) half_split(ar1, A2[(mi + 1) & 1], 1);
                }
                }
#pragma unroll
                for (; dpart < ((b + 1) * NDMA + NBLK - 1) / NBLK; dpart++) {
                    // the two stages of a pair are the two 64-byte halves of the same 128-byte lines (32 consecutive channels of
                    // a pixel): the same rows of both stages are requested back to back, so the second request meets the line
                    // in the CU's L1 (or its miss in flight) instead of fetching it from L2 again a stage later (layer3 /
                    // layer4 of the forward: 192.7 -> 202.2 and 211.3 -> 219.8 TFLOP/s, profiles/r3_detr_forward_B32_f32split_per_layer.txt).
                    // (A ring of THREE pairs for the 64-column tiles, whose pair is shorter than an L2 round trip, took the
                    // wait at the vmcnt from 1,349 to 106 cycles per 16 k and left the layer's time where it was: the wave
                    // is paced by issuing the DMAs, ~150 cycles each, not by their latency.  Gone.)
                    const int half_ = dpart < 2 * A_LD ? (dpart & 1) : (dpart - 2 * A_LD) / B3_HI;
                    const int part = dpart < 2 * A_LD ? (dpart >> 1) : A_LD + (dpart - 2 * A_LD) % B3_HI;
                    dma_part16(2 * kp + 2 + half_, (s0 ^ 2) + half_, part);
                }
#if CG_HACK & 2
                if (b % 4 == 0) {
                    const int step = kp * (NBLK / 4) + b / 4;      // 32 steps = the tile's 32 KB per wave, both ways
                    const int r = hk_row0 + (step >> 3) * 16 + (step & 7) * HK_RPI;
                    const unsigned vo = (step < (BM / WM / 16) * (16 / HK_RPI) && m0 + r < a.M) ? (unsigned)r * (unsigned)a.ldc * 4u + hk_col : OOB_VOFF;
                    asm volatile("" :: "v"(hk_l[(b / 4) % 8]));      // the load of the pair before is consumed here
                    __builtin_amdgcn_raw_buffer_store_b128(hk_l[(b / 4) % 8], hk_out, vo, 0, 0);
                    hk_l[(b / 4) % 8] = __builtin_amdgcn_raw_buffer_load_b128(hk_res, vo, 0, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);      // (without the fence: +2 % time; static priority for waves 4-7: no change)
                if (b == NBLK / 2 - 1) CG_STAMP_AT(2);
            }
        }
#if CG_STAMP
        CG_STAMP_AT(3);
        const unsigned long long st_loop1 = __builtin_readcyclecounter();
        auto stamps_out = [&](unsigned long long t_end) {      // at the very end of the kernel; one workgroup in 64 reports (2 M atomics on eight
            if (lane == 0 && (blockIdx.x & 63) == 0) {         // words would take longer than the layer: the round-5 stamps of every wave did)
                for (int i = 0; i < 4; i++) atomicAdd(&g_stamps[i], st_acc[i]);
                atomicAdd(&g_stamps[4], (unsigned long long)nk);
                atomicAdd(&g_stamps[5], 1ull);
                atomicAdd(&g_stamps[6], st_loop0 - st_entry);
                atomicAdd(&g_stamps[7], t_end - st_loop1);
            }
        };
        if constexpr (FUSE != 0) stamps_out(st_loop1);
#endif
#undef ODAM_MM16
#if CG_HACK & 3
        if constexpr (FUSE == 0) {
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) asm volatile("" :: "v"(c16[i][j]));
#if CG_HACK & 2
#pragma unroll
            for (int i = 0; i < 8; i++) asm volatile("" :: "v"(hk_l[i]));
#endif
            return;
        }
#endif
        if constexpr (FUSE != 0 || POOLABLE) __syncthreads();      // (the plain epilogue below has its own, behind its first residual requests)
        if constexpr (FUSE == 1) { fused_second(a, c16, smem, lds_base, m0, wave, lane); return; }
        if constexpr (FUSE == 2) { fused_chain(a, c16, smem, lds_base, m0, wave, lane); return; }
        if constexpr (FUSE == 3) { fused_second128(a, c16, smem, lds_base, m0, wave, lane); return; }
        if constexpr (FUSE == 4) { fused_chain128(a, c16, smem, lds_base, m0, wave, lane); return; }
        if constexpr (POOLABLE) {
            if (a.pool) {
                constexpr int SLD = BN + 4;
                static_assert(BM * SLD * 4 <= NST * STAGE, "the pooling slab fits the ring");
                float* S = smem;
                float scv[NJ], biv[NJ];
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const int n = j * 16 + r16;
                    scv[j] = a.scale ? a.scale[n] : 1.0f; biv[j] = a.bias ? a.bias[n] : 0.0f;
                }
#pragma unroll
                for (int mi = 0; mi < MI; mi++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int R = wm * (BM / WM) + mi * 16 + 4 * kq + r;
                        const bool ok = pool_row_ok(R);
#pragma unroll
                        for (int j = 0; j < NJ; j++) {
                            const float v = c16[mi][j][r] * scv[j] + biv[j];
                            S[R * SLD + j * 16 + r16] = (ok && v > 0.f) ? v : 0.f;
                        }
                    }
                __syncthreads();
                pool_store(S);
                return;
            }
        }
        // epilogue of the 16x16 accumulators (lane: column l & 15, rows 4 (l >> 4) + r): scale / bias (+ residual) (ReLU), fp32
        // out; a row tile of 16 rows goes through a per-wave LDS slab (rows padded by 4 floats: the four row groups of a
        // ds_write_b32 land on different banks) and leaves as 16-byte stores.
        // Round 5: NO BRANCHES.  Residual rows and output rows go through buffer descriptors over the tile (base = the tile's
        // first element: wave-uniform, per-lane byte offsets below 2^31 whatever the tensor's size); a lane whose row is past M
        // or whose channels are past Cout gets an out-of-range offset -- the load returns zeros, the store is dropped.  With
        // `if (m < M) load` every load and store sat in its own exec-masked block, and the compiler's wait insertion gives up
        // across those: it waited vmcnt(0) at the first use of a residual row -- i.e. for the row tile it had just requested as
        // well, and for every store issued so far (the .s of round 4: s_waitcnt vmcnt(0) at row tiles 0 and 2) -- so each wave
        // sat through two full memory round trips per tile with nothing in flight.  Now the waits are counted (the residual of
        // row tile i + 1 and the stores of row tile i - 1 stay in flight under row tile i), and the first two row tiles'
        // residual rows are requested BEFORE the barrier that ends the main loop.  Same arithmetic, same order: bit-identical.
        {
            // A wave's tile leaves in UNITS of 16 rows x UC columns (UC = its width, at most 128): unit u = row tile u / NH, column half u % NH.
            constexpr int WROWS = BM / WM, WCOLS = BN / WN, UC = WCOLS > 128 ? 128 : WCOLS, NH = WCOLS / UC, NU = MI * NH, EP_LD = UC + 4;
            static_assert(NW * 16 * EP_LD * 4 <= NST * STAGE, "epilogue slabs fit the ring");
            // (opaque copy of the thread id: nothing of the epilogue's lane arithmetic is computed ahead of the K loop and carried through it)
            int tid_e = tid;
            asm volatile("" : "+v"(tid_e));
            const int lane = tid_e & 63, wave = tid_e >> 6, wm = wave / WN, wn = wave % WN, r16 = lane & 15, kq = lane >> 4;
            float* Ep = smem + wave * (16 * EP_LD);
            constexpr int LPR = UC / 4, RPI = 64 / LPR, NIT = 16 / RPI;
            const int c4e = (lane % LPR) * 4;
            const int n = n0 + wn * WCOLS + c4e;
            bool n_ok[NH];
            float4 sc[NH], bi[NH];
#pragma unroll
            for (int h = 0; h < NH; h++) {
                n_ok[h] = n + h * UC < a.Cout;
                const int nc = n_ok[h] ? n + h * UC : 0;
                sc[h] = a.scale ? *reinterpret_cast<const float4*>(a.scale + nc) : float4{1.f, 1.f, 1.f, 1.f};
                bi[h] = a.bias ? *reinterpret_cast<const float4*>(a.bias + nc) : float4{0.f, 0.f, 0.f, 0.f};
                if (!n_ok[h]) { sc[h] = float4{1.f, 1.f, 1.f, 1.f}; bi[h] = float4{0.f, 0.f, 0.f, 0.f}; }
            }
            const size_t tile_off = ((size_t)m0 * a.ldc + n0) * sizeof(float);
            // (no residual: a descriptor of zero records -- every load through it returns zeros)
            const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(a.res ? a.res : a.C)) + tile_off, 0, a.res ? (int)OOB_VOFF : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.C) + tile_off, 0, (int)OOB_VOFF, 0x00020000);
            const int rows_left = a.M - m0;                       // rows of this tile that exist
            const int row0 = wm * WROWS + lane / LPR;             // + mi * 16 + it * RPI
            const unsigned col_b = (unsigned)(wn * WCOLS + c4e) * 4u;
            auto voff = [&](int u, int it) -> unsigned {
                const int r = row0 + (u / NH) * 16 + it * RPI;
                const unsigned ok = (unsigned)n_ok[u % NH] & (unsigned)(r < rows_left);      // (bitwise: a short-circuit && becomes a branch)
                return ok ? (unsigned)r * (unsigned)a.ldc * 4u + col_b + (unsigned)((u % NH) * UC * 4) : OOB_VOFF;
            };
            float4 rv[2][NIT];
            auto load_res = [&](int u, float4 (&dst)[NIT]) {
#pragma unroll
                for (int it = 0; it < NIT; it++)
                    dst[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, voff(u, it), 0, 0));
            };
            load_res(0, rv[0]);
            if (NU > 1) load_res(1, rv[1]);
            __syncthreads();      // every wave is past its last fragment read: the ring is free for the slabs
            const int relu_fb = relu_floor_bits(a.relu);
            auto act = [&](float v) { return relu_floor(v, relu_fb); };      // relu ? (v > 0 ? v : 0) : v, without a branch
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int mi = u / NH, h = u % NH;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < UC / 16; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Ep[(4 * kq + r) * EP_LD + j * 16 + r16] = c16[mi][h * (UC / 16) + j][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int row = it * RPI + lane / LPR;
                    const float4 v = *reinterpret_cast<const float4*>(&Ep[row * EP_LD + c4e]);
                    const float4 rr = rv[u & 1][it];
                    float4 o;
                    o.x = v.x * sc[h].x + bi[h].x; o.y = v.y * sc[h].y + bi[h].y; o.z = v.z * sc[h].z + bi[h].z; o.w = v.w * sc[h].w + bi[h].w;
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                    o.x = act(o.x); o.y = act(o.y); o.z = act(o.z); o.w = act(o.w);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rs, voff(u, it), 0, 0);
                }
                if (u + 2 < NU) load_res(u + 2, rv[u & 1]);      // into the registers this unit has just released
            }
        }
#if CG_STAMP
        __builtin_amdgcn_sched_barrier(0);
        stamps_out(__builtin_readcyclecounter());
#endif
        return;
    } else if constexpr (PRE) {
        // Pre-split filters, software-pipelined across the barrier.  Tile kt is consumed from REGISTERS: its fragments were
        // read from the ring during iteration kt - 1 -- the filter fragments of column tile j into the registers that
        // column's last matrix instruction has just released, the activation fragments into a second set, row tile 0
        // already split -- so after the barrier a wave goes straight to matrix instructions, and fragment latency,
        // splitting and DMA issue all sit between them (measured with CG_STAMP before this: of 4,330 cycles per k-tile a
        // wave spent 890 at the barrier and 850 on reads + DMA issue before its first matrix instruction).
        // Ring bookkeeping: at the barrier of iteration kt every wave's part of tile kt + 1 has landed (each waits with at
        // most one tile of its DMAs outstanding) and every read of tile kt - 1 was consumed by iteration kt - 1's matrix
        // instructions, so slot (kt + 3) % 4 = (kt - 1) % 4 is refilled with tile kt + 3.  Past the last tile the DMAs are
        // still issued, out of bounds (zeros into a dead slot): the outstanding count stays uniform.
        // The order of the matrix instructions on every accumulator is the one of the loop below: results are identical.
        struct Ops { uint4 h, m, l; };       // one row tile's activations as three bf16x8 operands
        Ops S[2];                            // row tile i of the tile in flight uses S[i & 1]; the next one is split into the other
        uint4 bh[TN], bm[TN], bl[TN];
        float4 ar0[TM], ar1[TM];
        static_assert(TM == 1 || TN >= 3, "the pipelined schedule places a row tile's split and re-read under three column tiles");
        constexpr int NC = TM * TN;          // chunks of six matrix instructions per k-tile
        auto read_a = [&](int s_, int i) {
            const char* Ab = lds_c + s_ * STAGE + a_off + i * 32 * ROWB;
            ar0[i] = *reinterpret_cast<const float4*>(Ab + fq[0]);
            ar1[i] = *reinterpret_cast<const float4*>(Ab + fq[1]);
        };
        auto read_b = [&](int s_, int j) {
            const char* B3 = lds_c + s_ * STAGE + b3_off + j * 32 * 32;
            bh[j] = *reinterpret_cast<const uint4*>(B3);
            bm[j] = *reinterpret_cast<const uint4*>(B3 + BN * 32);
            bl[j] = *reinterpret_cast<const uint4*>(B3 + 2 * BN * 32);
        };
        // half of split8: the four values of one 16-byte chunk -> elements (2 hf, 2 hf + 1) of the three operands
        auto half_split = [](const float4& c, Ops& o, int hf) {
            const float v[4] = {c.x, c.y, c.z, c.w};
            unsigned uh[4], um[4], ul[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                uh[e] = __builtin_bit_cast(unsigned, v[e]);
                const float r1 = v[e] - __builtin_bit_cast(float, uh[e] & 0xffff0000u);
                um[e] = __builtin_bit_cast(unsigned, r1);
                ul[e] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, um[e] & 0xffff0000u));
            }
            auto pk = [](unsigned lo_, unsigned hi_) { return __builtin_amdgcn_perm(hi_, lo_, 0x07060302u); };
            if (hf == 0) {
                o.h.x = pk(uh[0], uh[1]); o.h.y = pk(uh[2], uh[3]); o.m.x = pk(um[0], um[1]); o.m.y = pk(um[2], um[3]);
                o.l.x = pk(ul[0], ul[1]); o.l.y = pk(ul[2], ul[3]);
            } else {
                o.h.z = pk(uh[0], uh[1]); o.h.w = pk(uh[2], uh[3]); o.m.z = pk(um[0], um[1]); o.m.w = pk(um[2], um[3]);
                o.l.z = pk(ul[0], ul[1]); o.l.w = pk(ul[2], ul[3]);
            }
        };
        auto dma_ok = [&](const TilePrep& tp, int s_, int part, bool ok) {   // dma_part, out of bounds when !ok
            if (part < A_LD) {
                const unsigned vo = (ok && ((a_mask[part] >> tp.tap) & 1u)) ? a_voff[part] : OOB_VOFF;
                lds_dma16(a_rs4, lds_base + (unsigned)(s_ * STAGE + wave_u * 16 * ROWB + RPP * part * ROWB), vo, tp.soff_a);
            } else {
                const int j = part - A_LD;
                if (j + 1 < B3_HI || b_wave)
                    lds_dma16(w3_rs4, lds_base + (unsigned)(s_ * STAGE + A_BYTES + (wave_u + NW * j) * 1024),
                              ok ? w3_voff[j < B3_HI ? j : 0] : OOB_VOFF, tp.soff_b);
            }
        };
        constexpr int LPT_LO = LPT - 1;      // DMAs per tile of the waves that carry one filter piece less
#pragma unroll
        for (int s0 = 0; s0 < 3; s0++) {
            const TilePrep tp = tile_prep(s0 < nk ? s0 : 0);
#pragma unroll
            for (int part = 0; part < LPT; part++) dma_ok(tp, s0, part, s0 < nk);
        }
        if (b_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT_LO) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < TM; i++) read_a(0, i);
#pragma unroll
        for (int j = 0; j < TN; j++) read_b(0, j);
        half_split(ar0[0], S[0], 0); half_split(ar1[0], S[0], 1);
#if CG_STAMP
        st_prev = __builtin_readcyclecounter();
#endif
#define ODAM_MM(x, y) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc[i][j], 0, 0, 0)
        // One k-tile out of the registers, in NC chunks of six matrix instructions (row tile i x column tile j) fenced
        // against reordering; what rides under each chunk:
        //   chunks (i, 0), (i, 1)     the two halves of row tile i + 1's split           (i + 1 < TM)
        //   chunk  (i, 2)             NEXT: row tile i + 1's raw fragment of tile kt + 1   (its registers were just split)
        //   chunk  (0, 0)             NEXT: row tile 0's raw fragment of tile kt + 1
        //   chunks (TM - 1, j)        NEXT: column tile j's filter fragments of tile kt + 1 (their last use is this chunk)
        //   the last two chunks       NEXT: the halves of tile kt + 1's row tile 0 split
        //   every chunk               NEXT: its share of the LPT DMA instructions of tile kt + 3
        auto body = [&](auto next_c, int sn, const TilePrep& tp, int sf, bool ok) {
            constexpr bool NEXT = decltype(next_c)::value;
            int part = 0;
#pragma unroll
            for (int i = 0; i < TM; i++) {
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int c = i * TN + j;
                    const Ops& o = S[i & 1];
                    ODAM_MM(o.l, bh[j]); ODAM_MM(o.m, bm[j]); ODAM_MM(o.h, bl[j]);
                    ODAM_MM(o.m, bh[j]); ODAM_MM(o.h, bm[j]);
                    ODAM_MM(o.h, bh[j]);
                    if (i + 1 < TM) {
                        if (j == 0) half_split(ar0[i + 1], S[(i + 1) & 1], 0);
                        if (j == 1) half_split(ar1[i + 1], S[(i + 1) & 1], 1);
                    }
                    if constexpr (NEXT) {
                        if (c == 0) read_a(sn, 0);
                        if (i + 1 < TM && j == 2) read_a(sn, i + 1);
                        if (i == TM - 1) read_b(sn, j);
#pragma unroll
                        for (; part < ((c + 1) * LPT + NC - 1) / NC; part++) dma_ok(tp, sf, part, ok);
                        if (NC >= 4) {
                            if (c == NC - 2) half_split(ar0[0], S[TM & 1], 0);
                            if (c == NC - 1) half_split(ar1[0], S[TM & 1], 1);
                        } else if (c == NC - 1) {
                            half_split(ar0[0], S[TM & 1], 0); half_split(ar1[0], S[TM & 1], 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (c == TN - 1) CG_STAMP_AT(2);
                }
            }
            if constexpr (NEXT && (TM & 1)) S[0] = S[1];
        };
        int sn = 1, sf = 3;
        for (int kt = 0; kt + 1 < nk; ++kt) {
            CG_STAMP_AT(3);
            if (b_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");     // all but my newest tile (kt + 2): tile kt + 1 is in
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT_LO) : "memory");
            CG_STAMP_AT(1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            CG_STAMP_AT(0);
            const bool ok = kt + 3 < nk;
            const TilePrep tp = tile_prep(ok ? kt + 3 : 0);
            body(std::true_type{}, sn, tp, sf, ok);
            sn = (sn + 1) & 3; sf = (sf + 1) & 3;
        }
        CG_STAMP_AT(3);
        body(std::false_type{}, 0, TilePrep{0, 0, 0}, 0, false);
        // the out-of-bounds DMAs issued past the last tile still write (zeros) into the ring: they must have landed
        // before the epilogue stages its output there (the compiler does not count DMAs issued from inline assembly)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef ODAM_MM
    } else {
    bool s1_done = false;
    if constexpr (BF16) {
        // ---- 3x3, stride 1, pad 1 in bf16: one LDS window serves the three horizontal taps ------------------------------------
        // The bf16 ring is bound by LDS-DMA issue, not by the matrix pipe (layer3's 3x3: 71 us per 256 x 256 tile for 2.3 MB of
        // DMA, 37 us of matrix instructions).  With stride 1 and H x W preserved, tap (ky, kx) of output pixel m is input pixel
        // m + (ky - 1) W + (kx - 1) of the flattened [B H W] pixel array -- a pure shift; only validity (image borders) depends on
        // the pixel.  So a stage holds, for one 32-channel slice and one ky, the 258 consecutive input pixels m0 - 1 + (ky - 1) W
        // ... (rows 0 .. 257 of a 384-row region: three uniform DMA passes, rows past 263 are requested out of range = no
        // traffic) and the THREE filter k-tiles kx = 0, 1, 2: the A fragments of tap kx are the same rows read at row offset kx
        // (the 64-byte-row swizzle stays conflict-free under a shift of 1 or 2 rows), zeroed per lane where the tap falls outside
        // the image.  A-side DMA bytes and instructions drop 3x.  Stages per ring: 4 / 3 / 2 for BN = 64 / 128 / 256 (144 KB).
        // The k order differs from the generic loop's (ky outside the two 32-channel halves of a chunk): same products, a
        // different but fixed summation order.
        const long tensor_bytes = (long)a.B * a.H * a.W * lda * 2;
        const bool s1 = NTH == 512 && a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.dil == 1 && a.k_order == 1 && a.Ho == a.H && a.Wo == a.W &&
                        tensor_bytes < 0x7fffffffL && a.s1_window != 0;
        if constexpr (NTH == 512) if (s1) {
            constexpr int NSS = BN == 256 ? 2 : (BN == 128 ? 3 : 4);
            constexpr int AW_BYTES = 384 * ROWB;                  // window region: 3 passes of 128 rows
            constexpr int STG = AW_BYTES + 3 * BN * ROWB;
            static_assert(NSS * STG <= 160 * 1024, "S1 ring fits");
            constexpr int LPT_S = 3 + 3 * B_LD, LPT_S_LO = 3;      // DMA instructions per thread and stage (waves without filter rows: 3)
            const int n_stage = (a.Cin / 32) * 3;                  // (chunk, ky, half) triples
            const int npix = a.B * a.H * a.W;
            // descriptor over the input, rebased one row + one pixel before the tensor so that the wave-uniform window offset is >= 0
            const unsigned long long pw1 = (unsigned long long)(Ag - (size_t)(a.W + 1) * lda);
            const u32x4 aw_rs = u32x4{(unsigned)pw1, (unsigned)(pw1 >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
            int prow[3];                                           // window row of this lane in pass p, as a pixel index for ky = 0
            unsigned aw_voff[3];
#pragma unroll
            for (int p_ = 0; p_ < 3; p_++) {
                const int wr = RPP * p_ + r0;                      // window row 0 .. 383
                prow[p_] = wr < 264 ? m0 - 1 - a.W + wr : -0x40000000;
                aw_voff[p_] = (unsigned)(wr * lda * 2 + csrc * 16);
            }
            // validity of the nine taps for the rows this lane holds as A fragments
            unsigned tmask[TM];
#pragma unroll
            for (int i = 0; i < TM; i++) {
                const int m = m0 + wm * (BM / WM) + i * 32 + frow;
                unsigned mk = 0u;
                if (m < a.M) {
                    const int rem = m % hw, y = rem / a.W, x = rem - y * a.W;
#pragma unroll
                    for (int t = 0; t < 9; t++) {
                        const int ky = t / 3, kx = t - 3 * ky;
                        if ((unsigned)(y + ky - 1) < (unsigned)a.H && (unsigned)(x + kx - 1) < (unsigned)a.W) mk |= 1u << t;
                    }
                }
                tmask[i] = mk;
            }
            auto stage_of = [&](int s_, int& chunk, int& ky, int& hf) { chunk = s_ / 6; const int r_ = s_ - 6 * chunk; ky = r_ >> 1; hf = r_ & 1; };
            auto dma_stage = [&](int s_, int slot) {
                int chunk, ky, hf;
                stage_of(s_, chunk, ky, hf);
                const int cbase = chunk * 64 + hf * 32;
                const unsigned sbase = lds_base + (unsigned)(slot * STG);
                const int soff_a = ((m0 + ky * a.W) * lda + cbase) * 2;
#pragma unroll
                for (int p_ = 0; p_ < 3; p_++) {
                    const int pix = prow[p_] + ky * a.W;
                    const unsigned vo = (pix >= 0 && pix < npix) ? aw_voff[p_] : OOB_VOFF;
                    lds_dma16(aw_rs, sbase + (unsigned)((wave_u * 16 + RPP * p_) * ROWB), vo, soff_a);
                }
                if (b_wave) {
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const int kt = ((chunk * 9 + ky * 3 + kx) << 1) + hf;
#pragma unroll
                        for (int i = 0; i < B_LD; i++)
                            lds_dma16(w_rs4, sbase + (unsigned)(AW_BYTES + (kx * BN + wave_u * 16 + RPP * i) * ROWB), w_voff[i], kt * BKE * 2);
                    }
                }
            };
#pragma unroll
            for (int s0 = 0; s0 < NSS - 1; s0++)
                if (s0 < n_stage) dma_stage(s0, s0);
            int slot = 0, slot_next = NSS - 1;
            for (int s_ = 0; s_ < n_stage; ++s_) {
                if (s_ + NSS - 1 <= n_stage) {
                    if (b_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSS - 2) * LPT_S) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSS - 2) * LPT_S_LO) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (s_ + NSS - 1 < n_stage) dma_stage(s_ + NSS - 1, slot_next);
                int chunk, ky, hf;
                stage_of(s_, chunk, ky, hf);
                const char* Sb = lds_c + slot * STG;
#pragma unroll
                for (int kx = 0; kx < 3; kx++) {
                    const int tap = ky * 3 + kx;
                    const int sw_k = (((frow + kx) >> 2) & 3);
                    float4 af[2][TM], bf[2][TN];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
#pragma unroll
                        for (int i = 0; i < TM; i++) {
                            float4 v = *reinterpret_cast<const float4*>(Sb + (wm * (BM / WM) + i * 32 + frow + kx) * ROWB + (((2 * q + (lane >> 5)) ^ sw_k) << 4));
                            const unsigned ok = ((tmask[i] >> tap) & 1u) ? 0xffffffffu : 0u;
                            v.x = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.x) & ok);
                            v.y = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.y) & ok);
                            v.z = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.z) & ok);
                            v.w = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v.w) & ok);
                            af[q][i] = v;
                        }
#pragma unroll
                        for (int j = 0; j < TN; j++)
                            bf[q][j] = *reinterpret_cast<const float4*>(Sb + AW_BYTES + (kx * BN + wn * (BN / WN) + j * 32 + frow) * ROWB + fq[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 2; q++)
#pragma unroll
                        for (int i = 0; i < TM; i++)
#pragma unroll
                            for (int j = 0; j < TN; j++)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[q][i]),
                                                                                    __builtin_bit_cast(bf16x8, bf[q][j]), acc[i][j], 0, 0, 0);
                }
                slot = slot + 1 == NSS ? 0 : slot + 1;
                slot_next = slot_next + 1 == NSS ? 0 : slot_next + 1;
            }
            s1_done = true;
        }
    }
    if constexpr (big_deep_a(MODE, BN, NST, NTH)) if (!s1_done) {
        // Deep A ring (round 6).  The pair loop below keeps ONE pair of k-tiles in flight behind the pair being multiplied: 32 KB of
        // activations per CU, requested 0.85 us of matrix work ahead of their use -- less than a loaded HBM read takes, so a 1x1 layer
        // that streams its activations once (layer3's reduce: 512 KB per tile) ran at the LATENCY of a pair, 44 us per tile for 13.6 us
        // of products, 3.4 TB/s.  The filters come from L2 and need no such distance.  So the two operands get rings of their own:
        // six activation slots (three pairs) and four filter slots (two pairs) = 160 KB; iteration kp requests the filters of pair
        // kp + 1 and then the activations of pair kp + 2.  The wait at its top is for filters (kp) -- requested in iteration kp - 1 and
        // followed there by exactly the two activation pieces of pair kp + 1, which stay in flight (vmcnt(2)); everything older has
        // landed with them.  A pair's second tile past the end of K is requested out of range (no traffic, the count stays); pairs past
        // the end are not requested at all and the wait is then for everything, so nothing is in flight when the epilogue takes the LDS.
        // Same products in the same order: bit-identical.  MEASURED (profiles/r6_experiments.txt 18): R101 bf16 forward -0.4 % at 32 frames,
        // -0.8 % at 76 -- the distance was NOT what held these layers at 3.4 TB/s; nor is it the lockstep of the CUs on the same filter
        // lines (every workgroup starting K at a different pair: -1.5 % / +0.7 %).  What is left is the loop's own ratio: sixteen waves of
        // 64 x 64 read one LDS fragment per matrix instruction (1 KB per 32 cycles of the pipe per wave, 128 KB per k-tile and CU, plus
        // 32 KB of DMA writes), so the LDS is as busy as the matrix pipe and neither reaches half.
        static_assert(A_LD == 1 && B_LD == 1 && LPT == 2, "deep A ring: one activation and one filter piece per thread and tile");
        constexpr int A_SL = BM * ROWB, B_SL = BN * ROWB, B_BASE = 6 * A_SL;
        static_assert(B_BASE + 4 * B_SL <= big_lds_bytes<MODE, BN, NST, NTH>(), "deep A ring fits the dynamic LDS");
        const int npair = (nk + 1) >> 1;
        auto dma_a_pair = [&](int kp) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int kt = 2 * kp + h;
                const TilePrep tp = tile_prep(kt < nk ? kt : 0);
                const unsigned vo = (kt < nk && ((a_mask[0] >> tp.tap) & 1u)) ? a_voff[0] : OOB_VOFF;
                lds_dma16(a_rs4, lds_base + (unsigned)((2 * (kp % 3) + h) * A_SL + wave_u * 16 * ROWB), vo, tp.soff_a);
            }
        };
        auto dma_b_pair = [&](int kp) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int kt = 2 * kp + h;
                const TilePrep tp = tile_prep(kt < nk ? kt : 0);
                lds_dma16(w_rs4, lds_base + (unsigned)(B_BASE + (2 * (kp & 1) + h) * B_SL + wave_u * 16 * ROWB), kt < nk ? w_voff[0] : OOB_VOFF, tp.soff_b);
            }
        };
        dma_b_pair(0);
        dma_a_pair(0);
        if (1 < npair) dma_a_pair(1);
        for (int kp = 0; kp < npair; ++kp) {
            if (kp + 1 < npair) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* Ab = lds_c + (2 * (kp % 3)) * A_SL + a_off;
            const char* Bb = lds_c + B_BASE + (2 * (kp & 1)) * B_SL + (b_off - BM * ROWB);
            mma_ab(Ab, Bb, [&](int part) {
                if (part == 0) { if (kp + 1 < npair) dma_b_pair(kp + 1); }
                else if (kp + 2 < npair) dma_a_pair(kp + 2);
            });
            if (2 * kp + 1 < nk) mma_ab(Ab + A_SL, Bb + B_SL, [](int) {});
        }
        s1_done = true;
    }
    if (BF16 && NST == 4 && !s1_done) {
        // bf16, tiles in PAIRS: k-tiles 2 kp and 2 kp + 1 are the two 64-byte halves of the same 128-byte lines, and their DMAs
        // go out together (the same rows of both back to back), all during the first tile of the pair before; one wait and
        // one barrier per pair.  Against one tile per barrier with three tiles in flight (same box, tests/native/
        // perf_conv_shape.py bf16, two interleaved runs each): expand layers +2-3 % (layer2 199 -> 203, layer3 329 -> 340
        // TFLOP/s), the 3x3 of layer3 / layer4 when run alone +2.5 %, reduce layers unchanged.
        {
            const TilePrep t0 = tile_prep(0), t1 = tile_prep(nk > 1 ? 1 : 0);
#pragma unroll
            for (int part = 0; part < LPT; part++) {
                dma_part(t0, 0, part);
                if (nk > 1) dma_part(t1, 1, part);
            }
        }
        for (int kp = 0; 2 * kp < nk; ++kp) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int s0 = (2 * kp) & 3;
            const bool more0 = 2 * kp + 2 < nk, more1 = 2 * kp + 3 < nk;
            const TilePrep ta = tile_prep(more0 ? 2 * kp + 2 : 0), tb = tile_prep(more1 ? 2 * kp + 3 : 0);
            mma_tile(s0, [&](int part) {
                if (more0) dma_part(ta, s0 ^ 2, part);
                if (more1) dma_part(tb, (s0 ^ 2) + 1, part);
            });
            if (2 * kp + 1 < nk) mma_tile(s0 + 1, [](int) {});
        }
        s1_done = true;
    }
    if (!s1_done) {
#pragma unroll
    for (int s0 = 0; s0 < NST - 1; s0++)
        if (s0 < nk) load_tile(s0, s0);
    int st = 0, st_next = NST - 1;
#if CG_STAMP
    st_prev = __builtin_readcyclecounter();
#endif
    for (int kt = 0; kt < nk; ++kt) {
        if (kt) CG_STAMP_AT(3);
        // my DMAs retire in order: at most (NST - 2) tiles' worth outstanding = my part of tile kt has landed; the
        // barrier extends that to every wave and says ring slot (kt - 1) % NST has been read by all (it is refilled next)
        if (kt + NST - 1 <= nk) {
            if (b_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPT) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * (PRE ? LPT - 1 : A_LD)) : "memory");   // this wave stages fewer / no filter rows
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        CG_STAMP_AT(0);
        const bool more = kt + NST - 1 < nk;
        const TilePrep tp = tile_prep(more ? kt + NST - 1 : 0);
        mma_tile(st, [&](int part) { if (more) dma_part(tp, st_next, part); });
        st = st + 1 == NST ? 0 : st + 1;
        st_next = st_next + 1 == NST ? 0 : st_next + 1;
    }
    }
    }
#if CG_STAMP
    CG_STAMP_AT(3);
    if (lane == 0) {
        for (int i = 0; i < 4; i++) atomicAdd(&g_stamps[i], st_acc[i]);
        atomicAdd(&g_stamps[4], (unsigned long long)nk);
        atomicAdd(&g_stamps[5], 1ull);
    }
#endif
    // plain layers whose shape the branch-free epilogue takes: it issues its first residual requests, THEN the barrier
    bool ep2 = false;
    if constexpr (FUSE == 0) ep2 = !(POOLABLE && MODE == 0 && a.pool) && epilogue2_ok<BM, BN, WM, WN, BF16>(a);
    if (!ep2) __syncthreads();   // the last stage is fully read before the epilogue reuses the ring
    if constexpr (FUSE == 1) {
        fused_second(a, acc, smem, lds_base, m0, wave, lane);
        return;
    }
    if constexpr (FUSE == 2) {
        fused_chain(a, acc, smem, lds_base, m0, wave, lane);
        return;
    }
    if constexpr (FUSE == 3) {
        fused_second128(a, acc, smem, lds_base, m0, wave, lane);
        return;
    }
    if constexpr (FUSE == 4) {
        fused_chain128(a, acc, smem, lds_base, m0, wave, lane);
        return;
    }
    if constexpr (POOLABLE && MODE == 0) {
        if (a.pool) {      // 32x32 accumulators (lane: column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)); bf16 output: the maximum of
                           // fp32 values rounded once equals the maximum of the rounded values (rounding is monotonic)
            constexpr int SLD = BN + 4;
            static_assert(BM * SLD * 4 <= big_lds_bytes<MODE, BN, NST, NTH>(), "the pooling slab fits the ring");
            float* S = smem;
            const int half = lane >> 5, l32 = lane & 31;
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const int n = wn * (BN / WN) + j * 32 + l32;
                    const float sc = a.scale ? a.scale[n] : 1.0f, bi = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int R = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        float v = acc[i][j][r] * sc + bi;
                        v = bf2f(f2bf(v));      // rounded where the unfused path stores it
                        S[R * SLD + n] = (pool_row_ok(R) && v > 0.f) ? v : 0.f;
                    }
                }
            __syncthreads();
            pool_store(S);
            return;
        }
    }
    if constexpr (FUSE >= 5) {
        fused_bf16<BN, FUSE == 5 ? 0 : (32 << (FUSE - 5)), FUSE == 7 ? 2 : 3>(a, acc, smem, lds_base, m0, wm, wn, wave, lane);      // (128 chained outputs: no registers for a third residual group)
        return;
    }
    static_assert(X16 || (WM * WN * 32 * (BN / WN) * 4 <= big_lds_bytes<MODE, BN, NST, NTH>() && NST * STAGE <= big_lds_bytes<MODE, BN, NST, NTH>()),
                  "ring and epilogue staging fit the dynamic LDS");
    if (ep2) epilogue2<BM, BN, WM, WN, TM, TN, BF16>(a, acc, smem, m0, n0, wm, wn, wave, lane);
    else epilogue<BM, BN, WM, WN, TM, TN, BF16>(a, acc, smem, big_lds_bytes<MODE, BN, NST, NTH>() / 4, m0, n0, wm, wn, wave, lane);
}

#define g_s1_window odam_cfg::get(odam_cfg::CG_S1)      // bf16 3x3 stride 1: window main loop (0: the generic tap gather)

template <int MODE, int BN, int NST, int NTH = 512>
static int launch_big(const ConvGemmArgs& a_in, hipStream_t stream) {
    ConvGemmArgs a = a_in;
    a.s1_window = (MODE == 0 && NTH == 512) ? g_s1_window : 0;
    constexpr int bytes = big_lds_bytes<MODE, BN, NST, NTH>();
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, BN, NST, NTH>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    }();
    if (!attr_ok) return odam_fail(2, "conv_gemm: cannot raise the dynamic LDS limit");
    constexpr int bm = big_bm(BN, NTH);
    int tiles = ((a.M + bm - 1) / bm) * ((a.Cout + BN - 1) / BN);
    if (a.pool) {
        if (!(NTH == 1024 && BN == 64 && (MODE == 4 || MODE == 0)) || a.Cout != 64 || a.pool_ph != POOL_PH || a.pool_pw != POOL_PW ||
            a.ldc != a.Cout || a.res)
            return odam_fail(1, "conv_gemm: pooled stem asked for where it does not apply");
        tiles = a.B * ((a.Hp + POOL_PH - 1) / POOL_PH) * ((a.Wp + POOL_PW - 1) / POOL_PW);
    }
    hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, BN, NST, NTH>), dim3(tiles), dim3(NTH), bytes, stream, a);
    ODAM_HIP(hipGetLastError());
    return 0;
}


}  // namespace odam_cg
