// cg_tail_bf16.hpp -- the rest of a ResNet bottleneck on the ring kernel's output tile, bf16 mode (BASELINE config 4).
#pragma once
#include "cg_common.hpp"

namespace odam_cg {

// Timing builds of the tail (WRONG results; tests/native/tail_phases_bf16.sh): 1 no residual request and no store leaves the CU
// (out-of-range offsets), 2 no expand / reduce products, 4 the tail returns at once (the launch is the 3x3's main loop alone),
// 8 no expand-filter DMA.
#ifndef CG_TAILHACK
#define CG_TAILHACK 0
#endif

// ---- bf16 bottleneck on the tile (BASELINE config 4): 3x3 (P channels) + 1x1 expand (4 P) + residual [+ the next reduce] -----
// After the K loop of the 3x3 the workgroup holds a 256-row x P-channel tile (P = 64 / 128 / 256) in accumulators.  In bf16 the
// bottleneck's 1x1 layers are bound by bytes, not by the matrix pipe (layer3's expand as its own launch: 154 us for 56 GFLOP,
// 23 us per tile of which 4 are matrix instructions -- load, multiply, residual and store phases of ONE workgroup per CU run
// one after the other), so the rest of the block runs here, on rows that never leave the CU:
//   (1) scale / bias / ReLU -> bf16 -> LDS tile `mid` [256][P] (rounded exactly where the unfused path stores it);
//   (2) every wave takes 32 rows of it as A fragments into registers (P / 4 VGPRs), the LDS is then free;
//   (3) the 4 P output channels in groups of 64: the group's expand filters (64 x P bf16) -- and, chained, the k-slice of the
//       next block's reduce (PN x 64) -- arrive by LDS-DMA one group ahead (two buffers); 2 P / 16 matrix instructions per
//       wave, accumulators -> fp32 slab -> rows of 8 channels per lane: scale, bias, residual (requested one group ahead),
//       ReLU, bf16, one 16-byte store; chained: the bf16 row goes back to LDS, comes out as A fragments and is multiplied
//       into the reduce's accumulators (32 rows x PN), which after the last group are the next 3x3's input tile.
// Same products in the same order as the separate launches (k ascending in steps of 16), same rounding points: bit-identical
// to them.  vmcnt bookkeeping: the DMAs are inline assembly (the compiler does not count them); each group's top waits with
// vmcnt(12) -- the DMAs of this group were issued one group ago and exactly 4 residual loads + 4 scale / bias loads + 4 stores
// (buffer instructions with an out-of-range offset for rows past M: issued by every wave, never branched around) are younger.
// More compiler-visible loads in between would only make that wait stricter, never laxer.
template <int P, int PN, int RES_DEPTH = 3>
__device__ __forceinline__ void fused_bf16(const ConvGemmArgs& a, floatx16 (&acc)[256 / (P == 64 ? 8 : (P == 128 ? 4 : 2)) / 32][2],
                                           float* smem, unsigned lds_base, int m0, int wm, int wn, int wave, int lane) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int WM = P == 64 ? 8 : (P == 128 ? 4 : 2), WN = 8 / WM, TM = 256 / WM / 32, TN = 2;
    constexpr int PITCH = 2 * P;                       // bytes per row of `mid` and of an expand-filter group
    constexpr int NG = 4 * P / 64;                     // output groups
    constexpr int SLAB_F = 0, SLAB_B = 64 * 1024, WB = SLAB_B + (PN ? 32 * 1024 : 0);
    constexpr int GB = 128 * P + 128 * PN;             // bytes per weight buffer: 64 x P expand rows + PN x 64 reduce slice
    static_assert(WB + 2 * GB <= 160 * 1024, "fused bf16 bottleneck: LDS");
    static_assert(256 * PITCH <= 160 * 1024, "mid tile fits");
    if (CG_TAILHACK & 4) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) asm volatile("" :: "v"(acc[i][j]));
        return;
    }
    const int half = lane >> 5, l32 = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    char* lds = reinterpret_cast<char*>(smem);
    auto swz = [](int row, int chunk) { return P == 64 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ (row & 15)); };   // conflict-free ds_read_b128 by 32 rows
    const int m_base = m0 + wave * 32;
    // residual / output rows through buffer descriptors: a row past M gets an out-of-range offset (zeros / dropped), so every
    // wave issues every instruction
    const size_t out_bytes = (size_t)a.M * a.F_ldc * 2;
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.F_res), 0, a.F_res ? (int)out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.F_C, 0, (int)out_bytes, 0x00020000);
    const int c8 = lane & 7;
    unsigned row_off[4];                               // byte offset of (row it * 8 + lane / 8, channel 8 c8) of this wave's rows
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int m = m_base + it * 8 + (lane >> 3);
        row_off[it] = (m < a.M && !(CG_TAILHACK & 1)) ? (unsigned)((size_t)m * a.F_ldc * 2 + c8 * 16) : OOB_VOFF;
    }
    // residual rows are requested RD - 1 groups ahead of their use (a group's 4 KB per wave is 32 KB per CU in flight: at the
    // ~2 us a loaded HBM read takes that alone is 16 GB/s per CU; the tail's registers -- the 3x3's accumulators are dead here --
    // allow a second group where the chained reduce's accumulators leave room)
    constexpr int RD = RES_DEPTH;
    u32x4 rv[RD][4];
    auto load_res = [&](int g) {      // g >= NG: an out-of-range request (zeros, no traffic): the instruction count per group stays
#pragma unroll
        for (int it = 0; it < 4; it++)
            rv[g % RD][it] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, (g >= NG || row_off[it] == OOB_VOFF) ? OOB_VOFF : row_off[it] + (unsigned)(g * 128), 0, 0);
    };
#pragma unroll
    for (int g = 0; g < RD - 1; g++) load_res(g);
    // (1) this layer's epilogue -> mid
    {
        // (the lane's channel of column block j is the same for every row tile: one branch-free fetch per block -- TailAffine)
        const TailAffine t1(a.scale, a.bias, a.relu, P);
        float sc1[TN], bi1[TN];
#pragma unroll
        for (int j = 0; j < TN; j++) { sc1[j] = t1.scale1(wn * (P / WN) + j * 32 + l32); bi1[j] = t1.bias1(wn * (P / WN) + j * 32 + l32); }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int n = wn * (P / WN) + j * 32 + l32;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = wm * (256 / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = t1.act(acc[i][j][r] * sc1[j] + bi1[j]);
                    *reinterpret_cast<unsigned short*>(lds + row * PITCH + (swz(row, n >> 3) << 4) + (n & 7) * 2) = f2bf(v);
                }
            }
    }
    __syncthreads();
    // (2) my 32 rows as A fragments: lane (row l32, half) holds k = 16 s + 8 half .. + 7
    uint4 af[P / 16];
    {
        const int row = wave * 32 + l32;
#pragma unroll
        for (int s_ = 0; s_ < P / 16; s_++) af[s_] = *reinterpret_cast<const uint4*>(lds + row * PITCH + (swz(row, 2 * s_ + half) << 4));
    }
    __syncthreads();     // mid has been read by everybody: the LDS is free for the weight buffers and the slabs
    // (3) weights of group g -> buffer g & 1
    const unsigned long long pf = (unsigned long long)a.F_Wt, pg = (unsigned long long)a.G_Wt;
    const u32x4 rsf = u32x4{(unsigned)pf, (unsigned)(pf >> 32) & 0xffffu, (CG_TAILHACK & 8) ? 0u : OOB_VOFF, 0x00020000u};
    const u32x4 rsg = u32x4{(unsigned)pg, (unsigned)(pg >> 32) & 0xffffu, (CG_TAILHACK & 8) ? 0u : OOB_VOFF, 0x00020000u};
    auto dma_group = [&](int g) {
        const unsigned buf = lds_base + (unsigned)(WB + (g & 1) * GB);
        constexpr int RPC = 1024 / PITCH;              // filter rows per 1 KB piece
#pragma unroll
        for (int j = 0; j < P / 64; j++) {             // expand rows 64 g .. + 63: P / 8 pieces, wave + 8 j
            const int q = wave_u + 8 * j;
            const int row = q * RPC + (lane * 16) / PITCH, pos = ((lane * 16) % PITCH) >> 4;
            lds_dma16(rsf, buf + (unsigned)(q * 1024), (unsigned)((64 * g + row) * PITCH + (swz(row, pos) << 4)), 0);
        }
        if constexpr (PN > 0) {
#pragma unroll
            for (int j = 0; j < PN / 64; j++) {        // reduce slice k = 64 g .. + 63 of every output row: PN / 8 pieces of 8 rows x 128 B
                const int q = wave_u + 8 * j;
                const int row = q * 8 + (lane >> 3), pos = lane & 7;
                lds_dma16(rsg, buf + (unsigned)(128 * P + q * 1024), (unsigned)(row * (8 * P) + g * 128 + ((pos ^ ((row >> 1) & 7)) << 4)), 0);
            }
        }
    };
    dma_group(0);
    float* slab = smem + (SLAB_F / 4) + wave * (32 * 64);
    char* bslab = lds + SLAB_B + wave * (32 * 128);
    auto slab_at = [](int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    const __amdgpu_buffer_rsrc_t sc_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.F_scale), 0, a.F_scale ? 4 * P * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bi_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.F_bias), 0, a.F_bias ? 4 * P * 4 : 0, 0x00020000);
    const unsigned have_sc = a.F_scale ? 0xffffffffu : 0u;
    const int relu_fb = relu_floor_bits(a.F_relu);
    auto act = [&](float v) { return relu_floor(v, relu_fb); };      // F_relu ? (v > 0 ? v : 0) : v, without a branch
    floatx16 acc3[PN ? PN / 32 : 1];
#pragma unroll
    for (int jc = 0; jc < (PN ? PN / 32 : 1); jc++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc3[jc][r] = 0.0f;
#define ODAM_MMB(accv, x, y) accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accv, 0, 0, 0)
#pragma unroll
    for (int g = 0; g < NG; g++) {      // unrolled: the residual buffers rv[g & 1] must be compile-time register names
        if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");     // my pieces of group g's filters have landed (see the header)
        __syncthreads();                                             // ... and everybody's; everybody is done with group g - 1's buffer
        if (g + 1 < NG) dma_group(g + 1);
        load_res(g + RD - 1);
        const int n = g * 64 + c8 * 8;
        // scale / bias of the lane's 8 channels: two 16-byte loads each through descriptors that hold no records when the pointer is null
        // (the load then returns zeros; a missing scale becomes 1 by a select) -- no branch around a load, so the group stays ONE basic block
        // (round 6: `p ? p[n + c] : 1` per element was 16 scalar loads and 31 branches per group, which also fenced the scheduler in);
        // requested here, ahead of the matrix instructions that hide their round trip
        // (with 128 chained outputs the 16 registers are not there across the products: requested where they are used, as before)
        constexpr bool AFFINE_EARLY = PN != 128;
        u32x4 s0, s1, b0, b1;
        auto load_affine = [&] {
            s0 = __builtin_amdgcn_raw_buffer_load_b128(sc_rs, (unsigned)n * 4u, 0, 0); s1 = __builtin_amdgcn_raw_buffer_load_b128(sc_rs, (unsigned)n * 4u + 16u, 0, 0);
            b0 = __builtin_amdgcn_raw_buffer_load_b128(bi_rs, (unsigned)n * 4u, 0, 0); b1 = __builtin_amdgcn_raw_buffer_load_b128(bi_rs, (unsigned)n * 4u + 16u, 0, 0);
        };
        if constexpr (AFFINE_EARLY) load_affine();
        const char* wf = lds + WB + (g & 1) * GB;
        floatx16 acc2[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[jb][r] = 0.0f;
            const int nrow = jb * 32 + l32;
#pragma unroll
            for (int s_ = 0; s_ < ((CG_TAILHACK & 2) ? 0 : P / 16); s_++) {
                const uint4 b = *reinterpret_cast<const uint4*>(wf + nrow * PITCH + (swz(nrow, 2 * s_ + half) << 4));
                ODAM_MMB(acc2[jb], af[s_], b);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jb = 0; jb < 2; jb++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, jb * 32 + l32)] = acc2[jb][r];
        wave_sync();
        float sc[8], bi[8];
        if constexpr (!AFFINE_EARLY) load_affine();
        {
            const unsigned sw[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int c = 0; c < 8; c++) {
                sc[c] = __builtin_bit_cast(float, (sw[c] & have_sc) | (0x3f800000u & ~have_sc));
                bi[c] = __builtin_bit_cast(float, bw[c]);
            }
        }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 8 + (lane >> 3);
            const float4 v0 = *reinterpret_cast<const float4*>(&slab[row * 64 + (((2 * c8) ^ (row & 15)) << 2)]);
            const float4 v1 = *reinterpret_cast<const float4*>(&slab[row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2)]);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const u32x4 rr = rv[g % RD][it];
            const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
            unsigned ow[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float lo = v[2 * c] * sc[2 * c] + bi[2 * c], hi = v[2 * c + 1] * sc[2 * c + 1] + bi[2 * c + 1];
                lo += __builtin_bit_cast(float, rw[c] << 16);
                hi += __builtin_bit_cast(float, rw[c] & 0xffff0000u);
                lo = act(lo); hi = act(hi);
                ow[c] = pack_bf16x2(lo, hi);
            }
            const u32x4 o = u32x4{ow[0], ow[1], ow[2], ow[3]};
            __builtin_amdgcn_raw_buffer_store_b128(o, out_rs, row_off[it] == OOB_VOFF ? OOB_VOFF : row_off[it] + (unsigned)(g * 128), 0, 0);
            if constexpr (PN > 0) *reinterpret_cast<u32x4*>(bslab + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4)) = o;
        }
        if constexpr (PN > 0) {
            wave_sync();
            uint4 xf[4];
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) xf[s_] = *reinterpret_cast<const uint4*>(bslab + l32 * 128 + (((2 * s_ + half) ^ ((l32 >> 1) & 7)) << 4));
            const char* wg = wf + 128 * P;
#pragma unroll
            for (int jc = 0; jc < PN / 32; jc++) {
                const int nrow = jc * 32 + l32;
#pragma unroll
                for (int s_ = 0; s_ < ((CG_TAILHACK & 2) ? 0 : 4); s_++) {
                    const uint4 b = *reinterpret_cast<const uint4*>(wg + nrow * 128 + (((2 * s_ + half) ^ ((nrow >> 1) & 7)) << 4));
                    ODAM_MMB(acc3[jc], xf[s_], b);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef ODAM_MMB
    if constexpr (PN > 0) {
        // the reduce's own epilogue: scale / bias / ReLU -> bf16 [32 rows][PN], 64 channels at a time through the slab
        const size_t g_bytes = (size_t)a.M * PN * 2;
        const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(a.G_C, 0, (int)g_bytes, 0x00020000);
        const TailAffine tG(a.G_scale, a.G_bias, 1, PN);
#pragma unroll
        for (int h = 0; h < PN / 64; h++) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int jb = 0; jb < 2; jb++) {
                const int n = h * 64 + jb * 32 + l32;
                const float sc = a.G_scale ? a.G_scale[n] : 1.0f, bi = a.G_bias ? a.G_bias[n] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc3[2 * h + jb][r] * sc + bi;
                    v = v > 0.f ? v : 0.f;
                    slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, jb * 32 + l32)] = v;
                }
            }
            wave_sync();
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int row = it * 8 + (lane >> 3);
                const int m = m_base + row;
                const float4 v0 = *reinterpret_cast<const float4*>(&slab[row * 64 + (((2 * c8) ^ (row & 15)) << 2)]);
                const float4 v1 = *reinterpret_cast<const float4*>(&slab[row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2)]);
                const u32x4 o = u32x4{pack_bf16x2(v0.x, v0.y), pack_bf16x2(v0.z, v0.w), pack_bf16x2(v1.x, v1.y), pack_bf16x2(v1.z, v1.w)};
                __builtin_amdgcn_raw_buffer_store_b128(o, g_rs, m < a.M ? (unsigned)((size_t)m * PN * 2 + h * 128 + c8 * 16) : OOB_VOFF, 0, 0);
            }
        }
    }
}


}  // namespace odam_cg
