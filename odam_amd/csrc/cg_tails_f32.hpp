// cg_tails_f32.hpp -- the rest of a ResNet bottleneck on the ring kernel's output tile, fp32 split mode (layer1 / layer2).
#pragma once
#include "cg_common.hpp"

namespace odam_cg {

// ---- fused second layer of the 64-column ring kernel (ConvGemmArgs F_*) ---------------------------------------------------
// After the K loop a wave holds 32 rows x 64 channels of this layer's output in two accumulators.  It applies scale / bias
// / ReLU, turns the tile through a private 8 KB LDS slab (row-major, 16-byte chunk c of row r at c ^ (r & 15): conflict-
// free for the accumulator-layout writes, the A-fragment reads and the row-major reads of the store path) into A fragments
// -- 8 consecutive k per lane, split exactly into three bf16 operands as in the main loop -- and multiplies them by the
// expand layer's pre-split filters, which the workgroup has meanwhile pulled into the dead ring by LDS-DMA (96 KB:
// [k-step 4][plane 3][256 rows][32 bytes], the plane layout of the main loop).  192 matrix instructions per wave; the
// 256 output channels leave in four groups of 64 through the same slab: scale, bias, residual, ReLU, 16-byte stores.
// Why: a bottleneck's 3x3 is matrix-bound and its expand + residual HBM-bound; as two launches each leaves the other
// resource idle (one workgroup per CU), fused the CUs run out of phase and the two overlap across the chip.
template <class ACC>
__device__ __forceinline__ void fused_second(const ConvGemmArgs& a, ACC& acc, float* smem, unsigned lds_base,
                                             int m0, int wave, int lane) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const int half = lane >> 5, l32 = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // this layer's and the expand's scale / bias / ReLU: branch-free (TailAffine); this layer's are requested now
    const TailAffine t1(a.scale, a.bias, a.relu, a.Cout), tF(a.F_scale, a.F_bias, a.F_relu, a.F_ldc);
    float sc1[4], bi1[4];
    cols64_affine(acc, t1, 0, lane, sc1, bi1);
    // (1) expand filters -> LDS: piece q = wave + 8 j covers (k-step, plane) = q / 8, rows 32 (q % 8) .. + 31
    {
        const unsigned long long pf = (unsigned long long)a.F_Wt3;
        const u32x4 rs = u32x4{(unsigned)pf, (unsigned)(pf >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const int q = wave_u + 8 * j;
            const int sp = q >> 3, rb = q & 7, s_ = sp / 3, pl = sp - 3 * s_;
            const int row = rb * 32 + (lane >> 1);
            const unsigned vo = (unsigned)(row * 384 + s_ * 96 + pl * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) * 16));
            lds_dma16(rs, lds_base + (unsigned)(q * 1024), vo, 0);
        }
    }
    // the residual rows are requested two output groups ahead of their use (the first two now): their HBM latency runs
    // under the filter fetch and the matrix instructions (requested right before use they cost four exposed round trips
    // per tile; all four groups at once do not fit the registers)
    const int c4 = lane & 15;                              // 16-byte chunk of a 64-channel group this lane stores
    const int m_base = m0 + wave * 32;
    float4 rv[2][8];
    auto load_res = [&](int g) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int m = m_base + it * 4 + (lane >> 4);
            rv[g & 1][it] = float4{0.f, 0.f, 0.f, 0.f};
            if (a.F_res && m < a.M) rv[g & 1][it] = *reinterpret_cast<const float4*>(a.F_res + (size_t)m * a.F_ldc + g * 64 + c4 * 4);
        }
    };
    load_res(0); load_res(1);
    // (2) this layer's epilogue in registers, tile -> slab
    float* slab = smem + FUSE_W3_BYTES / 4 + wave * (32 * 64);
    auto slab_at = [](int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); };
    acc_cols64(acc, 0, lane, [&](int row, int c64, int j, float x) { slab[slab_at(row, c64)] = t1.act(x * sc1[j] + bi1[j]); });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // (3) A fragments: lane (row l32, half) takes k = 16 s + 8 half .. + 7
    uint4 ah[4], am[4], al[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
        const float4 c0 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half)]);
        const float4 c1 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half + 4)]);
        cg_split8(c0, c1, ah[s_], am[s_], al[s_]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of the filters have landed
    __syncthreads();                                       // ... and everybody's; also: all slabs have been read
    // (4) four groups of 64 output channels
    const char* w3 = reinterpret_cast<const char*>(smem);
    const int b_off = l32 * 32 + ((half ^ ((l32 >> 3) & 1)) * 16);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int n = g * 64 + c4 * 4;
        const float4 sc = tF.scale4(n), bi = tF.bias4(n);      // requested here: their round trip runs under the matrix instructions
        floatx16 acc2[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[jb][r] = 0.0f;
            const int nb = 2 * g + jb;
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) {
                const char* bp = w3 + (s_ * 3) * 8192 + nb * 1024 + b_off;
                const uint4 bh = *reinterpret_cast<const uint4*>(bp), bm = *reinterpret_cast<const uint4*>(bp + 8192),
                            bl = *reinterpret_cast<const uint4*>(bp + 16384);
#define ODAM_MM2(x, y) acc2[jb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), acc2[jb], 0, 0, 0)
                ODAM_MM2(al[s_], bh); ODAM_MM2(am[s_], bm); ODAM_MM2(ah[s_], bl);
                ODAM_MM2(am[s_], bh); ODAM_MM2(ah[s_], bm);
                ODAM_MM2(ah[s_], bh);
#undef ODAM_MM2
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jb = 0; jb < 2; jb++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, jb * 32 + l32)] = acc2[jb][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            const float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            float4 o = {v.x * sc.x + bi.x + rv[g & 1][it].x, v.y * sc.y + bi.y + rv[g & 1][it].y, v.z * sc.z + bi.z + rv[g & 1][it].z,
                        v.w * sc.w + bi.w + rv[g & 1][it].w};
            o.x = tF.act(o.x); o.y = tF.act(o.y); o.z = tF.act(o.z); o.w = tF.act(o.w);
            if (m < a.M) *reinterpret_cast<float4*>(a.F_C + (size_t)m * a.F_ldc + n) = o;
        }
        if (g + 2 < 4) load_res(g + 2);
    }
}

// ---- the chain continued: expand (F_*) AND the next bottleneck's reduce (G_*) on the tile -------------------------------
// Same data flow as fused_second, but the expand filters are streamed: output group g (64 of the 256 channels) needs only
// rows 64 g .. 64 g + 63 of them (24 KB as three planes), and the next block's 256 -> 64 reduce needs only k = 64 g .. + 63
// of ITS filters for that group (another 24 KB).  Both are double-buffered in LDS (2 x 2 x 24 KB) beside the eight 8 KB
// slabs -- 160 KB exactly -- and fetched one group ahead by LDS-DMA.  After a group's residual / ReLU its 32 x 64 final
// values go back into the slab, come out as A fragments and are multiplied into a 32 x 64 accumulator of the reduce;
// after the fourth group that accumulator is the next block's 3x3 input tile.  Saves the launch that re-reads the
// 256-channel tensor (the largest read of layer1).
template <class ACC>
__device__ __forceinline__ void fused_chain(const ConvGemmArgs& a, ACC& acc, float* smem, unsigned lds_base,
                                            int m0, int wave, int lane) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int SLABS = 8 * 32 * 64 * 4, GB = 4 * 3 * 64 * 32;      // 64 KB of slabs; 24 KB per filter group
    const int half = lane >> 5, l32 = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // this layer's and the expand's scale / bias / ReLU: branch-free (TailAffine); this layer's are requested now
    const TailAffine t1(a.scale, a.bias, a.relu, a.Cout), tF(a.F_scale, a.F_bias, a.F_relu, a.F_ldc);
    float sc1[4], bi1[4];
    cols64_affine(acc, t1, 0, lane, sc1, bi1);
    const unsigned long long pf = (unsigned long long)a.F_Wt3, pg = (unsigned long long)a.G_Wt3;
    const u32x4 rsf = u32x4{(unsigned)pf, (unsigned)(pf >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const u32x4 rsg = u32x4{(unsigned)pg, (unsigned)(pg >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const int swz = ((lane & 1) ^ ((lane >> 4) & 1)) * 16;
    // filters of output group g -> buffer g & 1: 24 + 24 pieces of 1 KB, piece q = wave + 8 j: (k-step, plane) = q / 2, rows 32 (q % 2) .. + 31
    auto dma_group = [&](int g) {
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int q = wave_u + 8 * j;
            const int sp = q >> 1, rb = q & 1, s_ = sp / 3, pl = sp - 3 * s_;
            const int row = rb * 32 + (lane >> 1);
            lds_dma16(rsf, lds_base + (unsigned)(SLABS + (g & 1) * GB + q * 1024),
                      (unsigned)((64 * g + row) * 384 + s_ * 96 + pl * 32 + swz), 0);
            lds_dma16(rsg, lds_base + (unsigned)(SLABS + 2 * GB + (g & 1) * GB + q * 1024),
                      (unsigned)(row * (16 * 96) + (4 * g + s_) * 96 + pl * 32 + swz), 0);
        }
    };
    const int c4 = lane & 15;
    const int m_base = m0 + wave * 32;
    // residual rows and output rows through buffer descriptors: a row past M gets an out-of-range offset (zeros / dropped),
    // so every wave issues every load and store and the group loop can wait with a COUNT (see its top)
    const size_t out_bytes = (size_t)a.M * a.F_ldc * 4;
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.F_res), 0, a.F_res ? (int)out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.F_C, 0, (int)out_bytes, 0x00020000);
    unsigned row_off[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int m = m_base + it * 4 + (lane >> 4);
        row_off[it] = m < a.M ? (unsigned)((size_t)m * a.F_ldc * 4 + c4 * 16) : OOB_VOFF;
    }
    float4 rv[8];              // residual rows of one output group, requested one group ahead
    auto load_res = [&](int g) {
#pragma unroll
        for (int it = 0; it < 8; it++)
            rv[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, row_off[it] == OOB_VOFF ? OOB_VOFF : row_off[it] + (unsigned)(g * 256), 0, 0));
    };
#if CG_STAMP
    unsigned long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ts_prev = __builtin_readcyclecounter();      // (not fenced: slot 6 also holds the main loop)
#endif
    load_res(0);
    dma_group(0);
    float* slab = smem + wave * (32 * 64);
    auto slab_at = [](int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); };
    auto frags = [&](uint4 (&xh)[4], uint4 (&xm)[4], uint4 (&xl)[4]) {     // the slab's 32 x 64 values as A operands
#pragma unroll
        for (int s_ = 0; s_ < 4; s_++) {
            const float4 c0 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half)]);
            const float4 c1 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half + 4)]);
            cg_split8(c0, c1, xh[s_], xm[s_], xl[s_]);
        }
    };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    acc_cols64(acc, 0, lane, [&](int row, int c64, int j, float x) { slab[slab_at(row, c64)] = t1.act(x * sc1[j] + bi1[j]); });
    wave_sync();
    uint4 ah[4], am[4], al[4];
    frags(ah, am, al);
    CG_TSTAMP(6);
    floatx16 acc3[2];
#pragma unroll
    for (int jc = 0; jc < 2; jc++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc3[jc][r] = 0.0f;
    const char* lds_c = reinterpret_cast<const char*>(smem);
    const int b_off = l32 * 32 + ((half ^ ((l32 >> 3) & 1)) * 16);
#define ODAM_MM2(accv, x, y) accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accv, 0, 0, 0)
#pragma unroll
    for (int g = 0; g < 4; g++) {
        // my pieces of group g's filters have landed: they were issued at the top of group g - 1, and exactly 8 stores + 8
        // residual loads (buffer instructions, never branched around) are younger -- those may stay in flight
        if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __syncthreads();                                        // ... and everybody's; everybody is done with group g - 1's buffers
        CG_TSTAMP(0);
        if (g + 1 < 4) dma_group(g + 1);
        const char* wf = lds_c + SLABS + (g & 1) * GB;
        const char* wg = lds_c + SLABS + 2 * GB + (g & 1) * GB;
        const int n = g * 64 + c4 * 4;
        const float4 sc = tF.scale4(n), bi = tF.bias4(n);      // requested here: their round trip runs under the matrix instructions
        floatx16 acc2[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[jb][r] = 0.0f;
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) {
                const char* bp = wf + (s_ * 3) * 2048 + jb * 1024 + b_off;
                const uint4 bh = *reinterpret_cast<const uint4*>(bp), bm = *reinterpret_cast<const uint4*>(bp + 2048),
                            bl = *reinterpret_cast<const uint4*>(bp + 4096);
                ODAM_MM2(acc2[jb], al[s_], bh); ODAM_MM2(acc2[jb], am[s_], bm); ODAM_MM2(acc2[jb], ah[s_], bl);
                ODAM_MM2(acc2[jb], am[s_], bh); ODAM_MM2(acc2[jb], ah[s_], bm);
                ODAM_MM2(acc2[jb], ah[s_], bh);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jb = 0; jb < 2; jb++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, jb * 32 + l32)] = acc2[jb][r];
        wave_sync();
        CG_TSTAMP(1);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            float4* sp_ = reinterpret_cast<float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            const float4 v = *sp_;
            float4 o = {v.x * sc.x + bi.x + rv[it].x, v.y * sc.y + bi.y + rv[it].y, v.z * sc.z + bi.z + rv[it].z,
                        v.w * sc.w + bi.w + rv[it].w};
            o.x = tF.act(o.x); o.y = tF.act(o.y); o.z = tF.act(o.z); o.w = tF.act(o.w);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rs, row_off[it] == OOB_VOFF ? OOB_VOFF : row_off[it] + (unsigned)(g * 256), 0, 0);
            *sp_ = o;                                           // the final values: this group's slice of the reduce's input
        }
        if (g + 1 < 4) load_res(g + 1);
        wave_sync();
        CG_TSTAMP(2);
        uint4 xh[4], xm[4], xl[4];
        frags(xh, xm, xl);
        CG_TSTAMP(3);
#pragma unroll
        for (int jc = 0; jc < 2; jc++)
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) {
                const char* bp = wg + (s_ * 3) * 2048 + jc * 1024 + b_off;
                const uint4 bh = *reinterpret_cast<const uint4*>(bp), bm = *reinterpret_cast<const uint4*>(bp + 2048),
                            bl = *reinterpret_cast<const uint4*>(bp + 4096);
                ODAM_MM2(acc3[jc], xl[s_], bh); ODAM_MM2(acc3[jc], xm[s_], bm); ODAM_MM2(acc3[jc], xh[s_], bl);
                ODAM_MM2(acc3[jc], xm[s_], bh); ODAM_MM2(acc3[jc], xh[s_], bm);
                ODAM_MM2(acc3[jc], xh[s_], bh);
            }
        __builtin_amdgcn_wave_barrier();
        CG_TSTAMP(4);
    }
#undef ODAM_MM2
    // the reduce's own epilogue: scale / bias / ReLU, [32 rows][64 channels] out through the slab
#pragma unroll
    for (int jc = 0; jc < 2; jc++) {
        const int n = jc * 32 + l32;
        const float sc = a.G_scale ? a.G_scale[n] : 1.0f, bi = a.G_bias ? a.G_bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float v = acc3[jc][r] * sc + bi;
            v = v > 0.f ? v : 0.f;
            slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, n)] = v;
        }
    }
    wave_sync();
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int row = it * 4 + (lane >> 4);
        const int m = m_base + row;
        const float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
        if (m < a.M) *reinterpret_cast<float4*>(a.G_C + (size_t)m * 64 + c4 * 4) = v;
    }
#if CG_STAMP
    CG_TSTAMP(7);
    if (lane == 0) {
        for (int i = 0; i < 8; i++) if (i != 5) atomicAdd(&g_tail_stamps[i], ts_acc[i]);
        atomicAdd(&g_tail_stamps[5], 1ull);
    }
#endif
}

// ---- the chain into the NEXT STAGE: layer1's last block + layer2's first reduce (256 -> 128 channels) ---------------------
// As fused_chain with 128 reduce outputs: four 32x32 accumulators, the reduce's k-slice of a group is 48 KB and has ONE
// buffer (slabs 64 + expand filters 2 x 24 + reduce slice 48 = 160 KB): it is fetched at the top of its own group and
// awaited (with one more workgroup barrier) before the group's second half; its A fragments are split one k-step at a time
// to stay inside 256 registers.
template <class ACC>
__device__ __forceinline__ void fused_chain128(const ConvGemmArgs& a, ACC& acc, float* smem, unsigned lds_base,
                                               int m0, int wave, int lane) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int SLABS = 8 * 32 * 64 * 4, GB = 4 * 3 * 64 * 32, GB2 = 4 * 3 * 128 * 32;      // 64 KB; 24 KB; 48 KB
    static_assert(SLABS + 2 * GB + GB2 == 160 * 1024, "slabs, two expand groups and one reduce slice fill the LDS");
    int half = lane >> 5, l32 = lane & 31;      // laundered inside the group loop: the ~60 slab addresses derived from them are
                                                // loop invariants the compiler would otherwise keep in registers (and spill others)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // this layer's and the expand's scale / bias / ReLU: branch-free (TailAffine); this layer's are requested now
    const TailAffine t1(a.scale, a.bias, a.relu, a.Cout), tF(a.F_scale, a.F_bias, a.F_relu, a.F_ldc);
    float sc1[4], bi1[4];
    cols64_affine(acc, t1, 0, lane, sc1, bi1);
    const unsigned long long pf = (unsigned long long)a.F_Wt3, pg = (unsigned long long)a.G_Wt3;
    const u32x4 rsf = u32x4{(unsigned)pf, (unsigned)(pf >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const u32x4 rsg = u32x4{(unsigned)pg, (unsigned)(pg >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const int swz = ((lane & 1) ^ ((lane >> 4) & 1)) * 16;
    auto dma_expand = [&](int g) {      // 24 pieces: q = wave + 8 j: (k-step, plane) = q / 2, rows 32 (q % 2) .. + 31 of the group
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int q = wave_u + 8 * j;
            const int sp = q >> 1, rb = q & 1, s_ = sp / 3, pl = sp - 3 * s_;
            const int row = rb * 32 + (lane >> 1);
            lds_dma16(rsf, lds_base + (unsigned)(SLABS + (g & 1) * GB + q * 1024),
                      (unsigned)((64 * g + row) * 384 + s_ * 96 + pl * 32 + swz), 0);
        }
    };
    auto dma_reduce = [&](int g) {      // 48 pieces: q = wave + 8 j: (k-step, plane) = q / 4, rows 32 (q % 4) .. + 31 of the 128 outputs
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int q = wave_u + 8 * j;
            const int sp = q >> 2, rb = q & 3, s_ = sp / 3, pl = sp - 3 * s_;
            const int row = rb * 32 + (lane >> 1);
            lds_dma16(rsg, lds_base + (unsigned)(SLABS + 2 * GB + q * 1024),
                      (unsigned)(row * (16 * 96) + (4 * g + s_) * 96 + pl * 32 + swz), 0);
        }
    };
    const int c4 = lane & 15;
    const int m_base = m0 + wave * 32;
    // residual rows and output rows through buffer descriptors (a row past M: out-of-range offset -- zeros / dropped), so that
    // every wave issues every load and store and the waits below can COUNT; the offsets are rebuilt from one register per use
    // (eight kept ones do not fit this tail's registers)
    const size_t out_bytes = (size_t)a.M * a.F_ldc * 4;
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.F_res), 0, a.F_res ? (int)out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.F_C, 0, (int)out_bytes, 0x00020000);
    const unsigned ldc_b = (unsigned)a.F_ldc * 4u;
    const int m_lane = m_base + (lane >> 4);
    const unsigned off_lane = (unsigned)m_lane * ldc_b + (unsigned)(c4 * 16);
    auto voff = [&](int it, int g) -> unsigned {
        return (m_lane + it * 4 < a.M) ? off_lane + (unsigned)(it * 4) * ldc_b + (unsigned)(g * 256) : OOB_VOFF;
    };
    float4 rv[8];
    auto load_res = [&](int g) {
#pragma unroll
        for (int it = 0; it < 8; it++) rv[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, voff(it, g), 0, 0));
    };
    dma_expand(0);
    float* slab = smem + wave * (32 * 64);
    auto slab_at = [](int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    acc_cols64(acc, 0, lane, [&](int row, int c64, int j, float x) { slab[slab_at(row, c64)] = t1.act(x * sc1[j] + bi1[j]); });
    wave_sync();
    uint4 ah[4], am[4], al[4];
#pragma unroll
    for (int s_ = 0; s_ < 4; s_++) {
        const float4 c0 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half)]);
        const float4 c1 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half + 4)]);
        cg_split8(c0, c1, ah[s_], am[s_], al[s_]);
    }
    floatx16 acc3[4];
#pragma unroll
    for (int jc = 0; jc < 4; jc++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc3[jc][r] = 0.0f;
    const char* lds_c = reinterpret_cast<const char*>(smem);
#define ODAM_MM2(accv, x, y) accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accv, 0, 0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // my pieces of group 0's expand filters
#pragma unroll 1
    for (int g = 0; g < 4; g++) {
        asm volatile("" : "+v"(half), "+v"(l32));
        // group g's expand filters are in: group 0's by the wait before the loop, the others' by the wait in the middle of
        // group g - 1 (they were requested before the reduce slice that wait is for); only that group's stores are in flight
        __syncthreads();                                        // ... everybody's; and everybody is done with group g - 1
        if (g + 1 < 4) dma_expand(g + 1);
        dma_reduce(g);
        load_res(g);                                            // this group's residual rows: under the expand's matrix instructions
        const int b_off = l32 * 32 + ((half ^ ((l32 >> 3) & 1)) * 16);
        const char* wf = lds_c + SLABS + (g & 1) * GB;
        const char* wg = lds_c + SLABS + 2 * GB;
        const int n = g * 64 + c4 * 4;
        const float4 sc = tF.scale4(n), bi = tF.bias4(n);      // requested here: their round trip runs under the matrix instructions
        floatx16 acc2[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[jb][r] = 0.0f;
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) {
                const char* bp = wf + (s_ * 3) * 2048 + jb * 1024 + b_off;
                const uint4 bh = *reinterpret_cast<const uint4*>(bp), bm = *reinterpret_cast<const uint4*>(bp + 2048),
                            bl = *reinterpret_cast<const uint4*>(bp + 4096);
                ODAM_MM2(acc2[jb], al[s_], bh); ODAM_MM2(acc2[jb], am[s_], bm); ODAM_MM2(acc2[jb], ah[s_], bl);
                ODAM_MM2(acc2[jb], am[s_], bh); ODAM_MM2(acc2[jb], ah[s_], bm);
                ODAM_MM2(acc2[jb], ah[s_], bh);
                __builtin_amdgcn_sched_barrier(0);              // keeps the filter fragments of later k-steps out of the registers
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" : "+v"(half), "+v"(l32));
#pragma unroll
        for (int jb = 0; jb < 2; jb++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, jb * 32 + l32)] = acc2[jb][r];
        wave_sync();
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            float4* sp_ = reinterpret_cast<float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            const float4 v = *sp_;
            float4 o = {v.x * sc.x + bi.x + rv[it].x, v.y * sc.y + bi.y + rv[it].y, v.z * sc.z + bi.z + rv[it].z,
                        v.w * sc.w + bi.w + rv[it].w};
            o.x = tF.act(o.x); o.y = tF.act(o.y); o.z = tF.act(o.z); o.w = tF.act(o.w);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rs, voff(it, g), 0, 0);
            *sp_ = o;
        }
        wave_sync();
        // the reduce slice of this group (requested at the top) is in: younger than its pieces are exactly 2 scale / bias
        // loads, 8 residual loads and 8 stores -- buffer instructions, never branched around; the stores may stay in flight
        // (the next group's expand filters are OLDER than the slice: this wait covers them as well)
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __syncthreads();
        asm volatile("" : "+v"(half), "+v"(l32));
#pragma unroll 1
        for (int s_ = 0; s_ < 4; s_++) {                        // not unrolled: four k-steps of filter fragments at once spill
            uint4 xh, xm, xl;
            {
                const float4 c0 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half)]);
                const float4 c1 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half + 4)]);
                cg_split8(c0, c1, xh, xm, xl);
            }
#pragma unroll
            for (int jc = 0; jc < 4; jc++) {
                const char* bp = wg + (s_ * 3) * 4096 + jc * 1024 + b_off;
                const uint4 bh = *reinterpret_cast<const uint4*>(bp), bm = *reinterpret_cast<const uint4*>(bp + 4096),
                            bl = *reinterpret_cast<const uint4*>(bp + 8192);
                ODAM_MM2(acc3[jc], xl, bh); ODAM_MM2(acc3[jc], xm, bm); ODAM_MM2(acc3[jc], xh, bl);
                ODAM_MM2(acc3[jc], xm, bh); ODAM_MM2(acc3[jc], xh, bm);
                ODAM_MM2(acc3[jc], xh, bh);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef ODAM_MM2
    // the reduce's own epilogue, 128 channels in two passes of 64 through the slab
    const TailAffine tG(a.G_scale, a.G_bias, 1, 128);
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = h2 * 64 + j * 32 + l32;
                const float sc = tG.scale1(n), bi = tG.bias1(n);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                float v = acc3[2 * h2 + j][r] * sc + bi;
                v = v > 0.f ? v : 0.f;
                slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, j * 32 + l32)] = v;
            }
        }
        wave_sync();
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            const float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            if (m < a.M) *reinterpret_cast<float4*>(a.G_C + (size_t)m * 128 + h2 * 64 + c4 * 4) = v;
        }
        wave_sync();
    }
}

// ---- the same for 128 -> 512 channels (layer2's 3x3 + expand + residual; 128-column tiles: a wave owns 32 rows x 128) ----
// The 3x3's tile goes through the 8 KB slab in two 64-channel halves (eight k-steps of A fragments stay in registers); the
// expand filters of an output group are 48 KB as three planes, double-buffered beside the slabs (64 + 2 x 48 = 160 KB);
// eight output groups of 64 channels.  No chained reduce here: its filter slices would need another 96 KB.
template <class ACC>
__device__ __forceinline__ void fused_second128(const ConvGemmArgs& a, ACC& acc, float* smem, unsigned lds_base,
                                                int m0, int wave, int lane) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int SLABS = 8 * 32 * 64 * 4, KS = 8, GB = KS * 3 * 64 * 32;      // 64 KB of slabs; 48 KB per filter group
    static_assert(SLABS + 2 * GB == 160 * 1024, "slabs and two filter groups fill the LDS");
    const int half = lane >> 5, l32 = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // this layer's and the expand's scale / bias / ReLU: branch-free (TailAffine); this layer's are requested now
    const TailAffine t1(a.scale, a.bias, a.relu, a.Cout), tF(a.F_scale, a.F_bias, a.F_relu, a.F_ldc);
    float sc1[2][4], bi1[2][4];
    cols64_affine(acc, t1, 0, lane, sc1[0], bi1[0]);
    cols64_affine(acc, t1, 1, lane, sc1[1], bi1[1]);
    const unsigned long long pf = (unsigned long long)a.F_Wt3;
    const u32x4 rsf = u32x4{(unsigned)pf, (unsigned)(pf >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const int swz = ((lane & 1) ^ ((lane >> 4) & 1)) * 16;
    const int NG = a.F_ldc >> 6;                            // output groups of 64 channels
    auto dma_group = [&](int g) {                           // 48 pieces of 1 KB: piece q = wave + 8 j: (k-step, plane) = q / 2, rows 32 (q % 2) .. + 31
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const int q = wave_u + 8 * j;
            const int sp = q >> 1, rb = q & 1, s_ = sp / 3, pl = sp - 3 * s_;
            const int row = rb * 32 + (lane >> 1);
            lds_dma16(rsf, lds_base + (unsigned)(SLABS + (g & 1) * GB + q * 1024),
                      (unsigned)((64 * g + row) * (KS * 96) + s_ * 96 + pl * 32 + swz), 0);
        }
    };
    const int c4 = lane & 15;
    const int m_base = m0 + wave * 32;
    // residual rows and output rows through buffer descriptors: a row past M gets an out-of-range offset (zeros / dropped),
    // so every wave issues every load and store and the group loop can wait with a COUNT (see its top)
    const size_t out_bytes = (size_t)a.M * a.F_ldc * 4;
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.F_res), 0, a.F_res ? (int)out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(a.F_C, 0, (int)out_bytes, 0x00020000);
    unsigned row_off[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const int m = m_base + it * 4 + (lane >> 4);
        row_off[it] = m < a.M ? (unsigned)((size_t)m * a.F_ldc * 4 + c4 * 16) : OOB_VOFF;
    }
    float4 rv[8];              // residual rows of one output group, requested one group ahead
    auto load_res = [&](int g) {
#pragma unroll
        for (int it = 0; it < 8; it++)
            rv[it] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(res_rs, row_off[it] == OOB_VOFF ? OOB_VOFF : row_off[it] + (unsigned)(g * 256), 0, 0));
    };
    load_res(0);
    dma_group(0);
    float* slab = smem + wave * (32 * 64);
    auto slab_at = [](int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    uint4 ah[KS], am[KS], al[KS];
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {                        // channels 64 h2 .. + 63 of the 3x3's tile
        acc_cols64(acc, h2, lane, [&](int row, int c64, int j, float x) { slab[slab_at(row, c64)] = t1.act(x * sc1[h2][j] + bi1[h2][j]); });
        wave_sync();
#pragma unroll
        for (int s_ = 0; s_ < 4; s_++) {
            const float4 c0 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half)]);
            const float4 c1 = *reinterpret_cast<const float4*>(&slab[slab_at(l32, 16 * s_ + 8 * half + 4)]);
            cg_split8(c0, c1, ah[4 * h2 + s_], am[4 * h2 + s_], al[4 * h2 + s_]);
        }
        wave_sync();
    }
    const char* lds_c = reinterpret_cast<const char*>(smem);
    const int b_off = l32 * 32 + ((half ^ ((l32 >> 3) & 1)) * 16);
#define ODAM_MM2(accv, x, y) accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accv, 0, 0, 0)
#pragma unroll 1
    for (int g = 0; g < NG; g++) {
        // group g's filters were issued at the top of group g - 1; exactly 8 stores + 8 residual loads (buffer instructions, never
        // branched around) are younger -- those may stay in flight
        if (g == 0 || (CG_HACK & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __syncthreads();
        if (g + 1 < NG) dma_group(g + 1);
        const char* wf = lds_c + SLABS + (g & 1) * GB;
        const int n = g * 64 + c4 * 4;
        const float4 sc = tF.scale4(n), bi = tF.bias4(n);      // requested here: their round trip runs under the matrix instructions
        floatx16 acc2[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[jb][r] = 0.0f;
#pragma unroll
            for (int s_ = 0; s_ < ((CG_HACK & 8) ? 0 : KS); s_++) {      // (CG_HACK & 8: timing build without the expand products)
                const char* bp = wf + (s_ * 3) * 2048 + jb * 1024 + b_off;
                const uint4 bh = *reinterpret_cast<const uint4*>(bp), bm = *reinterpret_cast<const uint4*>(bp + 2048),
                            bl = *reinterpret_cast<const uint4*>(bp + 4096);
                ODAM_MM2(acc2[jb], al[s_], bh); ODAM_MM2(acc2[jb], am[s_], bm); ODAM_MM2(acc2[jb], ah[s_], bl);
                ODAM_MM2(acc2[jb], am[s_], bh); ODAM_MM2(acc2[jb], ah[s_], bm);
                ODAM_MM2(acc2[jb], ah[s_], bh);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jb = 0; jb < 2; jb++)
#pragma unroll
            for (int r = 0; r < 16; r++)
                slab[slab_at((r & 3) + 8 * (r >> 2) + 4 * half, jb * 32 + l32)] = acc2[jb][r];
        wave_sync();
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            const float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            float4 o = {v.x * sc.x + bi.x + rv[it].x, v.y * sc.y + bi.y + rv[it].y, v.z * sc.z + bi.z + rv[it].z,
                        v.w * sc.w + bi.w + rv[it].w};
            o.x = tF.act(o.x); o.y = tF.act(o.y); o.z = tF.act(o.z); o.w = tF.act(o.w);
#if CG_HACK & 4      // timing build without the tail's memory traffic: the value is consumed, nothing is stored, no residual row is requested
            asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
#else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rs, row_off[it] == OOB_VOFF ? OOB_VOFF : row_off[it] + (unsigned)(g * 256), 0, 0);
#endif
        }
#if !(CG_HACK & 4)
        if (g + 1 < NG) load_res(g + 1);
#endif
        __builtin_amdgcn_wave_barrier();
    }
#undef ODAM_MM2
}


}  // namespace odam_cg
