// cg_common.hpp -- what every translation unit of the contraction kernels shares: vector types, the hand-written LDS-DMA
// instruction, the scale / bias / residual / ReLU epilogue, the exact three-way bf16 split of a fragment.  See conv_gemm.h.
#pragma once
#include "conv_gemm.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "odam_config.h"
#include "odam_err.h"

namespace odam_cg {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// One k-tile is 8 chunks of 16 bytes per row in both precisions: 32 fp32 or 64 bf16 values; LDS rows are
// padded by one chunk (144-byte rows: 16-byte aligned, conflict-free for ds_read_b128).
constexpr int LDS_STRIDE = 36;  // floats per LDS row
// CG_STAMP=1 (diagnostic build only, tests/native/ring_stamps.hip): the ring kernel accumulates s_memtime deltas of its
// loop phases per wave (1 s_waitcnt vmcnt, 0 s_barrier, 2 first half of the matrix instructions, 3 the rest; the
// unpipelined loops: 0 wait + barrier, 1 fragment reads + DMA issue) into g_stamps[0..3], k-tiles into [4], waves into [5].  No stamp executes in the product build.
#ifndef CG_STAMP
#define CG_STAMP 0
#endif
// CG_STAMP=2 (round 5): the LIGHT set for the 16x16x32 loop -- per pair only the wait (before the vmcnt -> behind the barrier: phases 1 + 0,
// the products go to phase 2), per tile entry -> loop ([6]) and loop end -> last store issued ([7]); every sum leaves through atomics at the
// very END of the kernel (round 4's set issued them before the epilogue, where they sit in front of the epilogue's counted waits).
#if CG_STAMP
static __device__ unsigned long long g_stamps[8];
#define CG_STAMP_AT(i) do { if (CG_STAMP == 1 || (i) != 2) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                            st_acc[(CG_STAMP == 2 && (i) == 3) ? 2 : (i)] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CG_STAMP_AT(i) do { } while (0)
#endif
#if CG_STAMP      // phase stamps of the layer1 bottleneck tail (fused_chain), read by tests/native/tail_stamps.py through odam_cg_tail_stamps
static __device__ unsigned long long g_tail_stamps[8];
#define CG_TSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                          ts_acc[i] += t_ - ts_prev; ts_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CG_TSTAMP(i) do { } while (0)
#endif
#ifndef CG_X16_BREG
#define CG_X16_BREG 1      // the 16x16x32 loop (cg_big.hpp): filter fragments of a column group stay in registers; 0 = round 4's block order (A/B builds)
#endif
#ifndef CG_X16_WM8
#define CG_X16_WM8 1       // the 256-column 16x16x32 tiles as eight waves of 32 x 256 (no activation row is split by two waves; every wave reads every filter fragment); 0 = 4 x 2 waves of 64 x 128 (A/B builds)
#endif
#ifndef CG_HACK
#define CG_HACK 0      // timing experiments (WRONG results; 0 in every product build): bits 1, 2 = the 16x16x32 loop's epilogue (cg_big.hpp); layer2's fused tail (cg_tails_f32.hpp): 4 = without its residual loads and stores, 8 = without its products
#endif
#ifndef CG_ABLATE
#define CG_ABLATE 0   // timing experiments only (tests/native/perf_conv_ablate.py): 1 no global loads, 2 no LDS stores, 4 no barrier, 8 no output stores, 16 no first-tile load
#endif

__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
// two floats -> one register of two bf16 (lo in the low half), round to nearest even: ONE v_cvt_pk_bf16_f32 -- the scalar form
// f2bf(lo) | f2bf(hi) << 16 compiles to two conversions, a shift and an or (round 6: four instructions per pair in every bf16 epilogue)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB_VOFF = 0x80000000u;   // >= num_records of every buffer descriptor below: the load returns zeros

// UT ("uniform tap"): Cin is a multiple of the k-tile, so a whole k-tile lies inside one filter tap and the tap,
// its input offset and the channel offset are wave-uniform (SGPRs).  The gather then costs three VALU
// instructions per 16-byte load: per-row tap-validity bit -> select the row's byte offset or an out-of-range one ->
// buffer_load_dwordx4 (hardware returns zeros for the padding halo and the ragged last tile; no branches).
// One LDS-DMA instruction written by hand: the compiler does not know it, so it neither counts it in its own
// s_waitcnt bookkeeping nor drains vmcnt before every ds_read that might alias the DMA's destination (which it does
// for the builtin) -- the deep pipeline below retires these loads itself with counted waits.  M0 (the LDS
// destination base) is compiler-reserved: saved, set and restored inside the statement.
__device__ __forceinline__ void lds_dma16(u32x4 rsrc, unsigned lds_byte, unsigned voff, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_byte), "s"(soff)
                 : "memory");
}

// ReLU under a run-time flag as ONE integer maximum (v_max_i32): a float is > 0 exactly when its bits, read as int32, are, so
// max(bits, 0) is relu(v) and max(bits, INT_MIN) is v; a positive NaN passes through either way (as in the reference's relu).
// (`keep = v > 0 ? ~0 : ~relu_mask; bits & keep` was a compare into an SGPR pair, a select and an AND per element.)
__device__ __forceinline__ int relu_floor_bits(int relu) { return relu ? 0 : (int)0x80000000; }
__device__ __forceinline__ float relu_floor(float v, int floor_bits) {
    return __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, v), floor_bits));      // (llvm.smax: written as a select the integer compare stays beside a float select)
}

// ---- epilogue shared by the kernels below: scale/bias (FrozenBN or Linear bias), residual, ReLU ----------------
// acc[i][j] is the wave's (i, j)-th 32x32 accumulator tile of its (BM / WM) x (BN / WN) sub-tile; `smem` (at least
// WM * WN * 32 * (BN / WN) floats, free after the K loop) is used to turn the sub-tile for wide stores.
template <int BM, int BN, int WM, int WN, int TM, int TN, bool BF16>
__device__ __forceinline__ void epilogue(const ConvGemmArgs& a, floatx16 (&acc)[TM][TN], float* smem, int smem_floats,
                                         int m0, int n0, int wm, int wn, int wave, int lane) {
    // ---- epilogue: scale/bias (FrozenBN or Linear bias), residual, ReLU ----------------------
    // bf16 mode: residual and output are bf16 unless out_f32 asks for an fp32 result (the prediction heads).
    const int half = lane >> 5;
    const bool res_f32 = !BF16;
    const bool out_f32 = !BF16 || a.out_f32;
    auto load_res1 = [&](size_t idx) -> float {
        return res_f32 ? reinterpret_cast<const float*>(a.res)[idx]
                       : bf2f(reinterpret_cast<const unsigned short*>(a.res)[idx]);
    };
    auto store1 = [&](size_t idx, float v) {
        if (out_f32) reinterpret_cast<float*>(a.C)[idx] = v;
        else reinterpret_cast<unsigned short*>(a.C)[idx] = f2bf(v);
    };
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;   // this wave's sub-tile
    if constexpr (BF16) {
        if (!a.out_f32 && (a.Cout & 7) == 0 && (a.ldc & 7) == 0) {
            // bf16 output: 8 channels per lane, so that residual loads and stores are 16 bytes per lane (the 8-byte form
            // of the path below moves bytes at 0.55-0.7 of the rate: the wide expand + residual layers are HBM-bound)
            float* Ep = smem + wave * (32 * WCOLS);
            constexpr int LPR = WCOLS / 8, RPI = 64 / LPR, NIT = 32 / RPI;
            const int c8e = (lane % LPR) * 8;
            const int n = n0 + wn * WCOLS + c8e;
            const bool n_ok = n < a.Cout;
            float sc[8], bi[8];
#pragma unroll
            for (int c = 0; c < 8; c++) { sc[c] = (n_ok && a.scale) ? a.scale[n + c] : 1.f; bi[c] = (n_ok && a.bias) ? a.bias[n + c] : 0.f; }
            const unsigned short* resp = reinterpret_cast<const unsigned short*>(a.res);
            unsigned short* outp = reinterpret_cast<unsigned short*>(a.C);
            // The residual rows of ALL the wave's row tiles are requested before the first one is used: the expand + residual
            // layers are bound by how many bytes a CU keeps in flight (one workgroup per CU: nothing else covers the epilogue's
            // HBM latency), and one row tile at a time was 32 KB per CU -- 13 GB/s per CU, 3.1 TB/s over the chip
            // (layer3's expand).  TM * NIT 16-byte registers per lane (64 at most) are free here: the K loop's fragments are dead.
            uint4 rv[TM][NIT];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int m = m0 + wm * WROWS + i * 32 + it * RPI + lane / LPR;
                    rv[i][it] = uint4{0u, 0u, 0u, 0u};
                    if (a.res && n_ok && m < a.M) rv[i][it] = *reinterpret_cast<const uint4*>(resp + (size_t)m * a.ldc + n);
                }
#pragma unroll
            for (int i = 0; i < TM; i++) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < TN; j++)
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        Ep[((r & 3) + 8 * (r >> 2) + 4 * half) * WCOLS + j * 32 + (lane & 31)] = acc[i][j][r];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int row = it * RPI + lane / LPR;
                    const int m = m0 + wm * WROWS + i * 32 + row;
                    const float4 v0 = *reinterpret_cast<const float4*>(&Ep[row * WCOLS + c8e]);
                    const float4 v1 = *reinterpret_cast<const float4*>(&Ep[row * WCOLS + c8e + 4]);
                    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    const unsigned rw[4] = {rv[i][it].x, rv[i][it].y, rv[i][it].z, rv[i][it].w};
                    unsigned ow[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        float lo = v[2 * c] * sc[2 * c] + bi[2 * c], hi = v[2 * c + 1] * sc[2 * c + 1] + bi[2 * c + 1];
                        lo += __builtin_bit_cast(float, rw[c] << 16);
                        hi += __builtin_bit_cast(float, rw[c] & 0xffff0000u);
                        if (a.relu) { lo = lo > 0.f ? lo : 0.f; hi = hi > 0.f ? hi : 0.f; }
                        ow[c] = pack_bf16x2(lo, hi);
                    }
                    if (n_ok && m < a.M) *reinterpret_cast<uint4*>(outp + (size_t)m * a.ldc + n) = uint4{ow[0], ow[1], ow[2], ow[3]};
                }
            }
            return;
        }
    }
    if ((a.Cout & 3) == 0 && (a.ldc & 3) == 0) {
        // Wide path: the accumulator layout (one column per lane) would give 4-byte accesses; the sub-tile is
        // turned through LDS (free after the K loop) so that every lane owns 4 consecutive channels of a row:
        // 16-byte (fp32) / 8-byte (bf16) residual loads and stores, 256 / 128 contiguous bytes per 16 lanes.
        // Unpadded rows: a wave writes 32 consecutive floats of one row per half (ds_write_b32: conflict-free), and
        // the 16-lane groups of ds_read_b128 ({0-3,12-15,20-27}, ...) land on 16 distinct 16-byte slots for both
        // 128-byte and 256-byte rows.
        constexpr int EP_LD = WCOLS;
        (void)smem_floats;
        float* Ep = smem + wave * (32 * EP_LD);     // one 32-row slab per wave, reused for each of its TM row tiles
        constexpr int LPR = WCOLS / 4;              // lanes per row
        constexpr int RPI = 64 / LPR;               // rows per wave-instruction
        constexpr int NIT = 32 / RPI;
        const int c4e = (lane % LPR) * 4;
        const int n = n0 + wn * WCOLS + c4e;
        const bool n_ok = n < a.Cout;               // Cout % 4 == 0: the four channels are in or out together
        float4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (n_ok && a.scale) sc = *reinterpret_cast<const float4*>(a.scale + n);
        if (n_ok && a.bias) bi = *reinterpret_cast<const float4*>(a.bias + n);
        // residual rows: row tile i + 1's are requested as soon as row tile i's accumulators have gone to LDS (their registers
        // are free then), i.e. before tile i is read back, combined and stored -- two row tiles of loads in flight per wave
        // instead of one (the expand + residual layers are bound by the bytes a CU keeps in flight)
        float4 rv[2][NIT];
        auto load_res = [&](int i, float4 (&dst)[NIT]) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int m = m0 + wm * WROWS + i * 32 + it * RPI + lane / LPR;
                dst[it] = float4{0.f, 0.f, 0.f, 0.f};
                if (a.res && n_ok && m < a.M) {
                    const size_t idx = (size_t)m * a.ldc + n;
                    if (res_f32) {
                        dst[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.res) + idx);
                    } else {
                        const ushort4 u = *reinterpret_cast<const ushort4*>(reinterpret_cast<const unsigned short*>(a.res) + idx);
                        dst[it] = float4{bf2f(u.x), bf2f(u.y), bf2f(u.z), bf2f(u.w)};
                    }
                }
            }
        };
        load_res(0, rv[0]);
#pragma unroll
        for (int i = 0; i < TM; i++) {
            // The slab is private to this wavefront and a wavefront's LDS operations execute in order, so the
            // transposition needs no workgroup barrier (the K loop ended with one): waves store as soon as they are done.
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < TN; j++)
#pragma unroll
                for (int r = 0; r < 16; r++)
                    Ep[((r & 3) + 8 * (r >> 2) + 4 * half) * EP_LD + j * 32 + (lane & 31)] = acc[i][j][r];
            if (i + 1 < TM) load_res(i + 1, rv[(i + 1) & 1]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                const int row = it * RPI + lane / LPR;
                const int m = m0 + wm * WROWS + i * 32 + row;
                const float4 v = *reinterpret_cast<const float4*>(&Ep[row * EP_LD + c4e]);
                const float4 rr = rv[i & 1][it];
                float4 o;
                o.x = v.x * sc.x + bi.x; o.y = v.y * sc.y + bi.y; o.z = v.z * sc.z + bi.z; o.w = v.w * sc.w + bi.w;
                o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                if (a.relu) {
                    o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
                    o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
                }
                if (n_ok && m < a.M && (!(CG_ABLATE & 8) || o.x == 12345.678f)) {   // 8: (almost) never store
                    const size_t idx = (size_t)m * a.ldc + n;
                    if (out_f32) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + idx) = o;
                    } else {
                        const uint2 u = uint2{pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w)};
                        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.C) + idx) = u;
                    }
                }
            }
        }
        return;
    }
    // Narrow path (channel counts that are not a multiple of 4: the prediction heads): residual values of a
    // whole 32x32 tile are fetched first, then combined and stored.
#pragma unroll
    for (int j = 0; j < TN; j++) {
        const int n = n0 + wn * (BN / WN) + j * 32 + (lane & 31);
        const bool n_ok = n < a.Cout;
        const float sc = (a.scale && n_ok) ? a.scale[n] : 1.0f;
        const float bi = (a.bias && n_ok) ? a.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; i++) {
            const int mb = m0 + wm * (BM / WM) + i * 32 + 4 * half;
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                rv[r] = (a.res && n_ok && m < a.M) ? load_res1((size_t)m * a.ldc + n) : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                float v = acc[i][j][r] * sc + bi;
                v += rv[r];
                if (a.relu) v = v > 0.0f ? v : 0.0f;
                if (n_ok && m < a.M) store1((size_t)m * a.ldc + n, v);
            }
        }
    }
}

// ---- the same epilogue without branches (round 5) ----------------------------------------------------------------------------
// epilogue() above guards every residual load and every store with `if (m < M && n_ok)`: each sits in its own exec-masked block,
// and the compiler's wait insertion does not count across those -- it waits vmcnt(0) at the first use of a residual row, i.e. for
// every load and store issued so far, the next row tile's residual included (the .s of the round-4 kernels).  Here residual and
// output go through buffer descriptors over the TILE (wave-uniform base, per-lane byte offsets; a lane past M or past Cout gets an
// out-of-range offset: the load returns zeros, the store is dropped), so the code is straight-line, the waits are counted, the
// residual rows of row tile i + 1 and the stores of row tile i - 1 stay in flight under row tile i, and the first two row tiles'
// residual rows are requested BEFORE the workgroup barrier that frees the staging buffers.  16 bytes per lane either way: four
// fp32 channels or eight bf16 channels.  Same operations in the same order as epilogue(): bit-identical results.
template <int BM, int BN, int WM, int WN, bool BF16>
__device__ __forceinline__ bool epilogue2_ok(const ConvGemmArgs& a) {
    constexpr int WCOLS = BN / WN, CPL = BF16 ? 8 : 4, LPR = WCOLS / CPL;
    constexpr bool shape_ok = LPR >= 1 && LPR <= 32 && (64 % LPR) == 0 && 32 / (64 / LPR) <= 8;      // at most 2 x 8 residual registers-of-16-bytes per lane
    if (!shape_ok) return false;
    if (BF16) return !a.out_f32 && (a.Cout & 7) == 0 && (a.ldc & 7) == 0;
    return (a.Cout & 3) == 0 && (a.ldc & 3) == 0;
}
template <int BM, int BN, int WM, int WN, int TM, int TN, bool BF16>
__device__ __forceinline__ void epilogue2(const ConvGemmArgs& a, floatx16 (&acc)[TM][TN], float* smem, int m0, int n0, int wm, int wn,
                                          int wave, int lane) {
    constexpr int WROWS = BM / WM, WCOLS = BN / WN;
    constexpr int CPL = BF16 ? 8 : 4, ESZ = BF16 ? 2 : 4;
    constexpr int LPR = WCOLS / CPL, RPI = 64 / LPR, NIT = (32 / RPI) > 0 ? 32 / RPI : 1;
    // (opaque copies: nothing of the epilogue's lane arithmetic is to be computed ahead of the K loop and carried through it)
    asm volatile("" : "+v"(lane), "+v"(wave), "+v"(wm), "+v"(wn));
    const int half = lane >> 5;
    float* Ep = smem + wave * (32 * WCOLS);
    const int c0 = (lane % LPR) * CPL;
    const int n = n0 + wn * WCOLS + c0;
    const bool n_ok = n < a.Cout;
    const int nc = n_ok ? n : 0;
    float sc[CPL], bi[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c += 4) {
        const float4 s4 = a.scale ? *reinterpret_cast<const float4*>(a.scale + nc + c) : float4{1.f, 1.f, 1.f, 1.f};
        const float4 b4 = a.bias ? *reinterpret_cast<const float4*>(a.bias + nc + c) : float4{0.f, 0.f, 0.f, 0.f};
        sc[c] = n_ok ? s4.x : 1.f; sc[c + 1] = n_ok ? s4.y : 1.f; sc[c + 2] = n_ok ? s4.z : 1.f; sc[c + 3] = n_ok ? s4.w : 1.f;
        bi[c] = n_ok ? b4.x : 0.f; bi[c + 1] = n_ok ? b4.y : 0.f; bi[c + 2] = n_ok ? b4.z : 0.f; bi[c + 3] = n_ok ? b4.w : 0.f;
    }
    const size_t tile_off = ((size_t)m0 * a.ldc + n0) * ESZ;
    const __amdgpu_buffer_rsrc_t res_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.res ? a.res : a.C)) + tile_off, 0, a.res ? (int)OOB_VOFF : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t out_rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.C) + tile_off, 0, (int)OOB_VOFF, 0x00020000);
    const int rows_left = a.M - m0;
    const int row0 = wm * WROWS + lane / LPR;
    const unsigned col_b = (unsigned)(wn * WCOLS + c0) * (unsigned)ESZ;
    auto voff = [&](int i, int it) -> unsigned {
        const int r = row0 + i * 32 + it * RPI;
        const unsigned ok = (unsigned)n_ok & (unsigned)(r < rows_left);
        return ok ? (unsigned)r * (unsigned)a.ldc * (unsigned)ESZ + col_b : OOB_VOFF;
    };
    u32x4 rv[2][NIT];
    auto load_res = [&](int i, u32x4 (&dst)[NIT]) {
#pragma unroll
        for (int it = 0; it < NIT; it++) dst[it] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, voff(i, it), 0, 0);
    };
    load_res(0, rv[0]);
    if (TM > 1) load_res(1, rv[1]);
    __syncthreads();      // every wave is done with the staging buffers: they become the slabs
    const int relu_fb = relu_floor_bits(a.relu);
    auto act = [&](float v) { return relu_floor(v, relu_fb); };      // relu ? (v > 0 ? v : 0) : v, without a branch
#pragma unroll
    for (int i = 0; i < TM; i++) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) Ep[((r & 3) + 8 * (r >> 2) + 4 * half) * WCOLS + j * 32 + (lane & 31)] = acc[i][j][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int row = it * RPI + lane / LPR;
            const u32x4 rw = rv[i & 1][it];
            if constexpr (BF16) {
                const float4 v0 = *reinterpret_cast<const float4*>(&Ep[row * WCOLS + c0]);
                const float4 v1 = *reinterpret_cast<const float4*>(&Ep[row * WCOLS + c0 + 4]);
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                u32x4 ow;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float lo = v[2 * c] * sc[2 * c] + bi[2 * c], hi = v[2 * c + 1] * sc[2 * c + 1] + bi[2 * c + 1];
                    lo += __builtin_bit_cast(float, rw[c] << 16);
                    hi += __builtin_bit_cast(float, rw[c] & 0xffff0000u);
                    lo = act(lo); hi = act(hi);
                    ow[c] = pack_bf16x2(lo, hi);
                }
                __builtin_amdgcn_raw_buffer_store_b128(ow, out_rs, voff(i, it), 0, 0);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(&Ep[row * WCOLS + c0]);
                const float4 rr = __builtin_bit_cast(float4, rw);
                float4 o;
                o.x = v.x * sc[0] + bi[0]; o.y = v.y * sc[1] + bi[1]; o.z = v.z * sc[2] + bi[2]; o.w = v.w * sc[3] + bi[3];
                o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                o.x = act(o.x); o.y = act(o.y); o.z = act(o.z); o.w = act(o.w);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rs, voff(i, it), 0, 0);
            }
        }
        if (i + 2 < TM) load_res(i + 2, rv[i & 1]);
    }
}

// The wave's 32-row accumulator tile as (row, column, value) triples, 64 columns at a time, for either matrix-instruction
// shape: 32x32 blocks (lane: column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)) or 16x16 blocks (lane: column l & 15,
// rows 4 (l >> 4) + r).  f(row 0..31, column 0..63 inside the group, column block j, value): the lane's channel inside
// block j is the same for every row, so a tail fetches its scale / bias once per block (cols64_affine) -- written as
// `a.scale ? a.scale[n] : 1` inside f they were re-fetched for EVERY element, each in its own branch with its own wait
// (the slab stores in between may alias them as far as the compiler knows).
template <int TN, class F>
__device__ __forceinline__ void acc_cols64(floatx16 (&acc)[1][TN], int h2, int lane, F&& f) {
    const int half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) f((r & 3) + 8 * (r >> 2) + 4 * half, j * 32 + l32, j, acc[0][2 * h2 + j][r]);
}
template <int NJ, class F>
__device__ __forceinline__ void acc_cols64(floatx4 (&acc)[2][NJ], int h2, int lane, F&& f) {
    const int kq = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) f(mi * 16 + 4 * kq + r, j * 16 + r16, j, acc[mi][4 * h2 + j][r]);
}

// Scale / bias / ReLU of a tail without a branch.  A missing scale or bias vector is a descriptor of zero records (every
// load returns zero bits; the scale's are replaced by 1.0f under a mask), ReLU is a mask on the value's bits: the loads are
// plain buffer instructions the wait counts of a tail can count, and a group of the tail is ONE basic block.
struct TailAffine {
    __amdgpu_buffer_rsrc_t sc_rs, bi_rs;
    unsigned have_sc;
    int relu_fb;
    __device__ __forceinline__ TailAffine(const float* scale, const float* bias, int relu, int n_ch)
        : sc_rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(scale), 0, scale ? n_ch * 4 : 0, 0x00020000)),
          bi_rs(__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? n_ch * 4 : 0, 0x00020000)),
          have_sc(scale ? 0xffffffffu : 0u), relu_fb(relu_floor_bits(relu)) {}
    __device__ __forceinline__ float one_or(unsigned w) const { return __builtin_bit_cast(float, (w & have_sc) | (0x3f800000u & ~have_sc)); }
    __device__ __forceinline__ float scale1(int n) const { return one_or(__builtin_amdgcn_raw_buffer_load_b32(sc_rs, n * 4, 0, 0)); }
    __device__ __forceinline__ float bias1(int n) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bi_rs, n * 4, 0, 0)); }
    __device__ __forceinline__ float4 scale4(int n) const {
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(sc_rs, n * 4, 0, 0);
        return float4{one_or(w.x), one_or(w.y), one_or(w.z), one_or(w.w)};
    }
    __device__ __forceinline__ float4 bias4(int n) const { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(bi_rs, n * 4, 0, 0)); }
    __device__ __forceinline__ float act(float v) const { return relu_floor(v, relu_fb); }      // relu ? (v > 0 ? v : 0) : v
};
// the lane's scale / bias of column block j of channels 64 h2 .. + 63, for either accumulator layout (see acc_cols64)
template <int TN>
__device__ __forceinline__ void cols64_affine(floatx16 (&)[1][TN], const TailAffine& t, int h2, int lane, float (&sc)[4], float (&bi)[4]) {
#pragma unroll
    for (int j = 0; j < 2; j++) { sc[j] = t.scale1(h2 * 64 + j * 32 + (lane & 31)); bi[j] = t.bias1(h2 * 64 + j * 32 + (lane & 31)); }
}
template <int NJ>
__device__ __forceinline__ void cols64_affine(floatx4 (&)[2][NJ], const TailAffine& t, int h2, int lane, float (&sc)[4], float (&bi)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++) { sc[j] = t.scale1(h2 * 64 + j * 16 + (lane & 15)); bi[j] = t.bias1(h2 * 64 + j * 16 + (lane & 15)); }
}

__device__ __forceinline__ void cg_split8(const float4& c0, const float4& c1, uint4& hi, uint4& mid, uint4& lo) {
    const float v[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    unsigned uh[8], um[8], ul[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
        uh[e] = __builtin_bit_cast(unsigned, v[e]);
        const float r1 = v[e] - __builtin_bit_cast(float, uh[e] & 0xffff0000u);
        um[e] = __builtin_bit_cast(unsigned, r1);
        ul[e] = __builtin_bit_cast(unsigned, r1 - __builtin_bit_cast(float, um[e] & 0xffff0000u));
    }
    auto pk = [](unsigned lo_, unsigned hi_) { return __builtin_amdgcn_perm(hi_, lo_, 0x07060302u); };
    hi = uint4{pk(uh[0], uh[1]), pk(uh[2], uh[3]), pk(uh[4], uh[5]), pk(uh[6], uh[7])};
    mid = uint4{pk(um[0], um[1]), pk(um[2], um[3]), pk(um[4], um[5]), pk(um[6], um[7])};
    lo = uint4{pk(ul[0], ul[1]), pk(ul[2], ul[3]), pk(ul[4], ul[5]), pk(ul[6], ul[7])};
}

constexpr int FUSE_W3_BYTES = 4 * 3 * 256 * 32, FUSE_LDS_BYTES = FUSE_W3_BYTES + 8 * 32 * 64 * 4;    // 96 KB + 64 KB = all of it

}  // namespace odam_cg
