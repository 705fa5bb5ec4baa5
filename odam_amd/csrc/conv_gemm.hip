// conv_gemm.hip -- see conv_gemm.h
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
#if CG_STAMP
__device__ unsigned long long g_stamps[8];
#define CG_STAMP_AT(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                            st_acc[i] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CG_STAMP_AT(i) do { } while (0)
#endif
#if CG_STAMP      // phase stamps of the layer1 bottleneck tail (fused_chain), read by tests/native/tail_stamps.py through odam_cg_tail_stamps
__device__ unsigned long long g_tail_stamps[8];
#define CG_TSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
                          ts_acc[i] += t_ - ts_prev; ts_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CG_TSTAMP(i) do { } while (0)
#endif
#ifndef CG_ABLATE
#define CG_ABLATE 0   // timing experiments only (tests/native/perf_conv_ablate.py): 1 no global loads, 2 no LDS stores, 4 no barrier, 8 no output stores, 16 no first-tile load
#endif

__device__ __forceinline__ float bf2f(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

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
                        ow[c] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
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
                        ushort4 u = {f2bf(o.x), f2bf(o.y), f2bf(o.z), f2bf(o.w)};
                        *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(a.C) + idx) = u;
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

// NST: LDS stages.  2 = one tile in flight under the tile being multiplied (two workgroups per CU hide the rest).
// 4 (UT only; small, latency-bound problems that leave most CUs with one workgroup): three tiles in flight, retired
// with a counted s_waitcnt so the DMA queue stays full across the per-tile barrier.
template <int BM, int BN, int WM, int WN, bool BF16, bool UT, int NST = 2>
__global__ __launch_bounds__(64 * WM * WN) void conv_gemm_kernel(ConvGemmArgs a) {
    static_assert(NST == 2 || (UT && (NST & (NST - 1)) == 0), "deep pipelines are LDS-DMA only");
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
    epilogue<BM, BN, WM, WN, TM, TN, BF16>(a, acc, smem, (int)sizeof(smem) / 4, m0, n0, wm, wn, wave, lane);
}


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
// The wave's 32-row accumulator tile as (row, column, value) triples, 64 columns at a time, for either matrix-instruction
// shape: 32x32 blocks (lane: column l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)) or 16x16 blocks (lane: column l & 15,
// rows 4 (l >> 4) + r).  f(row 0..31, column 0..63 inside the group, channel = 64 h2 + column, value).
template <int TN, class F>
__device__ __forceinline__ void acc_cols64(floatx16 (&acc)[1][TN], int h2, int lane, F&& f) {
    const int half = lane >> 5, l32 = lane & 31;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) f((r & 3) + 8 * (r >> 2) + 4 * half, j * 32 + l32, h2 * 64 + j * 32 + l32, acc[0][2 * h2 + j][r]);
}
template <int NJ, class F>
__device__ __forceinline__ void acc_cols64(floatx4 (&acc)[2][NJ], int h2, int lane, F&& f) {
    const int kq = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) f(mi * 16 + 4 * kq + r, j * 16 + r16, h2 * 64 + j * 16 + r16, acc[mi][4 * h2 + j][r]);
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

template <class ACC>
__device__ __forceinline__ void fused_second(const ConvGemmArgs& a, ACC& acc, float* smem, unsigned lds_base,
                                             int m0, int wave, int lane) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const int half = lane >> 5, l32 = lane & 31;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
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
    acc_cols64(acc, 0, lane, [&](int row, int c64, int n, float x) {
        float v = x * (a.scale ? a.scale[n] : 1.0f) + (a.bias ? a.bias[n] : 0.0f);
        if (a.relu) v = v > 0.f ? v : 0.f;
        slab[slab_at(row, c64)] = v;
    });
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
        const int n = g * 64 + c4 * 4;
        float4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (a.F_scale) sc = *reinterpret_cast<const float4*>(a.F_scale + n);
        if (a.F_bias) bi = *reinterpret_cast<const float4*>(a.F_bias + n);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            const float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            float4 o = {v.x * sc.x + bi.x + rv[g & 1][it].x, v.y * sc.y + bi.y + rv[g & 1][it].y, v.z * sc.z + bi.z + rv[g & 1][it].z,
                        v.w * sc.w + bi.w + rv[g & 1][it].w};
            if (a.F_relu) { o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f; }
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
    acc_cols64(acc, 0, lane, [&](int row, int c64, int n, float x) {
        float v = x * (a.scale ? a.scale[n] : 1.0f) + (a.bias ? a.bias[n] : 0.0f);
        if (a.relu) v = v > 0.f ? v : 0.f;
        slab[slab_at(row, c64)] = v;
    });
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
        const int n = g * 64 + c4 * 4;
        float4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (a.F_scale) sc = *reinterpret_cast<const float4*>(a.F_scale + n);
        if (a.F_bias) bi = *reinterpret_cast<const float4*>(a.F_bias + n);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            float4* sp_ = reinterpret_cast<float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            const float4 v = *sp_;
            float4 o = {v.x * sc.x + bi.x + rv[it].x, v.y * sc.y + bi.y + rv[it].y, v.z * sc.z + bi.z + rv[it].z,
                        v.w * sc.w + bi.w + rv[it].w};
            if (a.F_relu) { o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f; }
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
    float4 rv[8];
    auto load_res = [&](int g) {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int m = m_base + it * 4 + (lane >> 4);
            rv[it] = float4{0.f, 0.f, 0.f, 0.f};
            if (a.F_res && m < a.M) rv[it] = *reinterpret_cast<const float4*>(a.F_res + (size_t)m * a.F_ldc + g * 64 + c4 * 4);
        }
    };
    dma_expand(0);
    float* slab = smem + wave * (32 * 64);
    auto slab_at = [](int row, int col) { return row * 64 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3)); };
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    acc_cols64(acc, 0, lane, [&](int row, int c64, int n, float x) {
        float v = x * (a.scale ? a.scale[n] : 1.0f) + (a.bias ? a.bias[n] : 0.0f);
        if (a.relu) v = v > 0.f ? v : 0.f;
        slab[slab_at(row, c64)] = v;
    });
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
#pragma unroll 1
    for (int g = 0; g < 4; g++) {
        asm volatile("" : "+v"(half), "+v"(l32));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // group g's expand filters are in; everybody is done with group g - 1
        if (g + 1 < 4) dma_expand(g + 1);
        dma_reduce(g);
        load_res(g);                                            // this group's residual rows: under the expand's matrix instructions
        const int b_off = l32 * 32 + ((half ^ ((l32 >> 3) & 1)) * 16);
        const char* wf = lds_c + SLABS + (g & 1) * GB;
        const char* wg = lds_c + SLABS + 2 * GB;
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
        const int n = g * 64 + c4 * 4;
        float4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (a.F_scale) sc = *reinterpret_cast<const float4*>(a.F_scale + n);
        if (a.F_bias) bi = *reinterpret_cast<const float4*>(a.F_bias + n);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            float4* sp_ = reinterpret_cast<float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            const float4 v = *sp_;
            float4 o = {v.x * sc.x + bi.x + rv[it].x, v.y * sc.y + bi.y + rv[it].y, v.z * sc.z + bi.z + rv[it].z,
                        v.w * sc.w + bi.w + rv[it].w};
            if (a.F_relu) { o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f; }
            if (m < a.M) *reinterpret_cast<float4*>(a.F_C + (size_t)m * a.F_ldc + n) = o;
            *sp_ = o;
        }
        wave_sync();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the reduce slice of this group (requested at the top) is in
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
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int n = h2 * 64 + j * 32 + l32;
            const float sc = a.G_scale ? a.G_scale[n] : 1.0f, bi = a.G_bias ? a.G_bias[n] : 0.0f;
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
        acc_cols64(acc, h2, lane, [&](int row, int c64, int n, float x) {
            float v = x * (a.scale ? a.scale[n] : 1.0f) + (a.bias ? a.bias[n] : 0.0f);
            if (a.relu) v = v > 0.f ? v : 0.f;
            slab[slab_at(row, c64)] = v;
        });
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
        if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __syncthreads();
        if (g + 1 < NG) dma_group(g + 1);
        const char* wf = lds_c + SLABS + (g & 1) * GB;
        floatx16 acc2[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[jb][r] = 0.0f;
#pragma unroll
            for (int s_ = 0; s_ < KS; s_++) {
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
        const int n = g * 64 + c4 * 4;
        float4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (a.F_scale) sc = *reinterpret_cast<const float4*>(a.F_scale + n);
        if (a.F_bias) bi = *reinterpret_cast<const float4*>(a.F_bias + n);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int row = it * 4 + (lane >> 4);
            const int m = m_base + row;
            const float4 v = *reinterpret_cast<const float4*>(&slab[row * 64 + ((c4 ^ (row & 15)) << 2)]);
            float4 o = {v.x * sc.x + bi.x + rv[it].x, v.y * sc.y + bi.y + rv[it].y, v.z * sc.z + bi.z + rv[it].z,
                        v.w * sc.w + bi.w + rv[it].w};
            if (a.F_relu) { o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f; o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f; }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), out_rs, row_off[it] == OOB_VOFF ? OOB_VOFF : row_off[it] + (unsigned)(g * 256), 0, 0);
        }
        if (g + 1 < NG) load_res(g + 1);
        __builtin_amdgcn_wave_barrier();
    }
#undef ODAM_MM2
}

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
// vmcnt(8) -- the DMAs of this group were issued one group ago and exactly 4 residual loads + 4 stores (buffer instructions
// with an out-of-range offset for rows past M: issued by every wave, never branched around) are younger.  More compiler-visible
// loads in between only make that wait stricter, never laxer.
template <int P, int PN>
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
        row_off[it] = m < a.M ? (unsigned)((size_t)m * a.F_ldc * 2 + c8 * 16) : OOB_VOFF;
    }
    u32x4 rv[2][4];
    auto load_res = [&](int g) {
#pragma unroll
        for (int it = 0; it < 4; it++)
            rv[g & 1][it] = __builtin_amdgcn_raw_buffer_load_b128(res_rs, row_off[it] == OOB_VOFF ? OOB_VOFF : row_off[it] + (unsigned)(g * 128), 0, 0);
    };
    load_res(0);
    // (1) this layer's epilogue -> mid
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = wn * (P / WN) + j * 32 + l32;
            const float sc = a.scale ? a.scale[n] : 1.0f, bi = a.bias ? a.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = wm * (256 / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[i][j][r] * sc + bi;
                if (a.relu) v = v > 0.f ? v : 0.f;
                *reinterpret_cast<unsigned short*>(lds + row * PITCH + (swz(row, n >> 3) << 4) + (n & 7) * 2) = f2bf(v);
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
    const u32x4 rsf = u32x4{(unsigned)pf, (unsigned)(pf >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const u32x4 rsg = u32x4{(unsigned)pg, (unsigned)(pg >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
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
    floatx16 acc3[PN ? PN / 32 : 1];
#pragma unroll
    for (int jc = 0; jc < (PN ? PN / 32 : 1); jc++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc3[jc][r] = 0.0f;
#define ODAM_MMB(accv, x, y) accv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accv, 0, 0, 0)
#pragma unroll
    for (int g = 0; g < NG; g++) {      // unrolled: the residual buffers rv[g & 1] must be compile-time register names
        if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // my pieces of group g's filters have landed (see the header)
        __syncthreads();                                             // ... and everybody's; everybody is done with group g - 1's buffer
        if (g + 1 < NG) dma_group(g + 1);
        load_res(g + 1 < NG ? g + 1 : g);                            // (the last group repeats its own: the instruction count stays)
        const char* wf = lds + WB + (g & 1) * GB;
        floatx16 acc2[2];
#pragma unroll
        for (int jb = 0; jb < 2; jb++) {
#pragma unroll
            for (int r = 0; r < 16; r++) acc2[jb][r] = 0.0f;
            const int nrow = jb * 32 + l32;
#pragma unroll
            for (int s_ = 0; s_ < P / 16; s_++) {
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
        const int n = g * 64 + c8 * 8;
        float sc[8], bi[8];
#pragma unroll
        for (int c = 0; c < 8; c++) { sc[c] = a.F_scale ? a.F_scale[n + c] : 1.0f; bi[c] = a.F_bias ? a.F_bias[n + c] : 0.0f; }
#pragma unroll
        for (int it = 0; it < 4; it++) {
            const int row = it * 8 + (lane >> 3);
            const float4 v0 = *reinterpret_cast<const float4*>(&slab[row * 64 + (((2 * c8) ^ (row & 15)) << 2)]);
            const float4 v1 = *reinterpret_cast<const float4*>(&slab[row * 64 + (((2 * c8 + 1) ^ (row & 15)) << 2)]);
            const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const u32x4 rr = rv[g & 1][it];
            const unsigned rw[4] = {rr.x, rr.y, rr.z, rr.w};
            unsigned ow[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float lo = v[2 * c] * sc[2 * c] + bi[2 * c], hi = v[2 * c + 1] * sc[2 * c + 1] + bi[2 * c + 1];
                lo += __builtin_bit_cast(float, rw[c] << 16);
                hi += __builtin_bit_cast(float, rw[c] & 0xffff0000u);
                if (a.F_relu) { lo = lo > 0.f ? lo : 0.f; hi = hi > 0.f ? hi : 0.f; }
                ow[c] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
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
                for (int s_ = 0; s_ < 4; s_++) {
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
                const u32x4 o = u32x4{(unsigned)f2bf(v0.x) | ((unsigned)f2bf(v0.y) << 16), (unsigned)f2bf(v0.z) | ((unsigned)f2bf(v0.w) << 16),
                                      (unsigned)f2bf(v1.x) | ((unsigned)f2bf(v1.y) << 16), (unsigned)f2bf(v1.z) | ((unsigned)f2bf(v1.w) << 16)};
                __builtin_amdgcn_raw_buffer_store_b128(o, g_rs, m < a.M ? (unsigned)((size_t)m * PN * 2 + h * 128 + c8 * 16) : OOB_VOFF, 0, 0);
            }
        }
    }
}

// Tile rows and wave grid of the ring kernel.  Eight waves on 256 rows is the general shape; the 64-column loops need ~100
// registers, so sixteen waves on 512 x 64 put four waves on a SIMD (conv1: 1.44 -> 1.15 ms fp32, 0.95 -> 0.74 bf16).
// (The 256-column split loop with 32-row wave tiles was built both as twelve waves on 192 x 256 -- 140 registers, three per
// SIMD -- and as sixteen on 256 x 256 -- 126 registers, four per SIMD: the first loses to tile rounds what it gains (layer4's
// 3x3: 284 tiles = two rounds, 583 -> 822 us; expand layers +3-8 %), the second measured the same as eight waves (forward
// 29.76 vs 29.83 ms): that loop is paced by the matrix pipe and its barrier, not by waves waiting to issue.  Not kept.)
constexpr int big_bm(int BN, int NTH) { return (NTH == 1024 && BN == 64) ? 512 : 256; }
constexpr int big_wm(int MODE, int BN, int NTH) {
    return NTH == 1024 ? (BN == 64 ? 16 : 4)      // bf16, 256 columns: sixteen waves of 64 x 64
         : MODE >= 3 ? (BN == 256 ? 4 : 8) : (BN == 256 ? 2 : (BN == 128 ? 4 : 8));
}
// dynamic LDS of the ring kernel: the ring, or the epilogue's 32-row slab per wave if that is larger
template <int MODE, int BN, int NST, int NTH>
constexpr int big_lds_bytes() {
    const int bm = big_bm(BN, NTH);
    const int ring = NST * (MODE >= 3 ? bm * 64 + 96 * BN : (bm + BN) * 64);
    const int wm = big_wm(MODE, BN, NTH);
    const int wn = NTH / 64 / wm;
    const int slab = MODE == 4 ? (NTH / 64) * 16 * (BN / wn + 4) * 4 : (NTH / 64) * 32 * (BN / wn) * 4;      // the 16x16x32 loop's epilogue: 16-row slabs, rows padded by 4
    const int s1 = (MODE == 0 && NTH == 512) ? (BN == 256 ? 2 : (BN == 128 ? 3 : 4)) * (384 * 64 + 3 * BN * 64) : 0;   // the 3x3 stride-1 window ring (bf16)
    const int m = ring > slab ? ring : slab;
    return m > s1 ? m : s1;
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
    {
        const int nwg = gridDim.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tile_m = bid / tiles_n, tile_n = bid - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- gather bookkeeping (uniform-tap path of the kernel above) -------------------------------------------------
    const int c4 = tid & 3, r0 = tid >> 2;
    const int hw = a.Ho * a.Wo;
    const int lda = a.lda > 0 ? a.lda : a.Cin;
    const bool plain = a.KH * a.KW == 1 && a.stride == 1 && a.pad == 0;
    const int ntaps = a.KH * a.KW;
    const int kw_inv = (65536 + a.KW - 1) / a.KW;
    const int nt_inv = (65536 + ntaps - 1) / ntaps;
    const int b0 = m0 / hw;
    const int csrc = X16 ? (c4 ^ ((lane >> 4) & 1)) : (c4 ^ ((lane >> 4) & 3));     // 16-row fragments: rows r, r + 4 differ in the key (see X16)
    const int margin = (a.pad * a.W + a.pad) * lda;
    const unsigned long long pa = (unsigned long long)(Ag + (size_t)b0 * a.H * a.W * lda - margin), pw = (unsigned long long)Wg;
    const u32x4 a_rs4 = u32x4{(unsigned)pa, (unsigned)(pa >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    const u32x4 w_rs4 = u32x4{(unsigned)pw, (unsigned)(pw >> 32) & 0xffffu, OOB_VOFF, 0x00020000u};
    unsigned a_voff[A_LD], a_mask[A_LD], w_voff[B_LD];
#pragma unroll
    for (int i = 0; i < A_LD; i++) {
        const int m = m0 + r0 + RPP * i;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        int base, iy0 = 0, ix0 = 0;
        if (plain) {
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
#define ODAM_MM16(accv, x, y) accv = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), accv, 0, 0, 0)
#if CG_STAMP
        unsigned long long st_acc[4] = {0, 0, 0, 0}, st_prev = __builtin_readcyclecounter();
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
            read_a(0);
#pragma unroll
            for (int j = 0; j < BD; j++) read_b(j % NJ, B2[j]);
            half_split(ar0, A2[0], 0); half_split(ar1, A2[0], 1);
            int dpart = 0;
#pragma unroll
            for (int b = 0; b < NBLK; b++) {
                const int mi = b / NJ, nj = b % NJ;
                const Ops& oa = A2[mi & 1];
                const Ops& ob = B2[b % (BD + 1)];
                // smallest products first, as MODE 3
                ODAM_MM16(c16[mi][nj], oa.l, ob.h); ODAM_MM16(c16[mi][nj], oa.m, ob.m); ODAM_MM16(c16[mi][nj], oa.h, ob.l);
                ODAM_MM16(c16[mi][nj], oa.m, ob.h); ODAM_MM16(c16[mi][nj], oa.h, ob.m);
                ODAM_MM16(c16[mi][nj], oa.h, ob.h);
                if (b + BD < NBLK) read_b((b + BD) % NJ, B2[(b + BD) % (BD + 1)]);
                if (mi + 1 < MI) {
                    if (nj == 1) read_a(mi + 1);
                    if (nj == 3) half_split(ar0, A2[(mi + 1) & 1], 0);
                    if (nj == (NJ > 5 ? 5 : 3)) half_split(ar1, A2[(mi + 1) & 1], 1);
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
                __builtin_amdgcn_sched_barrier(0);      // (without the fence: +2 % time; static priority for waves 4-7: no change)
                if (b == NBLK / 2 - 1) CG_STAMP_AT(2);
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
#undef ODAM_MM16
        __syncthreads();
        if constexpr (FUSE == 1) { fused_second(a, c16, smem, lds_base, m0, wave, lane); return; }
        if constexpr (FUSE == 2) { fused_chain(a, c16, smem, lds_base, m0, wave, lane); return; }
        if constexpr (FUSE == 3) { fused_second128(a, c16, smem, lds_base, m0, wave, lane); return; }
        if constexpr (FUSE == 4) { fused_chain128(a, c16, smem, lds_base, m0, wave, lane); return; }
        // epilogue of the 16x16 accumulators (lane: column l & 15, rows 4 (l >> 4) + r): scale / bias (+ residual) (ReLU), fp32
        // out; a row tile of 16 rows goes through a per-wave LDS slab (rows padded by 4 floats: the four row groups of a
        // ds_write_b32 land on different banks) and leaves as 16-byte stores
        {
            constexpr int WROWS = BM / WM, WCOLS = BN / WN, EP_LD = WCOLS + 4;
            static_assert(NW * 16 * EP_LD * 4 <= NST * STAGE, "epilogue slabs fit the ring");
            float* Ep = smem + wave * (16 * EP_LD);
            constexpr int LPR = WCOLS / 4, RPI = 64 / LPR, NIT = 16 / RPI;
            const int c4e = (lane % LPR) * 4;
            const int n = n0 + wn * WCOLS + c4e;
            const bool n_ok = n < a.Cout;
            float4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
            if (n_ok && a.scale) sc = *reinterpret_cast<const float4*>(a.scale + n);
            if (n_ok && a.bias) bi = *reinterpret_cast<const float4*>(a.bias + n);
            const float* resp = reinterpret_cast<const float*>(a.res);
            float* outp = reinterpret_cast<float*>(a.C);
            float4 rv[2][NIT];
            auto load_res = [&](int mi, float4 (&dst)[NIT]) {
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int m = m0 + wm * WROWS + mi * 16 + it * RPI + lane / LPR;
                    dst[it] = float4{0.f, 0.f, 0.f, 0.f};
                    if (resp && n_ok && m < a.M) dst[it] = *reinterpret_cast<const float4*>(resp + (size_t)m * a.ldc + n);
                }
            };
            load_res(0, rv[0]);
#pragma unroll
            for (int mi = 0; mi < MI; mi++) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < NJ; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Ep[(4 * kq + r) * EP_LD + j * 16 + r16] = c16[mi][j][r];
                if (mi + 1 < MI) load_res(mi + 1, rv[(mi + 1) & 1]);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int it = 0; it < NIT; it++) {
                    const int row = it * RPI + lane / LPR;
                    const int m = m0 + wm * WROWS + mi * 16 + row;
                    const float4 v = *reinterpret_cast<const float4*>(&Ep[row * EP_LD + c4e]);
                    const float4 rr = rv[mi & 1][it];
                    float4 o;
                    o.x = v.x * sc.x + bi.x; o.y = v.y * sc.y + bi.y; o.z = v.z * sc.z + bi.z; o.w = v.w * sc.w + bi.w;
                    o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
                    if (a.relu) {
                        o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
                        o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
                    }
                    if (n_ok && m < a.M) *reinterpret_cast<float4*>(outp + (size_t)m * a.ldc + n) = o;
                }
            }
        }
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
    __syncthreads();   // the last stage is fully read before the epilogue reuses the ring
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
    if constexpr (FUSE >= 5) {
        fused_bf16<BN, FUSE == 5 ? 0 : (32 << (FUSE - 5))>(a, acc, smem, lds_base, m0, wm, wn, wave, lane);
        return;
    }
    static_assert(X16 || (WM * WN * 32 * (BN / WN) * 4 <= big_lds_bytes<MODE, BN, NST, NTH>() && NST * STAGE <= big_lds_bytes<MODE, BN, NST, NTH>()),
                  "ring and epilogue staging fit the dynamic LDS");
    epilogue<BM, BN, WM, WN, TM, TN, BF16>(a, acc, smem, big_lds_bytes<MODE, BN, NST, NTH>() / 4, m0, n0, wm, wn, wave, lane);
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
    const int tiles = ((a.M + bm - 1) / bm) * ((a.Cout + BN - 1) / BN);
    hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, BN, NST, NTH>), dim3(tiles), dim3(NTH), bytes, stream, a);
    ODAM_HIP(hipGetLastError());
    return 0;
}

template <int MODE>
static int launch_big_fused_as(const ConvGemmArgs& a, hipStream_t stream) {
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 64, 4, 512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 64, 4, 512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 128, 4, 512, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess &&
               hipFuncSetAttribute((const void*)conv_gemm_big_kernel<MODE, 64, 4, 512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess;
    }();
    if (!attr_ok) return odam_fail(2, "conv_gemm: cannot raise the dynamic LDS limit");
    const int tiles = (a.M + 255) / 256;
    if (a.Cout == 128) hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 128, 4, 512, 3>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    else if (a.G_Wt3 && a.G_N == 128) hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 64, 4, 512, 4>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    else if (a.G_Wt3) hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 64, 4, 512, 2>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    else hipLaunchKernelGGL((conv_gemm_big_kernel<MODE, 64, 4, 512, 1>), dim3(tiles), dim3(512), FUSE_LDS_BYTES, stream, a);
    ODAM_HIP(hipGetLastError());
    return 0;
}
static int launch_big_fused(const ConvGemmArgs& a, hipStream_t stream) {
    const int m16 = odam_cfg::get(odam_cfg::CG_MFMA16);      // 2: the 128-column bottleneck launches too, 3: the 64-column ones as well
    const bool x16 = a.Kpad % 32 == 0 && (m16 >= 3 || (m16 == 2 && a.Cout == 128));
    return x16 ? launch_big_fused_as<4>(a, stream) : launch_big_fused_as<3>(a, stream);
}

// bf16 bottleneck on the tile: (P, PN) combinations built -- P = 64: PN 0 / 64 / 128; P = 128: PN 0 / 128; P = 256: PN 0
template <int P, int FUSE>
static int launch_fused_bf16_as(const ConvGemmArgs& a, hipStream_t stream) {
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void*)conv_gemm_big_kernel<0, P, 4, 512, FUSE>, hipFuncAttributeMaxDynamicSharedMemorySize, FUSE_LDS_BYTES) == hipSuccess;
    }();
    if (!attr_ok) return odam_fail(2, "conv_gemm: cannot raise the dynamic LDS limit");
    ConvGemmArgs b = a;
    b.s1_window = g_s1_window;
    hipLaunchKernelGGL((conv_gemm_big_kernel<0, P, 4, 512, FUSE>), dim3((a.M + 255) / 256), dim3(512), FUSE_LDS_BYTES, stream, b);
    ODAM_HIP(hipGetLastError());
    return 0;
}
static int launch_fused_bf16(const ConvGemmArgs& a, hipStream_t stream) {
    const int pn = a.G_Wt ? a.G_N : 0;
    if (a.Cout == 64 && pn == 0) return launch_fused_bf16_as<64, 5>(a, stream);
    if (a.Cout == 64 && pn == 64) return launch_fused_bf16_as<64, 6>(a, stream);
    if (a.Cout == 64 && pn == 128) return launch_fused_bf16_as<64, 7>(a, stream);
    if (a.Cout == 128 && pn == 0) return launch_fused_bf16_as<128, 5>(a, stream);
    if (a.Cout == 128 && pn == 128) return launch_fused_bf16_as<128, 7>(a, stream);
    if (a.Cout == 256 && pn == 0) return launch_fused_bf16_as<256, 5>(a, stream);
    return odam_fail(1, "conv_gemm: fused bf16 bottleneck: channel combination not built");
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
        if (ut) hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, false, true, NST_UT>), g, b, 0, stream, a);
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

int launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream) {
    const int epc = (a.dtype == ODAM_CG_BF16) ? 8 : 4;
    if (a.Kpad % (8 * epc) != 0 || a.Cin < epc || (a.Cin & (a.Cin - 1)) != 0 || a.KW > 7 || a.KH * a.KW > 64)
        return odam_fail(1, "conv_gemm: unsupported shape (Kpad % k-tile, Cin a power of two >= one 16-byte chunk, KW <= 7)");
    if (a.k_order && (a.Cin % (8 * epc) != 0 || a.KH * a.KW > 32 || a.Kpad / (8 * epc) >= 2048))
        return odam_fail(1, "conv_gemm: k_order 1 needs Cin % k-tile == 0, at most 32 taps, fewer than 2048 k-tiles");
    if (a.M <= 0 || a.Cout <= 0) return 0;
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
                return bf ? ((odam_cfg::get(odam_cfg::CG_TILES) & 16) && !(a.KH * a.KW == 9 && a.stride == 1) ? launch_big<0, 256, 4, 1024>(a, stream) : launch_big<0, 256, 4>(a, stream))
                          : (pre ? (x16 ? launch_big<4, 256, 4>(a, stream)
                                        : launch_big<3, 256, 4>(a, stream)) : launch_big<2, 256, 4>(a, stream));
#ifndef CG_NO_BN128_FALLBACK
            // too few 256-wide tiles to fill the device (the encoder's N = 256 layers at M = 27,200: 107) but enough 128-wide
            // ones: the ring kernel on 256 x 128 tiles instead of the 128 x 128 tiles of the fp32 matrix instruction
            if (pre && big_eligible(a, 128)) return x16 ? launch_big<4, 128, 4>(a, stream) : launch_big<3, 128, 4>(a, stream);
#endif
        } else if (a.Cout > 64 && (g_big_mode >= 2 || x3) && big_eligible(a, 128)) {   // bf16 / fp32: measured slower than the 128x128 tiles (tests only)
            return bf ? launch_big<0, 128, 4>(a, stream)
                      : (pre ? (x16 ? launch_big<4, 128, 4>(a, stream) : launch_big<3, 128, 4>(a, stream)) : launch_big<2, 128, 4>(a, stream));
        } else if (x3 && a.Cout > 32 && a.Cout <= 64 && big_eligible(a, 64)) {
            // Sixteen waves on 512 x 64 tiles (cg.tiles bit 3): the 64-column loop uses ~105 registers, so FOUR waves fit a SIMD, and a
            // wave of this loop spends more time issuing (5 DMAs of ~150 cycles, two 44-instruction splits, 28 LDS reads per 768
            // cycles of matrix work) than the pipe needs -- with four per SIMD the pipe finds a ready wave more often, and the
            // filter tile is fetched once per 512 rows.  conv1: 1.44 -> 1.15 ms per 32 frames (same box, A/B by config).
            // (128-row tiles with two 4-wave workgroups per CU = the same two waves per SIMD: measured no change, not kept.)
            if (x16 && (odam_cfg::get(odam_cfg::CG_TILES) & 8) && (a.M + 511) / 512 >= 256) return launch_big<4, 64, 4, 1024>(a, stream);      // (fewer 512-row tiles than CUs: the 256-row ones)
            return pre ? (x16 ? launch_big<4, 64, 4>(a, stream) : launch_big<3, 64, 4>(a, stream)) : launch_big<2, 64, 4>(a, stream);
        } else if (bf && (g_big_mode >= 2 || (a.lda > 0 && a.lda < a.Cin)) && a.Cout > 32 && a.Cout <= 64 && big_eligible(a, 64)) {
            if ((odam_cfg::get(odam_cfg::CG_TILES) & 8) && !(a.KH * a.KW == 9 && a.stride == 1) && (a.M + 511) / 512 >= 256) return launch_big<0, 64, 4, 1024>(a, stream);
            return launch_big<0, 64, 4>(a, stream);     // conv1 as a row convolution (pixel stride < row length); tests: layer1's 3x3 alone
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

#if CG_STAMP
extern "C" int odam_cg_tail_stamps(unsigned long long* out8, int reset) {      // diagnostic builds only (-DCG_STAMP=1)
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(odam_cg::g_tail_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return 2;
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(odam_cg::g_tail_stamps), z, sizeof(z)) != hipSuccess) return 2;
    }
    return 0;
}
#endif
