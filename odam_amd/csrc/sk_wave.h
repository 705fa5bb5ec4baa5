// sk_wave.h -- cross-lane pieces of the one-wavefront Sinkhorn kernel (assoc.hip: sinkhorn_wave_kernel): exchanges over every lane bit
// of a 64-lane wavefront on DPP / v_permlane*_swap, and the transposing column reduction built from them.
#pragma once
#include <hip/hip_runtime.h>

__host__ __device__ constexpr int sk_lane(int c) { return (c & 0x13) | (((c >> 2) & 1) << 3) | (((c >> 3) & 1) << 2); }     // column <-> lane: bits 2, 3 trade places

template <int CTRL> __device__ __forceinline__ float sk_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// what the lane 1 << BIT away holds
template <int BIT> __device__ __forceinline__ float sk_xor(float v) {
    if constexpr (BIT == 0) return sk_dpp<0xB1>(v);
    else if constexpr (BIT == 1) return sk_dpp<0x4E>(v);
    else if constexpr (BIT == 2) return sk_dpp<0x1B>(sk_dpp<0x141>(v));        // (l ^ 7) ^ 3
    else if constexpr (BIT == 3) return sk_dpp<0x128>(v);                      // row_ror:8
    else {
        // (the swap instructions exchange IN PLACE between two registers: handed the same value twice the compiler gives it one
        //  register and the swap does nothing -- the second operand is made a copy it cannot see through)
        const unsigned x = __builtin_bit_cast(unsigned, v);
        unsigned y = x;
        asm volatile("" : "+v"(y));
        // (r[0], r[1] go through scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 with this compiler)
        unsigned r0, r1;
        if constexpr (BIT == 4) { auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); r0 = r[0]; r1 = r[1]; }
        else { auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); r0 = r[0]; r1 = r[1]; }
        return (threadIdx.x & (1 << BIT)) ? __builtin_bit_cast(float, r0) : __builtin_bit_cast(float, r1);
    }
}
// one transposing step over lane bit BIT: lanes with the bit clear end with A summed over the pair, the others with B
template <int BIT> __device__ __forceinline__ float sk_tr(float A, float B, bool bit) {
    if constexpr (BIT >= 4) {
        const unsigned a = __builtin_bit_cast(unsigned, A);
        unsigned b = __builtin_bit_cast(unsigned, B);
        asm volatile("" : "+v"(b));          // A and B may be the same value (all-reduce form): see sk_xor
        unsigned r0, r1;
        if constexpr (BIT == 4) { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); r0 = r[0]; r1 = r[1]; }
        else { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); r0 = r[0]; r1 = r[1]; }
        return __builtin_bit_cast(float, r0) + __builtin_bit_cast(float, r1);
    } else {
        const float keep = bit ? B : A, send = bit ? A : B;
        return keep + sk_xor<BIT>(send);
    }
}
// t[P] per lane (column c in t[c]) -> the total of column c over the 64 lanes, in every lane l with sk_lane(c) == l & 31 (mod the width)
template <int P> __device__ __forceinline__ float sk_colsum(float (&t)[32], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < P / 2; i++) t[i] = sk_tr<0>(t[2 * i], t[2 * i + 1], b0);
#pragma unroll
    for (int i = 0; i < P / 4; i++) t[i] = sk_tr<1>(t[2 * i], t[2 * i + 1], b1);
#pragma unroll
    for (int i = 0; i < P / 8; i++) t[i] = sk_tr<3>(t[2 * i], t[2 * i + 1], b3);
    float x;
    if constexpr (P == 8) { x = t[0]; x += sk_xor<2>(x); }
    else {
#pragma unroll
        for (int i = 0; i < P / 16; i++) t[i] = sk_tr<2>(t[2 * i], t[2 * i + 1], b2);
        x = t[0];
    }
    if constexpr (P == 32) x = sk_tr<4>(t[0], t[1], false);
    else x = sk_tr<4>(x, x, false);                 // swap with itself: both halves of every row pair hold the pair's sum
    return sk_tr<5>(x, x, false);
}
__device__ __forceinline__ float sk_wave_max(float x) {
    x = fmaxf(x, sk_xor<0>(x)); x = fmaxf(x, sk_xor<1>(x)); x = fmaxf(x, sk_xor<2>(x)); x = fmaxf(x, sk_xor<3>(x));
    x = fmaxf(x, sk_xor<4>(x)); return fmaxf(x, sk_xor<5>(x));
}
__device__ __forceinline__ float sk_wave_sum(float x) {
    x += sk_xor<0>(x); x += sk_xor<1>(x); x += sk_xor<2>(x); x += sk_xor<3>(x); x += sk_xor<4>(x); return x + sk_xor<5>(x);
}

