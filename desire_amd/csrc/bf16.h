// bf16.h -- bf16-operand MFMA helpers shared by kernels_bf16.hip and kernels_bf16_cl.hip (v_mfma_f32_32x32x16_bf16:
// lane (c = lane&31, hi = lane>>5) holds 8 values for k-slot (hi, 0..7) of A and of B; fp32 accumulate).
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned short u16;

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {            // v_cvt_pk_bf16_f32 (RNE): a -> low half
    const f32x2v v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ u16 bf16_of(float a) { return (u16)(pk_bf16(a, 0.f) & 0xffffu); }
__device__ __forceinline__ f32x16 mfma16(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 splat16h(float v) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = v;
    return z;
}

// acc[nb][m] += A_m[32 x 16G] . B_nb[16G x 32]: A fragments from LDS (bf16 row-major, ap[m] = row (lane&31) of M-tile m
// + 8*(lane>>5) elements), B fragments from packed weights (bl[nb] already offset by +lane, uint4 units, stride 64 per
// k-group).  Chunks of CH groups with the next chunk's B fragments in flight (two named register sets, fenced).
#define CH16 4
#ifndef IOC16_OCC
#define IOC16_OCC 2
#endif
#ifndef IOC16_SPLIT
#define IOC16_SPLIT 1
#endif
#ifndef IOC16_RD
#define IOC16_RD 4                                                   // gate-contraction ring depth (k-groups in flight)
#endif
#ifndef IOC16_TWO_SETS
#define IOC16_TWO_SETS 1
#endif
template <int MT, int NB>
__device__ __forceinline__ void mma16_chunk(f32x16 (&acc)[NB][MT], const u16* const (&ap)[MT], int g, const uint4 (&b)[NB][CH16]) {
#pragma unroll
    for (int j = 0; j < CH16; ++j) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint4 a = *reinterpret_cast<const uint4*>(ap[m] + (g + j) * 16);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb][m] = mfma16(a, b[nb][j], acc[nb][m]);
        }
    }
}
template <int NB>
__device__ __forceinline__ void load_b16(uint4 (&b)[NB][CH16], const uint4* const (&bl)[NB], int g) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int j = 0; j < CH16; ++j) b[nb][j] = bl[nb][(g + j) * 64];
}
template <int MT, int NB>
__device__ __forceinline__ void mma16_groups(f32x16 (&acc)[NB][MT], const u16* const (&ap)[MT], const uint4* const (&bl)[NB], int G) {
    const int nch = G / CH16;
    int c = 0;
    if (nch > 0) {
        uint4 b0[NB][CH16], b1[NB][CH16];
        load_b16<NB>(b0, bl, 0);
#pragma clang loop unroll(disable)
        for (; c + 2 <= nch; c += 2) {
            load_b16<NB>(b1, bl, CH16 * (c + 1));
            __builtin_amdgcn_sched_barrier(0);
            mma16_chunk<MT, NB>(acc, ap, CH16 * c, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < nch) load_b16<NB>(b0, bl, CH16 * (c + 2));
            __builtin_amdgcn_sched_barrier(0);
            mma16_chunk<MT, NB>(acc, ap, CH16 * (c + 1), b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < nch) { mma16_chunk<MT, NB>(acc, ap, CH16 * c, b0); ++c; }
    }
#pragma clang loop unroll(disable)
    for (int g = CH16 * c; g < G; ++g) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const uint4 a = *reinterpret_cast<const uint4*>(ap[m] + g * 16);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb][m] = mfma16(a, bl[nb][g * 64], acc[nb][m]);
        }
    }
}


