// split.h -- fp32 operands as bf16 pieces on the bf16 matrix pipe (shared by kernels_x3.hip and kernels_x6.hip).
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the flop rate of v_mfma_f32_32x32x2_f32 and has no xf32 / TF32 rate in between.
// An fp32 value is a sum of bf16 pieces, every piece product is exact in the fp32 accumulator, so an fp32 product becomes a few
// bf16 MFMAs: two pieces / three products (~2^-16 relative), or three pieces / six products (fp32-class, below).
#pragma once
#include "bf16.h"

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pk_bf16(a, b);
    lo = pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// NP bf16 pieces of a pair of fp32 values: x = p0 + p1 (+ p2) with p_{i} = bf16(x - p_0 - .. - p_{i-1}); every subtraction is exact.
// NP = 2 leaves |r| <= 2^-17 |x| (three products per fp32 product, ~2^-16 relative); NP = 3 represents an fp32 value EXACTLY
// (8 + 8 + 8 significant bits) and six products -- every term down to 2^-16 |a b| -- leave an error of <= 2^-23 |a b| per product,
// the class of the fp32 fmaf chain's own accumulation rounding (dims.bf16 = 3, "x6").
template <int NP>
__device__ __forceinline__ void splitp(float a, float b, unsigned (&p)[NP]) {
    p[0] = pk_bf16(a, b);
    float ra = a - __uint_as_float(p[0] << 16), rb = b - __uint_as_float(p[0] & 0xffff0000u);
    p[1] = pk_bf16(ra, rb);
    if constexpr (NP == 3) {
        ra -= __uint_as_float(p[1] << 16); rb -= __uint_as_float(p[1] & 0xffff0000u);
        p[2] = pk_bf16(ra, rb);
    }
}
template <int NP> struct FragP { uint4 p[NP]; };
template <int NP>
__device__ __forceinline__ FragP<NP> split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
    FragP<NP> f;
    unsigned a[NP], b[NP], c[NP], d[NP];
    splitp<NP>(v0, v1, a); splitp<NP>(v2, v3, b); splitp<NP>(v4, v5, c); splitp<NP>(v6, v7, d);
#pragma unroll
    for (int i = 0; i < NP; ++i) f.p[i] = make_uint4(a[i], b[i], c[i], d[i]);
    return f;
}
// the (piece of A, piece of B) products of one fp32 product, smallest terms first: NP = 2 -> lo.hi, hi.lo, hi.hi;
// NP = 3 -> (2,0) (0,2) (1,1) (1,0) (0,1) (0,0)
template <int NP> struct Pairs;
template <> struct Pairs<2> { static constexpr int N = 3; static constexpr int A[3] = {1, 0, 0}; static constexpr int B[3] = {0, 1, 0}; };
template <> struct Pairs<3> { static constexpr int N = 6; static constexpr int A[6] = {2, 0, 1, 1, 0, 0}; static constexpr int B[6] = {0, 2, 1, 0, 1, 0}; };
// acc += a . b with split operands
template <int NP>
__device__ __forceinline__ f32x16 mfma_xp(const uint4 (&a)[NP], const uint4 (&b)[NP], f32x16 c) {
#pragma unroll
    for (int i = 0; i < Pairs<NP>::N; ++i) c = mfma16(a[Pairs<NP>::A[i]], b[Pairs<NP>::B[i]], c);
    return c;
}


// One A fragment (lane = row, 8 consecutive k) straight out of an fp32 LDS tile: two 16-byte reads, split on the fly into its three
// pieces (44 VALU operations -- affordable where the fragment feeds six or more MFMAs, i.e. >= 192 matrix-pipe cycles).
template <int NP>
__device__ __forceinline__ FragP<NP> fragp(const float* p8) {
    const float4 x0 = *reinterpret_cast<const float4*>(p8), x1 = *reinterpret_cast<const float4*>(p8 + 4);
    return split8<NP>(x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w);
}
__device__ __forceinline__ FragP<3> frag6(const float* p8) { return fragp<3>(p8); }
// acc[nb] += A[32 x 16 G16] . B_nb[16 G16 x 32], six products per fp32 product: A from an fp32 LDS tile (ap = this lane's row + 8 * hi
// floats, 16-byte aligned), B from three-piece packs [p0 | p1 | p2]: piece i of n-tile nb at bl[nb] + i * plo (uint4 units, already
// + lane; 64 uint4 per k-group).  B fragments run one k-group ahead (two named register sets, fenced).
template <int NB, int NP = 3>
__device__ __forceinline__ void mma6_groups(f32x16 (&acc)[NB], const float* ap, const uint4* const (&bl)[NB], size_t plo, int G16) {
    uint4 b0[NB][NP], b1[NB][NP];
    auto ld = [&](uint4 (&b)[NB][NP], int g) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < NP; ++i) b[nb][i] = bl[nb][i * plo + (size_t)g * 64];
    };
    auto run = [&](const uint4 (&b)[NB][NP], int g) {
        const FragP<NP> a = fragp<NP>(ap + g * 16);
#pragma unroll
        for (int pr = 0; pr < Pairs<NP>::N; ++pr)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma16(a.p[Pairs<NP>::A[pr]], b[nb][Pairs<NP>::B[pr]], acc[nb]);
    };
    ld(b0, 0);
    int g = 0;
#pragma clang loop unroll(disable)
    for (; g + 2 <= G16; g += 2) {
        ld(b1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        run(b0, g);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < G16) ld(b0, g + 2);
        __builtin_amdgcn_sched_barrier(0);
        run(b1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (g < G16) run(b0, g);
}

// acc[nb] += A[32 x 16G] . B_nb[16G x 32]: piece i of A = the image at ap + i * alo (bf16 elements); piece i of B = the pack at
// bl[nb] + i * blo (uint4 units; bl already + lane).  One k-group at a time, the next group's fragments in flight (used by the
// regression head only: G = H/16 groups once per pass).
#ifndef RD4
#define RD4 2                                                       // ring depth (k-groups) of the gate contraction
#endif
// SWAP = true exchanges the roles of the two operands in every MFMA (the fragment layouts are symmetric): the accumulators then hold the
// TRANSPOSED tile -- lane = the image's row, registers = the pack's columns in runs of four (8 q + 4 (lane >> 5) + 0..3).
template <int NB, int NP, bool SWAP = false>
__device__ __forceinline__ void mmax_groups(f32x16 (&acc)[NB], const u16* ap, int alo, const uint4* const (&bl)[NB], size_t blo, int G) {
    uint4 b0[NB][NP], b1[NB][NP];
    auto ld = [&](uint4 (&b)[NB][NP], int g) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < NP; ++i) b[nb][i] = bl[nb][i * blo + (size_t)g * 64];
    };
    auto run = [&](const uint4 (&b)[NB][NP], int g) {
        uint4 av[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) av[i] = *reinterpret_cast<const uint4*>(ap + i * alo + g * 16);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = SWAP ? mfma_xp<NP>(b[nb], av, acc[nb]) : mfma_xp<NP>(av, b[nb], acc[nb]);
    };
    ld(b0, 0);
    int g = 0;
#pragma clang loop unroll(disable)
    for (; g + 2 <= G; g += 2) {
        ld(b1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
        run(b0, g);
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < G) ld(b0, g + 2);
        __builtin_amdgcn_sched_barrier(0);
        run(b1, g + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (g < G) run(b0, g);
}

// TWO row blocks against ONE n-tile of a pack: acc[m] += A_m[32 x 16G] . B[16G x 32], m = 0, 1 (images of block m at ap + m * amo), every pack
// fragment fetched ONCE and feeding both blocks (2 * Pairs MFMAs per fetched k-group instead of Pairs), D k-groups in flight (ring slot = g % D,
// ring loop rolled).  Per accumulator the products and their order are those of mmax_groups: bit-identical results.
template <int NP, int D>
__device__ __forceinline__ void mmax_rows2_ring(f32x16 (&acc)[2], const u16* ap, int amo, int alo, const uint4* bl, size_t blo, int G) {
    uint4 b[D][NP];
    auto ld = [&](uint4 (&bb)[NP], int g) {
#pragma unroll
        for (int i = 0; i < NP; ++i) bb[i] = bl[i * blo + (size_t)g * 64];
    };
#pragma unroll
    for (int d = 0; d < D; ++d) if (d < G) ld(b[d], d);
    __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(disable)
    for (int g0 = 0; g0 < G; g0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (g0 + d < G) {                                      // (wave-uniform)
                uint4 a0[NP], a1[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    a0[i] = *reinterpret_cast<const uint4*>(ap + i * alo + (g0 + d) * 16);
                    a1[i] = *reinterpret_cast<const uint4*>(ap + amo + i * alo + (g0 + d) * 16);
                }
#pragma unroll
                for (int pr = 0; pr < Pairs<NP>::N; ++pr) {        // the two blocks alternate on the pipe (independent accumulators)
                    acc[0] = mfma16(a0[Pairs<NP>::A[pr]], b[d][Pairs<NP>::B[pr]], acc[0]);
                    acc[1] = mfma16(a1[Pairs<NP>::A[pr]], b[d][Pairs<NP>::B[pr]], acc[1]);
                }
                if (g0 + d + D < G) ld(b[d], g0 + d + D);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// mmax_groups with D k-groups of pack fragments in flight and any group count G (guarded tail; ring loop rolled; addresses = uniform base + lane * 16,
// see mmax_ring below).  Same products in the same order per accumulator as mmax_groups.
template <int NB, int NP, int D>
__device__ __forceinline__ void mmaxp_ring(f32x16 (&acc)[NB], const u16* ap, int alo, const uint4* W, const unsigned (&t0)[NB], size_t blo, int G) {
    uint4 b[D][NB][NP];
    const unsigned lane16 = (unsigned)lane_id() * 16u;
    auto ld = [&](uint4 (&bb)[NB][NP], int g) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const char* base = reinterpret_cast<const char*>(W + ((size_t)t0[nb] + (size_t)i * blo + (size_t)g * 64));
                bb[nb][i] = *reinterpret_cast<const uint4*>(base + lane16);
            }
    };
#pragma unroll
    for (int dd = 0; dd < D; ++dd) if (dd < G) ld(b[dd], dd);
    __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(disable)
    for (int g0 = 0; g0 < G; g0 += D) {
#pragma unroll
        for (int dd = 0; dd < D; ++dd) {
            if (g0 + dd < G) {                                     // (wave-uniform)
                uint4 av[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) av[i] = *reinterpret_cast<const uint4*>(ap + i * alo + (g0 + dd) * 16);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma_xp<NP>(av, b[dd][nb], acc[nb]);
                if (g0 + dd + D < G) ld(b[dd], g0 + dd + D);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The same contraction with D k-groups of pack fragments in flight (ring slot = g % D; G a multiple of D; the loop over rings stays ROLLED so that
// the fragment addresses are formed per ring -- fully unrolled they are loop invariants of the caller's time loop, get hoisted and spill).
// One group is only NB * (3 | 6) bf16 MFMAs -- 100 to 200 matrix-pipe cycles against the ~1000 cycles a fragment takes to arrive from L2 under
// load -- so with ONE group ahead (mmax_groups) every group waits for its fragments: k_ioc_bwd_x3's per-bin contraction ran 8 groups in ~4000 cycles.
// Fragment addresses are (uniform base in SGPRs) + (lane * 16 in one VGPR): W and the uint4 indices t0[nb] (group 0 of n-tile nb, piece 0) and
// blo (piece stride) must be wave-uniform.  Per-lane 64-bit pointers per (n-tile, piece, ring slot) would be loop invariants of the caller's
// time loop -- 2 VGPRs each, hoisted and spilled.
template <int NB, int NP, bool SWAP, int G, int DW>
__device__ __forceinline__ void mmax_ring(f32x16 (&acc)[NB], const u16* ap, int alo, const uint4* W, const unsigned (&t0)[NB], unsigned blo) {
    constexpr int D = DW < G ? DW : G;
    static_assert(G % D == 0, "whole rings");
    uint4 b[D][NB][NP];
    const unsigned lane16 = (unsigned)lane_id() * 16u;
    auto ld = [&](uint4 (&bb)[NB][NP], int g) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const char* base = reinterpret_cast<const char*>(W + ((size_t)t0[nb] + (size_t)i * blo + (size_t)g * 64));
                bb[nb][i] = *reinterpret_cast<const uint4*>(base + lane16);
            }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) ld(b[d], d);
    __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(disable)
    for (int g0 = 0; g0 < G; g0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            uint4 av[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) av[i] = *reinterpret_cast<const uint4*>(ap + i * alo + (g0 + d) * 16);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = SWAP ? mfma_xp<NP>(b[d][nb], av, acc[nb]) : mfma_xp<NP>(av, b[d][nb], acc[nb]);
            if (g0 + D < G) ld(b[d], g0 + d + D);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
