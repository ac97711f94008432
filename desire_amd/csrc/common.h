// common.h -- gfx950 device primitives shared by every DESIRE kernel.
//
// All contractions run on the exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32: 32x32 output
// tile, K=2 per instruction, 64 cycles/SIMD, bitwise an fmaf chain).  One wave owns one or
// more 32x32 accumulator tiles; A fragments come from LDS as one ds_read_b128 per four MFMAs,
// B fragments come from HBM/L2 as one global_load_dwordx4 per four MFMAs out of weights that
// the host has re-laid in "fragment order" (see PackedB in api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DS_WG 256          // every kernel runs 4 waves per workgroup, one per SIMD
#define DS_TM 64           // rows per workgroup tile (2 MFMA M-tiles)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Accumulator element i (0..15) of this lane sits at row acc_row(i), column lane&31.
__device__ __forceinline__ int acc_row(int i) { return (i & 3) + 8 * (i >> 2) + 4 * (lane_id() >> 5); }

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// K-group g (8 consecutive k) of the contraction: lane (hi = lane>>5, c = lane&31) holds
//   A: A[row][8g + 4hi + i]      i = 0..3   (one 16-byte LDS read)
//   B: W[8g + 4hi + i][n0 + c]   i = 0..3   (one 16-byte global read from packed weights)
// and MFMA step i contracts k = {8g + i (lanes 0-31), 8g + 4 + i (lanes 32-63)}.
//
// The K loop runs in chunks of 4 groups (16 MFMAs per M-tile) with the next chunk's B fragments
// prefetched into registers while the current chunk computes; the loop is kept rolled so the
// compiler cannot hoist a whole K extent of loads (38 groups at K=304 would need 300+ VGPRs).
// SWAP = true contracts the same operands with their roles exchanged (A = the packed weights, B = the LDS rows): the
// accumulators then hold the TRANSPOSED tile (lane = LDS row, registers = weight columns in runs of four).
template <int MT, bool SWAP = false>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[MT], const float* const (&ap)[MT], int g,
                                          const float4 (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float4 a = *reinterpret_cast<const float4*>(ap[m] + (g + j) * 8);
            if (SWAP) {
                acc[m] = mfma32(b[j].x, a.x, acc[m]);
                acc[m] = mfma32(b[j].y, a.y, acc[m]);
                acc[m] = mfma32(b[j].z, a.z, acc[m]);
                acc[m] = mfma32(b[j].w, a.w, acc[m]);
            } else {
                acc[m] = mfma32(a.x, b[j].x, acc[m]);
                acc[m] = mfma32(a.y, b[j].y, acc[m]);
                acc[m] = mfma32(a.z, b[j].z, acc[m]);
                acc[m] = mfma32(a.w, b[j].w, acc[m]);
            }
        }
    }
}

// ap[m]: LDS pointer to A[row = lane&31 of M-tile m][4*(lane>>5)]; rows may be gathered (convs).
// Two statically named B register sets ping-pong (no copies): the loads of chunk c+1 are issued,
// fenced with sched_barrier so hipcc cannot sink them to their use, then chunk c computes; a load is
// therefore in flight for a whole 16*MT-MFMA chunk (>= 1024 cycles) before anything waits on it.
__device__ __forceinline__ void load_b4(float4 (&b)[4], const float4* __restrict__ b_lane, int g) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = b_lane[(g + j) * 64];
}

// B fragments a contraction starts with: the first chunk and the G % 4 tail groups.  mma_begin only issues the loads, so a
// caller can place it ahead of unrelated work (building the next LDS operand, a barrier) and have the L2 latency overlap it.
struct MmaHead { float4 b0[4]; float4 bt[3]; };
__device__ __forceinline__ void mma_begin(MmaHead& hd, const float4* __restrict__ b_lane, int G) {
    const int nch = G >> 2, rem = G & 3;
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (j < rem) hd.bt[j] = b_lane[(4 * nch + j) * 64];
    if (nch > 0) load_b4(hd.b0, b_lane, 0);
}
template <int MT, bool SWAP = false>
__device__ __forceinline__ void mma_run(f32x16 (&acc)[MT], const float* const (&ap)[MT],
                                        const float4* __restrict__ b_lane, int G, MmaHead& hd) {
    const int nch = G >> 2, rem = G & 3;
    int c = 0;
    if (nch > 0) {
        float4 b1[4];
#pragma clang loop unroll(disable)
        for (; c + 2 <= nch; c += 2) {
            load_b4(b1, b_lane, 4 * c + 4);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk<MT, SWAP>(acc, ap, 4 * c, hd.b0);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < nch) load_b4(hd.b0, b_lane, 4 * c + 8);
            __builtin_amdgcn_sched_barrier(0);
            mma_chunk<MT, SWAP>(acc, ap, 4 * c + 4, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c < nch) { mma_chunk<MT, SWAP>(acc, ap, 4 * c, hd.b0); ++c; }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < rem) {
            const int g = 4 * nch + j;
            const float4 b = hd.bt[j];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float4 a = *reinterpret_cast<const float4*>(ap[m] + g * 8);
                if (SWAP) {
                    acc[m] = mfma32(b.x, a.x, acc[m]); acc[m] = mfma32(b.y, a.y, acc[m]);
                    acc[m] = mfma32(b.z, a.z, acc[m]); acc[m] = mfma32(b.w, a.w, acc[m]);
                } else {
                    acc[m] = mfma32(a.x, b.x, acc[m]); acc[m] = mfma32(a.y, b.y, acc[m]);
                    acc[m] = mfma32(a.z, b.z, acc[m]); acc[m] = mfma32(a.w, b.w, acc[m]);
                }
            }
        }
    }
}
template <int MT, bool SWAP = false>
__device__ __forceinline__ void mma_groups_ptr(f32x16 (&acc)[MT], const float* const (&ap)[MT],
                                               const float4* __restrict__ b_lane, int G) {
    MmaHead hd;
    mma_begin(hd, b_lane, G);
    mma_run<MT, SWAP>(acc, ap, b_lane, G, hd);
}

// a_lane: LDS pointer to A[lane&31][4*(lane>>5)] of M-tile 0 (row stride lda floats,
//         lda % 4 == 0 and (lda/4) odd keeps ds_read_b128 conflict-free).
// b_lane: packed weights of this n-tile, already offset by +lane (float4 units).
template <int MT>
__device__ __forceinline__ void mma_groups(f32x16 (&acc)[MT], const float* a_lane, int lda,
                                           const float4* __restrict__ b_lane, int G) {
    const float* ap[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) ap[m] = a_lane + m * 32 * lda;
    mma_groups_ptr<MT>(acc, ap, b_lane, G);
}

// elementwise: hardware exp2 / rcp forms (v_exp_f32, v_rcp_f32: ~1 ulp each), absolute error ~1e-7
// against the libm forms of the numpy oracle -- three orders of magnitude inside the 1e-3 parity bar,
// and ~5x fewer VALU instructions than expf/tanhf/expm1f in the recurrent epilogues.
__device__ __forceinline__ float fexp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float frcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoidf_(float x) { return frcp_(1.0f + fexp_(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return fmaf(-2.0f, frcp_(1.0f + fexp_(2.0f * x)), 1.0f); }
// GRU blend h' = u h + (1 - u) c with the contraction spelled out: left to -ffp-contract the compiler fuses SOME unrolled
// elements and not others, which makes a row's result depend on the accumulator register it happens to sit in (1 ulp) --
// and tiles / shards must be bit-identical whatever row a sample lands in
__device__ __forceinline__ float gru_blend(float u, float h, float c) { return fmaf(u, h, (1.0f - u) * c); }
__device__ __forceinline__ float eluf_(float x) { return x > 0.f ? x : fexp_(x) - 1.0f; }

// Epilogue of every conv-family kernel.  mode 0 (forward): act(acc*scale + shift), act = ELU (or sigmoid when
// `sig`).  Backward data-gradient modes reuse the same kernels with the roles of the layers swapped and fold the
// DESTINATION layer's activation derivative and frozen-BN scale into the store:
//   1: acc * ELU'(y) * scale   (ELU'(y) = 1 if y > 0 else y + 1, y = saved post-activation)
//   2: acc * (y > 0)           (ReLU')          3: acc (linear)          4: acc * y (1 - y) * scale  (sigmoid')
__device__ __forceinline__ float conv_epilogue(float acc, float sc, float sh, int mode, bool sig, const float* yprev, size_t idx) {
    if (mode == 0) return sig ? sigmoidf_(acc * sc + sh) : eluf_(acc * sc + sh);
    if (mode == 3) return acc;
    const float y = yprev[idx];
    if (mode == 1) return acc * (y > 0.f ? 1.0f : y + 1.0f) * sc;
    if (mode == 2) return y > 0.f ? acc : 0.f;
    return acc * y * (1.0f - y) * sc;
}

// ---- integer paths: every float op is ONE IEEE fp32 operation (no contraction) ----------------
// scene cell: cy = clamp(floor(y*Gh), 0, Gh-1), cx likewise (oracle: scene_cell)
__device__ __forceinline__ void scene_cell_dev(float x, float y, int Gh, int Gw, int& cy, int& cx) {
    float fy = floorf(__fmul_rn(y, (float)Gh));
    float fx = floorf(__fmul_rn(x, (float)Gw));
    fy = fminf(fmaxf(fy, 0.f), (float)(Gh - 1));
    fx = fminf(fmaxf(fx, 0.f), (float)(Gw - 1));
    cy = (int)fy;
    cx = (int)fx;
}

// neighbour bin of `other` (xj,yj) seen from centre (xi,yi); -1 when outside the window
// (oracle: neighbor_bins; lineage: Social-LSTM getGridMask, the reference's missing grid.py)
// tab != nullptr selects the LOG-POLAR layout (the paper's): G rings x G sectors around the centre.  tab[0..G-1] =
// squared ring radii (ascending, the last one = r_max^2), tab[8 + 2k], tab[9 + 2k] = (cos, sin) of sector boundary k.
// ring = #{k : d^2 >= tab[k]} (outside when ring == G); sector = first k with cross(dir_k, v) >= 0 and
// cross(dir_{k+1}, v) < 0, else 0.  Comparisons and single IEEE operations only: bit-exact against the oracle.
__device__ __forceinline__ int neighbor_bin_dev(float xi, float yi, float xj, float yj,
                                                float nb_w, float nb_h, int G, const float* __restrict__ tab = nullptr) {
#pragma clang fp contract(off)      // d2 and the cross products are sums of products: they must NOT become fmas (bit-exactness)
    if (tab) {
        // plain operators under contract(off): the __f*_rn intrinsics inline library IR that may still carry the
        // `contract` flag, and mul + mul + add is exactly the shape the backend would fuse
        const float dx = xj - xi, dy = yj - yi;
        const float dxx = dx * dx, dyy = dy * dy;
        const float d2 = dxx + dyy;
        int ring = 0;
        for (int k = 0; k < G; ++k) ring += (d2 >= tab[k]) ? 1 : 0;
        if (ring >= G) return -1;
        int sector = 0;
        bool found = false;
        for (int k = 0; k < G; ++k) {
            const int k1 = (k + 1 == G) ? 0 : k + 1;
            const float a0 = tab[8 + 2 * k] * dy, b0 = tab[9 + 2 * k] * dx;
            const float a1 = tab[8 + 2 * k1] * dy, b1 = tab[9 + 2 * k1] * dx;
            const float c0 = a0 - b0, c1 = a1 - b1;
            if (!found && c0 >= 0.f && c1 < 0.f) { sector = k; found = true; }
        }
        return ring * G + sector;
    }
    const float hw = __fdiv_rn(nb_w, 2.0f), hh = __fdiv_rn(nb_h, 2.0f);
    const float lx = __fsub_rn(xi, hw), hx = __fadd_rn(xi, hw);
    const float ly = __fsub_rn(yi, hh), hy = __fadd_rn(yi, hh);
    if (!(xj < hx) || !(xj >= lx) || !(yj < hy) || !(yj >= ly)) return -1;
    float cx = floorf(__fmul_rn(__fdiv_rn(__fsub_rn(xj, lx), nb_w), (float)G));
    float cy = floorf(__fmul_rn(__fdiv_rn(__fsub_rn(yj, ly), nb_h), (float)G));
    cx = fminf(fmaxf(cx, 0.f), (float)(G - 1));
    cy = fminf(fmaxf(cy, 0.f), (float)(G - 1));
    return (int)cx + (int)cy * G;
}

// The rectangular layout with the loop invariants of a centre taken out of the pair loop and the two IEEE divisions by nb_w / nb_h
// replaced by their correctly rounded equivalent for a FIXED divisor: y = RN(1 / b) (one real division per kernel), q0 = RN(a y),
// r = a - q0 b (exact in one fma), q = RN(q0 + r y) = RN(a / b) -- Markstein's theorem; its one exception is a divisor whose significand
// is all ones (y then rounds to a power of two), which keeps the real division (`fast` is wave-uniform).  Checked against IEEE division
// over every significand of a in thirteen binades for twelve divisors (scratch/div_check.py of round 5: 0 differences outside the
// excepted divisors).  3 VALU instead of the ~12 (two of them quarter-rate) of v_div_scale / v_rcp / v_div_fmas / v_div_fixup.
struct DivBy { float b, y; bool fast; };
__device__ __forceinline__ DivBy div_by(float b) {
    DivBy d;
    d.b = b; d.y = __fdiv_rn(1.0f, b);
    const unsigned u = __float_as_uint(b), e = (u >> 23) & 0xffu;
    // |b| in 2^-62 .. 2^62: no over / underflow anywhere near the window.  (b is wave-uniform at every call site: the flag is made scalar)
    d.fast = __builtin_amdgcn_readfirstlane((int)((u & 0x7fffffu) != 0x7fffffu && e > 64u && e < 190u)) != 0;
    return d;
}
__device__ __forceinline__ float div_rn(float a, const DivBy& d) {
    if (!d.fast) return __fdiv_rn(a, d.b);
    const float q0 = __fmul_rn(a, d.y);
    const float r = fmaf(-q0, d.b, a);
    return fmaf(r, d.y, q0);
}
struct NbRect { float lx, hx, ly, hy; };
__device__ __forceinline__ NbRect nb_rect(float xi, float yi, float nb_w, float nb_h) {
    const float hw = __fdiv_rn(nb_w, 2.0f), hh = __fdiv_rn(nb_h, 2.0f);
    NbRect w;
    w.lx = __fsub_rn(xi, hw); w.hx = __fadd_rn(xi, hw); w.ly = __fsub_rn(yi, hh); w.hy = __fadd_rn(yi, hh);
    return w;
}
// == neighbor_bin_dev(xi, yi, xj, yj, nb_w, nb_h, G) bit for bit, w = nb_rect(xi, yi, ..), dw / dh = div_by(nb_w / nb_h)
__device__ __forceinline__ int neighbor_bin_rect(const NbRect& w, float xj, float yj, const DivBy& dw, const DivBy& dh, int G) {
    if (!(xj < w.hx) || !(xj >= w.lx) || !(yj < w.hy) || !(yj >= w.ly)) return -1;
    float cx = floorf(__fmul_rn(div_rn(__fsub_rn(xj, w.lx), dw), (float)G));
    float cy = floorf(__fmul_rn(div_rn(__fsub_rn(yj, w.ly), dh), (float)G));
    cx = fminf(fmaxf(cx, 0.f), (float)(G - 1));
    cy = fminf(fmaxf(cy, 0.f), (float)(G - 1));
    return (int)cx + (int)cy * G;
}

// branch-free form for batched pair loops (same value; ONLY when dw.fast && dh.fast -- the caller branches once, on wave-uniform flags,
// and keeps the loop over neighbor_bin_dev for the excepted divisors): the cell is computed whatever the window test says and selected
// at the end; a NaN position (how the batched loops mark absent agents) fails the window test
__device__ __forceinline__ float div_rn_fast(float a, const DivBy& d) {
    const float q0 = __fmul_rn(a, d.y);
    return fmaf(fmaf(-q0, d.b, a), d.y, q0);
}
__device__ __forceinline__ int neighbor_bin_rect_nb(const NbRect& w, float xj, float yj, const DivBy& dw, const DivBy& dh, int G) {
    const bool in = (xj < w.hx) & (xj >= w.lx) & (yj < w.hy) & (yj >= w.ly);
    float cx = floorf(__fmul_rn(div_rn_fast(__fsub_rn(xj, w.lx), dw), (float)G));
    float cy = floorf(__fmul_rn(div_rn_fast(__fsub_rn(yj, w.ly), dh), (float)G));
    cx = fminf(fmaxf(cx, 0.f), (float)(G - 1));
    cy = fminf(fmaxf(cy, 0.f), (float)(G - 1));
    const int b = (int)cx + (int)cy * G;
    return in ? b : -1;
}

// One tile row's neighbour search against the n_nb slots of its group: slots j_first, j_first + j_step, .. (the TPR threads of a row
// interleave).  hit(j, b) records slot j in bin b (the mask atomics); the return value is this LANE's set of bins that got a neighbour
// (the caller folds it over the wave with wave_or and issues ONE atomic per wave and word).  Rectangular layout with Markstein-safe
// divisors: NPB pairs per batch -- positions and presence flags read together, cells computed branch-free, only the atomics
// predicated; otherwise (log-polar table, excepted divisors) the pair-by-pair loop over neighbor_bin_dev.  nbw / nbh: per-step opaque
// copies of a.nb_w / a.nb_h (nb_opaque), so that nothing derived from them is an invariant of the time loop (it would be spilled and
// reloaded inside the pair loop, a memory round trip per pair).
__device__ __forceinline__ void nb_opaque(float w, float h, float& nbw, float& nbh) {
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(nbw), "=v"(nbh) : "s"(w), "s"(h));
}
template <int NPB, class Hit>
__device__ __forceinline__ unsigned long long nb_search(const float* pc, const unsigned char* vld, int grp_base, int n_nb, int j_first, int j_step,
                                                         int my_slot, float px, float py, float nbw, float nbh, int G,
                                                         const float* __restrict__ tab, Hit&& hit) {
    unsigned long long oc = 0ull;
    const DivBy dw = div_by(nbw), dh = div_by(nbh);
    if (tab || !dw.fast || !dh.fast) {
        for (int j = j_first; j < n_nb; j += j_step) {
            if (j == my_slot || !vld[grp_base + j]) continue;
            const int b = neighbor_bin_dev(px, py, pc[(grp_base + j) * 2], pc[(grp_base + j) * 2 + 1], nbw, nbh, G, tab);
            if (b >= 0) { hit(j, b); oc |= 1ull << b; }
        }
        return oc;
    }
    const NbRect win = nb_rect(px, py, nbw, nbh);
    for (int j0 = j_first; j0 < n_nb; j0 += NPB * j_step) {
        float2 pj[NPB];
        unsigned char vj[NPB];
#pragma unroll
        for (int m = 0; m < NPB; ++m) {
            const int j = grp_base + min(j0 + m * j_step, n_nb - 1);
            pj[m] = *reinterpret_cast<const float2*>(pc + j * 2);
            vj[m] = vld[j];
        }
#pragma unroll
        for (int m = 0; m < NPB; ++m) {
            const int j = j0 + m * j_step;
            const int b = neighbor_bin_rect_nb(win, pj[m].x, pj[m].y, dw, dh, G);
            if (b >= 0 && vj[m] && j != my_slot && j < n_nb) { hit(j, b); oc |= 1ull << b; }
        }
    }
    return oc;
}
// the lanes' bin sets of nb_search -> the tile's occ words (B = number of bins; every lane of the wave active)
__device__ __forceinline__ void nb_publish_occ(unsigned long long oc, unsigned* occ, int B);

// OR of v over the 64 lanes of a wave (every lane active), the same value in every lane: four row shifts, two row broadcasts (DPP), one
// readlane.  (An LDS atomicOr that every lane aims at ONE word is turned by the compiler into a scalar loop over the active lanes --
// ~8 instructions per lane, per call.)
__device__ __forceinline__ unsigned wave_or(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);       // row_shr:1
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);       // row_shr:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);       // row_shr:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);       // row_shr:8   -> lane 15 of a row holds its row
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);       // row_bcast:15 into rows 1, 3
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);       // row_bcast:31 into rows 2, 3 -> lane 63 holds the wave
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ void nb_publish_occ(unsigned long long oc, unsigned* occ, int B) {
    const unsigned o0 = wave_or((unsigned)oc);
    const unsigned o1 = B > 32 ? wave_or((unsigned)(oc >> 32)) : 0u;
    if (lane_id() == 0) { if (o0) atomicOr(&occ[0], o0); if (o1) atomicOr(&occ[1], o1); }
}

// the same for the IOC kernels' padded tiles (IocArgs.gpt > 0): -1 = a dead row
__device__ __forceinline__ int ioc_agent_of_row(int r, int K, int mno, int gpt, int ngrp);

// ---- device-side row counts (kernels.h: DynCount) ----------------------------------------------------------------------------------------------
// DYN_N(a, field, first): a.field = cnt[0] * mul when the launch carries a device-side count; the workgroup returns when its first unit `first` lies
// beyond it (wave-uniform, before any barrier).  DYN_P: the per-row stages' pseudo-scene of P present agents (mno = P, R = P * K).
#define DYN_N(a, field, first)                                                                     \
    if ((a).dyn.cnt) {                                                                             \
        (a).field = __builtin_amdgcn_readfirstlane((a).dyn.cnt[0]) * (a).dyn.mul;                  \
        if ((long)(first) >= (long)(a).field) return;                                              \
    }
#define DYN_P(a, first_row)                                                                        \
    if ((a).dyn.cnt) {                                                                             \
        (a).mno = __builtin_amdgcn_readfirstlane((a).dyn.cnt[0]);                                  \
        (a).R = (a).mno * (a).K;                                                                   \
        if ((long)(first_row) >= (long)(a).R) return;                                              \
    }
// an IOC launch over the windows of a slot class: returns false when the class is empty (the whole grid exits)
#define IOC_DYN(a)                                                                                 \
    if ((a).dyn.cnt) {                                                                             \
        const int n_c_ = __builtin_amdgcn_readfirstlane((a).dyn.cnt[0]);                           \
        (a).ngrp = n_c_ * (a).K;                                                                   \
        (a).R = (a).gpt ? (((a).ngrp + (a).gpt - 1) / (a).gpt) * 32 : (a).ngrp * (a).mno;          \
        if (n_c_ <= 0) return;                                                                     \
    }
// row r = (scene*K + k)*mno + slot  ->  agent = scene*mno + slot
__device__ __forceinline__ int agent_of_row(int r, int K, int mno) {
    const int per_scene = K * mno;
    const int scene = r / per_scene;
    return scene * mno + (r % mno);
}
__device__ __forceinline__ int ioc_agent_of_row(int r, int K, int mno, int gpt, int ngrp) {
    if (gpt == 0) return agent_of_row(r, K, mno);
    const int tile = r >> 5, i = r & 31, gi = i / mno;
    const int G = tile * gpt + gi;
    if (gi >= gpt || G >= ngrp) return -1;
    return (G / K) * mno + (i - gi * mno);
}
