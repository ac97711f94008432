// kernels_x3.hip -- fp32-equivalent contractions on the bf16 matrix pipe ("split" operands, dims.bf16 = 2).
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the flop rate of v_mfma_f32_32x32x2_f32.  An fp32 value splits exactly into
// bf16 pieces  x = hi + lo + r,  hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-17 |x|,  so
//      a . b  =  a_hi b_hi + a_lo b_hi + a_hi b_lo  +  O(2^-16 |a b|)
// with every product exact in the fp32 accumulator: three bf16 MFMAs per fp32 one, at 3/16 of its matrix time, and a
// result that differs from the fp32 kernels' by ~1e-5 relative instead of the ~4e-3 of plain bf16 operands.
//   * activations are kept in LDS as TWO bf16 images (hi, lo) of each operand tile -- the same bytes as one fp32 tile
//   * weights are packed as [hi pack | lo pack] ("ioc/*16" buffers of api.hip, lo = bf16(w - hi(w)))
//   * 0/1 neighbour masks are exact in bf16: the pooling's first link costs 2 MFMAs (hi, lo), every other contraction 3
// Structure = k_ioc_bf16's bin-split form (kernels_bf16.hip): 32-row tile = whole (scene,k) groups, wave cb owns hidden
// columns [32cb, 32cb+32), occupied bins dealt round-robin to the waves, partial e_r tiles summed in fixed order through
// LDS slots that alias the (then dead) h^T and r*h tiles.
#include "common.h"
#include "kernels.h"

#include "bf16.h"
#include "split.h"

// ------------------------------------------------------------------------------------------------------------------
// IOC scoring / refinement with split operands.  Tile = 32 rows = whole (scene,k) groups (mno divides 32), H in {64, 128}.
// Weight pointers of IocArgs point at the [hi | lo] bf16 packs.
// ------------------------------------------------------------------------------------------------------------------
#ifdef DESIRE_IOC_TIMING
#define TICKX(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#else
#define TICKX(k)
#endif
// NP = bf16 pieces per fp32 operand: 2 (dims.bf16 = 2, three products per fp32 product, two workgroups per CU) or 3 (dims.bf16 = 3,
// six products, fp32-class accuracy; three operand images = 120 KB of LDS, one workgroup per CU with the whole register file).
// PAD: padded tiles (IocArgs.gpt: slot classes that do not divide 32), a template parameter so that the packed-row instantiations stay as they were
template <int H, int EV, int C, bool TRAIN, int NP = 2, bool PAD = false>      // TRAIN: keep x_t = [e_v | e_s | e_r], r, u, c, h_t of every step (fp32) for the backward pass
__global__ __launch_bounds__((H / 32) * 64, NP == 2 ? 2 : 1) void k_ioc_x3(IocArgs a) {
#ifdef DESIRE_IOC_TIMING
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = H >> 5, TM = 32, E = EV + C + H, KX = E + H;
    constexpr int LDXB = KX + 8, LDRB = H + 8, LDT = TM + 8;          // bf16 elements; (ld/2) = 4 mod 8 dwords: conflict-free b128
    constexpr int XLO = TM * LDXB, RLO = TM * LDRB, TLO = H * LDT;    // element offset of an operand tile's lo image
    constexpr int NTHR = NT * 64, TPR = NTHR / TM;
    constexpr int G16 = KX >> 4, GX16 = E >> 4, GH16 = H >> 4;
    static_assert(NP * TLO * 2 >= NT * 4096 && NP * RLO * 2 >= NT * 4096, "exchange slots must fit the tiles they alias");
    static_assert(NP == 2 || !TRAIN, "the six-product form is inference only");
    const int B = a.G * a.G, LDM = B + 1;
    u16* Xb = reinterpret_cast<u16*>(smem_raw);                       // [NP][TM][LDXB]  e_v | e_s | e_r | h   (one image per piece)
    u16* RHb = Xb + NP * XLO;                                         // [NP][TM][LDRB]  r * h
    u16* Ht = RHb + NP * RLO;                                         // [NP][H][LDT]    h transposed (pooling operand)
    unsigned* masks = reinterpret_cast<unsigned*>(Ht + NP * TLO);     // [TM][B+1], bit = local row
    uint2* lut = reinterpret_cast<uint2*>(masks + ((TM * LDM + 1) & ~1));   // [16] nibble -> 4 bf16 (0.0 / 1.0)
    float* pc = reinterpret_cast<float*>(lut + 16);                   // [TM][2]
    float* pp = pc + TM * 2;                                          // [TM][2]
    float* wv = pp + TM * 2;                                          // [3][EV]
    float* red = wv + 3 * EV;                                         // [NT][TM]
    unsigned char* vld = reinterpret_cast<unsigned char*>(red + NT * TM);   // [TM]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + TM);                  // [2] bins that hold a neighbour anywhere in the tile
    float* EX0 = reinterpret_cast<float*>(Ht);                              // exchange set 0: inside the h^T tile (dead after the pooling)
    float* EX1 = reinterpret_cast<float*>(RHb);                             // set 1: inside the r*h tile (idle until the gates)

    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    const int row0 = blockIdx.x * TM;
    IOC_DYN(a)                                          // (a slot class counted on the device: kernels.h DynCount; the grid is the worst case's)
    if (a.dyn.cnt && row0 >= a.R) return;
    const int col = cb * 32 + c31;
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int my_row = min(row0 + r8, a.R - 1);
    const int tile = blockIdx.x;
    const int gpt = PAD ? a.gpt : 0;
    const bool dead_row = gpt && (r8 / a.mno >= gpt || tile * gpt + r8 / a.mno >= a.ngrp);            // padded tiles (kernels.h: IocArgs.gpt)
    const int my_scene = gpt ? min(tile * gpt + min(r8 / a.mno, gpt - 1), a.ngrp - 1) / a.K : my_row / (a.K * a.mno);
    const int grp_base = (r8 / a.mno) * a.mno;
    const int my_slot = r8 - grp_base;
    const int n_nb = dead_row ? 0 : a.mno;

    for (int i = tid; i < 3 * EV; i += NTHR) wv[i] = (i < 2 * EV) ? a.w_vel[i] : a.b_vel[i - 2 * EV];
    if (tid < 16) {
        const unsigned lo = ((tid & 1) ? 0x3F80u : 0u) | ((tid & 2) ? 0x3F800000u : 0u);
        const unsigned hi2 = ((tid & 4) ? 0x3F80u : 0u) | ((tid & 8) ? 0x3F800000u : 0u);
        lut[tid] = make_uint2(lo, hi2);
    }
    if (tid < TM) { const int ag = ioc_agent_of_row(min(row0 + tid, a.R - 1), a.K, a.mno, gpt, a.ngrp); vld[tid] = ag >= 0 ? a.valid[ag] : 0; }
    const float bgr = a.b_g[col], bgu = a.b_g[H + col], bcc = a.b_c[col], bso = a.b_soc[col], wsc = a.w_score[col];
    const float* grid = a.grids + (size_t)a.grid_of_scene[my_scene] * a.Gh * a.Gw * C;
    const uint4* Wg = reinterpret_cast<const uint4*>(a.Wg);
    const uint4* Wc = reinterpret_cast<const uint4*>(a.Wc);
    const uint4* Wsoc = reinterpret_cast<const uint4*>(a.Wsoc);
    const uint4* Wreg = reinterpret_cast<const uint4*>(a.Wreg);
    constexpr size_t WG_LO = (size_t)2 * NT * G16 * 64, WC_LO = (size_t)NT * G16 * 64;   // uint4 offset from one piece's pack to the next
    const size_t WS_LO = (size_t)B * NT * GH16 * 64, WR_LO = (size_t)a.NTreg * GH16 * 64;

    const u16* xp = Xb + c31 * LDXB + 8 * hi;
    const u16* rp = RHb + c31 * LDRB + 8 * hi;
    const int arow = 4 * hi;                                          // + (i&3) + 8(i>>2): local row of accumulator element i
    // training-mode saves: (uniform tile base) + (32-bit offset inside the tile) keeps the addresses out of the register budget
    const size_t sv_tb = TRAIN ? (size_t)row0 * a.T * H : 0;
    float* sv_r_t = TRAIN ? a.sv_r + sv_tb : nullptr; float* sv_u_t = TRAIN ? a.sv_u + sv_tb : nullptr;
    float* sv_c_t = TRAIN ? a.sv_c + sv_tb : nullptr; float* sv_h_t = TRAIN ? a.sv_h + sv_tb : nullptr;
    float* sv_x_t = TRAIN ? a.sv_x + (size_t)row0 * a.T * E : nullptr;
    auto sv_off = [&](int i, int t) { return (unsigned)(((arow + (i & 3) + 8 * (i >> 2)) * a.T + t) * H + col); };
    auto sv_ok = [&](int i) { return row0 + arow + (i & 3) + 8 * (i >> 2) < a.R; };
    // h (fp32, accumulator layout) -> hi / lo images of both operand tiles
    // four values of one accumulator column run (rows arow + 8q + 0..3) -> every piece's image of a row-major tile
    auto put4 = [&](u16* x, int ld, int lo, float v0, float v1, float v2, float v3, unsigned (&pa)[NP], unsigned (&pb)[NP]) {
        splitp<NP>(v0, v1, pa);
        splitp<NP>(v2, v3, pb);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            x[i * lo] = (u16)pa[i]; x[i * lo + ld] = (u16)(pa[i] >> 16); x[i * lo + 2 * ld] = (u16)pb[i]; x[i * lo + 3 * ld] = (u16)(pb[i] >> 16);
        }
    };
    auto publish_h = [&](const f32x16& h) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            unsigned pa[NP], pb[NP];
            put4(Xb + (arow + 8 * q) * LDXB + E + col, LDXB, XLO, h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3], pa, pb);
#pragma unroll
            for (int i = 0; i < NP; ++i) *reinterpret_cast<uint2*>(Ht + i * TLO + col * LDT + arow + 8 * q) = make_uint2(pa[i], pb[i]);
        }
    };

    for (int it = 0; it < a.iters; ++it) {
        int row0p;                                                    // opaque copy: keeps the prologue's address math out of the time loop's registers
        asm volatile("s_mov_b32 %0, %1" : "=s"(row0p) : "s"(row0));
        f32x16 h, sp = zero16();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min(row0p + arow + (i & 3) + 8 * (i >> 2), a.R - 1);
            const int ag = ioc_agent_of_row(row, a.K, a.mno, gpt, a.ngrp);
            h[i] = ag >= 0 ? a.Hx[(size_t)ag * a.ldhx + col] : 0.f;
        }
        __syncthreads();                                  // previous pass's readers of Xb / Ht are done
        publish_h(h);
        float2 ynext = make_float2(0.f, 0.f);
        if (tid < TM) {
            const int row = min(row0 + tid, a.R - 1);
            const int ag = ioc_agent_of_row(row, a.K, a.mno, gpt, a.ngrp);
            pp[tid * 2] = ag >= 0 ? a.p_last[(size_t)ag * 2] : 0.f; pp[tid * 2 + 1] = ag >= 0 ? a.p_last[(size_t)ag * 2 + 1] : 0.f;
            const float2 y0 = *reinterpret_cast<const float2*>(a.Y + ((size_t)row * a.T) * 2);
            pc[tid * 2] = y0.x; pc[tid * 2 + 1] = y0.y;
        }
        for (int i = tid; i < TM * LDM; i += NTHR) masks[i] = 0u;
        if (tid < 2) occ[tid] = 0;
        __syncthreads();

        for (int t = 0; t < a.T; ++t) {
            TICKX(0)
            if (tid < TM && t + 1 < a.T)
                ynext = *reinterpret_cast<const float2*>(a.Y + ((size_t)min(row0 + tid, a.R - 1) * a.T + t + 1) * 2);
            // ---- P1: e_v, e_s, neighbour bits (row threads) ----
            {
                const float px = pc[r8 * 2], py = pc[r8 * 2 + 1];
                const float vx = px - pp[r8 * 2], vy = py - pp[r8 * 2 + 1];
                for (int j = 2 * q8; j < EV; j += 2 * TPR) {
                    const float e0 = fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f);
                    const float e1 = fmaxf(fmaf(vy, wv[EV + j + 1], vx * wv[j + 1]) + wv[2 * EV + j + 1], 0.f);
                    unsigned ep[NP];
                    splitp<NP>(e0, e1, ep);
#pragma unroll
                    for (int i = 0; i < NP; ++i) *reinterpret_cast<unsigned*>(Xb + i * XLO + r8 * LDXB + j) = ep[i];
                    if (TRAIN && row0 + r8 < a.R) *reinterpret_cast<float2*>(sv_x_t + (unsigned)((r8 * a.T + t) * E + j)) = make_float2(e0, e1);
                }
                int cy, cx;
                scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
                const float* gsrc = grid + ((size_t)cy * a.Gw + cx) * C;
                for (int j = 4 * q8; j < C; j += 4 * TPR) {
                    const float4 g4 = *reinterpret_cast<const float4*>(gsrc + j);
                    unsigned ga[NP], gb[NP];
                    splitp<NP>(g4.x, g4.y, ga); splitp<NP>(g4.z, g4.w, gb);
#pragma unroll
                    for (int i = 0; i < NP; ++i) *reinterpret_cast<uint2*>(Xb + i * XLO + r8 * LDXB + EV + j) = make_uint2(ga[i], gb[i]);
                    if (TRAIN && row0 + r8 < a.R) *reinterpret_cast<float4*>(sv_x_t + (unsigned)((r8 * a.T + t) * E + EV + j)) = g4;
                }
                float nbw, nbh;
                nb_opaque(a.nb_w, a.nb_h, nbw, nbh);
                const unsigned long long oc = nb_search<4>(pc, vld, grp_base, n_nb, q8, TPR, my_slot, px, py, nbw, nbh, a.G, a.bin_tab,
                                                          [&](int j, int b) { atomicOr(&masks[r8 * LDM + b], 1u << (grp_base + j)); });
                nb_publish_occ(oc, occ, B);
            }
            TICKX(1)
            __syncthreads();
            TICKX(2)
            // ---- P2: social pooling chain -> e_r (occupied bins dealt round-robin to the waves) ----
            unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
            om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
            {
                unsigned long long mine = 0ull;
                {
                    int k = 0;
                    for (unsigned long long tmp = om; tmp; tmp &= tmp - 1, ++k)
                        if (k % NT == cb) mine |= tmp & (0ull - tmp);
                }
                f32x16 soc[NT];
#pragma unroll
                for (int k = 0; k < NT; ++k) soc[k] = zero16();
                auto wptr = [&](int b, int hb, int k) {           // hi fragments of W_b[hidden block hb][column block (cb+k)%NT], 2 k-groups
                    const int cbo = (cb + k) % NT;
                    return Wsoc + ((size_t)(b * NT + cbo) * GH16 + 2 * hb) * 64 + lane;
                };
                if constexpr (NP == 3) {
                    // one wave per SIMD (the whole register file, nobody else to cover a stall): two fragment sets, set (hb & 1), each
                    // refreshed right after its last use with the fragments of the iteration AFTER next; the first link of hidden block
                    // hb + 1 is issued before hb's accumulators are split and consumed; the second link's MFMAs walk the NT partial
                    // tiles round-robin, so an accumulator is revisited NT MFMAs later (per accumulator the order of the products is
                    // unchanged: results are bit-identical to the straight form)
                    uint4 wf[2][2 * NT][NP];
                    if (mine) {
                        const int b0 = __ffsll((long long)mine) - 1;
#pragma unroll
                        for (int st = 0; st < 2; ++st)
#pragma unroll
                            for (int k = 0; k < NT; ++k) {
                                const uint4* p = wptr(b0, st % NT, k);
#pragma unroll
                                for (int i = 0; i < NP; ++i) { wf[st][2 * k][i] = p[i * WS_LO]; wf[st][2 * k + 1][i] = p[i * WS_LO + 64]; }
                            }
                    }
#pragma clang loop unroll(disable)
                    while (mine) {
                        const int b = __ffsll((long long)mine) - 1;
                        mine &= mine - 1;
                        const int nb = mine ? __ffsll((long long)mine) - 1 : b;
                        uint4 mf[2];
                        const unsigned m32 = masks[c31 * LDM + b];
#pragma unroll
                        for (int jg = 0; jg < 2; ++jg) {
                            const unsigned bits = (m32 >> (16 * jg + 8 * hi)) & 0xffu;
                            const uint2 l0 = lut[bits & 15u], l1 = lut[bits >> 4];
                            mf[jg] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                        }
                        auto chain = [&](int hb) {
                            f32x16 d1 = zero16();
                            const u16* hp = Ht + (hb * 32 + c31) * LDT + 8 * hi;
#pragma unroll
                            for (int jg = 0; jg < 2; ++jg)
#pragma unroll
                                for (int i = NP - 1; i >= 0; --i) d1 = mfma16(*reinterpret_cast<const uint4*>(hp + i * TLO + 16 * jg), mf[jg], d1);
                            return d1;
                        };
                        f32x16 da = chain(0), dn;
#pragma unroll
                        for (int hb = 0; hb < NT; ++hb) {
                            const int st = hb & 1;
                            if (hb + 1 < NT) dn = chain(hb + 1);
                            const FragP<NP> p0 = split8<NP>(da[0], da[1], da[2], da[3], da[4], da[5], da[6], da[7]);
                            const FragP<NP> p1 = split8<NP>(da[8], da[9], da[10], da[11], da[12], da[13], da[14], da[15]);
#pragma unroll
                            for (int pr = 0; pr < Pairs<NP>::N; ++pr)
#pragma unroll
                                for (int k = 0; k < NT; ++k) soc[k] = mfma16(p0.p[Pairs<NP>::A[pr]], wf[st][2 * k][Pairs<NP>::B[pr]], soc[k]);
#pragma unroll
                            for (int pr = 0; pr < Pairs<NP>::N; ++pr)
#pragma unroll
                                for (int k = 0; k < NT; ++k) soc[k] = mfma16(p1.p[Pairs<NP>::A[pr]], wf[st][2 * k + 1][Pairs<NP>::B[pr]], soc[k]);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int k = 0; k < NT; ++k) {            // refresh set st: (b, hb + 2), or (next bin, hb + 2 - NT)
                                const uint4* p = (hb + 2 < NT) ? wptr(b, hb + 2, k) : wptr(nb, (hb + 2 - NT) % NT, k);
#pragma unroll
                                for (int i = 0; i < NP; ++i) { wf[st][2 * k][i] = p[i * WS_LO]; wf[st][2 * k + 1][i] = p[i * WS_LO + 64]; }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (hb + 1 < NT) da = dn;
                        }
                    }
                } else {
                uint4 wf[2 * NT][NP];                              // W fragments of one hidden block (every piece), refreshed in place
                    if (mine) {
                        const int b0 = __ffsll((long long)mine) - 1;
    #pragma unroll
                        for (int k = 0; k < NT; ++k) {
                            const uint4* p = wptr(b0, 0, k);
    #pragma unroll
                            for (int i = 0; i < NP; ++i) { wf[2 * k][i] = p[i * WS_LO]; wf[2 * k + 1][i] = p[i * WS_LO + 64]; }
                        }
                    }
    #pragma clang loop unroll(disable)
                    while (mine) {
                        const int b = __ffsll((long long)mine) - 1;
                        mine &= mine - 1;
                        const int nb = mine ? __ffsll((long long)mine) - 1 : b;
                        uint4 mf[2];
                        const unsigned m32 = masks[c31 * LDM + b];
    #pragma unroll
                        for (int jg = 0; jg < 2; ++jg) {
                            const unsigned bits = (m32 >> (16 * jg + 8 * hi)) & 0xffu;
                            const uint2 l0 = lut[bits & 15u], l1 = lut[bits >> 4];
                            mf[jg] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                        }
    #pragma unroll
                        for (int hb = 0; hb < NT; ++hb) {
                            // link 1: P_b^T[hidden block hb] = (sum of h's pieces)^T . M_b^T  (the 0/1 mask is exact in bf16: NP MFMAs per
                            // chunk, smallest piece first)
                            f32x16 da = zero16();
                            const u16* hp = Ht + (hb * 32 + c31) * LDT + 8 * hi;
    #pragma unroll
                            for (int jg = 0; jg < 2; ++jg) {
    #pragma unroll
                                for (int i = NP - 1; i >= 0; --i) da = mfma16(*reinterpret_cast<const uint4*>(hp + i * TLO + 16 * jg), mf[jg], da);
                            }
                            const FragP<NP> p0 = split8<NP>(da[0], da[1], da[2], da[3], da[4], da[5], da[6], da[7]);
                            const FragP<NP> p1 = split8<NP>(da[8], da[9], da[10], da[11], da[12], da[13], da[14], da[15]);
    #pragma unroll
                            for (int k = 0; k < NT; ++k) {            // slot k's fragments are re-requested right after their last use
                                soc[k] = mfma_xp<NP>(p0.p, wf[2 * k], soc[k]);
                                soc[k] = mfma_xp<NP>(p1.p, wf[2 * k + 1], soc[k]);
                                const uint4* p = (hb + 1 < NT) ? wptr(b, hb + 1, k) : wptr(nb, 0, k);
    #pragma unroll
                                for (int i = 0; i < NP; ++i) { wf[2 * k][i] = p[i * WS_LO]; wf[2 * k + 1][i] = p[i * WS_LO + 64]; }
                            }
                            __builtin_amdgcn_sched_barrier(0);         // one hidden block at a time: keeps the live set to one chain result
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                // fixed-order sum of the partial tiles: round s hands slot s to the wave s column blocks further on; rounds
                // alternate between the two slot sets, one barrier per round
                TICKX(3)
                if (om) {                                          // (workgroup-uniform)
                    __syncthreads();                               // every wave is done reading Ht: it now carries exchange set 0
#pragma unroll
                    for (int sft = 1; sft < NT; ++sft) {
                        float* ex = ((sft - 1) & 1) ? EX1 : EX0;
                        float4* dst = reinterpret_cast<float4*>(ex + (size_t)cb * 1024) + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            dst[q * 64] = make_float4(soc[sft][4 * q], soc[sft][4 * q + 1], soc[sft][4 * q + 2], soc[sft][4 * q + 3]);
                        __syncthreads();
                        const float4* src = reinterpret_cast<const float4*>(ex + (size_t)((cb + NT - sft) % NT) * 1024) + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = src[q * 64];
                            soc[0][4 * q] += v.x; soc[0][4 * q + 1] += v.y; soc[0][4 * q + 2] += v.z; soc[0][4 * q + 3] += v.w;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (TRAIN) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (sv_ok(4 * q + e))
                                sv_x_t[(unsigned)(((arow + e + 8 * q) * a.T + t) * E + EV + C + col)] = fmaxf(soc[0][4 * q + e] + bso, 0.f);
                    }
                    unsigned pa[NP], pb[NP];
                    put4(Xb + (arow + 8 * q) * LDXB + EV + C + col, LDXB, XLO, fmaxf(soc[0][4 * q] + bso, 0.f), fmaxf(soc[0][4 * q + 1] + bso, 0.f),
                         fmaxf(soc[0][4 * q + 2] + bso, 0.f), fmaxf(soc[0][4 * q + 3] + bso, 0.f), pa, pb);
                }
            }
            TICKX(4)
            __syncthreads();
            TICKX(5)
            // ---- P4: gates over [x | h], and the candidate's x part (same A fragments: three n-tiles per LDS read) ----
            // B fragments run through a ring of RD4 k-groups, requested RD4 groups (~ 850 matrix cycles) before their use
            f32x16 u, ac = zero16();
            {
                f32x16 g0 = zero16(), g1 = zero16();
                // wave-uniform bases (scalar registers) + the lane as a 32-bit offset: no per-fragment address registers
                // (the opaque zero is redefined every step: the 100+ fragment addresses below are cheap to form and must not be
                // hoisted out of the time loop, where they would sit in spilled registers)
                int z4;
                asm volatile("s_mov_b32 %0, 0" : "=s"(z4));
                const uint4* wg0 = Wg + ((size_t)cb * G16) * 64 + z4;
                const uint4* wg1 = Wg + ((size_t)(cb + NT) * G16) * 64 + z4;
                const uint4* wcx = Wc + ((size_t)cb * G16) * 64 + z4;
                const unsigned ul = (unsigned)lane;
                uint4 rb[RD4][3][NP];
                auto req = [&](int g) {                            // (g is a compile-time constant after unrolling)
                    const int sl = g % RD4;
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        rb[sl][0][i] = (wg0 + i * WG_LO + g * 64)[ul];
                        rb[sl][1][i] = (wg1 + i * WG_LO + g * 64)[ul];
                        if (g < GX16) rb[sl][2][i] = (wcx + i * WC_LO + g * 64)[ul];
                    }
                };
#pragma unroll
                for (int g = 0; g < RD4; ++g) req(g);
#pragma unroll
                for (int g = 0; g < G16; ++g) {
                    const int sl = g % RD4;
                    uint4 av[NP];
#pragma unroll
                    for (int i = 0; i < NP; ++i) av[i] = *reinterpret_cast<const uint4*>(xp + i * XLO + g * 16);
#pragma unroll
                    for (int pr = 0; pr < Pairs<NP>::N; ++pr) {     // smallest products first, the three n-tiles side by side
                        const int pa = Pairs<NP>::A[pr], pb = Pairs<NP>::B[pr];
                        g0 = mfma16(av[pa], rb[sl][0][pb], g0); g1 = mfma16(av[pa], rb[sl][1][pb], g1);
                        if (g < GX16) ac = mfma16(av[pa], rb[sl][2][pb], ac);
                    }
                    if (g + RD4 < G16) req(g + RD4);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float rhv[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float r = sigmoidf_(g0[4 * q + e] + bgr);
                        rhv[e] = r * h[4 * q + e];
                        u[4 * q + e] = sigmoidf_(g1[4 * q + e] + bgu);
                        if (TRAIN && sv_ok(4 * q + e)) { sv_r_t[sv_off(4 * q + e, t)] = r; sv_u_t[sv_off(4 * q + e, t)] = u[4 * q + e]; }
                    }
                    unsigned pa[NP], pb[NP];
                    put4(RHb + (arow + 8 * q) * LDRB + col, LDRB, RLO, rhv[0], rhv[1], rhv[2], rhv[3], pa, pb);
                }
            }
            // the candidate's r*h part: all of its B fragments are requested before the barrier
            uint4 chp[GH16][NP];
            {
                int z5;
                asm volatile("s_mov_b32 %0, 0" : "=s"(z5));
                const uint4* wch = Wc + ((size_t)cb * G16 + GX16) * 64 + z5;
                const unsigned ul = (unsigned)lane;
#pragma unroll
                for (int g = 0; g < GH16; ++g)
#pragma unroll
                    for (int i = 0; i < NP; ++i) chp[g][i] = (wch + i * WC_LO + g * 64)[ul];
            }
            TICKX(6)
            __syncthreads();
            TICKX(7)
            // ---- P5: candidate += (r*h) part, blend, score; publish h_t ----
            {
#pragma unroll
                for (int g = 0; g < GH16; ++g) {
                    uint4 av[NP];
#pragma unroll
                    for (int i = 0; i < NP; ++i) av[i] = *reinterpret_cast<const uint4*>(rp + i * RLO + g * 16);
                    ac = mfma_xp<NP>(av, chp[g], ac);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float c = tanhf_(ac[i] + bcc);
                    h[i] = gru_blend(u[i], h[i], c);
                    sp[i] = fmaf(h[i], wsc, sp[i]);
                    if (TRAIN && sv_ok(i)) { sv_c_t[sv_off(i, t)] = c; sv_h_t[sv_off(i, t)] = h[i]; }
                }
                publish_h(h);                              // h slots of Xb / Ht were last read before the previous barrier
            }
            if (tid < TM) {
                pp[tid * 2] = pc[tid * 2]; pp[tid * 2 + 1] = pc[tid * 2 + 1];
                pc[tid * 2] = ynext.x; pc[tid * 2 + 1] = ynext.y;
            }
            for (int i = tid; i < TM * LDM; i += NTHR) masks[i] = 0u;
            if (tid < 2) occ[tid] = 0;
            TICKX(8)
            __syncthreads();
            TICKX(9)
        }
        // ---- score ----
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = sp[i];
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
            if (c31 == 0) red[cb * TM + arow + (i & 3) + 8 * (i >> 2)] = v;
        }
        __syncthreads();
        asm volatile("s_mov_b32 %0, %1" : "=s"(row0p) : "s"(row0));
        if (tid < TM && row0p + tid < a.R && it == a.iters - 1) {
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < NT; ++c) sc += red[c * TM + tid];
            a.score[row0p + tid] = sc + (float)a.T * a.b_score[0];
        }
        // ---- regression: Y += h_T W_r + b_r ----
        for (int nt = cb; nt < a.NTreg; nt += NT) {
            f32x16 acc[1] = {zero16()};
            const uint4* br[1] = {Wreg + ((size_t)nt * GH16) * 64 + lane};
            mmax_groups<1, NP>(acc, xp + E, XLO, br, WR_LO, GH16);
            const int cc = nt * 32 + c31;
            if (cc < 2 * a.T) {
                const float bb = a.b_reg[cc];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = row0p + arow + (i & 3) + 8 * (i >> 2);
                    if (row < a.R) { float* y = a.Y + (size_t)row * 2 * a.T + cc; *y = *y + (acc[0][i] + bb); }
                }
            }
        }
        __syncthreads();
    }
#ifdef DESIRE_IOC_TIMING
    if (a.dbg && blockIdx.x == 7 && tid == 0)
        for (int k = 0; k < 10; ++k) a.dbg[k] = tacc[k];
#endif
}

static size_t iocx3_lds(const IocArgs& a, int np = 2) {
    const int H = a.H, TM = 32, KX = 16 + 32 + 2 * H, B = a.G * a.G, NT = H / 32;
    size_t b = np * ((size_t)TM * (KX + 8) * 2 + (size_t)TM * (H + 8) * 2 + (size_t)H * (TM + 8) * 2);
    b += (size_t)((TM * (B + 1) + 1) & ~1) * 4 + 16 * 8 + (size_t)TM * 4 * 4 + 3 * 16 * 4 + (size_t)NT * TM * 4 + TM + 16;
    return b;
}
bool ioc_x3_supported(int mno, int H, int bins) { return mno >= 1 && mno <= 32 && 32 % mno == 0 && (H == 64 || H == 128) && bins <= 64; }
template <int H>
static void launch_x3(const IocArgs& a, hipStream_t s) {
    const dim3 grid((a.R + 31) / 32), block((H / 32) * 64);
    if (a.sv_h) {                                              // training-mode forward
        if (a.gpt > 0) {                                       // padded tiles
            allow_big_lds(k_ioc_x3<H, 16, 32, true, 2, true>);
            hipLaunchKernelGGL((k_ioc_x3<H, 16, 32, true, 2, true>), grid, block, iocx3_lds(a), s, a);
            return;
        }
        allow_big_lds(k_ioc_x3<H, 16, 32, true>);
        hipLaunchKernelGGL((k_ioc_x3<H, 16, 32, true>), grid, block, iocx3_lds(a), s, a);
    } else {
        if (a.gpt > 0) {                                      // padded tiles
            allow_big_lds(k_ioc_x3<H, 16, 32, false, 2, true>);
            hipLaunchKernelGGL((k_ioc_x3<H, 16, 32, false, 2, true>), grid, block, iocx3_lds(a), s, a);
            return;
        }
        allow_big_lds(k_ioc_x3<H, 16, 32, false>);
        hipLaunchKernelGGL((k_ioc_x3<H, 16, 32, false>), grid, block, iocx3_lds(a), s, a);
    }
}
void launch_ioc_x3(const IocArgs& a, hipStream_t s) {
    // groups of 64 agents: 64-row tiles with two row blocks per wave on fp32 LDS tiles split on the fly (kernels_x6r2.hip with two pieces:
    // every weight fragment used twice).  For groups of <= 32 agents that form was measured slower (30.7 vs 29.1 ms at 512 windows -- with
    // two pieces the 32-row tiles' second workgroup per CU is worth more than the halved weight stream), so it serves only the shape
    // it alone can
    if (!a.sv_h && a.mno > 32 && ioc_x6r2_supported(a.mno, a.H, a.G * a.G)) { launch_ioc_x3r2(a, s); return; }
    if (a.H == 128) launch_x3<128>(a, s); else launch_x3<64>(a, s);
}
// three pieces per operand, six products per fp32 product (dims.bf16 = 3): same shapes, inference only
template <int H>
static void launch_x6(const IocArgs& a, hipStream_t s) {
    const dim3 grid((a.R + 31) / 32), block((H / 32) * 64);
    allow_big_lds(k_ioc_x3<H, 16, 32, false, 3>);
    hipLaunchKernelGGL((k_ioc_x3<H, 16, 32, false, 3>), grid, block, iocx3_lds(a, 3), s, a);
}
void launch_ioc_x6(const IocArgs& a, hipStream_t s) {
    // default: 64-row tiles, two row blocks per wave, fp32 operand tiles split on the fly (kernels_x6r2.hip; results within 1-2 ulp);
    // a.variant == 13 (DESIRE_IOC_X6_TILE32) keeps the 32-row / three-image form below (A/B), which also serves the shapes whose masks do not fit beside a
    // 64-row tile -- and launches that would leave CUs idle with 64-row tiles (a few windows: one window = 20 tiles of 32 rows on 20 CUs
    // takes 1.29 ms, 10 tiles of 64 rows 2.39 ms)
    if ((a.mno > 32 || (a.variant != 13 && ((a.R + 63) / 64 >= 256 || a.variant == 14))) && ioc_x6r2_supported(a.mno, a.H, a.G * a.G)) { launch_ioc_x6r2(a, s); return; }    // (14: always, A/B and tests)
    if (a.H == 128) launch_x6<128>(a, s); else launch_x6<64>(a, s);
}
