// kernels_bf16.hip -- bf16-operand / fp32-accumulate forms of the recurrent hot kernels (BASELINE configs[2]).
//
// v_mfma_f32_32x32x16_bf16 runs at 16x the fp32 matrix rate, so the fp32 kernels' structure (operands staged through
// LDS by VALU phases between barriers) would leave the matrix pipe idle: this is a re-tiling, not a type swap.
//   * recurrent state h, gate math and all accumulators stay fp32 in registers; only MFMA OPERANDS are bf16
//   * social pooling never touches LDS or a barrier per bin: for bin b and a 32-row tile
//         P_b^T [hidden, row i] = H^T [hidden, j] . M_b^T [j, i]            (MFMA 1: A = h^T from LDS, B = 0/1 bits)
//     lands in registers with lane = row i and 4-row-aligned runs of the hidden index -- which IS the A-fragment
//     layout (lane = row, 8 k per lane) of the next contraction up to a permutation of k, and the permutation is
//     absorbed into how W_b is packed (chain order, pack_b16 in api.hip):
//         e_r += P_b . W_b                                                  (MFMA 2: A = cvt(P_b^T regs), B = packed W_b)
//     every wave redoes MFMA 1 for all hidden blocks (the pipe has the headroom); 4 barriers per step instead of ~36
//   * the neighbour bits of (row, bin) expand to bf16 0/1 B fragments through a 16-entry LDS table (one nibble ->
//     four bf16), conflict-free by construction
// Layout reminders (see common.h): accumulator element i of a lane = row (i&3) + 8(i>>2) + 4(lane>>5), col lane&31.
// A/B fragment of the bf16 MFMA: lane (c = lane&31, hi = lane>>5) holds 8 values for k-slot (hi, 0..7); A and B only
// have to agree on which logical k a slot means, so any consistent k order is valid.
#include "common.h"
#include "kernels.h"

#include "bf16.h"

// ------------------------------------------------------------------------------------------------------------------
// IOC scoring / refinement, bf16 operands.  Tile = 32*WM rows = whole (scene,k) groups (mno divides 32 with WM = 1, or
// mno = 64 with WM = 2); workgroup = WM * H/32 waves, wave (mt, cb) owns rows [32mt, 32mt+32) x hidden columns
// [32cb, 32cb+32).  Weight pointers of IocArgs (Wg, Wc, Wsoc, Wreg) point at the bf16 packs ("ioc/*16" in api.hip).
// ------------------------------------------------------------------------------------------------------------------
#ifdef DESIRE_IOC_TIMING
#define TICK16(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#else
#define TICK16(k)
#endif
template <int H, int EV, int C, int WM, bool WIDE = (WM > 1)>      // WIDE: one 64-agent group spans both row blocks (compile-time: branch-free chains)
__global__ __launch_bounds__((H / 32) * WM * 64, ((H / 32) * WM <= 4) ? IOC16_OCC : 1) void k_ioc_bf16(IocArgs a) {
#ifdef DESIRE_IOC_TIMING
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = H >> 5, TM = 32 * WM, E = EV + C + H, KX = E + H;
    constexpr int LDXB = KX + 8, LDRB = H + 8, LDT = TM + 8;          // bf16 elements; (ld/2) = 4 mod 8 dwords: conflict-free b128
    constexpr int NTHR = NT * WM * 64, TPR = NTHR / TM;
    constexpr int G16 = KX >> 4, GX16 = E >> 4, GH16 = H >> 4;
    constexpr int JGM = 2 * WM;                                       // most 16-neighbour chunks a row can have
    const int B = a.G * a.G, LDM = B + 1;
    u16* Xb = reinterpret_cast<u16*>(smem_raw);                       // [TM][LDXB]  e_v | e_s | e_r | h
    u16* RHb = Xb + TM * LDXB;                                        // [TM][LDRB]  r * h
    u16* Ht = RHb + TM * LDRB;                                        // [H][LDT]    h transposed (pooling operand)
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(Ht + H * LDT);   // [TM][B+1], bit = local row
    uint2* lut = reinterpret_cast<uint2*>(masks + TM * LDM);          // [16] nibble -> 4 bf16 (0.0 / 1.0)
    float* pc = reinterpret_cast<float*>(lut + 16);                   // [TM][2]
    float* pp = pc + TM * 2;                                          // [TM][2]
    float* wv = pp + TM * 2;                                          // [3][EV]
    float* red = wv + 3 * EV;                                         // [NT][TM]
    unsigned char* vld = reinterpret_cast<unsigned char*>(red + NT * TM);   // [TM]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + TM);                  // [2] bins that hold a neighbour anywhere in the tile
    constexpr bool SPLIT = IOC16_SPLIT && NT <= 4;                          // pooling split over BINS between the waves of a row block
    float* EX = reinterpret_cast<float*>(smem_raw + ((reinterpret_cast<unsigned char*>(occ + 2) - smem_raw + 15) & ~15));   // [WM][NT][1024]
    float* EXB = EX + WM * NT * 1024;                                       // [WM * NT / 2][1024] second set's upper half (B <= 32)

    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int cb = w % NT, mt = w / NT;
    const int hi = lane >> 5, c31 = lane & 31;
    const int row0 = blockIdx.x * TM;
    IOC_DYN(a)                                          // (a slot class counted on the device: kernels.h DynCount; the grid is the worst case's)
    if (a.dyn.cnt && row0 >= a.R) return;
    const int col = cb * 32 + c31;
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int my_row = min(row0 + r8, a.R - 1);
    const int my_scene = my_row / (a.K * a.mno);
    const int grp_base = (r8 / a.mno) * a.mno;
    const int my_slot = r8 - grp_base;
    constexpr bool wide = WIDE;                                       // one group spans both row blocks
    constexpr int JG = wide ? JGM : 2;                                // 16-wide neighbour chunks of this wave's rows
    const int jbase = wide ? 0 : mt * 32;                             // first local row its neighbours can have

    for (int i = tid; i < 3 * EV; i += NTHR) wv[i] = (i < 2 * EV) ? a.w_vel[i] : a.b_vel[i - 2 * EV];
    if (tid < 16) {
        const unsigned lo = ((tid & 1) ? 0x3F80u : 0u) | ((tid & 2) ? 0x3F800000u : 0u);
        const unsigned hi2 = ((tid & 4) ? 0x3F80u : 0u) | ((tid & 8) ? 0x3F800000u : 0u);
        lut[tid] = make_uint2(lo, hi2);
    }
    if (tid < TM) vld[tid] = a.valid[agent_of_row(min(row0 + tid, a.R - 1), a.K, a.mno)];
    const float bgr = a.b_g[col], bgu = a.b_g[H + col], bcc = a.b_c[col], bso = a.b_soc[col], wsc = a.w_score[col];
    const float* grid = a.grids + (size_t)a.grid_of_scene[my_scene] * a.Gh * a.Gw * C;
    const uint4* Wg = reinterpret_cast<const uint4*>(a.Wg);
    const uint4* Wc = reinterpret_cast<const uint4*>(a.Wc);
    const uint4* Wsoc = reinterpret_cast<const uint4*>(a.Wsoc);
    const uint4* Wreg = reinterpret_cast<const uint4*>(a.Wreg);

    const u16* xp[1] = {Xb + (mt * 32 + c31) * LDXB + 8 * hi};
    const u16* rp[1] = {RHb + (mt * 32 + c31) * LDRB + 8 * hi};
    const int arow = mt * 32 + 4 * hi;                                // + (i&3) + 8(i>>2): local row of accumulator element i
    // h (fp32, accumulator layout) -> both bf16 images
    auto publish_h = [&](const f32x16& h) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Xb[(arow + (i & 3) + 8 * (i >> 2)) * LDXB + E + col] = bf16_of(h[i]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint2*>(Ht + col * LDT + arow + 8 * q) =
                make_uint2(pk_bf16(h[4 * q], h[4 * q + 1]), pk_bf16(h[4 * q + 2], h[4 * q + 3]));
    };

    for (int it = 0; it < a.iters; ++it) {
        // an opaque zero, redefined per pass: the prologue / epilogue address math below depends on it, so it cannot be hoisted
        // out of the pass and sit in (spilled) registers across the whole time loop
        int row0p;
        asm volatile("s_mov_b32 %0, %1" : "=s"(row0p) : "s"(row0));
        f32x16 h, sp = zero16();
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = min(row0p + arow + (i & 3) + 8 * (i >> 2), a.R - 1);
            h[i] = a.Hx[(size_t)agent_of_row(row, a.K, a.mno) * a.ldhx + col];
        }
        __syncthreads();                                  // previous iteration's readers of Xb / Ht are done
        publish_h(h);
        float2 ynext = make_float2(0.f, 0.f);
        if (tid < TM) {
            const int row = min(row0 + tid, a.R - 1);
            const int ag = agent_of_row(row, a.K, a.mno);
            pp[tid * 2] = a.p_last[(size_t)ag * 2]; pp[tid * 2 + 1] = a.p_last[(size_t)ag * 2 + 1];
            const float2 y0 = *reinterpret_cast<const float2*>(a.Y + ((size_t)row * a.T) * 2);
            pc[tid * 2] = y0.x; pc[tid * 2 + 1] = y0.y;
        }
        for (int i = tid; i < TM * LDM; i += NTHR) masks[i] = 0ull;
        if (tid < 2) occ[tid] = 0;
        __syncthreads();

        for (int t = 0; t < a.T; ++t) {
            TICK16(0)
            if (tid < TM && t + 1 < a.T)
                ynext = *reinterpret_cast<const float2*>(a.Y + ((size_t)min(row0 + tid, a.R - 1) * a.T + t + 1) * 2);
            // ---- P1: e_v, e_s, neighbour bits (row threads) ----
            {
                const float px = pc[r8 * 2], py = pc[r8 * 2 + 1];
                const float vx = px - pp[r8 * 2], vy = py - pp[r8 * 2 + 1];
                for (int j = 2 * q8; j < EV; j += 2 * TPR) {
                    const float e0 = fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f);
                    const float e1 = fmaxf(fmaf(vy, wv[EV + j + 1], vx * wv[j + 1]) + wv[2 * EV + j + 1], 0.f);
                    *reinterpret_cast<unsigned*>(Xb + r8 * LDXB + j) = pk_bf16(e0, e1);
                }
                int cy, cx;
                scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
                const float* gsrc = grid + ((size_t)cy * a.Gw + cx) * C;
                for (int j = 4 * q8; j < C; j += 4 * TPR) {
                    const float4 g4 = *reinterpret_cast<const float4*>(gsrc + j);
                    *reinterpret_cast<uint2*>(Xb + r8 * LDXB + EV + j) = make_uint2(pk_bf16(g4.x, g4.y), pk_bf16(g4.z, g4.w));
                }
                float nbw, nbh;
                nb_opaque(a.nb_w, a.nb_h, nbw, nbh);
                const unsigned long long oc = nb_search<4>(pc, vld, grp_base, a.mno, q8, TPR, my_slot, px, py, nbw, nbh, a.G, a.bin_tab,
                                                          [&](int j, int b) { atomicOr(&masks[r8 * LDM + b], 1ull << (grp_base + j)); });
                nb_publish_occ(oc, occ, B);
            }
            TICK16(1)
            __syncthreads();
            TICK16(2)
            // ---- P2: social pooling chain -> e_r ----
            unsigned long long om_all = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
            om_all |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
            if constexpr (SPLIT) {
                // The occupied bins are dealt round-robin to the NT waves of a row block.  A wave runs the whole chain of ITS
                // bins -- link 1 once per hidden block (no longer repeated by every wave), link 2 into all NT column blocks --
                // and keeps NT partial e_r tiles; slot k belongs to column block (cb + k) % NT, so every register index is
                // static.  The partials are then summed in a fixed order through a 4 KB-per-wave LDS exchange (NT - 1 rounds).
                const unsigned long long om = om_all;
                unsigned long long mine = 0ull;
                {
                    int k = 0;
                    for (unsigned long long tmp = om; tmp; tmp &= tmp - 1, ++k)
                        if (k % NT == cb) mine |= tmp & (0ull - tmp);
                }
                f32x16 soc[NT];
                soc[0] = zero16();                                 // (biases are added after the contractions: a splat would pin 16 registers)
#pragma unroll
                for (int k = 1; k < NT; ++k) soc[k] = zero16();
                auto wptr = [&](int b, int hb, int k) {           // fragments of W_b[hidden block hb][column block (cb+k)%NT], 2 k-groups
                    const int cbo = (cb + k) % NT;
                    return Wsoc + ((size_t)(b * NT + cbo) * GH16 + 2 * hb) * 64 + lane;
                };
                uint4 wq[2 * NT];                                   // W fragments of one hidden block, refreshed in place
                if (mine) {
                    const int b0 = __ffsll((long long)mine) - 1;
#pragma unroll
                    for (int k = 0; k < NT; ++k) { const uint4* p = wptr(b0, 0, k); wq[2 * k] = p[0]; wq[2 * k + 1] = p[64]; }
                }
#pragma clang loop unroll(disable)
                while (mine) {
                    const int b = __ffsll((long long)mine) - 1;
                    mine &= mine - 1;
                    const int nb = mine ? __ffsll((long long)mine) - 1 : b;
                    uint4 mf[JGM];
                    const unsigned long long m64 = masks[(mt * 32 + c31) * LDM + b];
#pragma unroll
                    for (int jg = 0; jg < JGM; ++jg) {
                        if (jg < JG) {
                            const unsigned bits = (unsigned)(m64 >> (jbase + 16 * jg + 8 * hi)) & 0xffu;
                            const uint2 l0 = lut[bits & 15u], l1 = lut[bits >> 4];
                            mf[jg] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                        }
                    }
                    auto chain = [&](int hb) {
                        f32x16 d1 = zero16();
                        const u16* hp = Ht + (hb * 32 + c31) * LDT + jbase + 8 * hi;
#pragma unroll
                        for (int jg = 0; jg < JGM; ++jg)
                            if (jg < JG) d1 = mfma16(*reinterpret_cast<const uint4*>(hp + 16 * jg), mf[jg], d1);
                        return d1;
                    };
#pragma unroll
                    for (int hb = 0; hb < NT; ++hb) {
                        const f32x16 da = chain(hb);
                        const uint4 p0 = make_uint4(pk_bf16(da[0], da[1]), pk_bf16(da[2], da[3]), pk_bf16(da[4], da[5]), pk_bf16(da[6], da[7]));
                        const uint4 p1 = make_uint4(pk_bf16(da[8], da[9]), pk_bf16(da[10], da[11]), pk_bf16(da[12], da[13]), pk_bf16(da[14], da[15]));
#pragma unroll
                        for (int k = 0; k < NT; ++k) {            // slot k's fragments are re-requested right after their last use:
                            soc[k] = mfma16(p0, wq[2 * k], soc[k]);          // next hidden block of this bin, or block 0 of my next bin
                            soc[k] = mfma16(p1, wq[2 * k + 1], soc[k]);
                            const uint4* p = (hb + 1 < NT) ? wptr(b, hb + 1, k) : wptr(nb, 0, k);
                            wq[2 * k] = p[0]; wq[2 * k + 1] = p[64];
                        }
                        __builtin_amdgcn_sched_barrier(0);         // one hidden block at a time: keeps the live set to one chain result
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // fixed-order sum of the partial tiles: round s hands slot s to the wave s column blocks further on.  Rounds
                // alternate between two slot sets (the second one lives partly in the r*h tile, idle during P2), so one barrier
                // per round is enough; with more than 32 bins the masks leave no room for the second set and a round costs two.
                if (om) {                                          // (workgroup-uniform)
                    const bool two_sets = IOC16_TWO_SETS && B <= 32;
                    auto slot = [&](int set, int wv) {
                        if (set == 0 || !two_sets) return EX + (size_t)wv * 1024;
                        constexpr int nh = NT * WM / 2;
                        return wv < nh ? reinterpret_cast<float*>(RHb) + (size_t)wv * 1024 : EXB + (size_t)(wv - nh) * 1024;
                    };
#pragma unroll
                    for (int sft = 1; sft < NT; ++sft) {
                        const int set = (sft - 1) & 1;
                        if (sft > 1 && !two_sets) __syncthreads();  // single set: the previous round's readers must be done with my slot
                        float4* dst = reinterpret_cast<float4*>(slot(set, mt * NT + cb)) + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            dst[q * 64] = make_float4(soc[sft][4 * q], soc[sft][4 * q + 1], soc[sft][4 * q + 2], soc[sft][4 * q + 3]);
                        __syncthreads();
                        const float4* src = reinterpret_cast<const float4*>(slot(set, mt * NT + (cb + NT - sft) % NT)) + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float4 v = src[q * 64];
                            soc[0][4 * q] += v.x; soc[0][4 * q + 1] += v.y; soc[0][4 * q + 2] += v.z; soc[0][4 * q + 3] += v.w;
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    Xb[(arow + (i & 3) + 8 * (i >> 2)) * LDXB + EV + C + col] = bf16_of(fmaxf(soc[0][i] + bso, 0.f));
            } else
            {
                f32x16 soc = zero16();
                // this wave's n-tile of W_b (H/16 k-groups, chain order) lives in ONE register set that is refreshed in
                // place: the fragments of bin b+1 are requested right after their last use in bin b, so every load is in
                // flight for a whole bin of MFMAs
                // Only bins that hold a neighbour somewhere in the tile are visited (an empty bin contributes exact zeros).
                unsigned long long om = om_all;
                uint4 wb[2 * NT];
                if (om) {
                    const uint4* wsrc = Wsoc + ((size_t)((__ffsll((long long)om) - 1) * NT + cb) * GH16) * 64 + lane;
#pragma unroll
                    for (int g = 0; g < 2 * NT; ++g) wb[g] = wsrc[g * 64];
                }
#pragma clang loop unroll(disable)
                while (om) {
                    const int b = __ffsll((long long)om) - 1;
                    om &= om - 1;
                    const uint4* wnext = Wsoc + ((size_t)((om ? __ffsll((long long)om) - 1 : b) * NT + cb) * GH16) * 64 + lane;
                    uint4 mf[JGM];                                      // neighbour bits -> bf16 B fragments (16 neighbours each)
                    const unsigned long long m64 = masks[(mt * 32 + c31) * LDM + b];
#pragma unroll
                    for (int jg = 0; jg < JGM; ++jg) {
                        if (jg < JG) {
                            const unsigned bits = (unsigned)(m64 >> (jbase + 16 * jg + 8 * hi)) & 0xffu;
                            const uint2 l0 = lut[bits & 15u], l1 = lut[bits >> 4];
                            mf[jg] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                        }
                    }
                    // software pipeline over the hidden blocks: MFMA 1 of block hb+1 is issued before block hb's
                    // accumulators are converted and consumed, so the convert never waits on the matrix pipe
                    auto chain = [&](int hb) {
                        f32x16 d1 = zero16();
                        const u16* hp = Ht + (hb * 32 + c31) * LDT + jbase + 8 * hi;
#pragma unroll
                        for (int jg = 0; jg < JGM; ++jg)
                            if (jg < JG) d1 = mfma16(*reinterpret_cast<const uint4*>(hp + 16 * jg), mf[jg], d1);
                        return d1;
                    };
                    f32x16 da = chain(0), dn;
#pragma unroll
                    for (int hb = 0; hb < NT; ++hb) {
                        if (hb + 1 < NT) dn = chain(hb + 1);
                        const uint4 p0 = make_uint4(pk_bf16(da[0], da[1]), pk_bf16(da[2], da[3]), pk_bf16(da[4], da[5]), pk_bf16(da[6], da[7]));
                        const uint4 p1 = make_uint4(pk_bf16(da[8], da[9]), pk_bf16(da[10], da[11]), pk_bf16(da[12], da[13]), pk_bf16(da[14], da[15]));
                        soc = mfma16(p0, wb[2 * hb], soc);
                        soc = mfma16(p1, wb[2 * hb + 1], soc);
                        if (hb + 1 < NT) da = dn;
                        wb[2 * hb] = wnext[(2 * hb) * 64];
                        wb[2 * hb + 1] = wnext[(2 * hb + 1) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    Xb[(arow + (i & 3) + 8 * (i >> 2)) * LDXB + EV + C + col] = bf16_of(fmaxf(soc[i] + bso, 0.f));
            }
            TICK16(3)
            __syncthreads();
            TICK16(4)
            // ---- P4: gates over [x | h], and the candidate's x part (same A fragments: three n-tiles per LDS read) ----
            // B fragments run through a ring of RD k-groups requested that many groups before their use; their addresses are formed per
            // step from wave-uniform bases (the opaque zero keeps ~60 of them from being hoisted out of the time loop into spilled
            // registers).  (Kept inline in both bf16 IOC kernels: as a shared helper the same code spills three times as much.)
            f32x16 u, ac = zero16();
            {
                f32x16 g0 = zero16(), g1 = zero16();
                int z4;
                asm volatile("s_mov_b32 %0, 0" : "=s"(z4));
                const uint4* wg0 = Wg + ((size_t)cb * G16) * 64 + z4;
                const uint4* wg1 = Wg + ((size_t)(cb + NT) * G16) * 64 + z4;
                const uint4* wcx = Wc + ((size_t)cb * G16) * 64 + z4;
                const unsigned ul = (unsigned)lane;
                constexpr int RD = (NT * WM > 8) ? 2 : IOC16_RD;
                uint4 rb[RD][3];
                auto req = [&](int g) {                            // (g is a compile-time constant after unrolling)
                    const int sl = g % RD;
                    rb[sl][0] = (wg0 + g * 64)[ul]; rb[sl][1] = (wg1 + g * 64)[ul];
                    if (g < GX16) rb[sl][2] = (wcx + g * 64)[ul];
                };
#pragma unroll
                for (int g = 0; g < RD && g < G16; ++g) req(g);
#pragma unroll
                for (int g = 0; g < G16; ++g) {
                    const int sl = g % RD;
                    const uint4 av = *reinterpret_cast<const uint4*>(xp[0] + g * 16);
                    g0 = mfma16(av, rb[sl][0], g0); g1 = mfma16(av, rb[sl][1], g1);
                    if (g < GX16) ac = mfma16(av, rb[sl][2], ac);
                    if (g + RD < G16) req(g + RD);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float r = sigmoidf_(g0[i] + bgr);
                    RHb[(arow + (i & 3) + 8 * (i >> 2)) * LDRB + col] = bf16_of(r * h[i]);
                    u[i] = sigmoidf_(g1[i] + bgu);
                }
            }
            // the candidate's r*h part: all of its B fragments are requested before the barrier
            uint4 ch[GH16];
            {
                int z5;
                asm volatile("s_mov_b32 %0, 0" : "=s"(z5));
                const uint4* wch = Wc + ((size_t)cb * G16 + GX16) * 64 + z5;
                const unsigned ul = (unsigned)lane;
#pragma unroll
                for (int g = 0; g < GH16; ++g) ch[g] = (wch + g * 64)[ul];
            }
            TICK16(5)
            __syncthreads();
            TICK16(6)
            // ---- P5: candidate += (r*h) part, blend, score; publish h_t ----
            {
#pragma unroll
                for (int g = 0; g < GH16; ++g) ac = mfma16(*reinterpret_cast<const uint4*>(rp[0] + g * 16), ch[g], ac);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float c = tanhf_(ac[i] + bcc);
                    h[i] = gru_blend(u[i], h[i], c);
                    sp[i] = fmaf(h[i], wsc, sp[i]);
                }
                publish_h(h);                              // h slots of Xb / Ht were last read before the previous barrier
            }
            if (tid < TM) {
                pp[tid * 2] = pc[tid * 2]; pp[tid * 2 + 1] = pc[tid * 2 + 1];
                pc[tid * 2] = ynext.x; pc[tid * 2 + 1] = ynext.y;
            }
            for (int i = tid; i < TM * LDM; i += NTHR) masks[i] = 0ull;
            if (tid < 2) occ[tid] = 0;
            TICK16(7)
            __syncthreads();
            TICK16(8)
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
            f32x16 acc[1][1] = {{zero16()}};
            const u16* hp2[1] = {xp[0] + E};
            const uint4* br[1] = {Wreg + ((size_t)nt * GH16) * 64 + lane};
            mma16_groups<1, 1>(acc, hp2, br, GH16);
            const int cc = nt * 32 + c31;
            if (cc < 2 * a.T) {
                const float bb = a.b_reg[cc];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = row0p + arow + (i & 3) + 8 * (i >> 2);
                    if (row < a.R) { float* y = a.Y + (size_t)row * 2 * a.T + cc; *y = *y + (acc[0][0][i] + bb); }
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

static size_t ioc16_lds(const IocArgs& a, int WM) {
    const int H = a.H, TM = 32 * WM, E = 16 + 32 + H, KX = E + H, B = a.G * a.G, NT = H / 32;
    size_t b = (size_t)TM * (KX + 8) * 2 + (size_t)TM * (H + 8) * 2 + (size_t)H * (TM + 8) * 2;
    b += (size_t)TM * (B + 1) * 8 + 16 * 8 + (size_t)TM * 4 * 4 + 3 * 16 * 4 + (size_t)NT * TM * 4 + TM + 64;
    if (IOC16_SPLIT && NT <= 4) b += (size_t)WM * NT * 4096 + 16 + (B <= 32 ? (size_t)WM * NT * 2048 : 0);   // partial-tile exchange
    return b;
}
template <int H, int WM>
static void launch16(const IocArgs& a, hipStream_t s) {
    const int TM = 32 * WM;
    const dim3 grid((a.R + TM - 1) / TM), block((H / 32) * WM * 64);
    if (WM > 1 && a.mno <= 32) {                              // (64-row tiles forced onto small groups: a.variant == 2)
        allow_big_lds(k_ioc_bf16<H, 16, 32, WM, false>);
        hipLaunchKernelGGL((k_ioc_bf16<H, 16, 32, WM, false>), grid, block, ioc16_lds(a, WM), s, a);
    } else {
        allow_big_lds(k_ioc_bf16<H, 16, 32, WM>);
        hipLaunchKernelGGL((k_ioc_bf16<H, 16, 32, WM>), grid, block, ioc16_lds(a, WM), s, a);
    }
}
// mno must divide 32 (32-row tiles, two workgroups per CU at H <= 128) or be 64 (64-row tiles, twice the waves);
// a.variant == 2 forces 64-row tiles (A/B)
void launch_ioc_bf16(const IocArgs& a, hipStream_t s) {
    // (64-row tiles with two row blocks per wave -- half the weight bytes per row, one workgroup per CU -- were bit-identical and SLOWER,
    //  3.67 vs 3.20 ms per 81 920 rows: one wave per SIMD leaves the position-only phase, the exchange and the epilogues uncovered; removed)
    const bool two = a.mno > 32 || a.variant == 2;
    if (a.H == 128) { if (two) launch16<128, 2>(a, s); else launch16<128, 1>(a, s); }
    else if (a.H == 64) { if (two) launch16<64, 2>(a, s); else launch16<64, 1>(a, s); }
    else { if (two) launch16<256, 2>(a, s); else launch16<256, 1>(a, s); }
}

// ------------------------------------------------------------------------------------------------------------------
// CVAE decoder convolutions with bf16 operands (same tilings as kernels_conv.hip; activations stay fp32 in HBM and are
// rounded on their way into fragments, accumulation and the BN/ELU epilogue are fp32).  a.Wp points at the bf16 packs.
// ------------------------------------------------------------------------------------------------------------------
// deconv2: [n,4,4,128] -> [n,8,8,64], 5x5 VALID stride 1, scatter form; wave = (co-half hf, sample pair sp), M-tile rows
// = (sample, input pixel); A (K = 128) is 8 register fragments per lane; 8 MFMAs per tap, then the plain LDS scatter.
__global__ __launch_bounds__(DS_WG) void k_deconv2_bf16(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float out_s16[];   // [4][64 px][64 co]
    const int lane = lane_id(), w = wave_id();
    const int hf = w & 1, sp = w >> 1;
    const int s0 = blockIdx.x * 4 + sp * 2;
    DYN_N(a, n, blockIdx.x * 4)
    const int c = lane & 31, hi = lane >> 5;
    float* my = out_s16 + (sp * 2) * 4096;
    // this wave's region = [2 samples x 64 px] x its 32 output channels; 8 lanes x float4 cover one row of it
    const int er = lane >> 3, ec = hf * 32 + (lane & 7) * 4;
    for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(my + (i * 8 + er) * 64 + ec) = make_float4(0.f, 0.f, 0.f, 0.f);
    uint4 af[8];
    {
        const int row = lane & 31;
        const int smp = min(s0 + (row >> 4), a.n - 1);
        const float* src = a.in + ((size_t)smp * 16 + (row & 15)) * 128 + 8 * hi;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 x0 = *reinterpret_cast<const float4*>(src + g * 16), x1 = *reinterpret_cast<const float4*>(src + g * 16 + 4);
            af[g] = make_uint4(pk_bf16(x0.x, x0.y), pk_bf16(x0.z, x0.w), pk_bf16(x1.x, x1.y), pk_bf16(x1.z, x1.w));
        }
    }
    const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
    auto load_b = [&](uint4 (&b)[8], int tap) {
        const uint4* bp = Wp + ((size_t)(tap * 2 + hf) * 8) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) b[g] = bp[g * 64];
    };
    auto do_tap = [&](const uint4 (&b)[8], int tap) {
        const int ky = tap / 5, kx = tap - ky * 5;
        f32x16 acc = zero16();
#pragma unroll
        for (int g = 0; g < 8; ++g) acc = mfma16(af[g], b[g], acc);
        // the 16 targets of a lane are distinct (different input pixels, same tap) and no other lane touches its column:
        // read all, then add and write all -- written as one dependent chain the compiler serialises 16 LDS round trips
        float* dst[16]; float old[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = (i & 3) + 8 * (i >> 2) + 4 * hi;
            const int s = rr >> 4, p = rr & 15;
            const int o = ((p >> 2) + ky) * 8 + (p & 3) + kx;
            dst[i] = my + (s * 64 + o) * 64 + hf * 32 + c;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) old[i] = *dst[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) *dst[i] = old[i] + acc[i];
    };
    uint4 b0[8], b1[8];
    load_b(b0, 0);
#pragma clang loop unroll(disable)
    for (int tap = 0; tap < 24; tap += 2) {
        load_b(b1, tap + 1);
        __builtin_amdgcn_sched_barrier(0);
        do_tap(b0, tap);
        __builtin_amdgcn_sched_barrier(0);
        load_b(b0, tap + 2);
        __builtin_amdgcn_sched_barrier(0);
        do_tap(b1, tap + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    do_tap(b0, 24);
    const float4 sc4 = *reinterpret_cast<const float4*>(a.scale + ec), sh4 = *reinterpret_cast<const float4*>(a.shift + ec);
    for (int i = 0; i < 16; ++i) {
        const int sp_px = i * 8 + er;                                  // 0..127 = (sample, pixel)
        const int smp = s0 + (sp_px >> 6);
        if (smp < a.n) {
            const float4 v = *reinterpret_cast<const float4*>(my + sp_px * 64 + ec);
            const size_t ix = ((size_t)smp * 64 + (sp_px & 63)) * 64 + ec;
            float4 o;
            o.x = eluf_(v.x * sc4.x + sh4.x); o.y = eluf_(v.y * sc4.y + sh4.y);      // inference only: BN + ELU
            o.z = eluf_(v.z * sc4.z + sh4.z); o.w = eluf_(v.w * sc4.w + sh4.w);
            *reinterpret_cast<float4*>(a.out + ix) = o;
        }
    }
}
// (Measured and dropped: two sample pairs per wave -- half the weight-fragment stream, 1.24 vs 0.95 ms: with two waves per workgroup nothing
// covers the 16-deep LDS read-add-write scatter after every tap.)
void launch_deconv2_bf16(const ConvArgs& a, hipStream_t s) {
    allow_big_lds(k_deconv2_bf16);
    hipLaunchKernelGGL(k_deconv2_bf16, dim3((a.n + 3) / 4), dim3(DS_WG), 4 * 4096 * sizeof(float), s, a);
}

// deconv3: [n,8,8,64] -> [n,16,16,32], 5x5 SAME stride 2, output-parity gather (see k_deconv3); one wave per sample,
// the sample's input staged in LDS as bf16, K = 64 = 4 fragments per tap, next tap's B fragments in flight.
__global__ __launch_bounds__(DS_WG) void k_deconv3_bf16(ConvArgs a) {
    constexpr int LDP = 72;                                            // bf16 elements; 36 dwords = 4 mod 8
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_d3[];
    u16* zero_row = reinterpret_cast<u16*>(smem_d3);
    u16* in_s = zero_row + LDP;                                        // [4][64][LDP]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int s0 = blockIdx.x * 4;
    DYN_N(a, n, s0)
    for (int i = tid; i < LDP / 2; i += DS_WG) reinterpret_cast<unsigned*>(zero_row)[i] = 0u;
    for (int i = tid; i < 4 * 64 * 8; i += DS_WG) {
        const int pix = i >> 3, c8 = i & 7;
        const int smp = s0 + (pix >> 6);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (smp < a.n) {
            const float* src = a.in + ((size_t)s0 * 64 + pix) * 64 + c8 * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
            v = make_uint4(pk_bf16(x0.x, x0.y), pk_bf16(x0.z, x0.w), pk_bf16(x1.x, x1.y), pk_bf16(x1.z, x1.w));
        }
        *reinterpret_cast<uint4*>(in_s + pix * LDP + c8 * 8) = v;
    }
    __syncthreads();
    const int smp = s0 + w;
    const u16* mine = in_s + w * 64 * LDP;
    const int c = lane & 31, hi = lane >> 5;
    const float sc = a.scale[c], sh = a.shift[c];
    const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
    int qy[2], qx[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { const int q = m * 32 + c; qy[m] = q >> 3; qx[m] = q & 7; }
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            f32x16 acc[2] = {zero16(), zero16()};
            const int ny = py ? 3 : 2, nx = px ? 3 : 2, ntap = ny * nx;
            auto tap_of = [&](int t) { const int iy = t / nx, ix = t - iy * nx; return (1 - py + 2 * iy) * 5 + (1 - px + 2 * ix); };
            uint4 bc[4], bn[4];
            {
                const uint4* bp = Wp + ((size_t)tap_of(0) * 4) * 64 + lane;
#pragma unroll
                for (int g = 0; g < 4; ++g) bc[g] = bp[g * 64];
            }
#pragma clang loop unroll(disable)
            for (int t = 0; t < ntap; ++t) {
                const int tap = tap_of(t), ky = tap / 5, kx = tap - ky * 5;
                {
                    const uint4* bp = Wp + ((size_t)tap_of(min(t + 1, ntap - 1)) * 4) * 64 + lane;
#pragma unroll
                    for (int g = 0; g < 4; ++g) bn[g] = bp[g * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;   // exact: numerators even
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int iy = qy[m] + dy, ix = qx[m] + dx;
                    const bool ok = iy >= 0 && iy < 8 && ix >= 0 && ix < 8;
                    const u16* ap = (ok ? mine + (iy * 8 + ix) * LDP : zero_row) + 8 * hi;
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[m] = mfma16(*reinterpret_cast<const uint4*>(ap + 16 * g), bc[g], acc[m]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) bc[g] = bn[g];
            }
            if (smp < a.n) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int q = m * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                        const int oy = 2 * (q >> 3) + py, ox = 2 * (q & 7) + px;
                        const size_t ix = ((size_t)smp * 256 + oy * 16 + ox) * 32 + c;
                        a.out[ix] = eluf_(acc[m][i] * sc + sh);
                    }
            }
        }
}
void launch_deconv3_bf16(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (72 + 4 * 64 * 72) * sizeof(u16);
    hipLaunchKernelGGL(k_deconv3_bf16, dim3((a.n + 3) / 4), dim3(DS_WG), lds, s, a);
}

// ------------------------------------------------------------------------------------------------------------------
// GRU decoder + head with bf16 recurrent operands.  32-row tiles, wave cb = hidden columns [32cb, 32cb+32).
// The constant-input half (x_z Wx, once per tile) stays on the exact fp32 pipe; per step only h Whg and (r*h) Whc run, as
// 16 + 8 bf16 MFMAs per wave.  h lives in fp32 registers; LDS holds a bf16 image of h (next step's operand), a bf16
// image of r*h, and an fp32 image of h for the 2-column head, which stays an fp32 VALU dot product (trajectory
// coordinates come straight out of it).  Two barriers per step (the fp32 kernel needs four: it reuses one operand tile).
// a.Whg / a.Whc point at the bf16 packs.
// ------------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__((H / 32) * 64, (H / 32) <= 4 ? 2 : 1) void k_decoder_bf16(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dec[];
    constexpr int TM = 32, LDH = H + 4, LDB = H + 8, NT = H >> 5, G = H >> 3, GH16 = H >> 4, NTHR = NT * 64, TPR = NTHR / TM;
    float* hs = reinterpret_cast<float*>(smem_dec);        // [32][LDH] fp32 h (head operand); x_z tile in the prologue
    float* wo = hs + TM * LDH;                             // [H][2]
    float* pl = wo + 2 * H;                                // [32][2]
    u16* hb = reinterpret_cast<u16*>(pl + TM * 2);         // [32][LDB] bf16 h
    u16* rb = hb + TM * LDB;                               // [32][LDB] bf16 r*h
    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    const int row0 = blockIdx.x * TM;
    DYN_P(a, row0)
    const int col = cb * 32 + c31;
    for (int i = tid; i < TM * (H >> 2); i += NTHR) {
        const int r = i / (H >> 2), c4 = i - r * (H >> 2);
        *reinterpret_cast<float4*>(hs + r * LDH + c4 * 4) =
            *reinterpret_cast<const float4*>(a.xz + (size_t)min(row0 + r, a.R - 1) * H + c4 * 4);
    }
    for (int i = tid; i < 2 * H; i += NTHR) wo[i] = a.w_head[i];
    if (tid < TM) {
        const int ag = agent_of_row(min(row0 + tid, a.R - 1), a.K, a.mno);
        pl[tid * 2] = a.p_last[(size_t)ag * 2]; pl[tid * 2 + 1] = a.p_last[(size_t)ag * 2 + 1];
    }
    __syncthreads();
    f32x16 xr[1] = {splat16h(a.b_g[col])}, xu[1] = {splat16h(a.b_g[H + col])}, xc[1] = {splat16h(a.b_c[col])};
    {
        const float* x_lane = hs + c31 * LDH + 4 * hi;
        mma_groups<1>(xr, x_lane, LDH, a.Wxg + ((size_t)cb * G) * 64 + lane, G);
        mma_groups<1>(xu, x_lane, LDH, a.Wxg + ((size_t)(cb + NT) * G) * 64 + lane, G);
        mma_groups<1>(xc, x_lane, LDH, a.Wxc + ((size_t)cb * G) * 64 + lane, G);
    }
    f32x16 h;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = min(row0 + (i & 3) + 8 * (i >> 2) + 4 * hi, a.R - 1);
        h[i] = a.Hx[(size_t)agent_of_row(row, a.K, a.mno) * a.ldhx + col];
    }
    __syncthreads();                                       // x_z tile consumed: hs becomes the fp32 h image
#pragma unroll
    for (int i = 0; i < 16; ++i) hb[((i & 3) + 8 * (i >> 2) + 4 * hi) * LDB + col] = bf16_of(h[i]);
    __syncthreads();
    const uint4* Whg = reinterpret_cast<const uint4*>(a.Whg);
    const uint4* Whc = reinterpret_cast<const uint4*>(a.Whc);
    const u16* hp[1] = {hb + c31 * LDB + 8 * hi};
    const u16* rp[1] = {rb + c31 * LDB + 8 * hi};
    const uint4* bg[2] = {Whg + ((size_t)cb * GH16) * 64 + lane, Whg + ((size_t)(cb + NT) * GH16) * 64 + lane};
    const uint4* bc[1] = {Whc + ((size_t)cb * GH16) * 64 + lane};
    const float bh0 = a.b_head[0], bh1 = a.b_head[1];
    const int hr = tid / TPR, hq = tid % TPR;
    for (int t = 0; t < a.T; ++t) {
        f32x16 g2[2][1] = {{xr[0]}, {xu[0]}};
        mma16_groups<1, 2>(g2, hp, bg, GH16);
        f32x16 u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float r = sigmoidf_(g2[0][0][i]);
            rb[((i & 3) + 8 * (i >> 2) + 4 * hi) * LDB + col] = bf16_of(r * h[i]);
            u[i] = sigmoidf_(g2[1][0][i]);
        }
        __syncthreads();
        f32x16 ac[1][1] = {{xc[0]}};
        mma16_groups<1, 1>(ac, rp, bc, GH16);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float c = tanhf_(ac[0][0][i]);
            h[i] = gru_blend(u[i], h[i], c);
            const int rl = (i & 3) + 8 * (i >> 2) + 4 * hi;
            hb[rl * LDB + col] = bf16_of(h[i]);
            hs[rl * LDH + col] = h[i];
        }
        __syncthreads();
        {   // head: y = p_last + h W_o + b_o from the fp32 image; TPR threads per row
            constexpr int per = H / TPR;
            float s0 = 0.f, s1 = 0.f;
            for (int c = hq * per; c < (hq + 1) * per; ++c) {
                const float hv = hs[hr * LDH + c];
                s0 = fmaf(hv, wo[c * 2], s0);
                s1 = fmaf(hv, wo[c * 2 + 1], s1);
            }
            s0 += __shfl_xor(s0, 1); s1 += __shfl_xor(s1, 1);
            s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
            if (TPR >= 8) { s0 += __shfl_xor(s0, 4); s1 += __shfl_xor(s1, 4); }
            if (TPR >= 16) { s0 += __shfl_xor(s0, 8); s1 += __shfl_xor(s1, 8); }
            if (hq == 0 && row0 + hr < a.R)
                *reinterpret_cast<float2*>(a.Y + ((size_t)(row0 + hr) * a.T + t) * 2) =
                    make_float2(pl[hr * 2] + (s0 + bh0), pl[hr * 2 + 1] + (s1 + bh1));
        }
        // hs is rewritten only after the NEXT step's first barrier, which every head reader reaches first
    }
}
template <int H>
static void launch_dec16(const DecArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(32 * (H + 4) + 2 * H + 64) * sizeof(float) + (size_t)2 * 32 * (H + 8) * sizeof(u16);
    allow_big_lds(k_decoder_bf16<H>);
    hipLaunchKernelGGL((k_decoder_bf16<H>), dim3((a.R + 31) / 32), dim3((H / 32) * 64), lds, s, a);
}
void launch_decoder_bf16(const DecArgs& a, hipStream_t s) {
    if (a.H == 256) launch_dec16<256>(a, s);
    else if (a.H == 128) launch_dec16<128>(a, s);
    else launch_dec16<64>(a, s);
}

// ------------------------------------------------------------------------------------------------------------------
// deconv3 + deconv4 fused (bf16 operands): [n,8,8,64] -> [n,32,32] without ever writing the [n,16,16,32] activation.
// deconv3 runs TRANSPOSED (A = packed W3 with lane = output channel, B = the gathered input pixel row from LDS), so its
// accumulators come out with lane = pixel and registers = channels -- the B-fragment layout (up to the chain order of k)
// of the next product.  deconv4 has ONE output channel, so it is done as "tap products first":
//     T[tap, pixel] = sum_c W4[tap, c] * d3[pixel, c]            (2 MFMAs per 32 pixels, A = W4 packed in chain order)
// and each of the 25 products of a pixel is then added to its output position o = 2i + k - 1 in an fp32 LDS image of the
// sample (plain read-add-write, the two half-waves in turn: inside a half every lane holds the SAME tap, so the
// addresses are distinct; deterministic).  One wave per sample.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DS_WG) void k_deconv34_bf16(ConvArgs a, const float* __restrict__ sc4p, const float* __restrict__ sh4p) {
    constexpr int LDP = 72;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_d34[];
    u16* zero_row = reinterpret_cast<u16*>(smem_d34);
    u16* in_s = zero_row + LDP;                                        // [4][64][LDP]
    float* xacc = reinterpret_cast<float*>(in_s + 4 * 64 * LDP);       // [4][1024]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int s0 = blockIdx.x * 4;
    DYN_N(a, n, s0)
    for (int i = tid; i < LDP / 2; i += DS_WG) reinterpret_cast<unsigned*>(zero_row)[i] = 0u;
    for (int i = tid; i < 4 * 1024; i += DS_WG) xacc[i] = 0.f;
    for (int i = tid; i < 4 * 64 * 8; i += DS_WG) {
        const int pix = i >> 3, c8 = i & 7;
        const int smp = s0 + (pix >> 6);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (smp < a.n) {
            const float* src = a.in + ((size_t)s0 * 64 + pix) * 64 + c8 * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(src), x1 = *reinterpret_cast<const float4*>(src + 4);
            v = make_uint4(pk_bf16(x0.x, x0.y), pk_bf16(x0.z, x0.w), pk_bf16(x1.x, x1.y), pk_bf16(x1.z, x1.w));
        }
        *reinterpret_cast<uint4*>(in_s + pix * LDP + c8 * 8) = v;
    }
    __syncthreads();
    const int smp = s0 + w;
    const u16* mine = in_s + w * 64 * LDP;
    float* xa = xacc + w * 1024;
    const int c = lane & 31, hi = lane >> 5;
    const uint4* W3 = reinterpret_cast<const uint4*>(a.Wp);
    const uint4* W4 = reinterpret_cast<const uint4*>(a.w_raw);         // [2 k-groups][64] chain order, rows = taps
    float sc3[16], sh3[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { const int co = (r & 3) + 8 * (r >> 2) + 4 * hi; sc3[r] = a.scale[co]; sh3[r] = a.shift[co]; }
    const uint4 w4a = W4[lane], w4b = W4[64 + lane];
    int qy[2], qx[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) { const int q = m * 32 + c; qy[m] = q >> 3; qx[m] = q & 7; }
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            f32x16 acc[2] = {zero16(), zero16()};
            const int ny = py ? 3 : 2, nx = px ? 3 : 2, ntap = ny * nx;
            auto tap_of = [&](int t) { const int iy = t / nx, ix = t - iy * nx; return (1 - py + 2 * iy) * 5 + (1 - px + 2 * ix); };
            uint4 bc[4], bn[4];
            {
                const uint4* bp = W3 + ((size_t)tap_of(0) * 4) * 64 + lane;
#pragma unroll
                for (int g = 0; g < 4; ++g) bc[g] = bp[g * 64];
            }
#pragma clang loop unroll(disable)
            for (int t = 0; t < ntap; ++t) {
                const int tap = tap_of(t), ky = tap / 5, kx = tap - ky * 5;
                {
                    const uint4* bp = W3 + ((size_t)tap_of(min(t + 1, ntap - 1)) * 4) * 64 + lane;
#pragma unroll
                    for (int g = 0; g < 4; ++g) bn[g] = bp[g * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
                const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int iy = qy[m] + dy, ix = qx[m] + dx;
                    const bool ok = iy >= 0 && iy < 8 && ix >= 0 && ix < 8;
                    const u16* xp = (ok ? mine + (iy * 8 + ix) * LDP : zero_row) + 8 * hi;
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[m] = mfma16(bc[g], *reinterpret_cast<const uint4*>(xp + 16 * g), acc[m]);   // D[co][pixel]
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 4; ++g) bc[g] = bn[g];
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = eluf_(acc[m][r] * sc3[r] + sh3[r]);
                const uint4 p0 = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
                const uint4 p1 = make_uint4(pk_bf16(v[8], v[9]), pk_bf16(v[10], v[11]), pk_bf16(v[12], v[13]), pk_bf16(v[14], v[15]));
                f32x16 tp = mfma16(w4a, p0, zero16());
                tp = mfma16(w4b, p1, tp);                               // D[tap][pixel]
                const int oy = 2 * qy[m] + py, ox = 2 * qx[m] + px;    // this lane's pixel in the 16x16 grid
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (hi == half) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int tap = (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (tap < 25) {
                                const int ky = tap / 5, kx = tap - ky * 5;
                                const int Y = 2 * oy + ky - 1, X = 2 * ox + kx - 1;
                                if (Y >= 0 && Y < 32 && X >= 0 && X < 32) xa[Y * 32 + X] += tp[r];
                            }
                        }
                    }
                }
            }
        }
    if (smp < a.n) {
        const float sc4 = sc4p[0], sh4 = sh4p[0];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = i * 64 + lane;
            a.out[(size_t)smp * 1024 + o] = sigmoidf_(xa[o] * sc4 + sh4);
        }
    }
}
void launch_deconv34_bf16(const ConvArgs& a, const float* sc4, const float* sh4, hipStream_t s) {
    const size_t lds = (72 + 4 * 64 * 72) * sizeof(u16) + 4 * 1024 * sizeof(float);
    hipLaunchKernelGGL(k_deconv34_bf16, dim3((a.n + 3) / 4), dim3(DS_WG), lds, s, a, sc4, sh4);
}

// ------------------------------------------------------------------------------------------------------------------
// Row-tiled GEMMs with bf16 operands for the two dense layers that see every (agent, k) row: deconv1
// ([R, L] x [L, 2048], BN + ELU) and the mask fc ([R, 1024] x [1024, H], ReLU -> softmax -> * Hx).  64-row tiles, the
// fp32 activations are rounded on their way into a bf16 LDS image; accumulation and epilogues fp32.
// ------------------------------------------------------------------------------------------------------------------
#define KC16 512
#define LDA16 (KC16 + 8)          // bf16 elements: 260 dwords = 4 mod 8
__device__ __forceinline__ void stage16(u16* dst, const float* __restrict__ src, int lds_ld, int ld, int row0, int M, int k0, int kc, int tid) {
    const int q = kc >> 3;                                             // 8-element chunks per row
    for (int i = tid; i < 64 * q; i += DS_WG) {
        const int r = i / q, c8 = i - r * q;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row0 + r < M) {
            const float* p = src + (size_t)(row0 + r) * ld + k0 + c8 * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(p), x1 = *reinterpret_cast<const float4*>(p + 4);
            v = make_uint4(pk_bf16(x0.x, x0.y), pk_bf16(x0.z, x0.w), pk_bf16(x1.x, x1.y), pk_bf16(x1.z, x1.w));
        }
        *reinterpret_cast<uint4*>(dst + r * lds_ld + c8 * 8) = v;
    }
}

// deconv1: out[row, col] = elu((z W)[row, col] * scale[col % 128] + shift[col % 128]); K = L <= 512 in one chunk.
// grid = (row tiles, column blocks of 16 n-tiles); wave w takes n-tiles w, w+4, w+8, w+12 of its block.
__global__ __launch_bounds__(DS_WG) void k_deconv1_bf16(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_g16[];
    u16* As = reinterpret_cast<u16*>(smem_g16);                        // [64][K + 8]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    const int row0 = blockIdx.x * 64, ld = a.K + 8, G16 = a.K >> 4;
    DYN_N(a, M, row0)
    stage16(As, a.A, ld, a.lda, row0, a.M, 0, a.K, tid);
    __syncthreads();
    const uint4* Bp = reinterpret_cast<const uint4*>(a.Bp);
    const u16* ap[2] = {As + c31 * ld + 8 * hi, As + (32 + c31) * ld + 8 * hi};
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const int nt = (blockIdx.y * 4 + j) * 4 + w;
        if (nt >= a.NT) continue;
        f32x16 acc[1][2] = {{zero16(), zero16()}};
        const uint4* bl[1] = {Bp + ((size_t)nt * G16) * 64 + lane};
        mma16_groups<2, 1>(acc, ap, bl, G16);
        const int col = nt * 32 + c31;
        const float sc = a.p0[col % a.chmod], sh = a.p1[col % a.chmod];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + m * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                if (row < a.M) a.out[(size_t)row * a.ldo + col] = eluf_(acc[0][m][i] * sc + sh);
            }
    }
}
void launch_deconv1_bf16(const GemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)64 * (a.K + 8) * sizeof(u16);
    hipLaunchKernelGGL(k_deconv1_bf16, dim3((a.M + 63) / 64, (a.NT + 15) / 16), dim3(DS_WG), lds, s, a);
}

// mask fc: x_z[r, :] = softmax(relu(xhat[r, :] Wm + bm)) * Hx[agent(r), :]; K = V = 1024 in two chunks, H <= 256.
__global__ __launch_bounds__(DS_WG) void k_mask_bf16(MaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_m16[];
    u16* As = reinterpret_cast<u16*>(smem_m16);                        // [64][LDA16]; reused as the fp32 softmax tile
    float* tile = reinterpret_cast<float*>(smem_m16);
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    const int row0 = blockIdx.x * 64;
    DYN_P(a, row0)
    const int NT = a.H >> 5, G16 = a.V >> 4;
    f32x16 acc[2][1][2] = {{{zero16(), zero16()}}, {{zero16(), zero16()}}};
    const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
    const u16* ap[2] = {As + c31 * LDA16 + 8 * hi, As + (32 + c31) * LDA16 + 8 * hi};
    for (int k0 = 0; k0 < a.V; k0 += KC16) {
        stage16(As, a.xhat, LDA16, a.V, row0, a.R, k0, KC16, tid);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nt = w + 4 * j;
            if (nt < NT) {
                const uint4* bl[1] = {Wp + ((size_t)nt * G16 + (k0 >> 4)) * 64 + lane};
                mma16_groups<2, 1>(acc[j], ap, bl, KC16 >> 4);
            }
        }
        __syncthreads();
    }
    const int LDT = a.H + 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nt = w + 4 * j;
        if (nt >= NT) continue;
        const int col = nt * 32 + c31;
        const float b = a.bias[col];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) tile[(m * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi) * LDT + col] = fmaxf(acc[j][0][m][i] + b, 0.f);
    }
    __syncthreads();
    // softmax over H per row: 4 threads per row, each owning every fourth float4 of it (16-byte accesses to the tile, xz and Hx; as k_mask)
    const int r = tid >> 2, q4 = tid & 3;
    const int row = row0 + r;
    const int nv = a.H >> 4;
    float4* trow = reinterpret_cast<float4*>(tile + r * LDT);
    float mx = -3.0e38f;
    for (int j = 0; j < nv; ++j) { const float4 v = trow[4 * j + q4]; mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w))); }
    mx = fmaxf(mx, __shfl_xor(mx, 1));
    mx = fmaxf(mx, __shfl_xor(mx, 2));
    float sum = 0.f;
    for (int j = 0; j < nv; ++j) {
        float4 v = trow[4 * j + q4];
        v.x = expf(v.x - mx); v.y = expf(v.y - mx); v.z = expf(v.z - mx); v.w = expf(v.w - mx);
        trow[4 * j + q4] = v;
        const int c0 = 16 * j + 4 * q4;         // padded columns (relu(0) = 0 <= mx) are not in the softmax
        sum += (c0 < a.Hl ? v.x : 0.f) + (c0 + 1 < a.Hl ? v.y : 0.f) + (c0 + 2 < a.Hl ? v.z : 0.f) + (c0 + 3 < a.Hl ? v.w : 0.f);
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (row < a.R) {
        const int ag = agent_of_row(row, a.K, a.mno);
        const float inv = 1.0f / sum;
        const float4* hx = reinterpret_cast<const float4*>(a.Hx + (size_t)ag * a.ldhx);
        float4* xz = reinterpret_cast<float4*>(a.xz + (size_t)row * a.H);
        for (int j = 0; j < nv; ++j) {
            const float4 e = trow[4 * j + q4], hv = hx[4 * j + q4];
            xz[4 * j + q4] = make_float4(e.x * inv * hv.x, e.y * inv * hv.y, e.z * inv * hv.z, e.w * inv * hv.w);
        }
    }
}
void launch_mask_bf16(const MaskArgs& a, hipStream_t s) {
    const size_t lds = (size_t)64 * LDA16 * sizeof(u16);               // 66.6 KB >= the [64][H+4] fp32 softmax tile
    allow_big_lds(k_mask_bf16);
    hipLaunchKernelGGL(k_mask_bf16, dim3((a.R + 63) / 64), dim3(DS_WG), lds, s, a);
}

// ------------------------------------------------------------------------------------------------------------------
// CVAE encoder convolutions conv2 / conv3 with bf16 operands (per AGENT; same gather tiling as k_conv_gather: the input
// image of SPW samples in LDS as bf16, rows = output pixels, taps gathered; G16 = CI/16 fragments per tap).
// ------------------------------------------------------------------------------------------------------------------
template <int CI, int IW, int OW, int STRIDE, int PAD, int CO>
__global__ __launch_bounds__(DS_WG) void k_conv_gather_bf16(ConvArgs a) {
    constexpr int PIX = OW * OW, SPW = 64 / PIX, LDP = CI + 8, NT = CO / 32, G16 = CI / 16;
    constexpr int MT = (NT == 4) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cg[];
    u16* zero_row = reinterpret_cast<u16*>(smem_cg);                   // LDP zeros
    u16* in_s = zero_row + LDP;                                        // [SPW][IW*IW][LDP]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    const int s0 = blockIdx.x * SPW;
    DYN_N(a, n, s0)
    for (int i = tid; i < LDP / 2; i += DS_WG) reinterpret_cast<unsigned*>(zero_row)[i] = 0u;
    constexpr int Q = CI / 8;
    for (int i = tid; i < SPW * IW * IW * Q; i += DS_WG) {
        const int pix = i / Q, c8 = i - pix * Q;
        const int smp = s0 + pix / (IW * IW);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (smp < a.n) {
            const float* p = a.in + ((size_t)s0 * IW * IW + pix) * CI + c8 * 8;
            const float4 x0 = *reinterpret_cast<const float4*>(p), x1 = *reinterpret_cast<const float4*>(p + 4);
            v = make_uint4(pk_bf16(x0.x, x0.y), pk_bf16(x0.z, x0.w), pk_bf16(x1.x, x1.y), pk_bf16(x1.z, x1.w));
        }
        *reinterpret_cast<uint4*>(in_s + pix * LDP + c8 * 8) = v;
    }
    __syncthreads();
    const int nt = (NT == 4) ? w : (w & 1);
    const int mt0 = (NT == 4) ? 0 : (w >> 1);
    f32x16 acc[1][MT];
    int oy[MT], ox[MT], sm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[0][m] = zero16();
        const int r = (mt0 + m) * 32 + c31;
        sm[m] = r / PIX;
        const int q = r - sm[m] * PIX;
        oy[m] = q / OW;
        ox[m] = q - oy[m] * OW;
    }
    const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx) {
            const u16* ap[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int iy = oy[m] * STRIDE + ky - PAD, ix = ox[m] * STRIDE + kx - PAD;
                const bool ok = iy >= 0 && iy < IW && ix >= 0 && ix < IW;
                ap[m] = (ok ? in_s + ((sm[m] * IW + iy) * IW + ix) * LDP : zero_row) + 8 * hi;
            }
            const uint4* bl[1] = {Wp + ((size_t)((ky * 5 + kx) * NT + nt) * G16) * 64 + lane};
            mma16_groups<MT, 1>(acc, ap, bl, G16);
        }
    const int co = nt * 32 + c31;
    const float sc = a.scale[co], sh = a.shift[co];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = (mt0 + m) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
            const int smp = s0 + r / PIX;
            if (smp < a.n) a.out[((size_t)s0 * PIX + r) * CO + co] = eluf_(acc[0][m][i] * sc + sh);
        }
}
void launch_conv2_bf16(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(40 + 256 * 40) * sizeof(u16);
    hipLaunchKernelGGL((k_conv_gather_bf16<32, 16, 8, 2, 1, 64>), dim3(a.n), dim3(DS_WG), lds, s, a);
}
void launch_conv3_bf16(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)(72 + 4 * 64 * 72) * sizeof(u16);
    hipLaunchKernelGGL((k_conv_gather_bf16<64, 8, 4, 1, 0, 128>), dim3((a.n + 3) / 4), dim3(DS_WG), lds, s, a);
}

// ------------------------------------------------------------------------------------------------------------------
// GRU encoders with bf16 recurrent operands (per AGENT, latency-bound: T sequential steps on A/32 tiles).  The 2-wide
// input contribution stays fp32 VALU (explicit fma chain like k_encoder); h and r*h are bf16 LDS images, h itself stays
// fp32 in registers; two barriers per step.  a.Whg / a.Whc point at the bf16 packs.
// ------------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__((H / 32) * 64) void k_encoder_bf16(EncArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_e16[];
    constexpr int TM = 32, LDB = H + 8, NT = H >> 5, GH16 = H >> 4, NTHR = NT * 64;
    u16* hb = reinterpret_cast<u16*>(smem_e16);            // [32][LDB] bf16 h
    u16* rb = hb + TM * LDB;                               // [32][LDB] bf16 r*h
    float* xs = reinterpret_cast<float*>(rb + TM * LDB);   // [2][32][2] normalised input, double-buffered over t
    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    DYN_N(a, mno, blockIdx.x * TM)
    const int A = a.n_scenes * a.mno;
    const int a0 = blockIdx.x * TM;
    const int col = cb * 32 + c31;
    const float wr0 = a.wx_g[col], wr1 = a.wx_g[2 * H + col], wu0 = a.wx_g[H + col], wu1 = a.wx_g[2 * H + H + col];
    const float wc0 = a.wx_c[col], wc1 = a.wx_c[H + col];
    const float br = a.b_g[col], bu = a.b_g[H + col], bc = a.b_c[col];
    f32x16 h = zero16();
    for (int i = tid; i < TM * LDB / 2; i += NTHR) reinterpret_cast<unsigned*>(hb)[i] = 0u;
    const uint4* Whg = reinterpret_cast<const uint4*>(a.Whg);
    const uint4* Whc = reinterpret_cast<const uint4*>(a.Whc);
    const u16* hp[1] = {hb + c31 * LDB + 8 * hi};
    const u16* rp[1] = {rb + c31 * LDB + 8 * hi};
    const uint4* bg[2] = {Whg + ((size_t)cb * GH16) * 64 + lane, Whg + ((size_t)(cb + NT) * GH16) * 64 + lane};
    const uint4* bcp[1] = {Whc + ((size_t)cb * GH16) * 64 + lane};
    auto load_x = [&](int t) {
        if (tid < TM) {
            const int ag = min(a0 + tid, A - 1);
            const int sc = ag / a.mno, slot = ag - sc * a.mno;
            const float* f = a.frames + (((size_t)sc * a.T + t) * a.mno + slot) * 3;
            float* x = xs + (t & 1) * TM * 2;
            x[tid * 2] = __fmul_rn(f[1], a.sx); x[tid * 2 + 1] = __fmul_rn(f[2], a.sy);
            if (t == a.T - 1 && a0 + tid < A) {
                if (a.p_last) { a.p_last[(size_t)ag * 2] = x[tid * 2]; a.p_last[(size_t)ag * 2 + 1] = x[tid * 2 + 1]; }
                if (a.valid) a.valid[ag] = (f[0] != 0.f) ? 1 : 0;
            }
        }
    };
    load_x(0);
    __syncthreads();
    for (int t = 0; t < a.T; ++t) {
        const float* x = xs + (t & 1) * TM * 2;
        if (t + 1 < a.T) load_x(t + 1);                    // other buffer: read after the next barrier pair
        f32x16 g2[2][1];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;
            g2[0][0][i] = fmaf(x[row * 2 + 1], wr1, fmaf(x[row * 2], wr0, br));
            g2[1][0][i] = fmaf(x[row * 2 + 1], wu1, fmaf(x[row * 2], wu0, bu));
        }
        mma16_groups<1, 2>(g2, hp, bg, GH16);
        f32x16 u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float r = sigmoidf_(g2[0][0][i]);
            rb[((i & 3) + 8 * (i >> 2) + 4 * hi) * LDB + col] = bf16_of(r * h[i]);
            u[i] = sigmoidf_(g2[1][0][i]);
        }
        __syncthreads();
        f32x16 ac[1][1];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * hi;
            ac[0][0][i] = fmaf(x[row * 2 + 1], wc1, fmaf(x[row * 2], wc0, bc));
        }
        mma16_groups<1, 1>(ac, rp, bcp, GH16);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            h[i] = gru_blend(u[i], h[i], tanhf_(ac[0][0][i]));
            hb[((i & 3) + 8 * (i >> 2) + 4 * hi) * LDB + col] = bf16_of(h[i]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int ag = a0 + (i & 3) + 8 * (i >> 2) + 4 * hi;
        if (ag < A) a.out[(size_t)ag * a.ldo + col] = h[i];
    }
}
void launch_encoder_bf16(const EncArgs& a, hipStream_t s) {
    const int A = a.n_scenes * a.mno;
    const dim3 grid((A + 31) / 32);
    const size_t lds = (size_t)2 * 32 * (a.H + 8) * sizeof(u16) + 2 * 32 * 2 * sizeof(float);
    if (a.H == 256) hipLaunchKernelGGL(k_encoder_bf16<256>, grid, dim3(512), lds, s, a);
    else if (a.H == 128) hipLaunchKernelGGL(k_encoder_bf16<128>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_encoder_bf16<64>, grid, dim3(128), lds, s, a);
}
