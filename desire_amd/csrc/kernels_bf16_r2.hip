// kernels_bf16_r2.hip -- bf16-operand IOC kernel with TWO row blocks per wave (64-row tiles, one workgroup per CU).
//
// What bounds k_ioc_bf16's 32-row tiles (kernels_bf16.hip) is not the matrix pipe but the per-CU vector-memory path: every
// tile-step streams ALL of W_soc (512 KB of bf16 fragments at 16 bins, H = 128) plus the gate / candidate kernels (233 KB)
// through it, two co-resident tiles per CU = 1.5 MB at 64 B/clk = 23 k cycles against 14 k cycles of MFMA work (TA_BUSY 66 %,
// profiles/r02_ta_busy.json).  The lever is weight bytes per row.  Here a wave owns hidden columns [32cb, 32cb+32) of TWO
// 32-row blocks and every weight fragment it fetches is used for both: half the weight bytes per row.  The second block's
// live state (its recurrent state, score partials and the four partial e_r tiles of the bin-split pooling: 128 more
// registers) does not fit two waves per SIMD -- so the workgroup keeps the CU to itself (one wave per SIMD, the whole
// 512-entry register file) and hides latency by software pipelining instead of a second wave: W_b fragments double-buffered
// two (bin, hidden block) iterations ahead, the gate ring six k-groups deep, LDS A fragments read one k-group ahead.
//
// Tile = 64 rows = whole (scene, k) groups: mno divides 32 (block m pools inside its own 32 rows) or mno = 64 (one group,
// both blocks pool over all 64 agents).  Pooling = the bin-split chain of k_ioc_bf16: occupied bins dealt round-robin to the
// waves, link 1 (P_b^T = Ht . M_b^T) once per (bin, hidden block, row block), link 2 into all column blocks, partial tiles
// summed in fixed order through exchange slots that alias the h^T and r*h tiles.  Same rounding points as k_ioc_bf16 (the
// oracle's q = bf16_round), same summation order per row: results are bit-identical to it.
#include "common.h"
#include "kernels.h"

#include "bf16.h"

#ifdef DESIRE_IOC_TIMING
#define TICKR(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#else
#define TICKR(k)
#endif

template <int H, int EV, int C, bool WIDE>          // WIDE: one 64-agent group spans both row blocks (compile-time: keeps the chains branch-free)
__global__ __launch_bounds__((H / 32) * 64, 1) void k_ioc_bf16_r2(IocArgs a) {
#ifdef DESIRE_IOC_TIMING
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = H >> 5, RB = 2, TM = 32 * RB, E = EV + C + H, KX = E + H;
    constexpr int LDXB = KX + 8, LDRB = H + 8, LDT = TM + 8;          // bf16 elements; (ld/2) = 4 mod 8 dwords: conflict-free b128
    constexpr int NTHR = NT * 64, TPR = NTHR / TM;
    constexpr int G16 = KX >> 4, GX16 = E >> 4, GH16 = H >> 4;
    constexpr int JGM = TM / 16;                                      // most 16-neighbour chunks a row can have (mno = 64)
    constexpr int RD = 6;                                             // gate ring depth (k-groups in flight)
    static_assert(H * LDT * 2 >= NT * 4096 && TM * LDRB * 2 >= NT * 4096, "exchange sets must fit the tiles they alias");
    const int B = a.G * a.G, LDM = B + 1;
    u16* Xb = reinterpret_cast<u16*>(smem_raw);                       // [TM][LDXB]  e_v | e_s | e_r | h
    u16* RHb = Xb + TM * LDXB;                                        // [TM][LDRB]  r * h
    u16* Ht = RHb + TM * LDRB;                                        // [H][LDT]    h transposed (pooling operand)
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(Ht + H * LDT);   // [TM][B+1], bit = tile-local row
    uint2* lut = reinterpret_cast<uint2*>(masks + TM * LDM);          // [16] nibble -> 4 bf16 (0.0 / 1.0)
    float* pc = reinterpret_cast<float*>(lut + 16);                   // [TM][2]
    float* pp = pc + TM * 2;                                          // [TM][2]
    float* wv = pp + TM * 2;                                          // [3][EV]
    float* red = wv + 3 * EV;                                         // [NT][TM]
    unsigned char* vld = reinterpret_cast<unsigned char*>(red + NT * TM);   // [TM]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + TM);                  // [2] bins that hold a neighbour anywhere in the tile
    float* EX0 = reinterpret_cast<float*>(Ht);                              // exchange set 0: inside the h^T tile (dead after the pooling chains)
    float* EX1 = reinterpret_cast<float*>(RHb);                             // set 1: inside the r*h tile (idle until the gates)

    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    const int row0 = blockIdx.x * TM;
    const int col = cb * 32 + c31;
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int my_row = min(row0 + r8, a.R - 1);
    const int my_scene = my_row / (a.K * a.mno);
    const int grp_base = (r8 / a.mno) * a.mno;
    const int my_slot = r8 - grp_base;
    constexpr bool wide = WIDE;                                       // one group spans both row blocks
    constexpr int JG = WIDE ? JGM : 2;                                // 16-wide neighbour chunks of a row block

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

    const int arow = 4 * hi;                                          // + 32m + (i&3) + 8(i>>2): tile-local row of accumulator element i of block m
    // h (fp32, accumulator layout) of row block m -> both bf16 images
    auto publish_h = [&](const f32x16& h, int m) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Xb[(32 * m + arow + (i & 3) + 8 * (i >> 2)) * LDXB + E + col] = bf16_of(h[i]);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<uint2*>(Ht + col * LDT + 32 * m + arow + 8 * q) =
                make_uint2(pk_bf16(h[4 * q], h[4 * q + 1]), pk_bf16(h[4 * q + 2], h[4 * q + 3]));
    };

    for (int it = 0; it < a.iters; ++it) {
        int row0p;                                                    // opaque copy: keeps the prologue's address math out of the time loop's registers
        asm volatile("s_mov_b32 %0, %1" : "=s"(row0p) : "s"(row0));
        f32x16 h[RB], sp[RB];
#pragma unroll
        for (int m = 0; m < RB; ++m) {
            sp[m] = zero16();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = min(row0p + 32 * m + arow + (i & 3) + 8 * (i >> 2), a.R - 1);
                h[m][i] = a.Hx[(size_t)agent_of_row(row, a.K, a.mno) * a.ldhx + col];
            }
        }
        __syncthreads();                                  // previous pass's readers of Xb / Ht are done
#pragma unroll
        for (int m = 0; m < RB; ++m) publish_h(h[m], m);
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
            TICKR(0)
            if (tid < TM && t + 1 < a.T)
                ynext = *reinterpret_cast<const float2*>(a.Y + ((size_t)min(row0 + tid, a.R - 1) * a.T + t + 1) * 2);
            // ---- P1: e_v, e_s, neighbour bits (row threads) ----
            {
                const float px = pc[r8 * 2], py = pc[r8 * 2 + 1];
                int cy, cx;
                scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
                const float* gsrc = grid + ((size_t)cy * a.Gw + cx) * C;
                float4 g4[(C + 4 * TPR - 1) / (4 * TPR)];             // the scene gather first: its L2 latency hides under the neighbour search
#pragma unroll
                for (int u = 0; u < (C + 4 * TPR - 1) / (4 * TPR); ++u)
                    if (4 * q8 + 4 * TPR * u < C) g4[u] = *reinterpret_cast<const float4*>(gsrc + 4 * q8 + 4 * TPR * u);
                const float vx = px - pp[r8 * 2], vy = py - pp[r8 * 2 + 1];
                for (int j = 2 * q8; j < EV; j += 2 * TPR) {
                    const float e0 = fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f);
                    const float e1 = fmaxf(fmaf(vy, wv[EV + j + 1], vx * wv[j + 1]) + wv[2 * EV + j + 1], 0.f);
                    *reinterpret_cast<unsigned*>(Xb + r8 * LDXB + j) = pk_bf16(e0, e1);
                }
                for (int j = q8; j < a.mno; j += TPR) {
                    if (j == my_slot || !vld[grp_base + j]) continue;
                    const int b = neighbor_bin_dev(px, py, pc[(grp_base + j) * 2], pc[(grp_base + j) * 2 + 1], a.nb_w, a.nb_h, a.G, a.bin_tab);
                    if (b >= 0) { atomicOr(&masks[r8 * LDM + b], 1ull << (grp_base + j)); atomicOr(&occ[b >> 5], 1u << (b & 31)); }
                }
#pragma unroll
                for (int u = 0; u < (C + 4 * TPR - 1) / (4 * TPR); ++u)
                    if (4 * q8 + 4 * TPR * u < C)
                        *reinterpret_cast<uint2*>(Xb + r8 * LDXB + EV + 4 * q8 + 4 * TPR * u) =
                            make_uint2(pk_bf16(g4[u].x, g4[u].y), pk_bf16(g4[u].z, g4[u].w));
            }
            TICKR(1)
            __syncthreads();
            TICKR(2)
            // ---- P2: social pooling chain -> e_r ----
            unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
            om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
            {
                // occupied bins dealt round-robin to the NT waves; a wave runs the whole chain of ITS bins for BOTH row blocks -- link 1
                // once per (hidden block, row block), link 2 into all NT column blocks -- and keeps RB x NT partial e_r tiles; slot k
                // belongs to column block (cb + k) % NT, so every register index is static
                unsigned long long mine = 0ull;
                {
                    int k = 0;
                    for (unsigned long long tmp = om; tmp; tmp &= tmp - 1, ++k)
                        if (k % NT == cb) mine |= tmp & (0ull - tmp);
                }
                f32x16 soc[RB][NT];
#pragma unroll
                for (int m = 0; m < RB; ++m)
#pragma unroll
                    for (int k = 0; k < NT; ++k) soc[m][k] = zero16();
                auto wptr = [&](int b, int hb, int k) {           // fragments of W_b[hidden block hb][column block (cb+k)%NT], 2 k-groups
                    const int cbo = (cb + k) % NT;
                    return Wsoc + ((size_t)(b * NT + cbo) * GH16 + 2 * hb) * 64 + lane;
                };
                // two fragment sets, set (hb & 1): a set is refreshed right after its last use with the fragments of the iteration
                // after next ((bin, hb + 2), or the first hidden blocks of my next bin), so a load is in flight for a whole
                // (bin, hidden block) iteration of MFMAs before anything waits on it
                uint4 wq[2][2 * NT];
                if (mine) {
                    const int b0 = __ffsll((long long)mine) - 1;
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int k = 0; k < NT; ++k) { const uint4* p = wptr(b0, s, k); wq[s][2 * k] = p[0]; wq[s][2 * k + 1] = p[64]; }
                }
#pragma clang loop unroll(disable)
                while (mine) {
                    const int b = __ffsll((long long)mine) - 1;
                    mine &= mine - 1;
                    const int nb = mine ? __ffsll((long long)mine) - 1 : b;
                    uint4 mf[RB][JGM];                                  // neighbour bits -> bf16 B fragments (16 neighbours each)
#pragma unroll
                    for (int m = 0; m < RB; ++m) {
                        const unsigned long long m64 = masks[(32 * m + c31) * LDM + b];
                        const int jb = wide ? 0 : 32 * m;
#pragma unroll
                        for (int jg = 0; jg < JGM; ++jg) {
                            if (jg < JG) {
                                const unsigned bits = (unsigned)(m64 >> (jb + 16 * jg + 8 * hi)) & 0xffu;
                                const uint2 l0 = lut[bits & 15u], l1 = lut[bits >> 4];
                                mf[m][jg] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                            }
                        }
                    }
#pragma unroll
                    for (int hb = 0; hb < NT; ++hb) {
                        const int s = hb & 1;
#pragma unroll
                        for (int m = 0; m < RB; ++m) {
                            f32x16 da = zero16();
                            const u16* hp = Ht + (hb * 32 + c31) * LDT + (wide ? 0 : 32 * m) + 8 * hi;
#pragma unroll
                            for (int jg = 0; jg < JGM; ++jg)
                                if (jg < JG) da = mfma16(*reinterpret_cast<const uint4*>(hp + 16 * jg), mf[m][jg], da);
                            const uint4 p0 = make_uint4(pk_bf16(da[0], da[1]), pk_bf16(da[2], da[3]), pk_bf16(da[4], da[5]), pk_bf16(da[6], da[7]));
                            const uint4 p1 = make_uint4(pk_bf16(da[8], da[9]), pk_bf16(da[10], da[11]), pk_bf16(da[12], da[13]), pk_bf16(da[14], da[15]));
#pragma unroll
                            for (int k = 0; k < NT; ++k) {
                                soc[m][k] = mfma16(p0, wq[s][2 * k], soc[m][k]);
                                soc[m][k] = mfma16(p1, wq[s][2 * k + 1], soc[m][k]);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < NT; ++k) {            // refresh set s: (b, hb + 2), or (next bin, hb + 2 - NT)
                            const uint4* p = (hb + 2 < NT) ? wptr(b, hb + 2, k) : wptr(nb, hb + 2 - NT, k);
                            wq[s][2 * k] = p[0]; wq[s][2 * k + 1] = p[64];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                TICKR(3)
                // fixed-order sum of the partial tiles: round (sft, m) hands slot sft of block m to the wave sft column blocks further
                // on; rounds alternate between the two slot sets, one barrier per round
                if (om) {                                          // (workgroup-uniform)
                    __syncthreads();                               // every wave is done reading Ht: it now carries exchange set 0
#pragma unroll
                    for (int sft = 1; sft < NT; ++sft) {
#pragma unroll
                        for (int m = 0; m < RB; ++m) {
                            float* ex = (((sft - 1) * RB + m) & 1) ? EX1 : EX0;
                            float4* dst = reinterpret_cast<float4*>(ex + (size_t)cb * 1024) + lane;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                dst[q * 64] = make_float4(soc[m][sft][4 * q], soc[m][sft][4 * q + 1], soc[m][sft][4 * q + 2], soc[m][sft][4 * q + 3]);
                            __syncthreads();
                            const float4* src = reinterpret_cast<const float4*>(ex + (size_t)((cb + NT - sft) % NT) * 1024) + lane;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 v = src[q * 64];
                                soc[m][0][4 * q] += v.x; soc[m][0][4 * q + 1] += v.y; soc[m][0][4 * q + 2] += v.z; soc[m][0][4 * q + 3] += v.w;
                            }
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < RB; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        Xb[(32 * m + arow + (i & 3) + 8 * (i >> 2)) * LDXB + EV + C + col] = bf16_of(fmaxf(soc[m][0][i] + bso, 0.f));
            }
            TICKR(4)
            __syncthreads();
            TICKR(5)
            // ---- P4: gates over [x | h], and the candidate's x part (same A fragments: three n-tiles per LDS read, both row blocks per
            //      weight fragment).  B fragments through a ring of RD k-groups; A fragments read one k-group ahead. ----
            f32x16 u[RB], ac[RB];
            {
                f32x16 g0[RB], g1[RB];
#pragma unroll
                for (int m = 0; m < RB; ++m) { g0[m] = zero16(); g1[m] = zero16(); ac[m] = zero16(); }
                int z4;
                asm volatile("s_mov_b32 %0, 0" : "=s"(z4));
                const uint4* wg0 = Wg + ((size_t)cb * G16) * 64 + z4;
                const uint4* wg1 = Wg + ((size_t)(cb + NT) * G16) * 64 + z4;
                const uint4* wcx = Wc + ((size_t)cb * G16) * 64 + z4;
                const unsigned ul = (unsigned)lane;
                uint4 rb[RD][3];
                auto req = [&](int g) {                            // (g is a compile-time constant after unrolling)
                    const int sl = g % RD;
                    rb[sl][0] = (wg0 + g * 64)[ul]; rb[sl][1] = (wg1 + g * 64)[ul];
                    if (g < GX16) rb[sl][2] = (wcx + g * 64)[ul];
                };
#pragma unroll
                for (int g = 0; g < RD && g < G16; ++g) req(g);
                const u16* xp0 = Xb + c31 * LDXB + 8 * hi;
                uint4 av[RB], an[RB];
#pragma unroll
                for (int m = 0; m < RB; ++m) av[m] = *reinterpret_cast<const uint4*>(xp0 + 32 * m * LDXB);
#pragma unroll
                for (int g = 0; g < G16; ++g) {
                    const int sl = g % RD;
                    if (g + 1 < G16) {
#pragma unroll
                        for (int m = 0; m < RB; ++m) an[m] = *reinterpret_cast<const uint4*>(xp0 + 32 * m * LDXB + (g + 1) * 16);
                    }
#pragma unroll
                    for (int m = 0; m < RB; ++m) {
                        g0[m] = mfma16(av[m], rb[sl][0], g0[m]); g1[m] = mfma16(av[m], rb[sl][1], g1[m]);
                        if (g < GX16) ac[m] = mfma16(av[m], rb[sl][2], ac[m]);
                    }
                    if (g + RD < G16) req(g + RD);
                    if (g + 1 < G16) {
#pragma unroll
                        for (int m = 0; m < RB; ++m) av[m] = an[m];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < RB; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float r = sigmoidf_(g0[m][i] + bgr);
                        RHb[(32 * m + arow + (i & 3) + 8 * (i >> 2)) * LDRB + col] = bf16_of(r * h[m][i]);
                        u[m][i] = sigmoidf_(g1[m][i] + bgu);
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
            TICKR(6)
            __syncthreads();
            TICKR(7)
            // ---- P5: candidate += (r*h) part, blend, score; publish h_t ----
            {
                const u16* rp0 = RHb + c31 * LDRB + 8 * hi;
#pragma unroll
                for (int g = 0; g < GH16; ++g)
#pragma unroll
                    for (int m = 0; m < RB; ++m) ac[m] = mfma16(*reinterpret_cast<const uint4*>(rp0 + 32 * m * LDRB + g * 16), ch[g], ac[m]);
#pragma unroll
                for (int m = 0; m < RB; ++m) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float c = tanhf_(ac[m][i] + bcc);
                        h[m][i] = gru_blend(u[m][i], h[m][i], c);
                        sp[m][i] = fmaf(h[m][i], wsc, sp[m][i]);
                    }
                    publish_h(h[m], m);                    // h slots of Xb / Ht were last read before the previous barrier
                }
            }
            if (tid < TM) {
                pp[tid * 2] = pc[tid * 2]; pp[tid * 2 + 1] = pc[tid * 2 + 1];
                pc[tid * 2] = ynext.x; pc[tid * 2 + 1] = ynext.y;
            }
            for (int i = tid; i < TM * LDM; i += NTHR) masks[i] = 0ull;
            if (tid < 2) occ[tid] = 0;
            TICKR(8)
            __syncthreads();
            TICKR(9)
        }
        // ---- score ----
#pragma unroll
        for (int m = 0; m < RB; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = sp[m][i];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
                if (c31 == 0) red[cb * TM + 32 * m + arow + (i & 3) + 8 * (i >> 2)] = v;
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
            const int cc = nt * 32 + c31;
            const float bb = cc < 2 * a.T ? a.b_reg[cc] : 0.f;
#pragma unroll
            for (int m = 0; m < RB; ++m) {
                f32x16 acc[1][1] = {{zero16()}};
                const u16* hp2[1] = {Xb + (32 * m + c31) * LDXB + 8 * hi + E};
                const uint4* br[1] = {Wreg + ((size_t)nt * GH16) * 64 + lane};
                mma16_groups<1, 1>(acc, hp2, br, GH16);
                if (cc < 2 * a.T) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = row0p + 32 * m + arow + (i & 3) + 8 * (i >> 2);
                        if (row < a.R) { float* y = a.Y + (size_t)row * 2 * a.T + cc; *y = *y + (acc[0][0][i] + bb); }
                    }
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

static size_t ioc16_r2_lds(const IocArgs& a) {
    const int H = a.H, TM = 64, E = 16 + 32 + H, KX = E + H, B = a.G * a.G, NT = H / 32;
    size_t b = (size_t)TM * (KX + 8) * 2 + (size_t)TM * (H + 8) * 2 + (size_t)H * (TM + 8) * 2;
    b += (size_t)TM * (B + 1) * 8 + 16 * 8 + (size_t)TM * 4 * 4 + 3 * 16 * 4 + (size_t)NT * TM * 4 + TM + 64;
    return b;
}
// whole (scene, k) groups in 64-row tiles: mno divides 32, or mno = 64; H in {64, 128} (the bin-split pooling keeps NT partial tiles
// per row block)
bool ioc_bf16_r2_supported(int mno, int H, int bins) {
    return (H == 64 || H == 128) && bins <= 64 && mno >= 1 && ((mno <= 32 && 32 % mno == 0) || mno == 64);
}
template <int H>
static void launch_r2(const IocArgs& a, hipStream_t s) {
    const dim3 grid((a.R + 63) / 64), block((H / 32) * 64);
    if (a.mno > 32) {
        allow_big_lds(k_ioc_bf16_r2<H, 16, 32, true>);
        hipLaunchKernelGGL((k_ioc_bf16_r2<H, 16, 32, true>), grid, block, ioc16_r2_lds(a), s, a);
    } else {
        allow_big_lds(k_ioc_bf16_r2<H, 16, 32, false>);
        hipLaunchKernelGGL((k_ioc_bf16_r2<H, 16, 32, false>), grid, block, ioc16_r2_lds(a), s, a);
    }
}
void launch_ioc_bf16_r2(const IocArgs& a, hipStream_t s) {
    if (a.H == 128) launch_r2<128>(a, s); else launch_r2<64>(a, s);
}
