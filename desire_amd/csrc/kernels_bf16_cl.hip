// kernels_bf16_cl.hip -- cluster form of the bf16-operand IOC kernel: (scene, k) groups of 64 / 96 / 128 agents
// (BASELINE configs[2]: a real SDD deathCircle window holds 65+ track ids, i.e. mno = 96 or 128).
//
// A group of mno = 32 * tpg agents spans tpg 32-row tiles = tpg workgroups (the tile body is k_ioc_bf16's 32-row form: wave
// cb owns hidden columns [32cb, 32cb+32) of the tile's 32 rows; state, gate math and accumulators fp32, MFMA operands bf16).
// What the members exchange once per step is exactly the pooling operand: the TRANSPOSED bf16 image of their hidden-state
// tile, Ht[hidden][row] (8 KB at H = 128), through a global buffer hex16[2][n_tiles][H][32] and the hand-off of cluster.h.
// Every member keeps the whole group's Ht[H][mno] in LDS and runs the pooling chain over mno/16 neighbour chunks:
//       P_b^T[hidden, i] = Ht[hidden, j] . M_b^T[j, i]      (B operand = the 128-bit neighbour mask of row i, expanded to bf16 0/1)
//       e_r            += cvt(P_b^T) . W_b                   (chain-ordered weights, no shuffle / LDS / barrier in between)
// The step's position-only work (velocity embedding, scene-feature gather, neighbour bins of my rows against all mno agents)
// runs BEFORE the wait for the neighbours' h_{t-1}, so part of the hand-off latency hides under it.
// Pooling is split over BINS between the waves of a tile (H <= 128): a bin's first chain link is run once, by its owner wave, and the
// partial e_r tiles are summed through exchange slots that live INSIDE the Ht tile (dead between the pooling and the end of the step).
// Grid: persistent, a multiple of tpg, never more workgroups than are co-resident (occupancy query), members adjacent.
// Round 5 (docs/DESIGN_DETAIL.md section 13, "Fourth step"): the exchange moves in 16-byte write-through units (a tile's column is stored in the order
// publish_h documents, so a lane's sixteen values are two stores), all of a thread's peer chunks in flight; the neighbour search is batched and
// branch-free (common.h: neighbor_bin_rect_nb over NaN-marked positions); the next step's positions / cleared masks are installed behind the
// barriers of the current step (no barrier at the top of a step); the file is built with -sink-insts-to-avoid-spills and forms its prologue /
// epilogue addresses from per-pass opaque row bases: 1 spilled register where there were 190.  6.15 -> 4.73 ms at configs[2]'s shape.
#include "bf16.h"
#include "cluster.h"
#include "kernels.h"

#define CLMAXM 128

#ifdef DESIRE_IOC_TIMING
#define TICKC(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#else
#define TICKC(k)
#endif
// TPGT: tiles per group as a compile-time constant (2, 3, 4: the neighbour-chunk loops of the pooling chains are then branch-free), or 0 = read
// it from the arguments
#ifndef IOC16CL_OCC
#define IOC16CL_OCC 2
#endif
template <int H, int EV, int C, bool SPLIT, int TPGT = 0>
__global__ __launch_bounds__((H / 32) * 64, ((H / 32) <= 4) ? IOC16CL_OCC : 1) void k_ioc_bf16_cl(IocArgs a, u16* __restrict__ hex16) {
#ifdef DESIRE_IOC_TIMING
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NT = H >> 5, TM = 32, E = EV + C + H, KX = E + H;
    constexpr int LDXB = KX + 8, LDRB = H + 8, LDT = CLMAXM + 8;      // bf16 elements; (ld/2) = 4 mod 8 dwords: conflict-free b128
    constexpr int NTHR = NT * 64, TPR = NTHR / TM;
    constexpr int G16 = KX >> 4, GX16 = E >> 4, GH16 = H >> 4;
    constexpr int JGM = CLMAXM / 16;                                   // most 16-neighbour chunks a row can have
    const int B = a.G * a.G, LDM = B + 1;
    const int tpg = TPGT ? TPGT : a.mno / 32;                          // tiles (workgroups) per group
    const int JG = TPGT ? 2 * TPGT : a.mno / 16;
    u16* Xb = reinterpret_cast<u16*>(smem_raw);                        // [TM][LDXB]  e_v | e_s | e_r | h   (my rows)
    u16* RHb = Xb + TM * LDXB;                                         // [TM][LDRB]  r * h
    u16* Ht = RHb + TM * LDRB;                                         // [H][LDT]    h transposed, the WHOLE group
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(Ht + H * LDT);   // [TM][B+1][2], bit = group-local slot
    uint2* lut = reinterpret_cast<uint2*>(masks + TM * LDM * 2);       // [16] nibble -> 4 bf16 (0.0 / 1.0)
    float* pg = reinterpret_cast<float*>(lut + 16);                    // [CLMAXM][2] positions of the whole group
    float* pp = pg + CLMAXM * 2;                                       // [TM][2] previous position of my rows
    float* wv = pp + TM * 2;                                           // [3][EV]
    float* red = wv + 3 * EV;                                          // [NT][TM]
    unsigned char* vld = reinterpret_cast<unsigned char*>(red + NT * TM);   // [CLMAXM]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + CLMAXM);              // [2] bins that hold a neighbour anywhere in the tile
    float2* pgv = reinterpret_cast<float2*>(occ + 2);                        // [CLMAXM] the same positions with NaN for absent agents (pair loop)
    // partial-tile exchange of the bin-split pooling: two sets of NT 4 KB slots INSIDE the Ht tile.  Ht is dead between the end of the
    // pooling chains and the end of the step (my columns are rewritten by publish_h, the other members' by the next step's copy), so
    // the exchange costs no LDS of its own and two workgroups still share a CU
    float* EX = reinterpret_cast<float*>(Ht);                               // [2][NT][1024] floats = 2 * NT * 4 KB <= H * LDT * 2 bytes

    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int hi = lane >> 5, c31 = lane & 31;
    const int col = cb * 32 + c31;
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int tile_pos = blockIdx.x % tpg;                             // my tile inside its group
    IOC_DYN(a)                                                         // (a slot class counted on the device: kernels.h DynCount)
    const int n_tiles = a.R / TM;
    static_assert(NTHR >= CLMAXM, "one thread per group slot for the position loads");
    const agent_buf hexr = agent_buffer(hex16, 2u * (unsigned)n_tiles * (H * TM * 2));
    const int my_slot = tile_pos * TM + r8;                            // group-local slot of my VALU row

    for (int i = tid; i < 3 * EV; i += NTHR) wv[i] = (i < 2 * EV) ? a.w_vel[i] : a.b_vel[i - 2 * EV];
    if (tid < 16) {
        const unsigned lo = ((tid & 1) ? 0x3F80u : 0u) | ((tid & 2) ? 0x3F800000u : 0u);
        const unsigned hi2 = ((tid & 4) ? 0x3F80u : 0u) | ((tid & 8) ? 0x3F800000u : 0u);
        lut[tid] = make_uint2(lo, hi2);
    }
    const float bgr = a.b_g[col], bgu = a.b_g[H + col], bcc = a.b_c[col], bso = a.b_soc[col], wsc = a.w_score[col];
    const uint4* Wg = reinterpret_cast<const uint4*>(a.Wg);
    const uint4* Wc = reinterpret_cast<const uint4*>(a.Wc);
    const uint4* Wsoc = reinterpret_cast<const uint4*>(a.Wsoc);
    const uint4* Wreg = reinterpret_cast<const uint4*>(a.Wreg);
    const u16* xp[1] = {Xb + c31 * LDXB + 8 * hi};
    const u16* rp[1] = {RHb + c31 * LDRB + 8 * hi};
    const int arow = 4 * hi;                                           // + (i&3) + 8(i>>2): local row of accumulator element i

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        const int grow0 = row0 - tile_pos * TM;                        // first row of the group
        const int group = grow0 / a.mno;
        int* cnt = a.grp_cnt + group;
        const int scene = grow0 / (a.K * a.mno);
        const float* grid = a.grids + (size_t)a.grid_of_scene[scene] * a.Gh * a.Gw * C;
        __syncthreads();
        for (int i = tid; i < a.mno; i += NTHR) vld[i] = a.valid[agent_of_row(grow0 + i, a.K, a.mno)];
        // h (fp32, accumulator layout) -> the bf16 images: my rows of Xb, my columns of Ht and (steps only) the exchange buffer
        // exchange order of a tile's column: position (i >> 3) * 16 + hi * 8 + (i & 7) for accumulator element i of lane half hi, so that
        // a lane's sixteen values are two 16-byte write-through stores and the two halves of a store instruction fill 32 contiguous bytes
        auto publish_h = [&](const f32x16& h, bool pub, unsigned gbyte) {
#pragma unroll
            for (int i = 0; i < 16; ++i) Xb[(arow + (i & 3) + 8 * (i >> 2)) * LDXB + E + col] = bf16_of(h[i]);
            unsigned pk[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pk[2 * q] = pk_bf16(h[4 * q], h[4 * q + 1]); pk[2 * q + 1] = pk_bf16(h[4 * q + 2], h[4 * q + 3]);
                *reinterpret_cast<uint2*>(Ht + col * LDT + tile_pos * TM + arow + 8 * q) = make_uint2(pk[2 * q], pk[2 * q + 1]);
            }
            if (pub) {
                const unsigned off = gbyte + (unsigned)col * (TM * 2) + 16u * hi;
                st_agent_u128(hexr, off, make_uint4(pk[0], pk[1], pk[2], pk[3]));            // write-through (cluster.h)
                st_agent_u128(hexr, off + 32u, make_uint4(pk[4], pk[5], pk[6], pk[7]));
            }
        };

        for (int it = 0; it < a.iters; ++it) {
            if (it > 0) group_wait(cnt, tpg * (it * (a.T + 1)), a.err);          // everybody's Y += dY has landed
            // opaque per-pass copies of the tile's row bases: the prologue / epilogue address math depends on them, so it cannot be hoisted
            // out of the pass and sit in (spilled) registers across the whole time loop (kernels_bf16.hip does the same)
            int row0p, grow0p;
            asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3" : "=s"(row0p), "=s"(grow0p) : "s"(row0), "s"(grow0));
            f32x16 h, sp = zero16();
#pragma unroll
            for (int i = 0; i < 16; ++i)
                h[i] = a.Hx[(size_t)agent_of_row(row0p + arow + (i & 3) + 8 * (i >> 2), a.K, a.mno) * a.ldhx + col];
            __syncthreads();                                  // previous pass's readers of Xb / Ht are done
            publish_h(h, false, 0u);
            // h_{-1} = Hx of the OTHER tiles' agents: every member computes it itself (no hand-off before step 0)
            for (int i = tid; i < a.mno * (H >> 2); i += NTHR) {
                const int j = i / (H >> 2), c4 = i - j * (H >> 2);
                if ((j >> 5) == tile_pos) continue;
                const float4 v = *reinterpret_cast<const float4*>(a.Hx + (size_t)agent_of_row(grow0p + j, a.K, a.mno) * a.ldhx + c4 * 4);
                Ht[(c4 * 4 + 0) * LDT + j] = bf16_of(v.x); Ht[(c4 * 4 + 1) * LDT + j] = bf16_of(v.y);
                Ht[(c4 * 4 + 2) * LDT + j] = bf16_of(v.z); Ht[(c4 * 4 + 3) * LDT + j] = bf16_of(v.w);
            }
            if (tid < TM) {
                const int ag = agent_of_row(row0p + tid, a.K, a.mno);
                pp[tid * 2] = a.p_last[(size_t)ag * 2]; pp[tid * 2 + 1] = a.p_last[(size_t)ag * 2 + 1];
            }

            // The step has NO barrier of its own at the top: what the position-only phase reads -- the group's positions (pg, pgv), the
            // previous positions of my rows (pp), cleared masks / occupancy words -- is prepared DURING the previous step, each behind the
            // barrier after its last reader (positions: after the barrier that closes P1; masks: after the barrier that closes the pooling).
            float2 ynx = make_float2(0.f, 0.f);                                  // my agent's position at the NEXT step, requested a step ahead
            auto put_positions = [&]() {
                pg[tid * 2] = ynx.x; pg[tid * 2 + 1] = ynx.y;
                const float qn = __int_as_float(0x7fc00000);
                pgv[tid] = vld[tid] ? ynx : make_float2(qn, qn);              // (vld[tid] was written by this thread)
            };
            if (tid < a.mno) { ynx = *reinterpret_cast<const float2*>(a.Y + (size_t)(grow0p + tid) * a.T * 2); put_positions(); }
            for (int i = tid; i < TM * LDM * 2; i += NTHR) masks[i] = 0ull;
            if (tid < 2) occ[tid] = 0;
            __syncthreads();
            for (int t = 0; t < a.T; ++t) {
                if (t + 1 < a.T && tid < a.mno) ynx = *reinterpret_cast<const float2*>(a.Y + ((size_t)(grow0 + tid) * a.T + t + 1) * 2);
                TICKC(0)
                // 16-byte loads (sc1) of the peers' h_{t-1} tiles (published at the end of step t-1, parity (t-1)&1), all of a thread's
                // chunks in flight: chunk i of a tile = column i >> 2, exchange positions 8 (i & 3) .. + 7 = lane half (i & 1), accumulator
                // elements 8 (i >> 1 & 1) .. + 7 (publish_h).  (Requesting them BEFORE the position-only phase was measured twice -- rounds 2 and 5: 4.73 ->
                // 5.1 ms, 90 spilled registers -- and is gone.)
                const unsigned pbyte = (unsigned)((t + 1) & 1) * (unsigned)n_tiles * (H * TM * 2);
                constexpr int CPT = (H * 4 + NTHR - 1) / NTHR;                 // chunks per thread and peer tile
                uint4 xv[TPGT > 0 ? (TPGT - 1) * CPT : 1];
                auto request_peers = [&]() {
                    if constexpr (TPGT > 0) {
#pragma unroll
                        for (int pi = 0; pi < TPGT - 1; ++pi) {
                            const int tp = pi + (pi >= tile_pos ? 1 : 0);
                            const unsigned base = pbyte + (unsigned)(tile - tile_pos + tp) * (H * TM * 2);
#pragma unroll
                            for (int k = 0; k < CPT; ++k) xv[pi * CPT + k] = ld_agent_u128(hexr, base + (unsigned)(tid + k * NTHR) * 16u);
                        }
                    }
                };
                // ---- P1: e_v, e_s, neighbour bits of my rows against the whole group (positions only: no hidden state needed) ----
                {
                    const float px = pg[my_slot * 2], py = pg[my_slot * 2 + 1];
                    const float vx = px - pp[r8 * 2], vy = py - pp[r8 * 2 + 1];
                    // the scene-feature line is requested first and lands under the pair loop
                    int cy, cx;
                    scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
                    const float* gsrc = grid + ((size_t)cy * a.Gw + cx) * C;
                    constexpr int NGQ = (C + 4 * TPR - 1) / (4 * TPR);
                    float4 g4[NGQ];
#pragma unroll
                    for (int k = 0; k < NGQ; ++k)
                        if (4 * (q8 + k * TPR) < C) g4[k] = *reinterpret_cast<const float4*>(gsrc + 4 * (q8 + k * TPR));
                    for (int j = 2 * q8; j < EV; j += 2 * TPR) {
                        const float e0 = fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f);
                        const float e1 = fmaxf(fmaf(vy, wv[EV + j + 1], vx * wv[j + 1]) + wv[2 * EV + j + 1], 0.f);
                        *reinterpret_cast<unsigned*>(Xb + r8 * LDXB + j) = pk_bf16(e0, e1);
                    }
                    // bins that hold a neighbour: collected per lane, ONE LDS atomic per lane and word at the end (a same-address atomic per
                    // pair serialises the wave)
                    unsigned oc0 = 0u, oc1 = 0u;
                    // (the divisors' reciprocals are formed per step from opaque copies: as invariants of the time loop they are spilled and
                    //  reloaded from scratch inside the pair loop, a memory round trip per pair)
                    float nbw, nbh;
                    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=v"(nbw), "=v"(nbh) : "s"(a.nb_w), "s"(a.nb_h));
                    const DivBy dvw = div_by(nbw), dvh = div_by(nbh);
                    if (a.bin_tab || !dvw.fast || !dvh.fast) {
                        for (int j = q8; j < a.mno; j += TPR) {
                            if (j == my_slot || !vld[j]) continue;
                            const int b = neighbor_bin_dev(px, py, pg[j * 2], pg[j * 2 + 1], a.nb_w, a.nb_h, a.G, a.bin_tab);
                            if (b >= 0) { atomicOr(&masks[(r8 * LDM + b) * 2 + (j >> 6)], 1ull << (j & 63)); if (b < 32) oc0 |= 1u << b; else oc1 |= 1u << (b - 32); }
                        }
                    } else {
                        const NbRect win = nb_rect(px, py, nbw, nbh);
                        // eight pairs at a time: positions read in one batch, cells branch-free, only the atomics predicated
                        const int mno_ = TPGT ? 32 * TPGT : a.mno;
                        unsigned long long oc = 0ull;
                        for (int j0 = q8; j0 < mno_; j0 += 8 * TPR) {
                            float2 pj[8];
#pragma unroll
                            for (int m = 0; m < 8; ++m) pj[m] = pgv[min(j0 + m * TPR, CLMAXM - 1)];
#pragma unroll
                            for (int m = 0; m < 8; ++m) {
                                const int j = j0 + m * TPR;
                                const int b = neighbor_bin_rect_nb(win, pj[m].x, pj[m].y, dvw, dvh, a.G);
                                if (b >= 0 && j != my_slot && j < mno_) {
                                    atomicOr(&masks[(r8 * LDM + b) * 2 + (j >> 6)], 1ull << (j & 63));
                                    oc |= 1ull << b;
                                }
                            }
                        }
                        oc0 = (unsigned)oc; oc1 = (unsigned)(oc >> 32);
                    }
                    oc0 = wave_or(oc0);
                    if (B > 32) oc1 = wave_or(oc1);
                    if (lane == 0) { if (oc0) atomicOr(&occ[0], oc0); if (oc1) atomicOr(&occ[1], oc1); }
#pragma unroll
                    for (int k = 0; k < NGQ; ++k)
                        if (4 * (q8 + k * TPR) < C)
                            *reinterpret_cast<uint2*>(Xb + r8 * LDXB + EV + 4 * (q8 + k * TPR)) = make_uint2(pk_bf16(g4[k].x, g4[k].y), pk_bf16(g4[k].z, g4[k].w));
                }
                TICKC(1)
                // ---- neighbours' h_{t-1} into the group's Ht ----
                if (t > 0) {
                    group_wait_wt(cnt, tpg * (it * (a.T + 1) + t), a.err);
                    TICKC(2)
                    if constexpr (TPGT > 0) {
                        request_peers();
#pragma unroll
                        for (int pi = 0; pi < TPGT - 1; ++pi) {
                            const int tp = pi + (pi >= tile_pos ? 1 : 0);
#pragma unroll
                            for (int k = 0; k < CPT; ++k) {
                                const int i = tid + k * NTHR;
                                u16* dst = Ht + (i >> 2) * LDT + tp * TM + 4 * (i & 1) + 16 * ((i >> 1) & 1);
                                *reinterpret_cast<uint2*>(dst) = make_uint2(xv[pi * CPT + k].x, xv[pi * CPT + k].y);
                                *reinterpret_cast<uint2*>(dst + 8) = make_uint2(xv[pi * CPT + k].z, xv[pi * CPT + k].w);
                            }
                        }
                    } else {
                        for (int tp = 0; tp < tpg; ++tp) {
                            if (tp == tile_pos) continue;
                            const unsigned base = pbyte + (unsigned)(tile - tile_pos + tp) * (H * TM * 2);
                            for (int i = tid; i < H * 4; i += NTHR) {
                                const uint4 v = ld_agent_u128(hexr, base + (unsigned)i * 16u);
                                u16* dst = Ht + (i >> 2) * LDT + tp * TM + 4 * (i & 1) + 16 * ((i >> 1) & 1);
                                *reinterpret_cast<uint2*>(dst) = make_uint2(v.x, v.y);
                                *reinterpret_cast<uint2*>(dst + 8) = make_uint2(v.z, v.w);
                            }
                        }
                    }
                }
                __syncthreads();
                TICKC(3)
                // every reader of this step's positions is past the barrier: the next step's go in, my rows' current ones become "previous"
                if (t + 1 < a.T && tid < a.mno) {
                    if ((tid >> 5) == tile_pos) { pp[(tid & 31) * 2] = pg[tid * 2]; pp[(tid & 31) * 2 + 1] = pg[tid * 2 + 1]; }
                    put_positions();
                }
                // ---- P2: social pooling chain -> e_r ----
                unsigned long long om_all = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
                om_all |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
                auto frag_bits = [&](int b, uint4 (&mf)[JGM]) {           // neighbour bits of row c31 in bin b -> bf16 B fragments
                    const unsigned long long m0 = masks[(c31 * LDM + b) * 2], m1 = masks[(c31 * LDM + b) * 2 + 1];
#pragma unroll
                    for (int jg = 0; jg < JGM; ++jg) {
                        if (jg < JG) {
                            const unsigned bits = (unsigned)((jg < 4 ? m0 : m1) >> (16 * (jg & 3) + 8 * hi)) & 0xffu;
                            const uint2 l0 = lut[bits & 15u], l1 = lut[bits >> 4];
                            mf[jg] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                        }
                    }
                };
                auto chain = [&](int hb, const uint4 (&mf)[JGM]) {
                    f32x16 d1 = zero16();
                    const u16* hp = Ht + (hb * 32 + c31) * LDT + 8 * hi;
#pragma unroll
                    for (int jg = 0; jg < JGM; ++jg)
                        if (jg < JG) d1 = mfma16(*reinterpret_cast<const uint4*>(hp + 16 * jg), mf[jg], d1);
                    return d1;
                };
                if constexpr (SPLIT) {
                    // occupied bins dealt round-robin to the NT waves; a wave runs the whole chain of ITS bins into NT partial e_r
                    // tiles (slot k = column block (cb + k) % NT), summed in fixed order through an LDS exchange (kernels_bf16.hip)
                    const unsigned long long om = om_all;
                    unsigned long long mine = 0ull;
                    {
                        int k = 0;
                        for (unsigned long long tmp = om; tmp; tmp &= tmp - 1, ++k)
                            if (k % NT == cb) mine |= tmp & (0ull - tmp);
                    }
                    f32x16 soc[NT];
#pragma unroll
                    for (int k = 0; k < NT; ++k) soc[k] = zero16();
                    auto wptr = [&](int b, int hb, int k) {
                        const int cbo = (cb + k) % NT;
                        return Wsoc + ((size_t)(b * NT + cbo) * GH16 + 2 * hb) * 64 + lane;
                    };
                    uint4 wq[2 * NT];
                    if (mine) {
                        const int b0 = __ffsll((long long)mine) - 1;
#pragma unroll
                        for (int k = 0; k < NT; ++k) { const uint4* p = wptr(b0, 0, k); wq[2 * k] = p[0]; wq[2 * k + 1] = p[64]; }
                    }
                    // wave priority by phase (round 6): the two workgroups of a CU put two waves on every SIMD, and a wave in its pooling chains (3) or gate
                    // contraction (2) now wins the issue arbitration against one in a position / exchange phase (0).  Same-box A/B at configs[2]'s shape:
                    // IOC 4.65 - 4.76 ms without -> 4.34 ms (equal priorities for both contractions: 4.52; the candidate phase raised too: 4.37 - 4.40)
                    __builtin_amdgcn_s_setprio(3);
#pragma clang loop unroll(disable)
                    while (mine) {
                        const int b = __ffsll((long long)mine) - 1;
                        mine &= mine - 1;
                        const int nb = mine ? __ffsll((long long)mine) - 1 : b;
                        uint4 mf[JGM];
                        frag_bits(b, mf);
                        // (the chain's h^T fragments through a ring of 3 / 4 / 6 register sets instead of one was measured neutral in round 5 -- the CU's
                        //  other workgroup already covers that LDS latency -- and is gone)
#pragma unroll
                        for (int hb = 0; hb < NT; ++hb) {
                            const f32x16 da = chain(hb, mf);
                            const uint4 p0 = make_uint4(pk_bf16(da[0], da[1]), pk_bf16(da[2], da[3]), pk_bf16(da[4], da[5]), pk_bf16(da[6], da[7]));
                            const uint4 p1 = make_uint4(pk_bf16(da[8], da[9]), pk_bf16(da[10], da[11]), pk_bf16(da[12], da[13]), pk_bf16(da[14], da[15]));
#pragma unroll
                            for (int k = 0; k < NT; ++k) {
                                soc[k] = mfma16(p0, wq[2 * k], soc[k]);
                                soc[k] = mfma16(p1, wq[2 * k + 1], soc[k]);
                                const uint4* p = (hb + 1 < NT) ? wptr(b, hb + 1, k) : wptr(nb, 0, k);
                                wq[2 * k] = p[0]; wq[2 * k + 1] = p[64];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    __builtin_amdgcn_s_setprio(0);
                    TICKC(4)
                    if (om) {                                          // (workgroup-uniform)
                        __syncthreads();                               // every wave is done reading Ht: it now carries the exchange slots
#pragma unroll
                        for (int sft = 1; sft < NT; ++sft) {
                            const int set = (sft - 1) & 1;
                            float4* dst = reinterpret_cast<float4*>(EX + (size_t)(set * NT + cb) * 1024) + lane;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                dst[q * 64] = make_float4(soc[sft][4 * q], soc[sft][4 * q + 1], soc[sft][4 * q + 2], soc[sft][4 * q + 3]);
                            __syncthreads();
                            const float4* src = reinterpret_cast<const float4*>(EX + (size_t)(set * NT + (cb + NT - sft) % NT) * 1024) + lane;
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
                } else {
                    f32x16 soc = zero16();
                    unsigned long long om = om_all;
                    uint4 wb[2 * NT];                                   // this wave's n-tile of W_b, refreshed in place one bin ahead
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
                        uint4 mf[JGM];
                        frag_bits(b, mf);
#pragma unroll
                        for (int hb = 0; hb < NT; ++hb) {
                            const f32x16 da = chain(hb, mf);
                            const uint4 p0 = make_uint4(pk_bf16(da[0], da[1]), pk_bf16(da[2], da[3]), pk_bf16(da[4], da[5]), pk_bf16(da[6], da[7]));
                            const uint4 p1 = make_uint4(pk_bf16(da[8], da[9]), pk_bf16(da[10], da[11]), pk_bf16(da[12], da[13]), pk_bf16(da[14], da[15]));
                            soc = mfma16(p0, wb[2 * hb], soc);
                            soc = mfma16(p1, wb[2 * hb + 1], soc);
                            wb[2 * hb] = wnext[(2 * hb) * 64];
                            wb[2 * hb + 1] = wnext[(2 * hb + 1) * 64];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        Xb[(arow + (i & 3) + 8 * (i >> 2)) * LDXB + EV + C + col] = bf16_of(fmaxf(soc[i] + bso, 0.f));
                }
                TICKC(5)
                __syncthreads();
                TICKC(6)
                // the pooling chains are done with the neighbour masks and every wave has read the occupancy words: cleared for the next step
                for (int i = tid; i < TM * LDM * 2; i += NTHR) masks[i] = 0ull;
                if (tid < 2) occ[tid] = 0;
                // ---- P4: gates over [x | h], and the candidate's x part (same A fragments: three n-tiles per LDS read) ----
                // B fragments run through a ring of RD k-groups requested that many groups before their use; their addresses are formed per
                // step from wave-uniform bases (the opaque zero keeps ~60 of them from being hoisted out of the time loop into spilled
                // registers).  (Kept inline in both bf16 IOC kernels: as a shared helper the same code spills three times as much.)
                f32x16 u, ac = zero16();
                __builtin_amdgcn_s_setprio(2);
                {
                    f32x16 g0 = zero16(), g1 = zero16();
                    int z4;
                    asm volatile("s_mov_b32 %0, 0" : "=s"(z4));
                    const uint4* wg0 = Wg + ((size_t)cb * G16) * 64 + z4;
                    const uint4* wg1 = Wg + ((size_t)(cb + NT) * G16) * 64 + z4;
                    const uint4* wcx = Wc + ((size_t)cb * G16) * 64 + z4;
                    const unsigned ul = (unsigned)lane;
                    constexpr int RD = (NT > 4) ? 2 : IOC16_RD;
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
                TICKC(7)
                __syncthreads();
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
                    __builtin_amdgcn_s_setprio(0);
                    publish_h(h, true, (unsigned)((t & 1) * n_tiles + tile) * (H * TM * 2));       // LDS images + exchange buffer, parity t & 1
                }
                TICKC(8)
                group_publish_wt(cnt);                       // includes the end-of-step __syncthreads
                TICKC(9)
            }
            // ---- score ----
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = sp[i];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
                if (c31 == 0) red[cb * TM + arow + (i & 3) + 8 * (i >> 2)] = v;
            }
            __syncthreads();
            asm volatile("s_mov_b32 %0, %1" : "=s"(row0p) : "s"(row0));           // (again opaque: the epilogue's addresses are formed here)
            if (tid < TM && it == a.iters - 1) {
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < NT; ++c) sc += red[c * TM + tid];
                a.score[row0p + tid] = sc + (float)a.T * a.b_score[0];
            }
            // ---- regression: Y += h_T W_r + b_r ----
            // (every member has published step T-1, i.e. is past its own load of the group's positions Y[.][T-1]: only now may my rows
            // of Y be updated in place)
            group_wait_wt(cnt, tpg * (it * (a.T + 1) + a.T), a.err);
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
                        float* y = a.Y + (size_t)(row0p + arow + (i & 3) + 8 * (i >> 2)) * 2 * a.T + cc;
                        *y = *y + (acc[0][0][i] + bb);
                    }
                }
            }
            group_publish(cnt);                              // pass end: my rows of Y are final for this pass
        }
    }
#ifdef DESIRE_IOC_TIMING
    if (a.dbg && blockIdx.x == 7 && tid == 0)
        for (int k = 0; k < 10; ++k) a.dbg[k] = tacc[k];
#endif
}

static size_t ioc16_cl_lds(const IocArgs& a, bool split) {
    const int H = a.H, TM = 32, E = 16 + 32 + H, KX = E + H, B = a.G * a.G, NT = H / 32;
    size_t b = (size_t)TM * (KX + 8) * 2 + (size_t)TM * (H + 8) * 2 + (size_t)H * (CLMAXM + 8) * 2;
    b += (size_t)TM * (B + 1) * 16 + 16 * 8 + (size_t)CLMAXM * 2 * 4 + TM * 2 * 4 + 3 * 16 * 4 + (size_t)NT * TM * 4 + CLMAXM + 8 + CLMAXM * 8 + 64;
    (void)split;                                           // the bin-split exchange lives inside the Ht tile
    return b;
}
template <int H, bool SPLIT, int TPGT = 0>
static int launch16_cl(const IocArgs& a, u16* hex16, hipStream_t s) {
    auto kern = k_ioc_bf16_cl<H, 16, 32, SPLIT, TPGT>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = ioc16_cl_lds(a, SPLIT);
    const int threads = (H / 32) * 64;
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds) != hipSuccess || per_cu < 1) return -1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    const int tpg = a.mno / 32, n_tiles = a.R / 32;
    // the 16-byte exchange goes through a raw-buffer descriptor (32-bit size and byte offsets): an exchange buffer of 4 GB or more would wrap, and
    // out-of-range raw-buffer accesses return 0 / are dropped instead of faulting -- refuse the shape (R * H >= 2^30 rows x columns)
    if ((unsigned long long)2 * (unsigned long long)n_tiles * (unsigned long long)(H * 32 * 2) >= (1ull << 32)) return -1;
    long cap = (long)per_cu * prop.multiProcessorCount;
    int grid = n_tiles < cap ? n_tiles : (int)cap;           // every workgroup of the grid is resident: members of a group never wait on an unscheduled one
    grid -= grid % tpg;
    if (grid < tpg) return -1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, a, hex16);
    return 0;
}
// a.hex = the exchange buffer (as bf16: 2 * n_tiles * H * 32 elements), a.grp_cnt / a.err as for the fp32 cluster form.
// Pooling split over BINS between the waves (H <= 128; a.variant == 4 selects the column-split form, A/B): the first chain link of
// a bin is run once, by its owner wave, instead of once per wave.
int launch_ioc_bf16_cluster(const IocArgs& a, hipStream_t s) {
    u16* hex16 = reinterpret_cast<u16*>(a.hex);
    const bool split = a.variant != 4 && a.H <= 128;
    if (a.H == 128 && split && a.mno == 128) return launch16_cl<128, true, 4>(a, hex16, s);
    if (a.H == 128 && split && a.mno == 96) return launch16_cl<128, true, 3>(a, hex16, s);
    if (a.H == 128 && split && a.mno == 64) return launch16_cl<128, true, 2>(a, hex16, s);
    if (a.H == 128) return split ? launch16_cl<128, true>(a, hex16, s) : launch16_cl<128, false>(a, hex16, s);
    if (a.H == 64) return split ? launch16_cl<64, true>(a, hex16, s) : launch16_cl<64, false>(a, hex16, s);
    return launch16_cl<256, false>(a, hex16, s);
}
