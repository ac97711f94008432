// kernels_rnn.hip -- persistent recurrent kernels: one workgroup owns a 64-row tile for the whole
// sequence, hidden state lives in registers (accumulator layout) + LDS (A-operand layout), the
// per-step contractions run on the fp32 matrix pipe with weights streamed from L2 in fragment
// order.  Wave w owns hidden columns [32w, 32w+32) of all 64 rows, so the r / u / candidate /
// blend arithmetic of the TF GRUCell is register-local.
//
//   k_encoder   GRU encoders X and Y      model/model.py:136-167,233-241   (static_rnn, zero state)
//   k_decoder   GRU decoder + head        model/model.py:279-289 (+ commented head :315-321)
//   k_ioc       IOC scoring/refinement    absent in the reference (model/model.py:312-313): paper
//   k_neighbor_bins / k_scene_cells       integer paths, bit-exact vs oracle
//
// TF GRUCell: [r,u] = sigmoid([x,h] Wg + bg); c = tanh([x, r*h] Wc + bc); h' = u*h + (1-u)*c.
#include "common.h"
#include "kernels.h"
#include <cstdlib>
#include <map>
#include <tuple>

#define RNN_WG 512         // 8 waves: wave = (column block cb = w&3, M-tile mt = w>>2), two per SIMD

__device__ __forceinline__ f32x16 splat16(float v) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = v;
    return z;
}

// one 32x32 output tile of this wave: acc += A[32 rows][K] . Wp(ntile)
__device__ __forceinline__ void mma1(f32x16& acc, const float* a_lane, const float4* __restrict__ b_lane, int G) {
    f32x16 t[1] = {acc};
    mma_groups<1>(t, a_lane, 0, b_lane, G);
    acc = t[0];
}

// ------------------------------------------------------------------------------------------------
// Encoder: tile = 64 agents, T steps, input (x,y) normalised in-kernel: one fp32 multiply each.
// ------------------------------------------------------------------------------------------------
template <int H, int TM, bool ROLL>
__device__ __forceinline__ void encoder_tile(const EncArgs& a, int blk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDH = H + 4, NT = H >> 5, G = H >> 3, NTHR = NT * (TM / 32) * 64;
    float* hs = smem;                      // [64][LDH]
    float* xs = smem + TM * LDH;        // [64][2]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int cb = w % NT, mt = w / NT;
    const int A = a.n_scenes * a.mno;
    const int a0 = blk * TM;
    const bool active = cb < NT;
    const int col = cb * 32 + (lane & 31);
    float wr0 = 0, wr1 = 0, wu0 = 0, wu1 = 0, wc0 = 0, wc1 = 0, br = 0, bu = 0, bc = 0;
    if (active) {
        wr0 = a.wx_g[col]; wr1 = a.wx_g[2 * H + col];
        wu0 = a.wx_g[H + col]; wu1 = a.wx_g[2 * H + H + col];
        wc0 = a.wx_c[col]; wc1 = a.wx_c[H + col];
        br = a.b_g[col]; bu = a.b_g[H + col]; bc = a.b_c[col];
    }
    f32x16 h = zero16();
    for (int i = tid; i < TM * LDH; i += NTHR) hs[i] = 0.f;
    const float* a_lane = hs + (mt * 32 + (lane & 31)) * LDH + 4 * (lane >> 5);
    float* my_h = hs + (mt * 32 + 4 * (lane >> 5)) * LDH + col;       // + acc-row offset * LDH

    const int n_steps = ROLL ? a.T + a.n_roll : a.T;
    // frame entries (id, x, y) of this thread's agent, one step ahead of their use: frames [n_scenes, T, mno, 3]
    const float* fbase = a.frames;
    float fr0 = 0.f, fr1 = 0.f, fr2 = 0.f;
    if (tid < TM) {
        const int ag = min(a0 + tid, A - 1);
        const int sc = ag / a.mno, slot = ag - sc * a.mno;
        fbase = a.frames + ((size_t)sc * a.T * a.mno + slot) * 3;
        if (a.T > 0) { fr0 = fbase[0]; fr1 = fbase[1]; fr2 = fbase[2]; }
    }
    for (int t = 0; t < n_steps; ++t) {
        if (ROLL && t >= a.T) {
            // sample() prediction step (model/model.py:643-681): the 5-wide head reads (mux, muy, log sx, log sy, corr) off the
            // state, a point is drawn from that bivariate Gaussian (:661-665, Cholesky form, caller's normals), clipped to <= 1.0
            // (:666-669) and becomes this step's input (:680-681 prev_data = newpos)
            __syncthreads();                               // hs holds h_{t-1} of every wave
            if (tid < TM) {
                const int ag = min(a0 + tid, A - 1);
                float p5[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) p5[j] = 0.f;
                for (int c = 0; c < H; ++c) {
                    const float hv = hs[tid * LDH + c];
#pragma unroll
                    for (int j = 0; j < 5; ++j) p5[j] = fmaf(hv, a.w5[c * 5 + j], p5[j]);
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) p5[j] += a.b5[j];
                const size_t ix = ((size_t)(t - a.T) * A + ag) * 2;
                const float sdx = expf(p5[2]), sdy = expf(p5[3]), rho = tanhf(p5[4]);
                const float n0 = a.normals[ix], n1 = a.normals[ix + 1];
                const float x = fminf(p5[0] + sdx * n0, 1.0f);
                const float y = fminf(p5[1] + sdy * (rho * n0 + sqrtf(fmaxf(1.0f - rho * rho, 0.f)) * n1), 1.0f);
                xs[tid * 2] = x; xs[tid * 2 + 1] = y;
                if (a0 + tid < A) { a.roll_out[ix] = x; a.roll_out[ix + 1] = y; }
            }
        } else
        if (tid < TM) {
            // this step's frame entry was requested a step ahead (fr: the load's round trip -- 1 - 2 us of a dependent chain whose step is 8 us at H = 128 --
            // used to sit in front of the step's first barrier); the next one is requested now and lands under this step's contractions
            const int ag = min(a0 + tid, A - 1);
            const float f0 = fr0, f1 = fr1, f2 = fr2;
            if (t + 1 < a.T) { const float* fn = fbase + (size_t)(t + 1) * a.mno * 3; fr0 = fn[0]; fr1 = fn[1]; fr2 = fn[2]; }
            xs[tid * 2 + 0] = __fmul_rn(f1, a.sx);
            xs[tid * 2 + 1] = __fmul_rn(f2, a.sy);
            if (a.sv_x && a0 + tid < A) { a.sv_x[((size_t)ag * a.T + t) * 2] = xs[tid * 2]; a.sv_x[((size_t)ag * a.T + t) * 2 + 1] = xs[tid * 2 + 1]; }
            if (t == a.T - 1 && a0 + tid < A) {
                if (a.p_last) { a.p_last[(size_t)ag * 2] = xs[tid * 2]; a.p_last[(size_t)ag * 2 + 1] = xs[tid * 2 + 1]; }
                if (a.valid) a.valid[ag] = (f0 != 0.f) ? 1 : 0;
            }
        }
        __syncthreads();                                   // xs ready, hs holds h_{t-1}
        f32x16 rh, u;
        if (active) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = mt * 32 + acc_row(i);
                rh[i] = fmaf(xs[row * 2 + 1], wr1, fmaf(xs[row * 2], wr0, br));
                u[i] = fmaf(xs[row * 2 + 1], wu1, fmaf(xs[row * 2], wu0, bu));
            }
            mma1(rh, a_lane, a.Whg + ((size_t)cb * G) * 64 + lane, G);
            mma1(u, a_lane, a.Whg + ((size_t)(cb + NT) * G) * 64 + lane, G);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float r = sigmoidf_(rh[i]);
                rh[i] = r * h[i]; u[i] = sigmoidf_(u[i]);
                if (a.sv_r) {
                    const int ag = a0 + mt * 32 + acc_row(i);
                    if (ag < A) { const size_t ix = ((size_t)ag * a.T + t) * H + col; a.sv_r[ix] = r; a.sv_u[ix] = u[i]; }
                }
            }
        }
        __syncthreads();                                   // every wave done reading h_{t-1}
        if (active) {
#pragma unroll
            for (int i = 0; i < 16; ++i) my_h[((i & 3) + 8 * (i >> 2)) * LDH] = rh[i];
        }
        __syncthreads();
        if (active) {
            f32x16 ac;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = mt * 32 + acc_row(i);
                ac[i] = fmaf(xs[row * 2 + 1], wc1, fmaf(xs[row * 2], wc0, bc));
            }
            mma1(ac, a_lane, a.Whc + ((size_t)cb * G) * 64 + lane, G);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float c = tanhf_(ac[i]);
                h[i] = gru_blend(u[i], h[i], c);
                if (a.sv_c) {
                    const int ag = a0 + mt * 32 + acc_row(i);
                    if (ag < A) { const size_t ix = ((size_t)ag * a.T + t) * H + col; a.sv_c[ix] = c; a.sv_h[ix] = h[i]; }
                }
            }
        }
        __syncthreads();                                   // every wave done reading r*h
        if (active) {
#pragma unroll
            for (int i = 0; i < 16; ++i) my_h[((i & 3) + 8 * (i >> 2)) * LDH] = h[i];
        }
    }
    if (active) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ag = a0 + mt * 32 + acc_row(i);
            if (ag < A) a.out[(size_t)ag * a.ldo + col] = h[i];
        }
    }
}
template <int H, int TM, bool ROLL = false>
__global__ __launch_bounds__((H / 32) * (TM / 32) * 64) void k_encoder(EncArgs a) {
    DYN_N(a, mno, blockIdx.x * TM)                 // (device-side count: one pseudo-scene of mno = P present agents, kernels.h: DynCount)
    encoder_tile<H, TM, ROLL>(a, blockIdx.x);
}
// past and future encoders in ONE launch (they are independent and each is latency-bound: A/32 workgroups stepping through T
// dependent GRU steps): the first nb0 workgroups run a0, the rest a1
template <int H, int TM>
__global__ __launch_bounds__((H / 32) * (TM / 32) * 64) void k_encoder_pair(EncArgs a0, EncArgs a1, int nb0) {
    if (a0.dyn.cnt) {
        // device-side count: both encoders run on the same P present agents; the grid is 2 * nb0 with nb0 sized from the count hint (kernels.h: DynCount.hint) or,
        // without one, for the worst case -- a workgroup strides over its encoder's tiles.  First half of the grid = past encoder, second half = future encoder,
        // as in the plain launch: consecutive workgroups go to consecutive XCDs, and dealing the two encoders' tiles alternately (tried) put every 40-step tile
        // of the future encoder on four of the eight XCDs
        const int P = __builtin_amdgcn_readfirstlane(a0.dyn.cnt[0]);
        a0.mno = P; a1.mno = P;
        const bool second = (int)blockIdx.x >= nb0;
        for (int blk = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x; blk * TM < P; blk += nb0) {
            if (second) encoder_tile<H, TM, false>(a1, blk); else encoder_tile<H, TM, false>(a0, blk);
            __syncthreads();                              // the next tile re-initialises the LDS state
        }
        return;
    }
    const int blk = (int)blockIdx.x < nb0 ? (int)blockIdx.x : (int)blockIdx.x - nb0;
    if ((int)blockIdx.x < nb0) encoder_tile<H, TM, false>(a0, blk);
    else encoder_tile<H, TM, false>(a1, blk);
}
void launch_encoder_pair(const EncArgs& a0, const EncArgs& a1, hipStream_t s) {
    constexpr int TM = 32;
    int nb0 = (a0.n_scenes * a0.mno + TM - 1) / TM, nb1 = (a1.n_scenes * a1.mno + TM - 1) / TM;
    if (a0.dyn.cnt) nb0 = nb1 = (dyn_units(a0.n_scenes * a0.mno, a0.dyn) + TM - 1) / TM;       // (device-side count: tiles dealt alternately, grid = 2 * nb0, strided)
    const size_t lds = (TM * (a0.H + 4) + TM * 2) * sizeof(float);
    const dim3 grid(nb0 + nb1);
    if (a0.H == 256) hipLaunchKernelGGL((k_encoder_pair<256, TM>), grid, dim3(512), lds, s, a0, a1, nb0);
    else if (a0.H == 128) hipLaunchKernelGGL((k_encoder_pair<128, TM>), grid, dim3(256), lds, s, a0, a1, nb0);
    else hipLaunchKernelGGL((k_encoder_pair<64, TM>), grid, dim3(128), lds, s, a0, a1, nb0);
}
void launch_encoder(const EncArgs& a, hipStream_t s) {
    // 32-agent tiles: the encoders are latency-bound (A/32 workgroups of H/32 waves), smaller tiles = more CUs busy
    const int A = a.n_scenes * a.mno;
    constexpr int TM = 32;
    const size_t lds = (TM * (a.H + 4) + TM * 2) * sizeof(float);
    const dim3 grid((A + TM - 1) / TM);
    if (a.n_roll > 0) {
        if (a.H == 256) hipLaunchKernelGGL((k_encoder<256, TM, true>), grid, dim3(512), lds, s, a);
        else if (a.H == 128) hipLaunchKernelGGL((k_encoder<128, TM, true>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_encoder<64, TM, true>), grid, dim3(128), lds, s, a);
        return;
    }
    if (a.H == 256) hipLaunchKernelGGL((k_encoder<256, TM>), grid, dim3(512), lds, s, a);
    else if (a.H == 128) hipLaunchKernelGGL((k_encoder<128, TM>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((k_encoder<64, TM>), grid, dim3(128), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// Decoder: tile = 64 rows; constant input x_z => its contribution (and the biases) is computed once
// and kept in 48 accumulator-layout registers per wave; per step only the h-part contracts (K = H).
// ------------------------------------------------------------------------------------------------
template <int H, int TM, bool SAVE>       // SAVE: training-mode forward (gates / candidate / hidden states kept for BPTT)
__global__ __launch_bounds__((H / 32) * (TM / 32) * 64, 2) void k_decoder(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDH = H + 4, NT = H >> 5, G = H >> 3, NTHR = NT * (TM / 32) * 64, TPR = NTHR / TM;
    float* hs = smem;                          // [64][LDH]  h / r*h as A operand
    float* xs = smem + TM * LDH;            // [64][LDH]  x_z tile (prologue only)
    float* wo = xs + TM * LDH;              // [H][2] head weights
    float* pl = wo + 2 * H;                    // [64][2] last observed position
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int cb = w % NT, mt = w / NT;
    const int row0 = blockIdx.x * TM;
    DYN_P(a, row0)
    const bool active = cb < NT;
    const int col = cb * 32 + (lane & 31);
    for (int i = tid; i < TM * (H >> 2); i += NTHR) {
        const int r = i / (H >> 2), c4 = i - r * (H >> 2);
        const int row = min(row0 + r, a.R - 1);
        *reinterpret_cast<float4*>(xs + r * LDH + c4 * 4) =
            *reinterpret_cast<const float4*>(a.xz + (size_t)row * H + c4 * 4);
        const int ag = agent_of_row(row, a.K, a.mno);
        *reinterpret_cast<float4*>(hs + r * LDH + c4 * 4) =
            *reinterpret_cast<const float4*>(a.Hx + (size_t)ag * a.ldhx + c4 * 4);
    }
    for (int i = tid; i < 2 * H; i += NTHR) wo[i] = a.w_head[i];
    if (tid < TM) {
        const int ag = agent_of_row(min(row0 + tid, a.R - 1), a.K, a.mno);
        pl[tid * 2] = a.p_last[(size_t)ag * 2];
        pl[tid * 2 + 1] = a.p_last[(size_t)ag * 2 + 1];
    }
    __syncthreads();
    // head weights of this thread's 16 columns, in the order it walks them.  The walk starts at chunk hrot: ds_read_b128 is serviced in
    // 16-lane groups over a 256-byte bank row, and with rows (H + 4) floats apart the threads q and q + 4 of a row would meet in one
    // 16-byte slot (the scalar form this replaces read h and both weights from LDS per column: 4- and 8-way conflicts, a fifth of the
    // kernel's LDS cycles)
    const int hq = tid % TPR, hrot = TPR >= 8 ? (hq >> 2) * (TPR == 8 ? 2 : 1) : 0;
    float hw0[16], hw1[16];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = hq * 16 + 4 * ((jj + hrot) & 3) + e;
            hw0[4 * jj + e] = wo[c * 2]; hw1[4 * jj + e] = wo[c * 2 + 1];
        }
    const float* x_lane = xs + (mt * 32 + (lane & 31)) * LDH + 4 * (lane >> 5);
    const float* a_lane = hs + (mt * 32 + (lane & 31)) * LDH + 4 * (lane >> 5);
    const float* r_lane = xs + (mt * 32 + (lane & 31)) * LDH + 4 * (lane >> 5);     // r*h operand: the x_z tile's space after the prologue
    float* my_h = hs + (mt * 32 + 4 * (lane >> 5)) * LDH + col;
    float* my_rh = xs + (mt * 32 + 4 * (lane >> 5)) * LDH + col;
    f32x16 xr, xu, xc, h;
    if (active) {
        xr = splat16(a.b_g[col]); xu = splat16(a.b_g[H + col]); xc = splat16(a.b_c[col]);
        mma1(xr, x_lane, a.Wxg + ((size_t)cb * G) * 64 + lane, G);
        mma1(xu, x_lane, a.Wxg + ((size_t)(cb + NT) * G) * 64 + lane, G);
        mma1(xc, x_lane, a.Wxc + ((size_t)cb * G) * 64 + lane, G);
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] = my_h[((i & 3) + 8 * (i >> 2)) * LDH];
    }
    __syncthreads();                                   // every wave is done with the x_z tile: its space now carries r*h
    const float bh0 = a.b_head[0], bh1 = a.b_head[1];
    // two barriers per step: h and r*h live in separate tiles, so the gates read h while nobody writes it, the candidate
    // reads r*h while nobody writes it, and h_t is published (for the head and the next step) after the first barrier
    for (int t = 0; t < a.T; ++t) {
        f32x16 rh, u;
        if (active) {
            rh = xr; u = xu;
            mma1(rh, a_lane, a.Whg + ((size_t)cb * G) * 64 + lane, G);
            mma1(u, a_lane, a.Whg + ((size_t)(cb + NT) * G) * 64 + lane, G);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float r = sigmoidf_(rh[i]);
                rh[i] = r * h[i]; u[i] = sigmoidf_(u[i]);
                if (SAVE && a.sv_r) {                           // training: keep the gates for BPTT
                    const int rl = mt * 32 + acc_row(i);
                    if (row0 + rl < a.R) {
                        const size_t ix = ((size_t)(row0 + rl) * a.T + t) * H + col;
                        a.sv_r[ix] = r; a.sv_u[ix] = u[i];
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) my_rh[((i & 3) + 8 * (i >> 2)) * LDH] = rh[i];
        }
        __syncthreads();
        if (active) {
            f32x16 ac = xc;
            mma1(ac, r_lane, a.Whc + ((size_t)cb * G) * 64 + lane, G);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float c = tanhf_(ac[i]);
                h[i] = gru_blend(u[i], h[i], c);
                if (SAVE && a.sv_c) {
                    const int rl = mt * 32 + acc_row(i);
                    if (row0 + rl < a.R) a.sv_c[((size_t)(row0 + rl) * a.T + t) * H + col] = c;
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                my_h[((i & 3) + 8 * (i >> 2)) * LDH] = h[i];
                const int rl = mt * 32 + acc_row(i);
                if (SAVE && a.hdump && row0 + rl < a.R) a.hdump[((size_t)(row0 + rl) * a.T + t) * H + col] = h[i];
            }
        }
        __syncthreads();
        {   // head: y = p_last + h W_o + b_o ; TPR threads per row, 16 columns each, their weights in registers (hw0 / hw1)
            const int r = tid / TPR;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float4 hv = *reinterpret_cast<const float4*>(hs + r * LDH + hq * 16 + 4 * ((jj + hrot) & 3));
                s0 = fmaf(hv.x, hw0[4 * jj], s0); s1 = fmaf(hv.x, hw1[4 * jj], s1);
                s0 = fmaf(hv.y, hw0[4 * jj + 1], s0); s1 = fmaf(hv.y, hw1[4 * jj + 1], s1);
                s0 = fmaf(hv.z, hw0[4 * jj + 2], s0); s1 = fmaf(hv.z, hw1[4 * jj + 2], s1);
                s0 = fmaf(hv.w, hw0[4 * jj + 3], s0); s1 = fmaf(hv.w, hw1[4 * jj + 3], s1);
            }
            const int q8 = hq;
            s0 += __shfl_xor(s0, 1); s1 += __shfl_xor(s1, 1);
            s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
            if (TPR >= 8) { s0 += __shfl_xor(s0, 4); s1 += __shfl_xor(s1, 4); }
            if (TPR >= 16) { s0 += __shfl_xor(s0, 8); s1 += __shfl_xor(s1, 8); }
            if (q8 == 0 && row0 + r < a.R) {
                float2 y = make_float2(pl[r * 2] + (s0 + bh0), pl[r * 2 + 1] + (s1 + bh1));
                *reinterpret_cast<float2*>(a.Y + ((size_t)(row0 + r) * a.T + t) * 2) = y;
            }
        }
        // the head's reads of h_t are ordered before the next rewrite of the h tile by the next step's first barrier
    }
}
template <int H, int TM>
static void launch_decoder_t(const DecArgs& a, hipStream_t s) {
    const size_t lds = (2 * TM * (H + 4) + 2 * H + TM * 2) * sizeof(float);
    if (a.sv_r || a.sv_c || a.hdump) {
        allow_big_lds(k_decoder<H, TM, true>);
        hipLaunchKernelGGL((k_decoder<H, TM, true>), dim3((a.R + TM - 1) / TM), dim3((H / 32) * (TM / 32) * 64), lds, s, a);
    } else {
        allow_big_lds(k_decoder<H, TM, false>);
        hipLaunchKernelGGL((k_decoder<H, TM, false>), dim3((a.R + TM - 1) / TM), dim3((H / 32) * (TM / 32) * 64), lds, s, a);
    }
}
void launch_decoder(const DecArgs& a, hipStream_t s) {
    if (a.H == 256) launch_decoder_t<256, 32>(a, s);
    else if (a.H == 128) {                              // 32-row tiles: 4 waves + 34 KB LDS -> two workgroups per CU
        launch_decoder_t<128, 32>(a, s);
    }
    else launch_decoder_t<64, 64>(a, s);
}

// ------------------------------------------------------------------------------------------------
// IOC scoring / refinement.  Tile = 64 rows = 64/mno complete (scene,k) groups, so social pooling
// never leaves the workgroup.  Per step:
//   e_v = relu(v W_v + b_v)                       VALU
//   e_s = grid[cell(y_t)]                         coalesced 128-B gather, cell index bit-exact
//   e_r = relu(sum_b pool_b(h_{t-1}) W_b + b_s)   per bin: VALU builds the pooled operand from a
//                                                 neighbour bitmask, MFMA contracts it (K = H)
//   h_t = GRU([e_v|e_s|e_r], h_{t-1})             K = E + H
//   score += h_t . w_s + b_s
// after T steps: dY = h_T W_r + b_r ; Y += dY.
// ------------------------------------------------------------------------------------------------
// TM = rows per workgroup (32 or 64; must hold whole (scene,k) groups: TM % mno == 0), 8 threads per
// row.  TM=32 runs 4 waves and ~77 KB of LDS so that TWO workgroups share a CU: their barrier/VALU
// phases interleave with each other's MFMA phases.
// neighbour bit-masks: 32 bits are enough when the tile holds 32 rows (mno <= 32), which keeps a 36-bin tile under half
// the LDS of a CU (two workgroups per CU); 64-row tiles use 64 bits
#include "cluster.h"
#include "split.h"
template <int TM> struct MaskT { typedef unsigned long long type; };
template <> struct MaskT<32> { typedef unsigned type; };
__device__ __forceinline__ int ffs_(unsigned m) { return __ffs((int)m); }
__device__ __forceinline__ int ffs_(unsigned long long m) { return __ffsll((long long)m); }
#ifdef DESIRE_IOC_TIMING
#define TICK(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#else
#define TICK(k)
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
// CP ("compact pooling", opt-in: DESIRE_IOC_VARIANT=8, TM = 32): only the rows that have a neighbour in bin b are built,
// packed into the first rows of the operand tile, contracted as 16-row v_mfma_f32_16x16x4_f32 tiles and added into the rows
// they belong to -- at the bench's density (about a quarter of the rows per bin) that halves the pooling MFMAs and builds a
// quarter of the operand rows.  Per-bin partial sums are added in ascending bin order (not one running accumulator), so the
// result differs from the default form by fp32 rounding only.
// NSPL > 1 ("bin split", few tiles: launch_ioc): NSPL workgroups per tile, on as many CUs.  Member m contracts the social bins b = m
// (mod NSPL) only, the members add their partial e_r pre-activations through global memory once per step (cluster.h write-through
// hand-off, fixed member order: every member holds bit-identical state afterwards) and everything else runs redundantly in each of
// them; member 0 writes the results.  One pass only (a second pass would need member 0's refined Y in every member).
// PAD: padded tiles (IocArgs.gpt; slot classes that do not divide 32).  A template parameter, so that the packed-row instantiations -- the headline's
// among them -- compile to exactly the code they had before.
template <int H, int EV, int C, int TM, bool TRAIN, bool CP = false, int NSPL = 1, bool PAD = false>
__global__ __launch_bounds__((H / 32) * (TM / 32) * 64, ((H / 32) * (TM / 32) <= 2) ? 1 : 2) void k_ioc(IocArgs a) {
#ifdef DESIRE_IOC_TIMING
    long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = H >> 5, E = EV + C + H, KX = E + H, LDX = KX + 4, LDB = H + 4;
    constexpr int NTHR = NT * (TM / 32) * 64, TPR = NTHR / TM;      // TPR threads per row in the VALU phases (8 or 16)
    constexpr int G8 = KX >> 3, GH = H >> 3;
    constexpr int NCH = H / (4 * TPR);                   // float4 chunks per thread in the pooled build
    const int B = a.G * a.G;
    float* XH = smem;                                   // [TM+1][LDX] [e_v | e_s | e_r | h]; row TM stays zero
    float* AB = XH + (TM + 1) * LDX;                    // [2][TM][LDB] pooled operand, double buffered;
                                                        //   buffer 0 doubles as the r*h operand of the candidate
    typedef typename MaskT<TM>::type mask_t;
    mask_t* masks = reinterpret_cast<mask_t*>(AB + 2 * TM * LDB);   // [TM][B]
    float* pc = reinterpret_cast<float*>(masks + TM * B);      // [TM][2] current position
    float* pp = pc + TM * 2;                            // [TM][2] previous position
    float* wv = pp + TM * 2;                            // [2][E_v] + [E_v]
    float* red = wv + 3 * EV;                           // [NT][TM] score reduction
    unsigned char* vld = reinterpret_cast<unsigned char*>(red + NT * TM);  // [TM]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + TM);                 // [2] bins that hold a neighbour anywhere in the tile
    unsigned* rowbits = occ + 2;                                           // CP: [B] rows of the tile with a neighbour in bin b
    unsigned* rowlist = rowbits + 36;                                      // CP: [2][TM/4] packed u8: tile row of operand row s

    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int cb = w % NT, mt = w / NT;
    const int tile = NSPL > 1 ? (int)blockIdx.x / NSPL : (int)blockIdx.x, member = NSPL > 1 ? (int)blockIdx.x % NSPL : 0;
    const int row0 = tile * TM;
    IOC_DYN(a)                                          // (a slot class counted on the device: kernels.h DynCount; the grid is the worst case's)
    if (a.dyn.cnt && row0 >= a.R) return;
    unsigned long long my_bins = ~0ull;
    if (NSPL > 1) { my_bins = 0ull; for (int b = member; b < 64; b += NSPL) my_bins |= 1ull << b; }
    const bool active = cb < NT;
    const int col = cb * 32 + (lane & 31);
    const int r8 = tid / TPR, q8 = tid % TPR;           // TPR threads per row for the VALU phases
    const int my_row = min(row0 + r8, a.R - 1);
    // (padded tiles, a.gpt > 0: the tile holds gpt whole groups and dead rows behind them -- kernels.h: IocArgs)
    const int gpt = PAD ? a.gpt : 0;
    const bool dead_row = gpt && (r8 / a.mno >= gpt || tile * gpt + r8 / a.mno >= a.ngrp);
    const int my_scene = gpt ? min(tile * gpt + min(r8 / a.mno, gpt - 1), a.ngrp - 1) / a.K : my_row / (a.K * a.mno);
    const int grp_base = (r8 / a.mno) * a.mno;          // first local row of my (scene,k) group
    const int my_slot = r8 - grp_base;
    const int n_nb = dead_row ? 0 : a.mno;              // slots my row looks for neighbours in

    for (int i = tid; i < 3 * EV; i += NTHR) wv[i] = (i < 2 * EV) ? a.w_vel[i] : a.b_vel[i - 2 * EV];
    for (int i = tid; i < LDX; i += NTHR) XH[TM * LDX + i] = 0.f;
    if (tid < TM) { const int ag = ioc_agent_of_row(min(row0 + tid, a.R - 1), a.K, a.mno, gpt, a.ngrp); vld[tid] = ag >= 0 ? a.valid[ag] : 0; }

    float bgr = 0, bgu = 0, bcc = 0, bso = 0, wsc = 0;
    if (active) { bgr = a.b_g[col]; bgu = a.b_g[H + col]; bcc = a.b_c[col]; bso = a.b_soc[col]; wsc = a.w_score[col]; }
    const float* x_lane = XH + (mt * 32 + (lane & 31)) * LDX + 4 * (lane >> 5);
    float* my_x = XH + (mt * 32 + 4 * (lane >> 5)) * LDX + col;        // + acc-row * LDX (+ column base)
    const float* grid = a.grids + (size_t)a.grid_of_scene[my_scene] * a.Gh * a.Gw * C;
    // training-mode saves: (uniform tile base) + (32-bit offset inside the tile) keeps the addresses out of the VGPR budget
    const size_t sv_tb = TRAIN ? (size_t)row0 * a.T * H : 0;
    float* sv_r_t = TRAIN ? a.sv_r + sv_tb : nullptr; float* sv_u_t = TRAIN ? a.sv_u + sv_tb : nullptr;
    float* sv_c_t = TRAIN ? a.sv_c + sv_tb : nullptr; float* sv_h_t = TRAIN ? a.sv_h + sv_tb : nullptr;
    auto sv_off = [&](int i, int t) { return (unsigned)(((mt * 32 + 4 * (lane >> 5) + (i & 3) + 8 * (i >> 2)) * a.T + t) * H + col); };

    // pooled operand of bin b: ab[r8][:] = sum_{j in mask} h_{t-1}[group row j][:].  Thread q8 owns the
    // float4 chunks q8, q8+8, ... so the 8 lanes of a row touch 128 contiguous bytes (no bank conflicts).
    // Branch-free for up to NS neighbours per (row, bin): the s-th set bit selects a source row, a missing
    // one selects the all-zero row TM, so the 4*NS LDS reads are independent and issue back to back; the
    // (rare) overflow is finished by a wave-uniform loop.  Summation order = ascending slot: deterministic.
    constexpr int NS = 2;
    auto build = [&](int b, int buf) {
        float* ab = AB + buf * TM * LDB + r8 * LDB;
        mask_t m2 = masks[r8 * B + b];
        int off[NS];
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            off[sl] = m2 ? (grp_base + ffs_(m2) - 1) * LDX : TM * LDX;
            m2 &= m2 - 1;
        }
        float4 s[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float4 v0 = *reinterpret_cast<const float4*>(XH + off[0] + E + q8 * 4 + c * 4 * TPR);
            const float4 v1 = *reinterpret_cast<const float4*>(XH + off[1] + E + q8 * 4 + c * 4 * TPR);
            s[c].x = v0.x + v1.x; s[c].y = v0.y + v1.y; s[c].z = v0.z + v1.z; s[c].w = v0.w + v1.w;
        }
        if (__any(m2 != 0)) {
            while (m2) {
                const int j = ffs_(m2) - 1;
                m2 &= m2 - 1;
                const float* src = XH + (grp_base + j) * LDX + E + q8 * 4;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float4 v = *reinterpret_cast<const float4*>(src + c * 4 * TPR);
                    s[c].x += v.x; s[c].y += v.y; s[c].z += v.z; s[c].w += v.w;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(ab + q8 * 4 + c * 4 * TPR) = s[c];
    };

    for (int it = 0; it < a.iters; ++it) {
        // h_0 = Hx[agent]
        for (int i = tid; i < TM * (H >> 2); i += NTHR) {
            const int r = i / (H >> 2), c4 = i - r * (H >> 2);
            const int ag = ioc_agent_of_row(min(row0 + r, a.R - 1), a.K, a.mno, gpt, a.ngrp);
            *reinterpret_cast<float4*>(XH + r * LDX + E + c4 * 4) =
                ag >= 0 ? *reinterpret_cast<const float4*>(a.Hx + (size_t)ag * a.ldhx + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < TM) {
            const int ag = ioc_agent_of_row(min(row0 + tid, a.R - 1), a.K, a.mno, gpt, a.ngrp);
            pp[tid * 2] = ag >= 0 ? a.p_last[(size_t)ag * 2] : 0.f;
            pp[tid * 2 + 1] = ag >= 0 ? a.p_last[(size_t)ag * 2 + 1] : 0.f;
        }
        __syncthreads();
        f32x16 h = zero16(), sp = zero16();
        if (active) {
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = my_x[((i & 3) + 8 * (i >> 2)) * LDX + E];
        }

        // positions of step 0 / cleared masks before the first P1
        float2 ynext = make_float2(0.f, 0.f);
        if (tid < TM) {
            const float2 y0 = *reinterpret_cast<const float2*>(a.Y + ((size_t)min(row0 + tid, a.R - 1) * a.T) * 2);
            pc[tid * 2] = y0.x; pc[tid * 2 + 1] = y0.y;
        }
        for (int i = tid; i < TM * B; i += NTHR) masks[i] = 0;
        if (tid < 2) occ[tid] = 0;
        if (CP && tid < B) rowbits[tid] = 0;
        __syncthreads();
        const float* rh_lane = AB + (mt * 32 + (lane & 31)) * LDB + 4 * (lane >> 5);     // r*h operand (AB buffer 0)
        float* my_rh = AB + (mt * 32 + 4 * (lane >> 5)) * LDB + col;
        constexpr int GX = E >> 3;                                                        // x-part groups

        for (int t = 0; t < a.T; ++t) {
            TICK(0)
            // prefetch next step's positions (consumed at the end of this step)
            if (tid < TM && t + 1 < a.T)
                ynext = *reinterpret_cast<const float2*>(a.Y + ((size_t)min(row0 + tid, a.R - 1) * a.T + t + 1) * 2);
            // ---- P1: e_v, e_s, neighbour masks ----
            {
                const float px = pc[r8 * 2], py = pc[r8 * 2 + 1];
                const float vx = px - pp[r8 * 2], vy = py - pp[r8 * 2 + 1];
                constexpr int per = EV / TPR;
#pragma unroll
                for (int j = q8 * per; j < (q8 + 1) * per; ++j)
                    XH[r8 * LDX + j] = fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f);
                int cy, cx;
                scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
                const float* gsrc = grid + ((size_t)cy * a.Gw + cx) * C;
                constexpr int cper = C / TPR;                                   // 8, 4 (float4s) or 2 (float2) channels per thread
                if (cper >= 4) {
#pragma unroll
                    for (int j = q8 * cper; j < (q8 + 1) * cper; j += 4)
                        *reinterpret_cast<float4*>(XH + r8 * LDX + EV + j) = *reinterpret_cast<const float4*>(gsrc + j);
                } else {
                    *reinterpret_cast<float2*>(XH + r8 * LDX + EV + q8 * 2) = *reinterpret_cast<const float2*>(gsrc + q8 * 2);
                }
                float nbw, nbh;
                nb_opaque(a.nb_w, a.nb_h, nbw, nbh);
                const unsigned long long oc = nb_search<4>(pc, vld, grp_base, n_nb, q8, TPR, my_slot, px, py, nbw, nbh, a.G, a.bin_tab,
                                                          [&](int j, int b) {
                                                              atomicOr(&masks[r8 * B + b], (mask_t)1 << j);
                                                              if (CP) atomicOr(&rowbits[b], 1u << r8);
                                                          });
                nb_publish_occ(oc, occ, B);
                if (CP) {                                           // e_r columns double as the accumulation tile of the bin loop
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
                        *reinterpret_cast<float4*>(XH + r8 * LDX + EV + C + q8 * 4 + c * 4 * TPR) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            __syncthreads();
            TICK(1)
            // ---- P2/P3 (compact form) ----
            if constexpr (CP) {
                unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
                om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
                constexpr int T16 = H / 16;                          // 16-column tiles per row / 16-k groups per contraction
                // operand row s of bin b = the s-th tile row that has a neighbour there (ascending rows)
                auto build_c = [&](int b, int buf) {
                    const unsigned rb = rowbits[b];
                    if (!((rb >> r8) & 1u)) return;
                    const int slot = __popc(rb & ((1u << r8) - 1u));
                    float* ab = AB + buf * TM * LDB + slot * LDB;
                    mask_t m2 = masks[r8 * B + b];
                    // first two neighbours branch-free (a missing second one reads the all-zero row TM), the rare rest in a loop
                    const int o0 = (grp_base + ffs_(m2) - 1) * LDX;
                    m2 &= m2 - 1;
                    const int o1 = m2 ? (grp_base + ffs_(m2) - 1) * LDX : TM * LDX;
                    m2 &= m2 - 1;
                    float4 sacc[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float4 v0 = *reinterpret_cast<const float4*>(XH + o0 + E + q8 * 4 + c * 4 * TPR);
                        const float4 v1 = *reinterpret_cast<const float4*>(XH + o1 + E + q8 * 4 + c * 4 * TPR);
                        sacc[c].x = v0.x + v1.x; sacc[c].y = v0.y + v1.y; sacc[c].z = v0.z + v1.z; sacc[c].w = v0.w + v1.w;
                    }
                    while (m2) {
                        const int j = ffs_(m2) - 1;
                        m2 &= m2 - 1;
                        const float* src = XH + (grp_base + j) * LDX + E + q8 * 4;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const float4 v = *reinterpret_cast<const float4*>(src + c * 4 * TPR);
                            sacc[c].x += v.x; sacc[c].y += v.y; sacc[c].z += v.z; sacc[c].w += v.w;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(ab + q8 * 4 + c * 4 * TPR) = sacc[c];
                    if (q8 == 0) reinterpret_cast<unsigned char*>(rowlist + buf * (TM / 4))[slot] = (unsigned char)r8;
                };
                int buf = 0;
                if (om) build_c(ffs_(om) - 1, 0);
                __syncthreads();
                while (om) {
                    const int b = ffs_(om) - 1;
                    om &= om - 1;
                    // this wave's two 16-column tiles of W_b, all T16 k-groups: requested now, consumed after the next build
                    const float4* w0 = a.Wsoc_c + ((size_t)(b * T16 + 2 * cb) * T16) * 64 + lane;
                    float4 wf[2][T16];
                    if (active) {
#pragma unroll
                        for (int g = 0; g < T16; ++g) { wf[0][g] = w0[g * 64]; wf[1][g] = w0[(T16 + g) * 64]; }
                    }
                    if (om) build_c(ffs_(om) - 1, buf ^ 1);
                    const int nrows = __popc(__builtin_amdgcn_readfirstlane((int)rowbits[b]));
                    if (active) {
                        for (int c16 = 0; c16 * 16 < nrows; ++c16) {
                            f32x4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                            const float* ap = AB + buf * TM * LDB + (16 * c16 + (lane & 15)) * LDB + 4 * (lane >> 4);
#pragma unroll
                            for (int g = 0; g < T16; ++g) {
                                const float4 av = *reinterpret_cast<const float4*>(ap + 16 * g);
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wf[0][g].x, a0, 0, 0, 0);
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wf[1][g].x, a1, 0, 0, 0);
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wf[0][g].y, a0, 0, 0, 0);
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wf[1][g].y, a1, 0, 0, 0);
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wf[0][g].z, a0, 0, 0, 0);
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wf[1][g].z, a1, 0, 0, 0);
                                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wf[0][g].w, a0, 0, 0, 0);
                                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wf[1][g].w, a1, 0, 0, 0);
                            }
                            // accumulator element i of a lane = operand row 16 c16 + 4 (lane>>4) + i, column (lane&15) of the tile
                            const unsigned rl4 = rowlist[buf * (TM / 4) + 4 * c16 + (lane >> 4)];
                            float* ex = XH + EV + C + cb * 32 + (lane & 15);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (16 * c16 + 4 * (lane >> 4) + i < nrows) {
                                    float* dst = ex + ((rl4 >> (8 * i)) & 0xffu) * LDX;
                                    dst[0] += a0[i];
                                    dst[16] += a1[i];
                                }
                            }
                        }
                    }
                    __syncthreads();
                    buf ^= 1;
                }
                if (active) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float* e = my_x + ((i & 3) + 8 * (i >> 2)) * LDX + EV + C;
                        *e = fmaxf(*e + bso, 0.f);
                    }
                }
            } else
            // ---- P2/P3: social pooling -> e_r, over the bins that hold a neighbour somewhere in this tile only
            //      (an empty bin's pooled operand is all zeros: skipping it drops exact-zero products, and on real
            //      tracks most of the window is empty).  build(next) and the contraction of the current bin sit between
            //      the same two barriers, so waves that finish building early start their MFMAs while others build
            {
                f32x16 soc = zero16();          // biases join after the contraction (a splat start value would pin 16 registers)
                unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
                om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
                if (NSPL > 1) om &= my_bins;
                int buf = 0;
                if (om) build(ffs_(om) - 1, 0);
                __syncthreads();
                TICK(2)
                // wave priority by phase (round 6): two workgroups per CU = two waves per SIMD; a wave inside its bin loop (3) or its gate / candidate
                // contraction (2) wins the issue arbitration against one that builds operands or sits in the position phase (0).  Same-box ABAB on the
                // headline: k_ioc 78.77 - 79.10 ms without, 78.55 - 78.65 with (the reverse order, 1 / 3: 79.1 - 79.25).  Where it pays properly is the
                // bf16 cluster kernel (kernels_bf16_cl.hip: -6.5 %); k_ioc_bf16 and k_ioc_x3 measured neutral and carry none.
                __builtin_amdgcn_s_setprio(3);
                while (om) {
                    const int b = ffs_(om) - 1;
                    om &= om - 1;
                    const float4* wb = a.Wsoc + ((size_t)(b * NT + cb) * GH) * 64 + lane;
                    MmaHead hd;
                    if (active) mma_begin(hd, wb, GH);             // this bin's first weight fragments travel while the next operand is built
                    if (om) build(ffs_(om) - 1, buf ^ 1);
                    TICK(3)
                    if (active) {
                        f32x16 t1[1] = {soc};
                        const float* ap1[1] = {AB + buf * TM * LDB + (mt * 32 + (lane & 31)) * LDB + 4 * (lane >> 5)};
                        mma_run<1>(t1, ap1, wb, GH, hd);
                        soc = t1[0];
                    }
                    TICK(4)
                    __syncthreads();
                    TICK(5)
                    buf ^= 1;
                }
                __builtin_amdgcn_s_setprio(0);
                if constexpr (NSPL > 1) {
                    // partial sums of this member's bins -> slot [tile][step parity][member]; total = fixed-order sum over the members
                    const int sidx = it * a.T + t;
                    float* slot = a.hex + (((size_t)tile * 2 + (sidx & 1)) * NSPL) * (TM * H);
                    uint2* mine = reinterpret_cast<uint2*>(slot + (size_t)member * TM * H) + (size_t)w * 8 * 64 + lane;
#pragma unroll
                    for (int q = 0; q < 8; ++q) st_agent_u64(mine + q * 64, make_uint2(__float_as_uint(soc[2 * q]), __float_as_uint(soc[2 * q + 1])));
                    group_publish_wt(a.grp_cnt + tile);
                    group_wait_wt(a.grp_cnt + tile, NSPL * (sidx + 1), a.err);
                    f32x16 tot = zero16();
#pragma unroll
                    for (int m = 0; m < NSPL; ++m) {
                        if (m == member) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) tot[i] += soc[i];
                        } else {
                            const uint2* src = reinterpret_cast<const uint2*>(slot + (size_t)m * TM * H) + (size_t)w * 8 * 64 + lane;
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const uint2 v = ld_agent_u64(src + q * 64);
                                tot[2 * q] += __uint_as_float(v.x); tot[2 * q + 1] += __uint_as_float(v.y);
                            }
                        }
                    }
                    soc = tot;
                }
                if (active) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) my_x[((i & 3) + 8 * (i >> 2)) * LDX + EV + C] = fmaxf(soc[i] + bso, 0.f);
                }
            }
            __syncthreads();
            TICK(6)
            if (TRAIN) {                                                 // keep x_t = [e_v | e_s | e_r] for the weight gradients
                for (int i = tid; i < TM * (E >> 2); i += NTHR) {
                    const int r = i / (E >> 2), c4 = i - r * (E >> 2);
                    if (row0 + r < a.R)
                        *reinterpret_cast<float4*>(a.sv_x + ((size_t)(row0 + r) * a.T + t) * E + c4 * 4) =
                            *reinterpret_cast<const float4*>(XH + r * LDX + c4 * 4);
                }
            }
            __builtin_amdgcn_s_setprio(2);
            // ---- P4: gates over [x | h]; r*h goes to its own LDS tile so no "done reading h" barrier is needed ----
            f32x16 rh, u;
            if (active) {
                rh = zero16(); u = zero16();
                mma1(rh, x_lane, a.Wg + ((size_t)cb * G8) * 64 + lane, G8);
                mma1(u, x_lane, a.Wg + ((size_t)(cb + NT) * G8) * 64 + lane, G8);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float r = sigmoidf_(rh[i] + bgr);
                    if (TRAIN && row0 + mt * 32 + acc_row(i) < a.R) sv_r_t[sv_off(i, t)] = r;
                    rh[i] = r * h[i];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) my_rh[((i & 3) + 8 * (i >> 2)) * LDB] = rh[i];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    u[i] = sigmoidf_(u[i] + bgu);
                    if (TRAIN && row0 + mt * 32 + acc_row(i) < a.R) sv_u_t[sv_off(i, t)] = u[i];
                }
            }
            __syncthreads();
            TICK(7)
            // ---- P5: candidate over [x | r*h], blend, score ----
            if (active) {
                f32x16 ac = zero16();
                mma1(ac, x_lane, a.Wc + ((size_t)cb * G8) * 64 + lane, GX);
                mma1(ac, rh_lane, a.Wc + ((size_t)cb * G8 + GX) * 64 + lane, GH);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float c = tanhf_(ac[i] + bcc);
                    h[i] = gru_blend(u[i], h[i], c);
                    sp[i] = fmaf(h[i], wsc, sp[i]);
                    if (TRAIN && row0 + mt * 32 + acc_row(i) < a.R) { sv_c_t[sv_off(i, t)] = c; sv_h_t[sv_off(i, t)] = h[i]; }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i)                               // h slot: last read by the gates, one barrier ago
                    my_x[((i & 3) + 8 * (i >> 2)) * LDX + E] = h[i];
            }
            __builtin_amdgcn_s_setprio(0);
            // next step's P0: positions and cleared masks (masks were last read in the bin loop)
            if (tid < TM) {
                pp[tid * 2] = pc[tid * 2]; pp[tid * 2 + 1] = pc[tid * 2 + 1];
                pc[tid * 2] = ynext.x; pc[tid * 2 + 1] = ynext.y;
            }
            for (int i = tid; i < TM * B; i += NTHR) masks[i] = 0;
            if (tid < 2) occ[tid] = 0;
            if (CP && tid < B) rowbits[tid] = 0;
            __syncthreads();
            TICK(8)
        }
        // ---- score: sum the per-lane partials over the 32 columns of this wave, then over column blocks ----
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = active ? sp[i] : 0.f;
            v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
            if ((lane & 31) == 0) red[cb * TM + mt * 32 + acc_row(i)] = v;
        }
        __syncthreads();
        // bin-split form: a hand-off that timed out (members of a tile not co-resident) has set the error word; its tile's results are
        // then made NaN instead of silently wrong (the host reports the word on the next call: api.hip)
        bool poisoned = false;
        if constexpr (NSPL > 1) poisoned = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        if (tid < TM && row0 + tid < a.R && it == a.iters - 1 && member == 0) {
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < NT; ++c) sc += red[c * TM + tid];
            a.score[row0 + tid] = poisoned ? __builtin_nanf("") : sc + (float)a.T * a.b_score[0];
        }
        // ---- regression: Y += h_T W_r + b_r   (columns = (t, xy) flattened) ----
        for (int nt = cb; nt < a.NTreg; nt += NT) {
            f32x16 acc = zero16();
            mma1(acc, x_lane + E, a.Wreg + ((size_t)nt * GH) * 64 + lane, GH);
            const int cc = nt * 32 + (lane & 31);
            if (cc < 2 * a.T) {
                const float bb = a.b_reg[cc];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = row0 + mt * 32 + acc_row(i);
                    if (row < a.R && member == 0) {
                        float* y = a.Y + (size_t)row * 2 * a.T + cc;
                        *y = poisoned ? __builtin_nanf("") : *y + (acc[i] + bb);
                    }
                }
            }
        }
        __syncthreads();
    }
#ifdef DESIRE_IOC_TIMING
    if (a.dbg && tile == 7 && member == 0 && tid == 0)
        for (int k = 0; k < 10; ++k) a.dbg[k] = tacc[k];
#endif
}
static size_t ioc_lds_bytes(const IocArgs& a, int TM) {
    const int EV = 16, H = a.H, NT = H / 32, E = EV + 32 + H, LDX = E + H + 4, LDB = H + 4, B = a.G * a.G;
    size_t f = (size_t)(TM + 1) * LDX + 2 * TM * LDB + (size_t)TM * B * (TM == 32 ? 1 : 2) + TM * 4 + 3 * EV + NT * TM;
    return f * sizeof(float) + TM + 64 + 256;              // + occupancy words, CP row bitmaps / row lists
}
template <int H, int TM>
static void launch_ioc_t(const IocArgs& a, hipStream_t s) {
    const dim3 grid((a.R + TM - 1) / TM), block((H / 32) * (TM / 32) * 64);
    if (a.sv_h) {                                              // training-mode forward: keeps x_t, r, u, c, h per step
        if constexpr (TM == 32 && H <= 128) {
            if (a.gpt > 0) {                                   // padded tiles (slot classes that do not divide 32): the row-compacted pooling form
                allow_big_lds(k_ioc<H, 16, 32, 32, true, true, 1, true>);
                hipLaunchKernelGGL((k_ioc<H, 16, 32, 32, true, true, 1, true>), grid, block, ioc_lds_bytes(a, TM), s, a);
                return;
            }
        }
        if constexpr (TM == 32 && H <= 128) {                  // 32-row tiles: the row-compacted pooling (k_ioc CP) also while training
            if (a.variant != 9) {                              // (DESIRE_IOC_TRAIN_DENSE: dense pooling, A/B)
                allow_big_lds(k_ioc<H, 16, 32, 32, true, true>);
                hipLaunchKernelGGL((k_ioc<H, 16, 32, 32, true, true>), grid, block, ioc_lds_bytes(a, TM), s, a);
                return;
            }
        }
        if constexpr (H <= 128 || TM == 32) {
            allow_big_lds(k_ioc<H, 16, 32, TM, true>);
            hipLaunchKernelGGL((k_ioc<H, 16, 32, TM, true>), grid, block, ioc_lds_bytes(a, TM), s, a);
        }
        return;
    }
    if constexpr (TM == 32 && H <= 128) {
        if (a.variant == 8) {                                  // opt-in (DESIRE_IOC_COMPACT): row-compacted pooling (see k_ioc)
            allow_big_lds(k_ioc<H, 16, 32, 32, false, true>);
            hipLaunchKernelGGL((k_ioc<H, 16, 32, 32, false, true>), grid, block, ioc_lds_bytes(a, TM), s, a);
            return;
        }
    }
    if constexpr (TM == 32 && H <= 128) {
        if (a.nspl > 1) {                                      // few tiles: the social bins of a tile split over nspl workgroups (see k_ioc)
            const dim3 gs(grid.x * a.nspl);
            switch (a.nspl) {
                case 3: allow_big_lds(k_ioc<H, 16, 32, 32, false, false, 3>); hipLaunchKernelGGL((k_ioc<H, 16, 32, 32, false, false, 3>), gs, block, ioc_lds_bytes(a, TM), s, a); return;
                case 4: allow_big_lds(k_ioc<H, 16, 32, 32, false, false, 4>); hipLaunchKernelGGL((k_ioc<H, 16, 32, 32, false, false, 4>), gs, block, ioc_lds_bytes(a, TM), s, a); return;
                default: allow_big_lds(k_ioc<H, 16, 32, 32, false, false, 2>); hipLaunchKernelGGL((k_ioc<H, 16, 32, 32, false, false, 2>), gs, block, ioc_lds_bytes(a, TM), s, a); return;
            }
        }
    }
    if constexpr (TM == 32 && H <= 128) {
        if (a.gpt > 0) {                                       // padded tiles (slot classes that do not divide 32)
            allow_big_lds(k_ioc<H, 16, 32, 32, false, false, 1, true>);
            hipLaunchKernelGGL((k_ioc<H, 16, 32, 32, false, false, 1, true>), grid, block, ioc_lds_bytes(a, TM), s, a);
            return;
        }
    }
    allow_big_lds(k_ioc<H, 16, 32, TM, false>);
    hipLaunchKernelGGL((k_ioc<H, 16, 32, TM, false>), grid, block, ioc_lds_bytes(a, TM), s, a);
}
// Workgroups of the bin-split k_ioc<H, ..., NSPL = n> this device keeps resident at once (occupancy x compute units); 0 when the shape has
// no such form or the query fails.  The members of a tile wait for each other inside the kernel, so a launch is only safe when ALL of
// its workgroups are co-resident: a partition with fewer CUs (CPX mode, CU masking) or a lower occupancy must fall back to the plain form.
int ioc_bin_split_capacity(const IocArgs& a, int n) {
    if (a.mno > 32 || a.H > 128 || n < 2 || n > 4) return 0;
    static std::map<std::tuple<int, int, int, size_t>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    const size_t lds = ioc_lds_bytes(a, 32);
    const auto key = std::make_tuple(dev, a.H, n, lds);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const void* kern = nullptr;
    if (a.H == 128) kern = n == 2 ? (const void*)k_ioc<128, 16, 32, 32, false, false, 2> : n == 3 ? (const void*)k_ioc<128, 16, 32, 32, false, false, 3> : (const void*)k_ioc<128, 16, 32, 32, false, false, 4>;
    else kern = n == 2 ? (const void*)k_ioc<64, 16, 32, 32, false, false, 2> : n == 3 ? (const void*)k_ioc<64, 16, 32, 32, false, false, 3> : (const void*)k_ioc<64, 16, 32, 32, false, false, 4>;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int per_cu = 0, cus = 0;
    int cap = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, (a.H / 32) * 64, lds) == hipSuccess && per_cu > 0 &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
        cap = per_cu * cus;
    cache[key] = cap;
    return cap;
}
void launch_ioc_cluster(const IocArgs& a, hipStream_t s);
void launch_ioc(const IocArgs& a, hipStream_t s) {
    // groups larger than one workgroup tile (mno > 64, or mno = 64 at H = 256) or variant=4: cluster form
    if (ioc_uses_cluster(a.mno, a.H, a.G * a.G, a.variant)) { launch_ioc_cluster(a, s); return; }
    // 32-row tiles (two workgroups per CU at H <= 128) whenever whole (scene,k) groups fit; variant=2 forces 64 rows (A/B)
    const bool small = (a.mno <= 32) && a.variant != 2;
    if (a.H == 256) launch_ioc_t<256, 32>(a, s);                  // mno = 64 at H = 256 exceeds the 160 KB LDS tile
    else if (a.H == 128) { if (small) launch_ioc_t<128, 32>(a, s); else launch_ioc_t<128, 64>(a, s); }
    else { if (small) launch_ioc_t<64, 32>(a, s); else launch_ioc_t<64, 64>(a, s); }
}

// ------------------------------------------------------------------------------------------------
// IOC, cluster form: a (scene,k) group of mno = 32*tpg agents spans tpg 32-row tiles = tpg workgroups
// that exchange their hidden-state tile once per step through global memory (the per-XCD L2s are not
// coherent, so every hand-off is: plain stores -> s_waitcnt vmcnt(0) -> __syncthreads -> one lane
// agent-scope release -> relaxed agent atomic add on the group's arrival counter; the consumers poll
// that ONE word relaxed, then one agent-scope acquire, __syncthreads, plain loads).  The grid is
// persistent (<= one workgroup per CU, a multiple of tpg) so all members of a group are co-resident by
// construction; every spin is bounded (err word) and the counters are zeroed by a memset node per launch.
// Exchange buffer Hex[2][R][H]: h_t goes to parity t&1; a tile can only reach the write of parity p again
// after every group member has published the step in between, i.e. finished reading parity p.
// Serves mno in {64, 96, 128} (and H = 256 with mno = 64, which does not fit one workgroup's LDS).
// ------------------------------------------------------------------------------------------------
template <int H, int EV, int C, bool TRAIN = false>      // TRAIN: keeps x_t, r, u, c, h per step for the cluster-form BPTT (k_ioc_bwd_cl)
__global__ __launch_bounds__((H / 32) * 64, (H / 32) <= 4 ? 1 : 2) void k_ioc_cl(IocArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 32, MAXM = 128;
    constexpr int NT = H >> 5, E = EV + C + H, KX = E + H, LDX = KX + 4, LDB = H + 4;
    constexpr int NTHR = NT * 64, TPR = NTHR / TM;
    constexpr int G8 = KX >> 3, GH = H >> 3, GX = E >> 3;
    constexpr int NCH = H / (4 * TPR);
    const int B = a.G * a.G;
    const int tpg = a.mno / 32;                         // tiles (workgroups) per group
    float* XH = smem;                                   // [TM][LDX]
    float* AB = XH + TM * LDX;                          // [2][TM][LDB]
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(AB + 2 * TM * LDB);   // [TM][B][2]
    float* pg = reinterpret_cast<float*>(masks + TM * B * 2);  // [MAXM][2] positions of the whole group
    float* pp = pg + MAXM * 2;                          // [TM][2] previous position of my rows
    float* wv = pp + TM * 2;                            // [2][E_v] + [E_v]
    float* red = wv + 3 * EV;                           // [NT][TM]
    unsigned char* vld = reinterpret_cast<unsigned char*>(red + NT * TM);  // [MAXM]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + MAXM);               // [2] bins that hold a neighbour anywhere in the tile

    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int cb = w;
    const int col = cb * 32 + (lane & 31);
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int tile_pos = blockIdx.x % tpg;              // my tile inside its group
    IOC_DYN(a)                                          // (a slot class counted on the device: kernels.h DynCount; the persistent grid is the worst case's)
    const int n_tiles = a.R / TM;

    for (int i = tid; i < 3 * EV; i += NTHR) wv[i] = (i < 2 * EV) ? a.w_vel[i] : a.b_vel[i - 2 * EV];
    const float bgr = a.b_g[col], bgu = a.b_g[H + col], bcc = a.b_c[col], bso = a.b_soc[col], wsc = a.w_score[col];
    const float* x_lane = XH + (lane & 31) * LDX + 4 * (lane >> 5);
    float* my_x = XH + (4 * (lane >> 5)) * LDX + col;
    const float* rh_lane = AB + (lane & 31) * LDB + 4 * (lane >> 5);
    float* my_rh = AB + (4 * (lane >> 5)) * LDB + col;
    int epoch = 0;                                      // publishes this workgroup has made so far in this launch

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        const int grow0 = row0 - tile_pos * TM;         // first row of the group
        const int group = grow0 / a.mno;
        int* cnt = a.grp_cnt + group;
        const int scene = grow0 / (a.K * a.mno);
        const float* grid = a.grids + (size_t)a.grid_of_scene[scene] * a.Gh * a.Gw * C;
        const int my_slot = tile_pos * TM + r8;         // group-local index of my VALU row
        int base = 0;                                   // counter value when this group started = 0 (fresh group)
        __syncthreads();
        for (int i = tid; i < a.mno; i += NTHR) vld[i] = a.valid[agent_of_row(grow0 + i, a.K, a.mno)];

        for (int it = 0; it < a.iters; ++it) {
            if (it > 0) group_wait(cnt, tpg * (base + it * (a.T + 1)), a.err);     // everybody's Y += dY has landed
            for (int i = tid; i < TM * (H >> 2); i += NTHR) {
                const int r = i / (H >> 2), c4 = i - r * (H >> 2);
                const int ag = agent_of_row(row0 + r, a.K, a.mno);
                *reinterpret_cast<float4*>(XH + r * LDX + E + c4 * 4) =
                    *reinterpret_cast<const float4*>(a.Hx + (size_t)ag * a.ldhx + c4 * 4);
            }
            if (tid < TM) {
                const int ag = agent_of_row(row0 + tid, a.K, a.mno);
                pp[tid * 2] = a.p_last[(size_t)ag * 2];
                pp[tid * 2 + 1] = a.p_last[(size_t)ag * 2 + 1];
            }
            __syncthreads();
            f32x16 h, sp = zero16();
#pragma unroll
            for (int i = 0; i < 16; ++i) h[i] = my_x[((i & 3) + 8 * (i >> 2)) * LDX + E];

            for (int t = 0; t < a.T; ++t) {
                // neighbours' h_{t-1}: published by their tiles at the end of step t-1
                if (t > 0) group_wait_wt(cnt, tpg * (base + it * (a.T + 1) + t), a.err);     // fence-free hand-off (cluster.h: *_wt)
                for (int i = tid; i < a.mno; i += NTHR) {
                    const float2 y = *reinterpret_cast<const float2*>(a.Y + ((size_t)(grow0 + i) * a.T + t) * 2);
                    pg[i * 2] = y.x; pg[i * 2 + 1] = y.y;
                }
                for (int i = tid; i < TM * B * 2; i += NTHR) masks[i] = 0ull;
                if (tid < 2) occ[tid] = 0;
                __syncthreads();
                {
                    const float px = pg[my_slot * 2], py = pg[my_slot * 2 + 1];
                    const float vx = px - pp[r8 * 2], vy = py - pp[r8 * 2 + 1];
                    constexpr int per = EV / TPR;
#pragma unroll
                    for (int j = q8 * per; j < (q8 + 1) * per; ++j)
                        XH[r8 * LDX + j] = fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f);
                    int cy, cx;
                    scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
                    const float* gsrc = grid + ((size_t)cy * a.Gw + cx) * C;
                    constexpr int cper = C / TPR;
                    if (cper >= 4) {
#pragma unroll
                        for (int j = q8 * cper; j < (q8 + 1) * cper; j += 4)
                            *reinterpret_cast<float4*>(XH + r8 * LDX + EV + j) = *reinterpret_cast<const float4*>(gsrc + j);
                    } else {
                        *reinterpret_cast<float2*>(XH + r8 * LDX + EV + q8 * 2) = *reinterpret_cast<const float2*>(gsrc + q8 * 2);
                    }
                    for (int j = q8; j < a.mno; j += TPR) {
                        if (j == my_slot || !vld[j]) continue;
                        const int b = neighbor_bin_dev(px, py, pg[j * 2], pg[j * 2 + 1], a.nb_w, a.nb_h, a.G, a.bin_tab);
                        if (b >= 0) { atomicOr(&masks[(r8 * B + b) * 2 + (j >> 6)], 1ull << (j & 63)); atomicOr(&occ[b >> 5], 1u << (b & 31)); }
                    }
                }
                __syncthreads();
                // pooled operand: local rows from LDS, other tiles' rows from the exchange buffer (Hx at t = 0)
                const float* hex = a.hex + (size_t)((t + 1) & 1) * a.R * H;          // parity (t-1)&1
                auto build = [&](int b, int buf) {
                    float* ab = AB + buf * TM * LDB + r8 * LDB;
                    float4 s[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int wd = 0; wd < 2; ++wd) {
                        unsigned long long m2 = masks[(r8 * B + b) * 2 + wd];
                        while (m2) {
                            const int j = wd * 64 + __ffsll((long long)m2) - 1;
                            m2 &= m2 - 1;
                            if ((j >> 5) != tile_pos && t > 0) {                   // another member's row: written through, read past L1
                                const float* src = hex + (size_t)(grow0 + j) * H;
#pragma unroll
                                for (int c = 0; c < NCH; ++c) {
                                    const uint2 lo = ld_agent_u64(src + q8 * 4 + c * 4 * TPR), hi2 = ld_agent_u64(src + q8 * 4 + c * 4 * TPR + 2);
                                    s[c].x += __uint_as_float(lo.x); s[c].y += __uint_as_float(lo.y);
                                    s[c].z += __uint_as_float(hi2.x); s[c].w += __uint_as_float(hi2.y);
                                }
                                continue;
                            }
                            const float* src = ((j >> 5) == tile_pos) ? XH + (j & 31) * LDX + E
                                                                      : a.Hx + (size_t)agent_of_row(grow0 + j, a.K, a.mno) * a.ldhx;
#pragma unroll
                            for (int c = 0; c < NCH; ++c) {
                                const float4 v = *reinterpret_cast<const float4*>(src + q8 * 4 + c * 4 * TPR);
                                s[c].x += v.x; s[c].y += v.y; s[c].z += v.z; s[c].w += v.w;
                            }
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(ab + q8 * 4 + c * 4 * TPR) = s[c];
                };
                f32x16 soc = zero16();          // biases join after the contraction (a splat start value would pin 16 registers)
                unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
                om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
                int buf = 0;
                if (om) build(ffs_(om) - 1, 0);
                __syncthreads();
                while (om) {                                              // occupied bins only (see k_ioc)
                    const int b = ffs_(om) - 1;
                    om &= om - 1;
                    if (om) build(ffs_(om) - 1, buf ^ 1);
                    mma1(soc, AB + buf * TM * LDB + (lane & 31) * LDB + 4 * (lane >> 5),
                         a.Wsoc + ((size_t)(b * NT + cb) * GH) * 64 + lane, GH);
                    __syncthreads();
                    buf ^= 1;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) my_x[((i & 3) + 8 * (i >> 2)) * LDX + EV + C] = fmaxf(soc[i] + bso, 0.f);
                __syncthreads();
                if (TRAIN) {
                    for (int i = tid; i < TM * (E >> 2); i += NTHR) {
                        const int r = i / (E >> 2), c4 = i - r * (E >> 2);
                        *reinterpret_cast<float4*>(a.sv_x + ((size_t)(row0 + r) * a.T + t) * E + c4 * 4) =
                            *reinterpret_cast<const float4*>(XH + r * LDX + c4 * 4);
                    }
                }
                f32x16 rh = zero16(), u = zero16();
                mma1(rh, x_lane, a.Wg + ((size_t)cb * G8) * 64 + lane, G8);
                mma1(u, x_lane, a.Wg + ((size_t)(cb + NT) * G8) * 64 + lane, G8);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float r = sigmoidf_(rh[i] + bgr);
                    if (TRAIN) a.sv_r[((size_t)(row0 + acc_row(i)) * a.T + t) * H + col] = r;
                    rh[i] = r * h[i];
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) my_rh[((i & 3) + 8 * (i >> 2)) * LDB] = rh[i];
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    u[i] = sigmoidf_(u[i] + bgu);
                    if (TRAIN) a.sv_u[((size_t)(row0 + acc_row(i)) * a.T + t) * H + col] = u[i];
                }
                __syncthreads();
                {
                    f32x16 ac = zero16();
                    mma1(ac, x_lane, a.Wc + ((size_t)cb * G8) * 64 + lane, GX);
                    mma1(ac, rh_lane, a.Wc + ((size_t)cb * G8 + GX) * 64 + lane, GH);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float c = tanhf_(ac[i] + bcc);
                        h[i] = gru_blend(u[i], h[i], c);
                        sp[i] = fmaf(h[i], wsc, sp[i]);
                        if (TRAIN) {
                            const size_t ix = ((size_t)(row0 + acc_row(i)) * a.T + t) * H + col;
                            a.sv_c[ix] = c; a.sv_h[ix] = h[i];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        my_x[((i & 3) + 8 * (i >> 2)) * LDX + E] = h[i];
                    }
                }
                if (tid < TM) { pp[tid * 2] = pg[(tile_pos * TM + tid) * 2]; pp[tid * 2 + 1] = pg[(tile_pos * TM + tid) * 2 + 1]; }
                __syncthreads();                         // h_t tile complete in LDS
                {   // publish it: row-major copy with 8-byte write-through stores (parity t&1), then the fence-free arrival
                    float* hout = a.hex + (size_t)(t & 1) * a.R * H + (size_t)row0 * H;
                    for (int i = tid; i < TM * (H >> 1); i += NTHR) {
                        const int r = i / (H >> 1), c2 = i - r * (H >> 1);
                        const float2 v = *reinterpret_cast<const float2*>(XH + r * LDX + E + 2 * c2);
                        st_agent_u64(hout + (size_t)r * H + 2 * c2, make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)));
                    }
                }
                group_publish_wt(cnt);                   // includes the end-of-step __syncthreads
                ++epoch;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = sp[i];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
                if ((lane & 31) == 0) red[cb * TM + acc_row(i)] = v;
            }
            __syncthreads();
            if (tid < TM && it == a.iters - 1) {
                float sc = 0.f;
#pragma unroll
                for (int c = 0; c < NT; ++c) sc += red[c * TM + tid];
                a.score[row0 + tid] = sc + (float)a.T * a.b_score[0];
            }
            // every member has published step T-1, i.e. is past its own load of the group's positions Y[.][T-1]: only now may my rows of
            // Y be updated in place (a peer that only waited for the T-2 publishes could otherwise read next-pass positions)
            group_wait_wt(cnt, tpg * (base + it * (a.T + 1) + a.T), a.err);
            for (int nt = cb; nt < a.NTreg; nt += NT) {
                f32x16 acc = zero16();
                mma1(acc, x_lane + E, a.Wreg + ((size_t)nt * GH) * 64 + lane, GH);
                const int cc = nt * 32 + (lane & 31);
                if (cc < 2 * a.T) {
                    const float bb = a.b_reg[cc];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float* y = a.Y + (size_t)(row0 + acc_row(i)) * 2 * a.T + cc;
                        *y = *y + (acc[i] + bb);
                    }
                }
            }
            group_publish(cnt);                          // pass end: my rows of Y are final for this pass
            ++epoch;
        }
        (void)epoch;
    }
}
static size_t ioc_cl_lds_bytes(const IocArgs& a) {
    const int EV = 16, H = a.H, NT = H / 32, E = EV + 32 + H, LDX = E + H + 4, LDB = H + 4, B = a.G * a.G, TM = 32;
    size_t f = (size_t)TM * LDX + 2 * TM * LDB + (size_t)TM * B * 4 + 128 * 2 + TM * 2 + 3 * EV + NT * TM;
    return f * sizeof(float) + 128 + 64;
}
template <int H>
static void launch_ioc_cl_t(const IocArgs& a, hipStream_t s) {
    allow_big_lds(k_ioc_cl<H, 16, 32>);
    const int tpg = a.mno / 32, n_tiles = a.R / 32;
    int grid = n_tiles < 256 ? n_tiles : 256;            // one workgroup per CU: all of them resident
    grid -= grid % tpg;
    if (a.sv_h) hipLaunchKernelGGL((k_ioc_cl<H, 16, 32, true>), dim3(grid), dim3((H / 32) * 64), ioc_cl_lds_bytes(a), s, a);
    else hipLaunchKernelGGL((k_ioc_cl<H, 16, 32>), dim3(grid), dim3((H / 32) * 64), ioc_cl_lds_bytes(a), s, a);
}
void launch_ioc_cluster(const IocArgs& a, hipStream_t s) {
    if (a.H == 256) launch_ioc_cl_t<256>(a, s);
    else if (a.H == 128) launch_ioc_cl_t<128>(a, s);
    else launch_ioc_cl_t<64>(a, s);
}

// ------------------------------------------------------------------------------------------------
// IOC, agent-sharded form (SURVEY.md 8(e) E1 / BASELINE north_star): the agents of every scene are block-sharded over G
// ranks (rank g owns slots [g*m_loc, (g+1)*m_loc) of every scene).  Everything before the IOC is per-agent, so each rank
// runs it on its own slots; the IOC couples agents through social pooling once per step, so ONE step is one launch of
// k_ioc_step and the host all-gathers the hidden states in between (RCCL over xGMI; dist.ShardedIoc).  Positions of all
// agents are gathered once (they are the decoder's output, fixed during a pass).
//   Yall  [G][n_groups][m_loc][T][2]   all ranks' decoded positions      (gathered before the pass)
//   Hall  [G][n_groups][m_loc][H]      all ranks' h_{t-1}                (gathered before every step)
//   vall  [G][n_scenes][m_loc]         presence flags
// Local state between launches: st_h [R_loc][H], st_score [R_loc].  Tile = 32 local rows (any mix of groups); the
// pooled operand is built from Hall in global memory (L2), neighbours in ascending global slot order -- the same
// summation order as the single-GPU kernels, so the result is bit-identical to them.
// ------------------------------------------------------------------------------------------------
// NP = 0: fp32 operands on the fp32 matrix pipe.  NP = 2 / 3 (dims.bf16 = 2 / 3 at shapes the persistent split kernels do not serve:
// groups of 96 .. 256 agents, H = 256): the same step with every contraction as three / six bf16 MFMAs per fp32 product -- A fragments
// split on the fly out of the fp32 LDS tiles (split.h: mma6_groups), weights = the [hi | lo (| lo2)] packs in plain k order
// ("ioc/Wg16", "ioc/Wc16", "ioc/Wsoc16l"); accumulators, state, gate math and layouts are the fp32 kernel's.
#ifndef STEP_RING
#define STEP_RING 8               // k-groups of weight fragments in flight in the split forms (0: the one-ahead mma6_groups of round 4)
#endif
// (split forms: ONE workgroup per CU is all the LDS tile allows at H = 256 anyway -- asking for two capped the kernel at 128 registers, which is what
//  kept its fragment prefetch one group deep)
template <int H, int EV, int C, int NP = 0>
__global__ __launch_bounds__((H / 32) * 64, ((H / 32) <= 4 || NP != 0) ? 1 : 2) void k_ioc_step(IocStepArgs a) {
#ifdef STEP_TIMING
    long long tk[10]; int nk = 0; tk[nk++] = clock64();
#define STK() { tk[nk++] = clock64(); }
#else
#define STK()
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 32;
    const int MW = (a.m_loc * a.nranks + 63) >> 6;      // 64-bit mask words per (row, bin): 1 .. 4 (up to 256 agents per scene) -- sized by the scene, so
                                                        // that at H <= 128 and <= 64 agents the tile stays under 80 KB and two workgroups share a CU
    constexpr int NT = H >> 5, E = EV + C + H, KX = E + H, LDX = KX + 4, LDB = H + 4;
    constexpr int NTHR = NT * 64, TPR = NTHR / TM;
    constexpr int G8 = KX >> 3, GH = H >> 3, GX = E >> 3;
    constexpr int NCH = H / (4 * TPR);
    const int B = a.G * a.G;
    float* XH = smem;                                   // [TM][LDX]
    float* AB = XH + TM * LDX;                          // [2][TM][LDB]
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(AB + 2 * TM * LDB);   // [TM][B][MW]
    float* wv = reinterpret_cast<float*>(masks + TM * B * MW);    // [3][EV]
    float* red = wv + 3 * EV;                           // [NT][TM]
    unsigned* occ = reinterpret_cast<unsigned*>(red + NT * TM);   // [2] bins that hold a neighbour anywhere in the tile
    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int col = cb * 32 + (lane & 31);
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int row0 = blockIdx.x * TM;
    const int mall = a.m_loc * a.nranks;
    const int n_groups = a.R / a.m_loc;
    for (int i = tid; i < 3 * EV; i += NTHR) wv[i] = (i < 2 * EV) ? a.w_vel[i] : a.b_vel[i - 2 * EV];
    for (int i = tid; i < TM * B * MW; i += NTHR) masks[i] = 0ull;
    if (tid < 2) occ[tid] = 0;
    for (int i = tid; i < TM * (H >> 2); i += NTHR) {
        const int r = i / (H >> 2), c4 = i - r * (H >> 2);
        const float* sp = a.st_h + (size_t)min(row0 + r, a.R - 1) * H + c4 * 4;
        *reinterpret_cast<float4*>(XH + r * LDX + E + c4 * 4) = a.peer ? ld_sys_f4(sp) : *reinterpret_cast<const float4*>(sp);
    }
    const float bgr = a.b_g[col], bgu = a.b_g[H + col], bcc = a.b_c[col], bso = a.b_soc[col], wsc = a.w_score[col];
    const float* x_lane = XH + (lane & 31) * LDX + 4 * (lane >> 5);
    float* my_x = XH + (4 * (lane >> 5)) * LDX + col;
    const float* rh_lane = AB + (lane & 31) * LDB + 4 * (lane >> 5);
    float* my_rh = AB + (4 * (lane >> 5)) * LDB + col;
    // my VALU row: local row -> (group, local slot) -> global slot
    const int my_row = min(row0 + r8, a.R - 1);
    const int grp = my_row / a.m_loc, sl = my_row - grp * a.m_loc;
    const int scene = grp / a.K;
    const int my_gslot = a.rank * a.m_loc + sl;
    auto pos_of = [&](int j, int t) {                    // position of global slot j of my group at step t (t = -1: last observed)
        const int rk = j / a.m_loc, s = j - rk * a.m_loc;
        // peer form: everything that lives in an exchange region is read with system-scope loads (cluster.h), never through a cache
        if (t < 0) {
            if (a.peer) return ld_sys_f2(a.plp[rk] + ((size_t)scene * a.m_loc + s) * 2);
            return *reinterpret_cast<const float2*>(a.plast_all + ((size_t)(rk * a.n_scenes + scene) * a.m_loc + s) * 2);
        }
        if (a.peer) return ld_sys_f2(a.Yp[rk] + (((size_t)grp * a.m_loc + s) * a.T + t) * 2);
        return *reinterpret_cast<const float2*>(a.Yall + ((((size_t)rk * n_groups + grp) * a.m_loc + s) * a.T + t) * 2);
    };
    __syncthreads();
    STK()          // 1: prologue (weights, masks clear, h tile)
    f32x16 h;
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = my_x[((i & 3) + 8 * (i >> 2)) * LDX + E];
    {
        const float2 pcur = pos_of(my_gslot, a.t), pprev = pos_of(my_gslot, a.t - 1);
        const float px = pcur.x, py = pcur.y;
        const float vx = px - pprev.x, vy = py - pprev.y;
        constexpr int per = EV / TPR;
#pragma unroll
        for (int j = q8 * per; j < (q8 + 1) * per; ++j)
            XH[r8 * LDX + j] = fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f);
        int cy, cx;
        scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
        const float* gsrc = a.grids + (size_t)a.grid_of_scene[scene] * a.Gh * a.Gw * C + ((size_t)cy * a.Gw + cx) * C;
        constexpr int cper = C / TPR;
        if (cper >= 4) {
#pragma unroll
            for (int j = q8 * cper; j < (q8 + 1) * cper; j += 4)
                *reinterpret_cast<float4*>(XH + r8 * LDX + EV + j) = *reinterpret_cast<const float4*>(gsrc + j);
        } else {
            *reinterpret_cast<float2*>(XH + r8 * LDX + EV + q8 * 2) = *reinterpret_cast<const float2*>(gsrc + q8 * 2);
        }
        for (int j = q8; j < mall; j += TPR) {
            const int rk = j / a.m_loc, s = j - rk * a.m_loc;
            bool there;
            if (a.peer) { const size_t ix = (size_t)scene * a.m_loc + s; there = (ld_sys_u32(a.vp[rk] + (ix & ~(size_t)3)) >> (8 * (ix & 3))) & 0xffu; }
            else there = a.valid_all[(size_t)(rk * a.n_scenes + scene) * a.m_loc + s];
            if (j == my_gslot || !there) continue;
            const float2 pj = pos_of(j, a.t);
            const int b = neighbor_bin_dev(px, py, pj.x, pj.y, a.nb_w, a.nb_h, a.G, a.bin_tab);
            if (b >= 0) { atomicOr(&masks[(r8 * B + b) * MW + (j >> 6)], 1ull << (j & 63)); atomicOr(&occ[b >> 5], 1u << (b & 31)); }
        }
    }
    __syncthreads();
    STK()          // 2: P1 (positions, e_v, e_s, neighbour search)
    // pooled operand of bin b for my row: sum of the neighbours' h_{t-1} in ascending global-slot order.  The rows come from L2 / a peer's HBM, one
    // dependent round trip per neighbour if taken one at a time (63 of them at 64 agents per scene: that chain, not the MFMAs, was the step's length);
    // NBAT neighbours' loads are issued together and added in slot order afterwards -- the same sums, bit for bit.
    constexpr int NBAT = (NP != 0 || NT <= 4) ? 4 : 2;
    auto build = [&](int b, int buf) {
        float* ab = AB + buf * TM * LDB + r8 * LDB;
        float4 s[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int wd = 0; wd < MW; ++wd) {
            unsigned long long m2 = masks[(r8 * B + b) * MW + wd];
            while (m2) {
                const float* src[NBAT];
#pragma unroll
                for (int q = 0; q < NBAT; ++q) {
                    src[q] = nullptr;
                    if (m2) {
                        const int j = wd * 64 + __ffsll((long long)m2) - 1;
                        m2 &= m2 - 1;
                        const int rk = j / a.m_loc, sj = j - rk * a.m_loc;
                        const float* hb = a.peer ? a.Hp[rk] : a.Hall + (size_t)rk * n_groups * a.m_loc * H;
                        src[q] = hb + ((size_t)grp * a.m_loc + sj) * H + q8 * 4;
                    }
                }
                float4 v[NBAT][NCH];
#pragma unroll
                for (int q = 0; q < NBAT; ++q)
                    if (src[q]) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c)
                            v[q][c] = a.peer ? ld_sys_f4(src[q] + c * 4 * TPR) : *reinterpret_cast<const float4*>(src[q] + c * 4 * TPR);
                    }
#pragma unroll
                for (int q = 0; q < NBAT; ++q)
                    if (src[q]) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) { s[c].x += v[q][c].x; s[c].y += v[q][c].y; s[c].z += v[q][c].z; s[c].w += v[q][c].w; }
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(ab + q8 * 4 + c * 4 * TPR) = s[c];
    };
    f32x16 soc = zero16();          // biases join after the contraction (a splat start value would pin 16 registers)
    unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
    om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
    int buf = 0;
    if (om) build(ffs_(om) - 1, 0);
    __syncthreads();
    STK()          // 3: first build
    while (om) {                                                          // occupied bins only (see k_ioc)
        const int b = ffs_(om) - 1;
        om &= om - 1;
        if (om) build(ffs_(om) - 1, buf ^ 1);
        if constexpr (NP == 0) mma1(soc, AB + buf * TM * LDB + (lane & 31) * LDB + 4 * (lane >> 5), a.Wsoc + ((size_t)(b * NT + cb) * GH) * 64 + lane, GH);
        else {
            f32x16 t1[1] = {soc};
            // (three pieces: one k-group ahead; a deeper fragment ring measured slower here, 13.55 vs 12.64 ms -- 96 more registers of fragments.  Two pieces run
            //  k_ioc_step_x2 below)
            const uint4* bl[1] = {reinterpret_cast<const uint4*>(a.Wsoc) + ((size_t)(b * NT + cb) * (H / 16)) * 64 + lane};
            mma6_groups<1, NP>(t1, AB + buf * TM * LDB + (lane & 31) * LDB + 8 * (lane >> 5), bl, a.plo_soc, H / 16);
            soc = t1[0];
        }
        __syncthreads();
        buf ^= 1;
    }
    STK()          // 4: bins loop
#pragma unroll
    for (int i = 0; i < 16; ++i) my_x[((i & 3) + 8 * (i >> 2)) * LDX + EV + C] = fmaxf(soc[i] + bso, 0.f);
    __syncthreads();
    f32x16 rh = zero16(), u = zero16();
    if constexpr (NP == 0) {
        mma1(rh, x_lane, a.Wg + ((size_t)cb * G8) * 64 + lane, G8);
        mma1(u, x_lane, a.Wg + ((size_t)(cb + NT) * G8) * 64 + lane, G8);
    } else {
        f32x16 t2[2] = {rh, u};
        const uint4* wg = reinterpret_cast<const uint4*>(a.Wg);
        const uint4* bl[2] = {wg + ((size_t)cb * (KX / 16)) * 64 + lane, wg + ((size_t)(cb + NT) * (KX / 16)) * 64 + lane};
        mma6_groups<2, NP>(t2, XH + (lane & 31) * LDX + 8 * (lane >> 5), bl, a.plo_g, KX / 16);
        rh = t2[0]; u = t2[1];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) rh[i] = sigmoidf_(rh[i] + bgr) * h[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) my_rh[((i & 3) + 8 * (i >> 2)) * LDB] = rh[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) u[i] = sigmoidf_(u[i] + bgu);
    __syncthreads();
    STK()          // 5: gates
    f32x16 ac = zero16();
    if constexpr (NP == 0) {
        mma1(ac, x_lane, a.Wc + ((size_t)cb * G8) * 64 + lane, GX);
        mma1(ac, rh_lane, a.Wc + ((size_t)cb * G8 + GX) * 64 + lane, GH);
    } else {
        f32x16 t1[1] = {ac};
        const uint4* wc = reinterpret_cast<const uint4*>(a.Wc);
        const uint4* bx[1] = {wc + ((size_t)cb * (KX / 16)) * 64 + lane};
        mma6_groups<1, NP>(t1, XH + (lane & 31) * LDX + 8 * (lane >> 5), bx, a.plo_c, E / 16);
        const uint4* bh[1] = {wc + ((size_t)cb * (KX / 16) + E / 16) * 64 + lane};
        mma6_groups<1, NP>(t1, AB + (lane & 31) * LDB + 8 * (lane >> 5), bh, a.plo_c, H / 16);
        ac = t1[0];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        h[i] = gru_blend(u[i], h[i], tanhf_(ac[i] + bcc));
        const int row = row0 + acc_row(i);
        if (row < a.R) {
            if (a.peer) st_sys_f32(a.st_h_out + (size_t)row * H + col, h[i]); else a.st_h_out[(size_t)row * H + col] = h[i];
            if (a.st_h_copy) a.st_h_copy[(size_t)row * H + col] = h[i];      // peer form, last step: h_T for the regression head, in ordinary memory
        }
        float v = h[i] * wsc;
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
        if ((lane & 31) == 0) red[cb * TM + acc_row(i)] = v;
    }
    __syncthreads();
    STK()          // 6: candidate + epilogue
#ifdef STEP_TIMING
    if (blockIdx.x == 7 && tid == 0 && a.t == 20)
        printf("k_ioc_step<%d,NP=%d> t=20 block 7: prologue %lld  P1 %lld  build0 %lld  bins %lld  gates %lld  cand+epi %lld  total %lld cycles\n", H, NP,
               tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5], tk[6] - tk[0]);
#endif
    if (tid < TM && row0 + tid < a.R) {
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) sc += red[c * TM + tid];
        a.st_score[row0 + tid] = (a.t == 0 ? 0.f : a.st_score[row0 + tid]) + sc;
    }
}
// ------------------------------------------------------------------------------------------------
// k_ioc_step with two-piece operands (dims.bf16 = 2), round 5: the same step with the operand tiles in LDS as bf16 PIECE IMAGES (hi | lo), written
// once by whoever produces the values, instead of fp32 tiles that each of the NT waves splits again for every k-group (44 VALU instructions per
// fragment next to the three MFMAs it feeds), and with STEP_RING k-groups of weight fragments in flight.  Phase counters of the fp32-tile form at
// configs[3]'s shape (-DSTEP_TIMING): 245 k cycles per step, 168 k of them in the bin loop = 10.5 k per bin for 3 k matrix cycles.
// Same pieces (splitp) and the same products per accumulator as k_ioc_step<H, EV, C, 2>; measured against it (profiles/ab/ab_x2b.sh, three shapes): positions
// within 1.5e-6, scores within 5e-6 -- the fp32 rounding class, not bit-identical.  configs[3]'s per-GPU shape: IOC 8.6 -> 8.0 ms, step 12.75 -> 12.16 ms.
// ------------------------------------------------------------------------------------------------
// (Round 5 also built this kernel with extra PRODUCER waves gathering bin b + 1 while the others contract bin b: bit-identical, 8.0 -> 10.5 ms at configs[3]'s
//  shape -- twelve waves at H = 256 leave 168 registers each -- and removed in round 6; docs/DESIGN_DETAIL.md section 13 item 7 has the numbers.)
template <int H, int EV, int C>
__global__ __launch_bounds__((H / 32) * 64, 1) void k_ioc_step_x2(IocStepArgs a) {
#ifdef STEP_TIMING
    long long tk[10]; int nk = 0; tk[nk++] = clock64();
    long long tbl = 0, tmm = 0, tbar = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int TM = 32, NP = 2, RD = STEP_RING > 0 ? STEP_RING : 1;
    const int MW = (a.m_loc * a.nranks + 63) >> 6;
    constexpr int NT = H >> 5, E = EV + C + H, KX = E + H, LDXB = KX + 8, LDBB = H + 8;      // bf16 elements; (ld / 2) = 4 mod 8 dwords: conflict-free b128 reads
    constexpr int XLO = TM * LDXB, BLO = TM * LDBB;                                            // elements between the two piece images of a tile
    constexpr int NTHR = NT * 64, TPR = NTHR / TM;
    constexpr int TPB = TPR;                                               // threads per row in the pooled-operand build
    constexpr int NCH = H / (4 * TPB);
    const int B = a.G * a.G;
    u16* Xb = reinterpret_cast<u16*>(smem_raw);                         // [NP][TM][LDXB]   e_v | e_s | e_r | h
    u16* ABb = Xb + NP * XLO;                                           // [2][NP][TM][LDBB] pooled operand (double-buffered), then r * h in buffer 0
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(ABb + 2 * NP * BLO);   // [TM][B][MW]
    float* wv = reinterpret_cast<float*>(masks + TM * B * MW);          // [3][EV]
    float* red = wv + 3 * EV;                                           // [NT][TM]
    unsigned* occ = reinterpret_cast<unsigned*>(red + NT * TM);
    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int col = cb * 32 + (lane & 31);
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int rb = r8, qb = q8;                                            // row / quarter of this thread in the pooled-operand build
    const int row0 = blockIdx.x * TM;
    const int mall = a.m_loc * a.nranks;
    const int n_groups = a.R / a.m_loc;
    auto st1 = [](u16* img, int plo, float v) {                         // one value -> its two pieces
        const u16 hi = bf16_of(v);
        img[0] = hi; img[plo] = bf16_of(v - __uint_as_float((unsigned)hi << 16));
    };
    auto st2 = [](u16* img, int plo, float v0, float v1) {              // two adjacent values (4-byte aligned)
        unsigned pp[2];
        splitp<2>(v0, v1, pp);
        *reinterpret_cast<unsigned*>(img) = pp[0]; *reinterpret_cast<unsigned*>(img + plo) = pp[1];
    };
    for (int i = tid; i < 3 * EV; i += NTHR) wv[i] = (i < 2 * EV) ? a.w_vel[i] : a.b_vel[i - 2 * EV];
    for (int i = tid; i < TM * B * MW; i += NTHR) masks[i] = 0ull;
    if (tid < 2) occ[tid] = 0;
    for (int i = tid; i < TM * (H >> 2); i += NTHR) {
        const int r = i / (H >> 2), c4 = i - r * (H >> 2);
        const float* sp = a.st_h + (size_t)min(row0 + r, a.R - 1) * H + c4 * 4;
        const float4 v = a.peer ? ld_sys_f4(sp) : *reinterpret_cast<const float4*>(sp);
        st2(Xb + r * LDXB + E + c4 * 4, XLO, v.x, v.y); st2(Xb + r * LDXB + E + c4 * 4 + 2, XLO, v.z, v.w);
    }
    const float bgr = a.b_g[col], bgu = a.b_g[H + col], bcc = a.b_c[col], bso = a.b_soc[col], wsc = a.w_score[col];
    const u16* x_lane = Xb + (lane & 31) * LDXB + 8 * (lane >> 5);
    const int my_row = min(row0 + r8, a.R - 1);
    const int grp = my_row / a.m_loc, sl = my_row - grp * a.m_loc;
    const int scene = grp / a.K;
    const int my_gslot = a.rank * a.m_loc + sl;
    const int grp_b = min(row0 + rb, a.R - 1) / a.m_loc;
    auto pos_of = [&](int j, int t) {
        const int rk = j / a.m_loc, s = j - rk * a.m_loc;
        if (t < 0) {
            if (a.peer) return ld_sys_f2(a.plp[rk] + ((size_t)scene * a.m_loc + s) * 2);
            return *reinterpret_cast<const float2*>(a.plast_all + ((size_t)(rk * a.n_scenes + scene) * a.m_loc + s) * 2);
        }
        if (a.peer) return ld_sys_f2(a.Yp[rk] + (((size_t)grp * a.m_loc + s) * a.T + t) * 2);
        return *reinterpret_cast<const float2*>(a.Yall + ((((size_t)rk * n_groups + grp) * a.m_loc + s) * a.T + t) * 2);
    };
    // h_{t-1} of my accumulator elements straight from the state (the LDS tile holds pieces only)
    f32x16 h = zero16();
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* sp = a.st_h + (size_t)min(row0 + acc_row(i), a.R - 1) * H + col;
            h[i] = a.peer ? __uint_as_float(ld_sys_u32(sp)) : *sp;
        }
    }
    {
        const float2 pcur = pos_of(my_gslot, a.t), pprev = pos_of(my_gslot, a.t - 1);
        const float px = pcur.x, py = pcur.y;
        const float vx = px - pprev.x, vy = py - pprev.y;
        constexpr int per = EV / TPR;
#pragma unroll
        for (int j = q8 * per; j < (q8 + 1) * per; ++j)
            st1(Xb + r8 * LDXB + j, XLO, fmaxf(fmaf(vy, wv[EV + j], vx * wv[j]) + wv[2 * EV + j], 0.f));
        int cy, cx;
        scene_cell_dev(px, py, a.Gh, a.Gw, cy, cx);
        const float* gsrc = a.grids + (size_t)a.grid_of_scene[scene] * a.Gh * a.Gw * C + ((size_t)cy * a.Gw + cx) * C;
        constexpr int cper = C / TPR;
        if (cper >= 4) {
#pragma unroll
            for (int j = q8 * cper; j < (q8 + 1) * cper; j += 4) {
                const float4 g4 = *reinterpret_cast<const float4*>(gsrc + j);
                st2(Xb + r8 * LDXB + EV + j, XLO, g4.x, g4.y); st2(Xb + r8 * LDXB + EV + j + 2, XLO, g4.z, g4.w);
            }
        } else {
            const float2 g2 = *reinterpret_cast<const float2*>(gsrc + q8 * 2);
            st2(Xb + r8 * LDXB + EV + q8 * 2, XLO, g2.x, g2.y);
        }
        for (int j = q8; j < mall; j += TPR) {
            const int rk = j / a.m_loc, s = j - rk * a.m_loc;
            bool there;
            if (a.peer) { const size_t ix = (size_t)scene * a.m_loc + s; there = (ld_sys_u32(a.vp[rk] + (ix & ~(size_t)3)) >> (8 * (ix & 3))) & 0xffu; }
            else there = a.valid_all[(size_t)(rk * a.n_scenes + scene) * a.m_loc + s];
            if (j == my_gslot || !there) continue;
            const float2 pj = pos_of(j, a.t);
            const int b = neighbor_bin_dev(px, py, pj.x, pj.y, a.nb_w, a.nb_h, a.G, a.bin_tab);
            if (b >= 0) { atomicOr(&masks[(r8 * B + b) * MW + (j >> 6)], 1ull << (j & 63)); atomicOr(&occ[b >> 5], 1u << (b & 31)); }
        }
    }
    __syncthreads();
#ifdef STEP_TIMING
    tk[nk++] = clock64();
#endif
    // (the gather was also tried in two halves around the bin's MFMAs -- rows requested before the contraction, added after it -- and LOST, 9.4 vs 8.0 ms:
    //  vmcnt retires loads in issue order, so the first wait for a weight fragment also waits for the gather issued before it)
    constexpr int NBAT = 4;
    auto build = [&](int b, int buf) {
        u16* ab = ABb + buf * NP * BLO + rb * LDBB;
        float4 s[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int wd = 0; wd < MW; ++wd) {
            unsigned long long m2 = masks[(rb * B + b) * MW + wd];
            while (m2) {
                const float* src[NBAT];
#pragma unroll
                for (int q = 0; q < NBAT; ++q) {
                    src[q] = nullptr;
                    if (m2) {
                        const int j = wd * 64 + __ffsll((long long)m2) - 1;
                        m2 &= m2 - 1;
                        const int rk = j / a.m_loc, sj = j - rk * a.m_loc;
                        const float* hb = a.peer ? a.Hp[rk] : a.Hall + (size_t)rk * n_groups * a.m_loc * H;
                        src[q] = hb + ((size_t)grp_b * a.m_loc + sj) * H + qb * 4;
                    }
                }
                float4 v[NBAT][NCH];
#pragma unroll
                for (int q = 0; q < NBAT; ++q)
                    if (src[q]) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c)
                            v[q][c] = a.peer ? ld_sys_f4(src[q] + c * 4 * TPB) : *reinterpret_cast<const float4*>(src[q] + c * 4 * TPB);
                    }
#pragma unroll
                for (int q = 0; q < NBAT; ++q)
                    if (src[q]) {
#pragma unroll
                        for (int c = 0; c < NCH; ++c) { s[c].x += v[q][c].x; s[c].y += v[q][c].y; s[c].z += v[q][c].z; s[c].w += v[q][c].w; }
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            u16* dst = ab + qb * 4 + c * 4 * TPB;
            st2(dst, BLO, s[c].x, s[c].y); st2(dst + 2, BLO, s[c].z, s[c].w);
        }
    };
    f32x16 soc = zero16();
    unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
    om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
    int buf = 0;
    if (om) build(ffs_(om) - 1, 0);
    __syncthreads();
#ifdef STEP_TIMING
    tk[nk++] = clock64();
#endif
    const uint4* Wsoc = reinterpret_cast<const uint4*>(a.Wsoc);
    while (om) {
        const int b = ffs_(om) - 1;
        om &= om - 1;
#ifdef STEP_TIMING
        const long long c0 = clock64();
#endif
        if (om) build(ffs_(om) - 1, buf ^ 1);
#ifdef STEP_TIMING
        const long long c1 = clock64();
#endif
        {
            f32x16 t1[1] = {soc};
            const unsigned t0[1] = {(unsigned)((b * NT + cb) * (H / 16)) * 64u};
            mmaxp_ring<1, NP, RD>(t1, ABb + buf * NP * BLO + (lane & 31) * LDBB + 8 * (lane >> 5), BLO, Wsoc, t0, a.plo_soc, H / 16);
            soc = t1[0];
        }
#ifdef STEP_TIMING
        const long long c2 = clock64();
#endif
        __syncthreads();
#ifdef STEP_TIMING
        tbl += c1 - c0; tmm += c2 - c1; tbar += clock64() - c2;
#endif
        buf ^= 1;
    }
#ifdef STEP_TIMING
    tk[nk++] = clock64();
#endif
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) st1(Xb + acc_row(i) * LDXB + EV + C + col, XLO, fmaxf(soc[i] + bso, 0.f));
    }
    __syncthreads();
    f32x16 rh = zero16(), u = zero16();
    {
        f32x16 t2[2] = {rh, u};
        const unsigned t0[2] = {(unsigned)(cb * (KX / 16)) * 64u, (unsigned)((cb + NT) * (KX / 16)) * 64u};
        mmaxp_ring<2, NP, (RD > 4 ? 4 : RD)>(t2, x_lane, XLO, reinterpret_cast<const uint4*>(a.Wg), t0, a.plo_g, KX / 16);
        rh = t2[0]; u = t2[1];
    }
    {
#pragma unroll
        for (int i = 0; i < 16; ++i) rh[i] = sigmoidf_(rh[i] + bgr) * h[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) st1(ABb + acc_row(i) * LDBB + col, BLO, rh[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) u[i] = sigmoidf_(u[i] + bgu);
    }
    __syncthreads();
#ifdef STEP_TIMING
    tk[nk++] = clock64();
#endif
    f32x16 ac = zero16();
    {
        f32x16 t1[1] = {ac};
        const uint4* wc = reinterpret_cast<const uint4*>(a.Wc);
        const unsigned tx[1] = {(unsigned)(cb * (KX / 16)) * 64u}, th[1] = {(unsigned)(cb * (KX / 16) + E / 16) * 64u};
        mmaxp_ring<1, NP, RD>(t1, x_lane, XLO, wc, tx, a.plo_c, E / 16);
        mmaxp_ring<1, NP, RD>(t1, ABb + (lane & 31) * LDBB + 8 * (lane >> 5), BLO, wc, th, a.plo_c, H / 16);
        ac = t1[0];
    }
    {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        h[i] = gru_blend(u[i], h[i], tanhf_(ac[i] + bcc));
        const int row = row0 + acc_row(i);
        if (row < a.R) {
            if (a.peer) st_sys_f32(a.st_h_out + (size_t)row * H + col, h[i]); else a.st_h_out[(size_t)row * H + col] = h[i];
            if (a.st_h_copy) a.st_h_copy[(size_t)row * H + col] = h[i];
        }
        float v = h[i] * wsc;
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
        if ((lane & 31) == 0) red[cb * TM + acc_row(i)] = v;
    }
    }
    __syncthreads();
#ifdef STEP_TIMING
    tk[nk++] = clock64();
    if (blockIdx.x == 7 && tid == 0 && a.t == 20)
        printf("k_ioc_step_x2<%d> t=20 block 7: prologue+P1 %lld  build0 %lld  bins %lld (wave 0: gather+split %lld, mfma %lld, barrier %lld)  gates %lld  cand+epi %lld  total %lld cycles\n", H,
               tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tbl, tmm, tbar, tk[4] - tk[3], tk[5] - tk[4], tk[5] - tk[0]);
#endif
    if (tid < TM && row0 + tid < a.R) {
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) sc += red[c * TM + tid];
        a.st_score[row0 + tid] = (a.t == 0 ? 0.f : a.st_score[row0 + tid]) + sc;
    }
}
static size_t ioc_step_x2_lds(const IocStepArgs& a) {
    const int EV = 16, H = a.H, NT = H / 32, E = EV + 32 + H, KX = E + H, B = a.G * a.G, TM = 32;
    const int MW = (a.m_loc * a.nranks + 63) >> 6;
    return (size_t)2 * TM * (KX + 8) * 2 + (size_t)2 * 2 * TM * (H + 8) * 2 + (size_t)TM * B * MW * 8 + (3 * EV + NT * TM) * sizeof(float) + 64;
}
static size_t ioc_step_lds(const IocStepArgs& a) {
    const int EV = 16, H = a.H, NT = H / 32, E = EV + 32 + H, LDX = E + H + 4, LDB = H + 4, B = a.G * a.G, TM = 32;
    const int MW = (a.m_loc * a.nranks + 63) >> 6;
    return ((size_t)TM * LDX + 2 * TM * LDB + (size_t)TM * B * MW * 2 + 3 * EV + NT * TM) * sizeof(float) + 64;
}
void launch_ioc_step(const IocStepArgs& a, hipStream_t s) {
    const dim3 grid((a.R + 31) / 32), block((a.H / 32) * 64);
    const size_t lds = ioc_step_lds(a);
#define STEP_LAUNCH(HH, NPP) { allow_big_lds(k_ioc_step<HH, 16, 32, NPP>); hipLaunchKernelGGL((k_ioc_step<HH, 16, 32, NPP>), grid, block, lds, s, a); }
    if (a.np == 2) {                 // two-piece operands: the piece-image form (the fp32-tile form of round 4, k_ioc_step<.., 2>, measured slower and is gone)
        const size_t l2 = ioc_step_x2_lds(a);
#define STEP2_LAUNCH(HH) { allow_big_lds(k_ioc_step_x2<HH, 16, 32>); hipLaunchKernelGGL((k_ioc_step_x2<HH, 16, 32>), grid, block, l2, s, a); }
        if (a.H == 256) STEP2_LAUNCH(256) else if (a.H == 128) STEP2_LAUNCH(128) else STEP2_LAUNCH(64)
#undef STEP2_LAUNCH
        return;
    }
    if (a.np == 3) { if (a.H == 256) STEP_LAUNCH(256, 3) else if (a.H == 128) STEP_LAUNCH(128, 3) else STEP_LAUNCH(64, 3) return; }
    if (a.H == 256) STEP_LAUNCH(256, 0) else if (a.H == 128) STEP_LAUNCH(128, 0) else STEP_LAUNCH(64, 0)
#undef STEP_LAUNCH
}
// ---- peer exchange: progress counters ------------------------------------------------------------------------------
// Every rank owns one 32-bit counter in its exchange region; value = epoch * per_pass + stage, monotonic over the passes.  A rank's
// kernels of stage s + 1 may start once every peer's counter has reached (epoch, s): k_peer_wait is ONE wave (lane = peer) polling with
// relaxed system-scope loads and a bounded sleep loop, then a system-scope acquire; k_peer_set publishes with a system-scope release.
// The kernels in between are ordinary launches -- nothing compute-sized ever spins, so two ranks that share a GPU (the one-box test)
// cannot starve each other.  The epoch lives in device memory (k_peer_epoch bumps it) so that a captured pass can be replayed.
__global__ void k_peer_wait(PeerFlags flags, int nranks, const unsigned* epoch, unsigned per_pass, unsigned stage, int* err) {
    const int r = threadIdx.x;
    if (r < nranks && !__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) {      // (one time-out per pass, not one per step)
        const unsigned target = ld_sys_u32(epoch) * per_pass + stage;
        const unsigned* f = flags.f[r];
        long spins = 0;
        while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (++spins > 2000000L) { __hip_atomic_fetch_or(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }      // ~4 s
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
__global__ void k_peer_set(unsigned* flag, const unsigned* epoch, unsigned per_pass, unsigned stage) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __hip_atomic_store(flag, ld_sys_u32(epoch) * per_pass + stage, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_peer_epoch(unsigned* epoch) { __hip_atomic_store(epoch, ld_sys_u32(epoch) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
void launch_peer_wait(const PeerFlags& flags, int nranks, const unsigned* epoch, unsigned per_pass, unsigned stage, int* err, hipStream_t s) {
    hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, s, flags, nranks, epoch, per_pass, stage, err);
}
void launch_peer_set(unsigned* flag, const unsigned* epoch, unsigned per_pass, unsigned stage, hipStream_t s) {
    hipLaunchKernelGGL(k_peer_set, dim3(1), dim3(1), 0, s, flag, epoch, per_pass, stage);
}
void launch_peer_epoch(unsigned* epoch, hipStream_t s) { hipLaunchKernelGGL(k_peer_epoch, dim3(1), dim3(1), 0, s, epoch); }
__global__ void k_peer_publish(const uint8_t* __restrict__ valid, const float* __restrict__ p_last, const float* __restrict__ Y,
                               const float* __restrict__ HxHy, int ldhx, uint8_t* __restrict__ o_valid, float* __restrict__ o_plast,
                               float* __restrict__ o_Y, float* __restrict__ o_H, int n_scenes, int K, int mno, int T, int H) {
    const size_t A = (size_t)n_scenes * mno, R = A * K;
    const size_t nY = R * T * 2, nH = R * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nY + nH; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nY) st_sys_f32(o_Y + i, Y[i]);
        else {
            const size_t j = i - nY, r = j / H, c = j - r * H;
            st_sys_f32(o_H + j, HxHy[(size_t)agent_of_row((int)r, K, mno) * ldhx + c]);
        }
        if (i < (A + 3) / 4) {                                  // presence flags, four to a word (the region is zero-padded)
            unsigned wv_ = 0;
            for (int b = 0; b < 4; ++b) if (4 * i + b < A) wv_ |= (unsigned)valid[4 * i + b] << (8 * b);
            __hip_atomic_store(reinterpret_cast<unsigned*>(o_valid) + i, wv_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (i < 2 * A) st_sys_f32(o_plast + i, p_last[i]);
    }
}
void launch_peer_publish(const uint8_t* valid, const float* p_last, const float* Y, const float* HxHy, int ldhx, uint8_t* o_valid,
                         float* o_plast, float* o_Y, float* o_H, int n_scenes, int K, int mno, int T, int H, hipStream_t s) {
    hipLaunchKernelGGL(k_peer_publish, dim3(1024), dim3(256), 0, s, valid, p_last, Y, HxHy, ldhx, o_valid, o_plast, o_Y, o_H, n_scenes, K, mno, T, H);
}

// h_{-1} of every row = Hx of the row's agent: out [R, H] from HxHy [A, ldhx]
__global__ void k_hx_rows(float* __restrict__ out, const float* __restrict__ HxHy, int ldhx, int R, int K, int mno, int H) {
    const size_t n = (size_t)R * H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / H, c = i - r * H;
        out[i] = HxHy[(size_t)agent_of_row((int)r, K, mno) * ldhx + c];
    }
}
void launch_hx_rows(float* out, const float* HxHy, int ldhx, int n_scenes, int K, int mno, int H, hipStream_t s) {
    hipLaunchKernelGGL(k_hx_rows, dim3(1024), dim3(256), 0, s, out, HxHy, ldhx, n_scenes * K * mno, K, mno, H);
}
// end of a pass: Y += dY (dY [R, 2T] from the regression GEMM), score = accumulated + T * b_score
__global__ void k_ioc_finish(float* __restrict__ Y, const float* __restrict__ dY, const float* __restrict__ st_score,
                             const float* __restrict__ b_score, float* __restrict__ score, int R, int T2, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R * T2) Y[i] = Y[i] + dY[i];
    if (i < R) score[i] = st_score[i] + (float)T * b_score[0];
}
void launch_ioc_finish(float* Y, const float* dY, const float* st_score, const float* b_score, float* score, int R, int T, hipStream_t s) {
    const int n = R * 2 * T;
    hipLaunchKernelGGL(k_ioc_finish, dim3((n + 255) / 256), dim3(256), 0, s, Y, dY, st_score, b_score, score, R, 2 * T, T);
}

// ------------------------------------------------------------------------------------------------
// integer paths (standalone, for bit-exact tests)
// ------------------------------------------------------------------------------------------------
__global__ void k_neighbor_bins(const float* __restrict__ pos, const uint8_t* __restrict__ valid,
                                int32_t* __restrict__ bins, int n_groups, int mno, float nb_w, float nb_h, int G,
                                const float* __restrict__ bin_tab) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_groups * mno * mno) return;
    const int g = idx / (mno * mno), ij = idx - g * mno * mno;
    const int i = ij / mno, j = ij - i * mno;
    int b = -1;
    if (i != j && valid[g * mno + j]) {
        const float* p = pos + (size_t)g * mno * 2;
        b = neighbor_bin_dev(p[i * 2], p[i * 2 + 1], p[j * 2], p[j * 2 + 1], nb_w, nb_h, G, bin_tab);
    }
    bins[idx] = b;
}
void launch_neighbor_bins(const float* pos, const uint8_t* valid, int32_t* bins, int n_groups, int mno,
                          float nb_w, float nb_h, int G, const float* bin_tab, hipStream_t s) {
    const int n = n_groups * mno * mno;
    hipLaunchKernelGGL(k_neighbor_bins, dim3((n + 255) / 256), dim3(256), 0, s, pos, valid, bins, n_groups, mno, nb_w, nb_h, G, bin_tab);
}

__global__ void k_scene_cells(const float* __restrict__ pos, int32_t* __restrict__ cells, int n, int Gh, int Gw) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int cy, cx;
    scene_cell_dev(pos[i * 2], pos[i * 2 + 1], Gh, Gw, cy, cx);
    cells[i * 2] = cy;
    cells[i * 2 + 1] = cx;
}
void launch_scene_cells(const float* pos, int32_t* cells, int n, int Gh, int Gw, hipStream_t s) {
    hipLaunchKernelGGL(k_scene_cells, dim3((n + 255) / 256), dim3(256), 0, s, pos, cells, n, Gh, Gw);
}
