// kernels_x6.hip -- sample generation with three-piece operands (dims.bf16 = 3): the GRU decoder and the two large CVAE-decoder
// transposed convolutions as six bf16 MFMAs per fp32 product (split.h).
//
// Why these three: the fp32 matrix pipe (157 TFLOP/s) is 1/16 of the bf16 one, and deconv2 + deconv3 + decoder are 48 of the
// 58 ms the fp32 sample generation takes per 327 680 samples.  Two-piece operands (dims.bf16 = 2, ~1e-5) are NOT an option here:
// the refinement that follows is a discontinuous function of the sampled positions (scene cell and social bin are floors of
// them), so sample generation has to stay in the fp32 kernels' own accuracy class -- which three exact pieces and six products
// are (<= 2^-23 |a b| dropped per product, the size of the fmaf chain's own rounding).
//   * activations stay fp32 in HBM and in LDS; an A fragment is split into its pieces on the fly (44 VALU per fragment, against
//     the >= 192 matrix-pipe cycles of the six MFMAs it feeds), or once per tile where it lives in registers (deconv2)
//   * weights are three-piece bf16 packs [p0 | p1 | p2] in bf16-MFMA fragment order ("*/W6", "dec/Wh?6": api.hip)
//   * tilings, epilogues and the summation order over k are those of the fp32 kernels (kernels_conv.hip, kernels_rnn.hip)
#include "common.h"
#include "kernels.h"

#include "split.h"

// ------------------------------------------------------------------------------------------------------------------
// GRU decoder + head (k_decoder<H, 32>): the constant-input half (x_z W_x, once per tile) stays on the exact fp32 pipe, the per-step
// contractions h W_hg and (r*h) W_hc run as six-product bf16 MFMAs over three bf16 images of h and of r*h, written once per step by the
// wave that owns the columns (splitting the fp32 tile on the fly in every wave: 8.9 ms per 327 680 samples; the images: 8.6 ms).  a.Whg / a.Whc point at the
// three-piece packs.
// ------------------------------------------------------------------------------------------------------------------
template <int H, bool SAVE, int NP = 3>      // NP = 2: three products per fp32 product (the training-mode forward: the first two pieces of the same packs)           // SAVE: training-mode forward (gates / candidate / hidden states kept for BPTT, fp32)
__global__ __launch_bounds__((H / 32) * 64, (H / 32) <= 4 ? 2 : 1) void k_decoder_x6(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 32, LDH = H + 4, LDB = H + 8, NT = H >> 5, G = H >> 3, GH16 = H >> 4, NTHR = NT * 64, TPR = NTHR / TM;
    constexpr int ILO = TM * LDB;                                      // bf16 elements from one piece's image to the next
    float* hs = smem;                       // [32][LDH]  fp32 h (head operand); initial state in the prologue
    float* wo = hs + TM * LDH;              // [H][2] head weights
    float* pl = wo + 2 * H;                 // [32][2] last observed position
    u16* hb = reinterpret_cast<u16*>(pl + TM * 2);     // [3][32][LDB]  the three bf16 pieces of h      (A operand of the gates)
    u16* rb = hb + NP * ILO;                            // [3][32][LDB]  ... of r*h                        (A operand of the candidate)
    float* xs = reinterpret_cast<float*>(hb);          // [32][LDH]  x_z tile: prologue only, in the space the images take afterwards
    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int row0 = blockIdx.x * TM;
    DYN_P(a, row0)
    const int col = cb * 32 + (lane & 31), hi = lane >> 5;
    for (int i = tid; i < TM * (H >> 2); i += NTHR) {
        const int r = i / (H >> 2), c4 = i - r * (H >> 2);
        const int row = min(row0 + r, a.R - 1);
        *reinterpret_cast<float4*>(xs + r * LDH + c4 * 4) = *reinterpret_cast<const float4*>(a.xz + (size_t)row * H + c4 * 4);
        const int ag = agent_of_row(row, a.K, a.mno);
        *reinterpret_cast<float4*>(hs + r * LDH + c4 * 4) = *reinterpret_cast<const float4*>(a.Hx + (size_t)ag * a.ldhx + c4 * 4);
    }
    for (int i = tid; i < 2 * H; i += NTHR) wo[i] = a.w_head[i];
    if (tid < TM) {
        const int ag = agent_of_row(min(row0 + tid, a.R - 1), a.K, a.mno);
        pl[tid * 2] = a.p_last[(size_t)ag * 2];
        pl[tid * 2 + 1] = a.p_last[(size_t)ag * 2 + 1];
    }
    __syncthreads();
    // head weights of this thread's 16 columns in registers, walked from chunk hrot on (k_decoder: conflict-free 16-byte reads of h)
    const int hq = tid % TPR, hrot = TPR >= 8 ? (hq >> 2) * (TPR == 8 ? 2 : 1) : 0;
    // (the head weights are read from LDS every step: held in 32 registers they pushed the training-mode form into spills;
    //  deeper fragment rings in the two contractions -- 4 / 8 k-groups in flight -- were measured with the room this makes: 8.75 - 8.95 vs 8.76 - 8.81 ms)
    float* my_h = hs + (4 * hi) * LDH + col;
    f32x16 xr[1] = {splat16h(a.b_g[col])}, xu[1] = {splat16h(a.b_g[H + col])}, xc[1] = {splat16h(a.b_c[col])};
    {
        const float* x_lane = xs + (lane & 31) * LDH + 4 * hi;
        mma_groups<1>(xr, x_lane, LDH, a.Wxg + ((size_t)cb * G) * 64 + lane, G);
        mma_groups<1>(xu, x_lane, LDH, a.Wxg + ((size_t)(cb + NT) * G) * 64 + lane, G);
        mma_groups<1>(xc, x_lane, LDH, a.Wxc + ((size_t)cb * G) * 64 + lane, G);
    }
    f32x16 h;
#pragma unroll
    for (int i = 0; i < 16; ++i) h[i] = my_h[((i & 3) + 8 * (i >> 2)) * LDH];
    __syncthreads();                                   // every wave is done with the x_z tile: its space now carries the images
    // four values of one accumulator column run (rows 4hi + 8q + 0..3) -> every piece's image of a row-major bf16 tile
    auto put4 = [&](u16* img, int q, float v0, float v1, float v2, float v3) {
        unsigned pa[NP], pb[NP];
        splitp<NP>(v0, v1, pa);
        splitp<NP>(v2, v3, pb);
        u16* x = img + (4 * hi + 8 * q) * LDB + col;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            x[i * ILO] = (u16)pa[i]; x[i * ILO + LDB] = (u16)(pa[i] >> 16); x[i * ILO + 2 * LDB] = (u16)pb[i]; x[i * ILO + 3 * LDB] = (u16)(pb[i] >> 16);
        }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) put4(hb, q, h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
    __syncthreads();
    const uint4* Whg = reinterpret_cast<const uint4*>(a.Whg);
    const uint4* Whc = reinterpret_cast<const uint4*>(a.Whc);
    constexpr size_t PLG = (size_t)2 * NT * GH16 * 64, PLC = (size_t)NT * GH16 * 64;      // uint4 offset from one piece's pack to the next
    const uint4* bg[2] = {Whg + ((size_t)cb * GH16) * 64 + lane, Whg + ((size_t)(cb + NT) * GH16) * 64 + lane};
    const uint4* bc[1] = {Whc + ((size_t)cb * GH16) * 64 + lane};
    const u16* a8 = hb + (lane & 31) * LDB + 8 * hi;
    const u16* r8p = rb + (lane & 31) * LDB + 8 * hi;
    const float bh0 = a.b_head[0], bh1 = a.b_head[1];
    for (int t = 0; t < a.T; ++t) {
        f32x16 g2[2] = {xr[0], xu[0]};
        __builtin_amdgcn_s_setprio(1);                    // (wave priority by phase: 8.72 -> 8.45 ms; deconv3_x6i measured neutral)
        mmax_groups<2, NP>(g2, a8, ILO, bg, PLG, GH16);
        __builtin_amdgcn_s_setprio(0);
        f32x16 u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float rh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float r = sigmoidf_(g2[0][4 * q + e]);
                rh[e] = r * h[4 * q + e];
                u[4 * q + e] = sigmoidf_(g2[1][4 * q + e]);
                if (SAVE && row0 + 4 * hi + 8 * q + e < a.R) {
                    const size_t ix = ((size_t)(row0 + 4 * hi + 8 * q + e) * a.T + t) * H + col;
                    a.sv_r[ix] = r; a.sv_u[ix] = u[4 * q + e];
                }
            }
            put4(rb, q, rh[0], rh[1], rh[2], rh[3]);
        }
        __syncthreads();
        f32x16 ac[1] = {xc[0]};
        __builtin_amdgcn_s_setprio(1);
        mmax_groups<1, NP>(ac, r8p, ILO, bc, PLC, GH16);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float c = tanhf_(ac[0][i]);
            h[i] = gru_blend(u[i], h[i], c);
            if (SAVE && row0 + 4 * hi + (i & 3) + 8 * (i >> 2) < a.R) {
                const size_t ix = ((size_t)(row0 + 4 * hi + (i & 3) + 8 * (i >> 2)) * a.T + t) * H + col;
                a.sv_c[ix] = c; a.hdump[ix] = h[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) my_h[((i & 3) + 8 * (i >> 2)) * LDH] = h[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) put4(hb, q, h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
        __syncthreads();
        {   // head: y = p_last + h W_o + b_o ; TPR threads per row, 16 columns each, their weights in registers (k_decoder: hw0 / hw1;
            // fp32 VALU: trajectory coordinates come straight out of it)
            const int r = tid / TPR;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int c = hq * 16 + 4 * ((jj + hrot) & 3);
                const float4 hv = *reinterpret_cast<const float4*>(hs + r * LDH + c);
                const float4 wa = *reinterpret_cast<const float4*>(wo + 2 * c), wb = *reinterpret_cast<const float4*>(wo + 2 * c + 4);
                s0 = fmaf(hv.x, wa.x, s0); s1 = fmaf(hv.x, wa.y, s1);
                s0 = fmaf(hv.y, wa.z, s0); s1 = fmaf(hv.y, wa.w, s1);
                s0 = fmaf(hv.z, wb.x, s0); s1 = fmaf(hv.z, wb.y, s1);
                s0 = fmaf(hv.w, wb.z, s0); s1 = fmaf(hv.w, wb.w, s1);
            }
            const int q8 = hq;
            s0 += __shfl_xor(s0, 1); s1 += __shfl_xor(s1, 1);
            s0 += __shfl_xor(s0, 2); s1 += __shfl_xor(s1, 2);
            if (TPR >= 8) { s0 += __shfl_xor(s0, 4); s1 += __shfl_xor(s1, 4); }
            if (TPR >= 16) { s0 += __shfl_xor(s0, 8); s1 += __shfl_xor(s1, 8); }
            if (q8 == 0 && row0 + r < a.R)
                *reinterpret_cast<float2*>(a.Y + ((size_t)(row0 + r) * a.T + t) * 2) =
                    make_float2(pl[r * 2] + (s0 + bh0), pl[r * 2 + 1] + (s1 + bh1));
        }
        // the head's reads of h_t (and the gates' of its images) are ordered before the next rewrite by the next step's first barrier
    }
}
template <int H>
static void launch_dec6(const DecArgs& a, hipStream_t s, int np) {
    const size_t lds = (size_t)(32 * (H + 4) + 2 * H + 64) * sizeof(float) + (size_t)6 * 32 * (H + 8) * sizeof(u16);
    if (a.sv_r && np == 2) {                                   // training-mode forward with two-piece operands (DESIRE_FLAG_TRAIN_FWD_3P)
        const size_t lds2 = (size_t)(32 * (H + 4) + 2 * H + 64) * sizeof(float) + (size_t)4 * 32 * (H + 8) * sizeof(u16);
        allow_big_lds(k_decoder_x6<H, true, 2>);
        hipLaunchKernelGGL((k_decoder_x6<H, true, 2>), dim3((a.R + 31) / 32), dim3((H / 32) * 64), lds2, s, a);
        return;
    }
    if (a.sv_r) {                                              // training-mode forward (api.hip sets all four save streams together)
        allow_big_lds(k_decoder_x6<H, true>);
        hipLaunchKernelGGL((k_decoder_x6<H, true>), dim3((a.R + 31) / 32), dim3((H / 32) * 64), lds, s, a);
        return;
    }
    allow_big_lds(k_decoder_x6<H, false>);
    hipLaunchKernelGGL((k_decoder_x6<H, false>), dim3((a.R + 31) / 32), dim3((H / 32) * 64), lds, s, a);
}
bool decoder_x6_supported(int H) { return H == 64 || H == 128 || H == 256; }
void launch_decoder_x6(const DecArgs& a, hipStream_t s, int np) {
    if (a.H == 256) launch_dec6<256>(a, s, np); else if (a.H == 128) launch_dec6<128>(a, s, np); else launch_dec6<64>(a, s, np);
}

// ------------------------------------------------------------------------------------------------------------------
// deconv2: [n,4,4,128] -> [n,8,8,64], 5x5 VALID stride 1, scatter form (k_deconv2): wave = (co-half hf, sample pair sp), M-tile rows =
// (sample, input pixel).  A (K = 128) is split ONCE into 3 x 8 register fragments; the B fragments of all 25 taps x 8 k-groups run
// through one ring RD k-groups deep; 48 MFMAs per tap in two interleaved accumulator chains, then the plain LDS scatter.
// ------------------------------------------------------------------------------------------------------------------
template <int NP>                                        // 3: six products (inference), 2: three (the training-mode forward)
__global__ __launch_bounds__(DS_WG, 2) void k_deconv2_x6(ConvArgs a, size_t plo) {
    extern __shared__ __attribute__((aligned(16))) float out_s6[];    // [4][64 px][64 co]
    constexpr int RD = 4;
    const int lane = lane_id(), w = wave_id();
    const int hf = w & 1, sp = w >> 1;
    // tiles bx, bx + gridDim.x, ..: one per workgroup unless the grid was sized from a count HINT (kernels.h: DynCount.hint) that the real count exceeds
    DYN_N(a, n, blockIdx.x * 4)
    for (int bx = blockIdx.x; bx * 4 < a.n; bx += gridDim.x) {
        const int s0 = bx * 4 + sp * 2;
        const int c = lane & 31, hi = lane >> 5;
        float* my = out_s6 + (sp * 2) * 4096;
        const int er = lane >> 3, ec = hf * 32 + (lane & 7) * 4;
        for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(my + (i * 8 + er) * 64 + ec) = make_float4(0.f, 0.f, 0.f, 0.f);
        uint4 af[8][NP];
        {
            const int row = lane & 31;
            const int smp = min(s0 + (row >> 4), a.n - 1);
            const float* src = a.in + ((size_t)smp * 16 + (row & 15)) * 128 + 8 * hi;
    #pragma unroll
            for (int g = 0; g < 8; ++g) {
                const FragP<NP> f = fragp<NP>(src + g * 16);
    #pragma unroll
                for (int i = 0; i < NP; ++i) af[g][i] = f.p[i];
            }
        }
        const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
        uint4 rb[RD][NP];
        auto req = [&](int tap, int g) {                       // g: compile-time after unrolling; the slot is g % RD (8 % RD == 0)
            const uint4* bp = Wp + ((size_t)(min(tap, 24) * 2 + hf) * 8 + g) * 64 + lane;
    #pragma unroll
            for (int i = 0; i < NP; ++i) rb[g % RD][i] = bp[i * plo];
        };
    #pragma unroll
        for (int g = 0; g < RD; ++g) req(0, g);
    #pragma clang loop unroll(disable)
        for (int tap = 0; tap < 25; ++tap) {
            const int ky = tap / 5, kx = tap - ky * 5;
            f32x16 accA = zero16(), accB = zero16();
            // wave priority by phase (round 6, as in k_ioc): 1 while the tap's 48 MFMAs issue, 0 for the LDS scatter -- the CU's other workgroup is
            // usually in the other phase.  Same-box ABAB: 11.35 -> 10.72 ms (327 680 rows)
            __builtin_amdgcn_s_setprio(1);
    #pragma unroll
            for (int g = 0; g < 8; ++g) {
                const int sl = g % RD;
    #pragma unroll
                for (int pr = 0; pr < Pairs<NP>::N; ++pr) {             // two accumulator chains, alternating
                    if (pr & 1) accB = mfma16(af[g][Pairs<NP>::A[pr]], rb[sl][Pairs<NP>::B[pr]], accB);
                    else accA = mfma16(af[g][Pairs<NP>::A[pr]], rb[sl][Pairs<NP>::B[pr]], accA);
                }
                if (g + RD < 8) req(tap, g + RD); else req(tap + 1, g + RD - 8);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_setprio(0);
            // the 16 targets of a lane are distinct (different input pixels, same tap) and no other lane touches its column
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
            for (int i = 0; i < 16; ++i) *dst[i] = old[i] + (accA[i] + accB[i]);
        }
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
}
void launch_deconv2_x6(const ConvArgs& a, hipStream_t s, int np) {
    const size_t plo = (size_t)25 * 2 * 8 * 64;                        // uint4 per piece: 25 taps x 2 n-tiles x 8 k-groups x 64 lanes
    if (np == 2) {
        allow_big_lds(k_deconv2_x6<2>);
        hipLaunchKernelGGL(k_deconv2_x6<2>, dim3((dyn_units(a.n, a.dyn) + 3) / 4), dim3(DS_WG), 4 * 4096 * sizeof(float), s, a, plo);
        return;
    }
    allow_big_lds(k_deconv2_x6<3>);
    hipLaunchKernelGGL(k_deconv2_x6<3>, dim3((dyn_units(a.n, a.dyn) + 3) / 4), dim3(DS_WG), 4 * 4096 * sizeof(float), s, a, plo);
}

// ------------------------------------------------------------------------------------------------------------------
// deconv3: [n,8,8,64] -> [n,16,16,32], 5x5 SAME stride 2, output-parity gather (k_deconv3), contracted TRANSPOSED (weights = A operand:
// lane = output channel) so the accumulators hold D[co][pixel] with lane = pixel and runs of four channels -> float4 stores.  A pixel
// fragment (8 input channels) feeds six MFMAs; K = 64 = 4 k-groups per tap; the next tap's weight fragments are in flight.
// The operand comes from three pre-split bf16 images of the sample's input in LDS (one conversion per element instead of one per
// fragment use: 25 taps read every pixel ~6 times; the form that split fragments on the fly was measured slower and is gone): two
// samples per workgroup, two waves per sample -- wave (sample, half) owns the output parity classes {(0,0), (1,1)} (4 + 9 taps) or
// {(0,1), (1,0)} (6 + 6) -- so that the images (56 KB) still leave two workgroups per CU.
// ------------------------------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(DS_WG, 2) void k_deconv3_x6i(ConvArgs a, size_t plo) {
    // bf16 elements per piece image: 128 pixels, then 384 bytes of zeros for the taps outside.  A lane whose tap is outside reads the
    // zeros at the byte offset (mod 256) its pixel WOULD have had: ds_read_b128 is serviced in 16-lane groups over a 256-byte bank row,
    // consecutive pixels (144 bytes apart) fill its sixteen 16-byte slots exactly once, and one shared zero address would sit on the
    // slot of some valid lane of the group (a second LDS cycle for every group of a border tap: 31 % of this kernel's LDS cycles)
    constexpr int LDB = 72, NPX = 2 * 64, ZB = NPX * LDB, IMG = ZB + 192;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* img = reinterpret_cast<u16*>(smem);                           // [3][NPX + 1][LDB]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    // tiles bx, bx + gridDim.x, ..: one per workgroup unless the grid was sized from a count HINT (kernels.h: DynCount.hint) that the real count exceeds
    DYN_N(a, n, blockIdx.x * 2)
    for (int bx = blockIdx.x; bx * 2 < a.n; bx += gridDim.x) {
        const int s0 = bx * 2;
        for (int i = tid; i < NP * 192; i += DS_WG) img[(i / 192) * IMG + ZB + (i % 192)] = 0;
        for (int i = tid; i < NPX * 16; i += DS_WG) {
            const int pix = i >> 4, c4 = i & 15;
            const int smp = s0 + (pix >> 6);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (smp < a.n) v = *reinterpret_cast<const float4*>(a.in + ((size_t)s0 * 64 + pix) * 64 + c4 * 4);
            unsigned p0[NP], p1[NP];
            splitp<NP>(v.x, v.y, p0); splitp<NP>(v.z, v.w, p1);
    #pragma unroll
            for (int k = 0; k < NP; ++k) *reinterpret_cast<uint2*>(img + k * IMG + pix * LDB + c4 * 4) = make_uint2(p0[k], p1[k]);
        }
        __syncthreads();
        const int ls = w >> 1, half = w & 1, smp = s0 + ls;
        const int hi = lane >> 5;
        float4 sc[4], sh[4];
    #pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = *reinterpret_cast<const float4*>(a.scale + 8 * q + 4 * hi);
            sh[q] = *reinterpret_cast<const float4*>(a.shift + 8 * q + 4 * hi);
        }
        const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
        int qy[2], qx[2];
    #pragma unroll
        for (int m = 0; m < 2; ++m) { const int q = m * 32 + (lane & 31); qy[m] = q >> 3; qx[m] = q & 7; }
        for (int cls = 0; cls < 2; ++cls) {
            const int py = cls, px = half ? 1 - cls : cls;                 // half 0: (0,0), (1,1);  half 1: (0,1), (1,0)
            f32x16 acc[2] = {zero16(), zero16()};
            const int ny = py ? 3 : 2, nx = px ? 3 : 2, ntap = ny * nx;
            auto tap_of = [&](int t) { const int iy = t / nx, ix = t - iy * nx; return (1 - py + 2 * iy) * 5 + (1 - px + 2 * ix); };
            uint4 bc[4][NP], bn[4][NP];
            auto ldw = [&](uint4 (&b)[4][NP], int tap) {
                const uint4* bp = Wp + ((size_t)tap * 4) * 64 + lane;
    #pragma unroll
                for (int g = 0; g < 4; ++g)
    #pragma unroll
                    for (int i = 0; i < NP; ++i) b[g][i] = bp[i * plo + g * 64];
            };
            auto run = [&](const uint4 (&b)[4][NP], int t) {
                const int tap = tap_of(t), ky = tap / 5, kx = tap - ky * 5;
                const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;   // exact: numerators even
                const u16* xp[2];
    #pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int iy = qy[m] + dy, ix = qx[m] + dx;
                    const bool ok = iy >= 0 && iy < 8 && ix >= 0 && ix < 8;
                    xp[m] = img + (ok ? (ls * 64 + iy * 8 + ix) * LDB : ZB + ((((ls * 64 + iy * 8 + ix + 64) * LDB * 2) & 255) >> 1)) + 8 * hi;
                }
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint4 f0[NP], f1[NP];
    #pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        f0[i] = *reinterpret_cast<const uint4*>(xp[0] + i * IMG + 16 * g);
                        f1[i] = *reinterpret_cast<const uint4*>(xp[1] + i * IMG + 16 * g);
                    }
    #pragma unroll
                    for (int pr = 0; pr < Pairs<NP>::N; ++pr) {            // D[co][pixel]: the two row blocks alternate on the pipe
                        acc[0] = mfma16(b[g][Pairs<NP>::B[pr]], f0[Pairs<NP>::A[pr]], acc[0]);
                        acc[1] = mfma16(b[g][Pairs<NP>::B[pr]], f1[Pairs<NP>::A[pr]], acc[1]);
                    }
                }
            };
            // weight fragments of the next tap in flight while this tap's 48 MFMAs run: two named register sets, no copies
            ldw(bc, tap_of(0));
            int t = 0;
    #pragma clang loop unroll(disable)
            for (; t + 2 <= ntap; t += 2) {
                ldw(bn, tap_of(t + 1));
                __builtin_amdgcn_sched_barrier(0);
                run(bc, t);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 2 < ntap) ldw(bc, tap_of(t + 2));
                __builtin_amdgcn_sched_barrier(0);
                run(bn, t + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t < ntap) run(bc, t);
            if (smp < a.n) {
    #pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int oy = 2 * qy[m] + py, ox = 2 * qx[m] + px;       // this lane's output pixel
                    const size_t base = ((size_t)smp * 256 + oy * 16 + ox) * 32 + 4 * hi;
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float4 o;
                        o.x = eluf_(acc[m][4 * q] * sc[q].x + sh[q].x); o.y = eluf_(acc[m][4 * q + 1] * sc[q].y + sh[q].y);
                        o.z = eluf_(acc[m][4 * q + 2] * sc[q].z + sh[q].z); o.w = eluf_(acc[m][4 * q + 3] * sc[q].w + sh[q].w);
                        *reinterpret_cast<float4*>(a.out + base + 8 * q) = o;
                    }
                }
            }
        }
        __syncthreads();                                   // (the next tile of this workgroup restages the LDS tiles)
    }
}
void launch_deconv3_x6(const ConvArgs& a, hipStream_t s, int np) {
    const size_t plo = (size_t)25 * 1 * 4 * 64;                       // uint4 per piece: 25 taps x 1 n-tile x 4 k-groups x 64 lanes
    if (np == 2) {
        const size_t ldsi = (size_t)2 * (2 * 64 * 72 + 192) * sizeof(u16);
        allow_big_lds(k_deconv3_x6i<2>);
        hipLaunchKernelGGL(k_deconv3_x6i<2>, dim3((dyn_units(a.n, a.dyn) + 1) / 2), dim3(DS_WG), ldsi, s, a, plo);
        return;
    }
    const size_t ldsi = (size_t)3 * (2 * 64 * 72 + 192) * sizeof(u16);
    allow_big_lds(k_deconv3_x6i<3>);
    hipLaunchKernelGGL(k_deconv3_x6i<3>, dim3((dyn_units(a.n, a.dyn) + 1) / 2), dim3(DS_WG), ldsi, s, a, plo);
}

// ------------------------------------------------------------------------------------------------------------------
// Row GEMMs of sample generation in six-product form: deconv1 (k_gemm_rows<EPI_SCALE_SHIFT_ELU>: [R, L] x [L, 2048], folded
// batch-norm + ELU) and the mask fc (k_mask: softmax(relu(xhat W + b)) * Hx, K = 1024).  The 64-row A tile is split once, when a
// 128-column chunk of it is staged: three bf16 images [64][136] in LDS (52 KB: two workgroups per CU), every A fragment read feeds
// all of a wave's n-tiles; weights are [p0 | p1 | p2] packs ("vae_dec/deconv1/W6", "mask/W6").  Same k order as the fp32 kernels.
// ------------------------------------------------------------------------------------------------------------------
constexpr int X6_ROWS2_RD = 2;          // k-groups of weight fragments in flight in the two-row-block ring (1.66 / 0.73 ms at 2; 4 and 8 measured slower)
namespace {
constexpr int KC6 = 128, LDB6 = KC6 + 8, ILO6 = 64 * LDB6;
// stages columns [k0, k0 + KC6) of 64 rows (row0..) of A [M, lda] into the three images
__device__ __forceinline__ void stage6(u16* img, const float* __restrict__ A, int lda, int M, int row0, int k0, int kc, int tid) {
    const int q = kc >> 2;
    for (int i = tid; i < 64 * q; i += DS_WG) {
        const int r = i / q, c4 = i - r * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < M) v = *reinterpret_cast<const float4*>(A + (size_t)(row0 + r) * lda + k0 + c4 * 4);
        unsigned p0[3], p1[3];
        splitp<3>(v.x, v.y, p0); splitp<3>(v.z, v.w, p1);
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<uint2*>(img + k * ILO6 + r * LDB6 + c4 * 4) = make_uint2(p0[k], p1[k]);
    }
}
template <int NTW>
__global__ __launch_bounds__(DS_WG, 2) void k_deconv1_x6(GemmArgs a) {          // K <= KC6: the whole A tile is staged once
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* img = reinterpret_cast<u16*>(smem);
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    DYN_N(a, M, row0)
    const uint4* Bp = reinterpret_cast<const uint4*>(a.Bp);
    const int G16 = a.K >> 4;
    const size_t plo = (size_t)a.NT * G16 * 64;
    const u16* a8 = img + (lane & 31) * LDB6 + 8 * (lane >> 5);
    stage6(img, a.A, a.lda, a.M, row0, 0, a.K, tid);
    __syncthreads();
    // one n-tile at a time (two row blocks = two accumulators): eight live accumulators plus two fragment sets would not fit the register file
#pragma unroll 1
    for (int j = 0; j < NTW; ++j) {
        const int nt = (blockIdx.y * NTW + j) * 4 + w;
        if (nt >= a.NT) continue;
        f32x16 acc2[2] = {zero16(), zero16()};
        // both row blocks per fetched weight fragment, X6_ROWS2_RD k-groups in flight (was: the blocks one after the other, each streaming the
        // n-tile's fragments again).  Same-box A/B, 327 680 rows: deconv1 1.67 (old) / 1.66 (2 in flight) / 1.86 (4, 8) ms -- the kernel waits for its
        // 2.7 GB of output stores, not for fragments; the mask fc 0.79 / 0.73 / 0.70 ms.  (Also measured: the ring run ACROSS the wave's n-tiles, so that
        // no fragment is requested behind a tile's 32 stores per lane -- vmcnt retires in order -- 1.60 -> 1.90 ms.  More loads in flight only hurt here.)
        mmax_rows2_ring<3, X6_ROWS2_RD>(acc2, a8, 32 * LDB6, ILO6, Bp + ((size_t)nt * G16) * 64 + lane, plo, G16);
        f32x16 acc[2][1] = {{acc2[0]}, {acc2[1]}};
        const int col = nt * 32 + (lane & 31);
        if (col >= a.N) continue;
        const int ch = col % a.chmod;
        const float p0 = a.p0[ch], p1 = a.p1[ch];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + m * 32 + acc_row(i);
                if (row < a.M) a.out[(size_t)row * a.ldo + col] = eluf_(acc[m][0][i] * p0 + p1);
            }
    }
}

__global__ __launch_bounds__(DS_WG, 2) void k_mask_x6(MaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u16* img = reinterpret_cast<u16*>(smem);
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int row0 = blockIdx.x * 64;
    DYN_P(a, row0)
    const int NT = a.H >> 5;                                 // 2 or 4 column tiles: wave w takes tile w
    f32x16 acc[2][1] = {{zero16()}, {zero16()}};
    const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
    const int G16 = a.V >> 4;
    const size_t plo = (size_t)NT * G16 * 64;
    const u16* a8 = img + (lane & 31) * LDB6 + 8 * (lane >> 5);
    for (int k0 = 0; k0 < a.V; k0 += KC6) {
        stage6(img, a.xhat, a.V, a.R, row0, k0, KC6, tid);
        __syncthreads();
        if (w < NT) {
            f32x16 acc2[2] = {acc[0][0], acc[1][0]};
            mmax_rows2_ring<3, X6_ROWS2_RD>(acc2, a8, 32 * LDB6, ILO6, Wp + ((size_t)w * G16 + (k0 >> 4)) * 64 + lane, plo, KC6 >> 4);
            acc[0][0] = acc2[0]; acc[1][0] = acc2[1];
        }
        __syncthreads();
    }
    // relu(acc + b) -> LDS tile [64][H+4] (the images are dead), then the row softmax of k_mask
    const int LDT = a.H + 4;
    if (w < NT) {
        const int col = w * 32 + (lane & 31);
        const float b = a.bias[col];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = fmaxf(acc[m][0][i] + b, 0.f);
                smem[(m * 32 + acc_row(i)) * LDT + col] = pv;
                if (a.sv_p && row0 + m * 32 + acc_row(i) < a.R) a.sv_p[(size_t)(row0 + m * 32 + acc_row(i)) * a.H + col] = pv;
            }
    }
    __syncthreads();
    // softmax over H per row: 4 threads per row, each owning every fourth float4 of it (a row's four threads touch 64 contiguous bytes per
    // instruction, in LDS and in the xz / Hx rows alike)
    const int r = tid >> 2, q4 = tid & 3;
    const int row = row0 + r;
    const int nv = a.H >> 4;                    // float4 per thread (H = 64 / 128 / 256: 4 / 8 / 16)
    float4* trow = reinterpret_cast<float4*>(smem + r * LDT);
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
}  // namespace
bool rows_x6_supported(int K, int NT) { return (K % 16) == 0 && K <= 128 && NT % 16 == 0; }
// a.Bp = the three-piece pack; a.K a multiple of 16; a.NT a multiple of 16 (deconv1: 64 n-tiles)
void launch_deconv1_x6(const GemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)3 * ILO6 * sizeof(u16);
    allow_big_lds(k_deconv1_x6<4>);
    hipLaunchKernelGGL((k_deconv1_x6<4>), dim3((a.M + 63) / 64, a.NT / 16), dim3(DS_WG), lds, s, a);
}
// a.Wp = the three-piece pack "mask/W6"; H = 64 or 128, V a multiple of 128
void launch_mask_x6(const MaskArgs& a, hipStream_t s) {
    const size_t lds = (size_t)3 * ILO6 * sizeof(u16);
    allow_big_lds(k_mask_x6);
    hipLaunchKernelGGL(k_mask_x6, dim3((a.R + 63) / 64), dim3(DS_WG), lds, s, a);
}
