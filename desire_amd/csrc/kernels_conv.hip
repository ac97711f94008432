// kernels_conv.hip -- the conv-CVAE stack of DESIRE on the fp32 matrix pipe.
// Replaces vae_encoder (model/model.py:471-492) and vae_decoder (model/model.py:453-469 +
// utils/convolutional_vae_util.py:27-135), batched over agents / (agent,k) rows instead of the
// reference's one-object-at-a-time unroll (model/model.py:211).  NHWC activations, frozen
// batch-norm + bias folded to per-channel (scale, shift), ELU / sigmoid fused in the epilogue.
//
//   conv1   VALU   [32,32,1]  -s2 SAME->  [16,16,32]     (0.4 MF/agent)
//   conv2   MFMA   gather, rows = output pixels, K = 25 taps x 32 ci
//   conv3   MFMA   gather, VALID, K = 25 taps x 64 ci
//   deconv2 MFMA   scatter: G[(sample,in-pixel), (tap,co)] = In @ W, accumulated into an
//                  LDS output tile (gather form would waste 75% of the MFMAs on a 4->8 VALID
//                  transposed conv); each wave owns (sample-pair, co-half) so the read-modify-
//                  write is race-free and deterministic
//   deconv3 MFMA   stride-2 transposed conv split in its 4 output-parity classes, each a
//                  stride-1 gather conv with 2x2 / 2x3 / 3x2 / 3x3 taps: accumulators ARE the
//                  output, no scatter
//   deconv4 VALU   one output channel: 32-wide dot per tap, parity class per wave (+sigmoid)
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// conv1 (VALU): one workgroup per agent
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DS_WG) void k_conv1(ConvArgs a) {
    // image with a one-pixel zero border ([35][40], image at +1,+1): no bounds tests, the five taps of a row come from two
    // aligned 16-byte LDS reads; the lane's 25 weights (its output channel) live in registers
    __shared__ __attribute__((aligned(16))) float in_s[36 * 40];
    const int n = blockIdx.x, tid = threadIdx.x;
    DYN_N(a, n, n)
    for (int i = tid; i < 36 * 40; i += DS_WG) in_s[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 1024; i += DS_WG) in_s[((i >> 5) + 1) * 40 + (i & 31) + 1] = a.in[(size_t)n * 1024 + i];
    const int co = tid & 31, pg = tid >> 5;
    float wr[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) wr[k] = a.w_raw[k * 32 + co];       // [ky][kx][0][co]
    __syncthreads();
    const float sc = a.scale[co], sh = a.shift[co];
    for (int p = pg; p < 256; p += 8) {
        const int oy = p >> 4, ox = p & 15;
        const int base = (2 * ox) & ~3;
        const bool odd = ox & 1;                                      // uniform in the wave (one pixel per wave)
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            const float* row = in_s + (2 * oy + ky) * 40 + base;
            const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
            const float t0 = odd ? v0.z : v0.x, t1 = odd ? v0.w : v0.y, t2 = odd ? v1.x : v0.z, t3 = odd ? v1.y : v0.w, t4 = odd ? v1.z : v1.x;
            acc = fmaf(t0, wr[ky * 5 + 0], acc);                      // same tap order as before: ky outer, kx inner
            acc = fmaf(t1, wr[ky * 5 + 1], acc);
            acc = fmaf(t2, wr[ky * 5 + 2], acc);
            acc = fmaf(t3, wr[ky * 5 + 3], acc);
            acc = fmaf(t4, wr[ky * 5 + 4], acc);
        }
        { const size_t ix = ((size_t)n * 256 + p) * 32 + co; a.out[ix] = conv_epilogue(acc, sc, sh, a.mode, false, a.yprev, ix); }
    }
}
void launch_conv1(const ConvArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_conv1, dim3(a.n), dim3(DS_WG), 0, s, a);
}

// ------------------------------------------------------------------------------------------------
// forward conv, gather form.  Workgroup = 64 output rows = SPW samples x OW*OW pixels.
// NT = CO/32 n-tiles:  NT==2 -> wave = (nt = w&1, m-tile = w>>1), MT = 1
//                      NT==4 -> wave = (nt = w), MT = 2
// ------------------------------------------------------------------------------------------------
template <int CI, int IW, int OW, int STRIDE, int PAD, int CO>
__global__ __launch_bounds__(DS_WG) void k_conv_gather(ConvArgs a) {
    constexpr int PIX = OW * OW, SPW = DS_TM / PIX, LDP = CI + 4, NT = CO / 32, G = CI / 8;
    constexpr int MT = (NT == 4) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* zero_row = smem;                       // LDP zeros
    float* in_s = smem + LDP;                     // [SPW][IW*IW][LDP]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int s0 = blockIdx.x * SPW;
    DYN_N(a, n, s0)
    for (int i = tid; i < LDP; i += DS_WG) zero_row[i] = 0.f;
    constexpr int Q = CI / 4;
    // position of pixel (y, x) in a sample's LDS image.  Stride 2: the four parity classes are kept as separate (IW/2)^2 sub-images -- a tap
    // reads ONE class, so the lanes of a fragment read neighbouring pixels LDP floats apart (conflict-free 16-byte reads) instead of
    // every second pixel (2 LDP = 288 bytes apart: eight lanes per bank group, half of the LDS cycles were conflicts)
    auto pos = [](int y, int x) {
        if (STRIDE == 2) return (((y & 1) * 2 + (x & 1)) * (IW / 2) + (y >> 1)) * (IW / 2) + (x >> 1);
        return y * IW + x;
    };
    for (int i = tid; i < SPW * IW * IW * Q; i += DS_WG) {
        const int pix = i / Q, c4 = i - pix * Q;
        const int ls = pix / (IW * IW), pp = pix - ls * IW * IW;
        const int smp = s0 + ls;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (smp < a.n) v = *reinterpret_cast<const float4*>(a.in + ((size_t)s0 * IW * IW + pix) * CI + c4 * 4);
        *reinterpret_cast<float4*>(in_s + (ls * IW * IW + pos(pp / IW, pp % IW)) * LDP + c4 * 4) = v;
    }
    __syncthreads();
    const int nt = (NT == 4) ? w : (w & 1);
    const int mt0 = (NT == 4) ? 0 : (w >> 1);
    f32x16 acc[MT];
    int oy[MT], ox[MT], sm[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        acc[m] = zero16();
        const int r = (mt0 + m) * 32 + (lane & 31);
        sm[m] = r / PIX;
        const int q = r - sm[m] * PIX;
        oy[m] = q / OW;
        ox[m] = q - oy[m] * OW;
    }
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx) {
            const float* ap[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int iy = oy[m] * STRIDE + ky - PAD, ix = ox[m] * STRIDE + kx - PAD;
                const bool ok = iy >= 0 && iy < IW && ix >= 0 && ix < IW;
                ap[m] = (ok ? in_s + (sm[m] * IW * IW + pos(iy, ix)) * LDP : zero_row) + 4 * (lane >> 5);
            }
            mma_groups_ptr<MT>(acc, ap, a.Wp + ((size_t)((ky * 5 + kx) * NT + nt) * G) * 64 + lane, G);
        }
    const int co = nt * 32 + (lane & 31);
    const float sc = a.scale[co], sh = a.shift[co];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = (mt0 + m) * 32 + acc_row(i);
            const int smp = s0 + r / PIX;
            if (smp < a.n) { const size_t ix = ((size_t)s0 * PIX + r) * CO + co; a.out[ix] = conv_epilogue(acc[m][i], sc, sh, a.mode, false, a.yprev, ix); }
        }
}
void launch_conv2(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (36 + 256 * 36) * sizeof(float);
    hipLaunchKernelGGL((k_conv_gather<32, 16, 8, 2, 1, 64>), dim3(a.n), dim3(DS_WG), lds, s, a);
}
void launch_conv3(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (68 + 4 * 64 * 68) * sizeof(float);
    allow_big_lds(k_conv_gather<64, 8, 4, 1, 0, 128>);
    hipLaunchKernelGGL((k_conv_gather<64, 8, 4, 1, 0, 128>), dim3((a.n + 3) / 4), dim3(DS_WG), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// deconv2: [n,4,4,128] -> [n,8,8,64], 5x5 VALID stride 1, scatter form.
// Workgroup = 4 samples; wave = (co-half hf = w&1, sample pair sp = w>>1).
// M-tile rows = (sample s in {0,1}, input pixel p in 0..15); A (K=128) lives in 64 VGPRs.
// ------------------------------------------------------------------------------------------------
// (Measured and dropped, round 5: a one-off half-tap stagger of the second workgroup of every CU -- by dispatch order or by the hardware wave slot --
// in case the two co-resident workgroups ran their scatters in lock-step: 19.64 -> 19.64 - 20.06 ms, the phases are not aligned to begin with.)
// (Measured and dropped, round 5: the tap's K = 128 as two interleaved chains over its halves, summed at the end -- 19.65 -> 23.3 ms: the chain consumes
// the B fragments in the order they were requested, the interleaved form needs fragment 8 first; `-amdgpu-sched-strategy=max-ilp` for this file: 20.8 ms.)
// (Measured and dropped: two taps interleaved into two accumulators -- the 64-deep dependent chain per tap is ~12 % of the kernel, but
// with the LDS scatter the paired form was slower)
template <bool FWD>
__global__ __launch_bounds__(DS_WG) void k_deconv2(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float out_s[];     // [4][64 px][64 co]
    const int lane = lane_id(), w = wave_id();
    const int hf = w & 1, sp = w >> 1;
    // tiles bx, bx + gridDim.x, ..: one per workgroup unless the grid was sized from a count HINT (kernels.h: DynCount.hint) that the real count exceeds
    DYN_N(a, n, blockIdx.x * 4)
    for (int bx = blockIdx.x; bx * 4 < a.n; bx += gridDim.x) {
        const int s0 = bx * 4 + sp * 2;
        const int c = lane & 31, hi = lane >> 5;
        float* my = out_s + (sp * 2) * 4096;
        // zero this wave's region: 2 samples x 64 px x 32 co
        // this wave's region = [2 samples x 64 px] x its 32 output channels; 8 lanes x float4 cover one row of it
        const int er = lane >> 3, ec = hf * 32 + (lane & 7) * 4;
        for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(my + (i * 8 + er) * 64 + ec) = make_float4(0.f, 0.f, 0.f, 0.f);
        // A fragments: row = lane&31 -> (s = row>>4, p = row&15)
        float4 af[16];
        {
            const int row = lane & 31;
            const int smp = min(s0 + (row >> 4), a.n - 1);
            const float* src = a.in + ((size_t)smp * 16 + (row & 15)) * 128 + 4 * hi;
    #pragma unroll
            for (int g = 0; g < 16; ++g) af[g] = *reinterpret_cast<const float4*>(src + g * 8);
        }
        // taps: B fragments of tap t+1 are fetched while tap t contracts (two register sets, loop unrolled
        // by 2 so the sets are addressed statically).  The scatter is a plain LDS read-add-write: this wave
        // is the only writer of its region (deterministic), and ds_add_f32 measured 1.4x SLOWER on the whole
        // kernel (LDS atomics retire at a fraction of the plain ds_read/ds_write rate).
        auto load_b = [&](float4 (&b)[16], int tap) {
            const float4* bp = a.Wp + ((size_t)(tap * 2 + hf) * 16) * 64 + lane;
    #pragma unroll
            for (int g = 0; g < 16; ++g) b[g] = bp[g * 64];
        };
        auto do_tap = [&](const float4 (&b)[16], int tap) {
            const int ky = tap / 5, kx = tap - ky * 5;
            f32x16 acc = zero16();
    #pragma unroll
            for (int g = 0; g < 16; ++g) {
                acc = mfma32(af[g].x, b[g].x, acc);
                acc = mfma32(af[g].y, b[g].y, acc);
                acc = mfma32(af[g].z, b[g].z, acc);
                acc = mfma32(af[g].w, b[g].w, acc);
            }
            // the 16 targets of a lane are distinct (different input pixels, same tap) and no other lane touches its column:
            // read all, then add and write all -- written as one dependent chain the compiler serialises 16 LDS round trips
            float* dst[16]; float old[16];
    #pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rr = acc_row(i);
                const int s = rr >> 4, p = rr & 15;
                const int o = ((p >> 2) + ky) * 8 + (p & 3) + kx;
                dst[i] = my + (s * 64 + o) * 64 + hf * 32 + c;
            }
    #pragma unroll
            for (int i = 0; i < 16; ++i) old[i] = *dst[i];
    #pragma unroll
            for (int i = 0; i < 16; ++i) *dst[i] = old[i] + acc[i];
        };
        float4 b0[16], b1[16];
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
                if (FWD) {                                                 // forward: BN + ELU, nothing else in the loop
                    o.x = eluf_(v.x * sc4.x + sh4.x); o.y = eluf_(v.y * sc4.y + sh4.y);
                    o.z = eluf_(v.z * sc4.z + sh4.z); o.w = eluf_(v.w * sc4.w + sh4.w);
                } else {
                    o.x = conv_epilogue(v.x, sc4.x, sh4.x, a.mode, false, a.yprev, ix);
                    o.y = conv_epilogue(v.y, sc4.y, sh4.y, a.mode, false, a.yprev, ix + 1);
                    o.z = conv_epilogue(v.z, sc4.z, sh4.z, a.mode, false, a.yprev, ix + 2);
                    o.w = conv_epilogue(v.w, sc4.w, sh4.w, a.mode, false, a.yprev, ix + 3);
                }
                *reinterpret_cast<float4*>(a.out + ix) = o;
            }
        }
    }
}
void launch_deconv2(const ConvArgs& a, hipStream_t s) {
    allow_big_lds(k_deconv2<true>); allow_big_lds(k_deconv2<false>);
    const int ng = dyn_units(a.n, a.dyn);                  // (device-side count: the grid follows the count hint, the kernel strides over what is left)
    if (a.mode == 0) hipLaunchKernelGGL(k_deconv2<true>, dim3((ng + 3) / 4), dim3(DS_WG), 4 * 4096 * sizeof(float), s, a);
    else hipLaunchKernelGGL(k_deconv2<false>, dim3((ng + 3) / 4), dim3(DS_WG), 4 * 4096 * sizeof(float), s, a);
}

// ------------------------------------------------------------------------------------------------
// deconv3: [n,8,8,64] -> [n,16,16,32], 5x5 SAME stride 2:  o = 2i + k - 1.
// Output parity class p = o&1 uses taps k with k = p+1 (mod 2): p=0 -> k in {1,3}, p=1 -> {0,2,4};
// input shift d = (p + 1 - k)/2, i = q + d for o = 2q + p.  One wave per sample, MT = 2 covers the
// 64 outputs of a class.
// ------------------------------------------------------------------------------------------------
// NS = samples (= waves) per workgroup (4; NS = 2 -- four small workgroups per CU instead of two -- measured 15 % slower).
template <bool FWD, int NS>
__global__ __launch_bounds__(NS * 64) void k_deconv3(ConvArgs a) {
    constexpr int LDP = 68;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // 384 bytes of zeros for the taps outside the image.  A lane whose tap is outside reads them at the byte offset (mod 256) its pixel
    // WOULD have had: ds_read_b128 is serviced in 16-lane groups over a 256-byte bank row, consecutive pixels (272 bytes apart) fill its
    // sixteen 16-byte slots exactly once, and ONE shared zero address sits on the slot of some valid lane of the group -- a second LDS
    // cycle for every group of a border tap (19 % of this kernel's LDS cycles were such conflicts)
    float* zero_row = smem;                                            // [96] floats, 256-byte aligned (dynamic LDS base)
    float* in_s = smem + 128;                                          // [NS][64][LDP]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    // tiles bx, bx + gridDim.x, ..: one per workgroup unless the grid was sized from a count HINT (kernels.h: DynCount.hint) that the real count exceeds
    DYN_N(a, n, blockIdx.x * NS)
    for (int bx = blockIdx.x; bx * NS < a.n; bx += gridDim.x) {
        const int s0 = bx * NS;
        for (int i = tid; i < 128; i += NS * 64) zero_row[i] = 0.f;
        for (int i = tid; i < NS * 64 * 16; i += NS * 64) {
            const int pix = i >> 4, c4 = i & 15;
            const int smp = s0 + (pix >> 6);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (smp < a.n) v = *reinterpret_cast<const float4*>(a.in + ((size_t)s0 * 64 + pix) * 64 + c4 * 4);
            *reinterpret_cast<float4*>(in_s + pix * LDP + c4 * 4) = v;
        }
        __syncthreads();
        const int smp = s0 + w;
        const float* mine = in_s + w * 64 * LDP;
        const int c = lane & 31, hi = lane >> 5;
        // transposed contraction (weights as the A operand): the accumulators hold D[co][pixel] with lane = pixel and the 16
        // registers = output channels in runs of four, so the epilogue stores float4 (4x fewer store instructions)
        float4 sc[4], sh[4];
    #pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc[q] = *reinterpret_cast<const float4*>(a.scale + 8 * q + 4 * hi);
            sh[q] = *reinterpret_cast<const float4*>(a.shift + 8 * q + 4 * hi);
        }
        int qy[2], qx[2];
    #pragma unroll
        for (int m = 0; m < 2; ++m) { const int q = m * 32 + (lane & 31); qy[m] = q >> 3; qx[m] = q & 7; }
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                f32x16 acc[2] = {zero16(), zero16()};
                for (int ky = 1 - py; ky < 5; ky += 2)
                    for (int kx = 1 - px; kx < 5; kx += 2) {
                        const int dy = (py + 1 - ky) / 2, dx = (px + 1 - kx) / 2;   // exact: numerators even
                        const float* ap[2];
    #pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int iy = qy[m] + dy, ix = qx[m] + dx;
                            const bool ok = iy >= 0 && iy < 8 && ix >= 0 && ix < 8;
                            ap[m] = (ok ? mine + (iy * 8 + ix) * LDP : zero_row + ((((w * 64 + iy * 8 + ix + 64) * LDP + 128) & 63))) + 4 * hi;
                        }
                        mma_groups_ptr<2, true>(acc, ap, a.Wp + ((size_t)(ky * 5 + kx) * 8) * 64 + lane, 8);
                    }
                if (smp < a.n) {
    #pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int oy = 2 * qy[m] + py, ox = 2 * qx[m] + px;       // this lane's output pixel
                        const size_t base = ((size_t)smp * 256 + oy * 16 + ox) * 32 + 4 * hi;
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const size_t ix = base + 8 * q;
                            float4 o;
                            if (FWD) {
                                o.x = eluf_(acc[m][4 * q] * sc[q].x + sh[q].x); o.y = eluf_(acc[m][4 * q + 1] * sc[q].y + sh[q].y);
                                o.z = eluf_(acc[m][4 * q + 2] * sc[q].z + sh[q].z); o.w = eluf_(acc[m][4 * q + 3] * sc[q].w + sh[q].w);
                            } else {
                                o.x = conv_epilogue(acc[m][4 * q], sc[q].x, sh[q].x, a.mode, false, a.yprev, ix);
                                o.y = conv_epilogue(acc[m][4 * q + 1], sc[q].y, sh[q].y, a.mode, false, a.yprev, ix + 1);
                                o.z = conv_epilogue(acc[m][4 * q + 2], sc[q].z, sh[q].z, a.mode, false, a.yprev, ix + 2);
                                o.w = conv_epilogue(acc[m][4 * q + 3], sc[q].w, sh[q].w, a.mode, false, a.yprev, ix + 3);
                            }
                            *reinterpret_cast<float4*>(a.out + ix) = o;
                        }
                    }
                }
            }
        __syncthreads();                                   // (the next tile of this workgroup restages the LDS tiles)
    }
}
void launch_deconv3(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (128 + (size_t)4 * 64 * 68) * sizeof(float);
    allow_big_lds(k_deconv3<true, 4>); allow_big_lds(k_deconv3<false, 4>);
    const int ng = dyn_units(a.n, a.dyn);
    if (a.mode == 0) hipLaunchKernelGGL((k_deconv3<true, 4>), dim3((ng + 3) / 4), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((k_deconv3<false, 4>), dim3((ng + 3) / 4), dim3(256), lds, s, a);
}

// ------------------------------------------------------------------------------------------------
// deconv4 (VALU): [n,16,16,32] -> [n,32,32], 5x5 SAME stride 2, BN + sigmoid.
// One workgroup per sample; wave = output parity class (uniform tap set), 4 pixels per lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DS_WG) void k_deconv4(ConvArgs a) {
    constexpr int LDP = 36;
    __shared__ __attribute__((aligned(16))) float in_s[256 * LDP];
    __shared__ __attribute__((aligned(16))) float w_s[25 * 32];
    const int n = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = wave_id();
    for (int i = tid; i < 256 * 8; i += DS_WG) {
        const int pix = i >> 3, c4 = i & 7;
        *reinterpret_cast<float4*>(in_s + pix * LDP + c4 * 4) =
            *reinterpret_cast<const float4*>(a.in + ((size_t)n * 256 + pix) * 32 + c4 * 4);
    }
    for (int i = tid; i < 800; i += DS_WG) w_s[i] = a.w_raw[i];        // [ky][kx][0][ci]
    __syncthreads();
    const int py = w >> 1, px = w & 1;
    const float sc = a.scale[0], sh = a.shift[0];
    for (int j = 0; j < 4; ++j) {
        const int q = j * 64 + lane;                                   // 0..255 within the class
        const int qy = q >> 4, qx = q & 15;
        float acc = 0.f;
        for (int ky = 1 - py; ky < 5; ky += 2) {
            const int iy = qy + (py + 1 - ky) / 2;
            if (iy < 0 || iy >= 16) continue;
            for (int kx = 1 - px; kx < 5; kx += 2) {
                const int ix = qx + (px + 1 - kx) / 2;
                if (ix < 0 || ix >= 16) continue;
                const float* ip = in_s + (iy * 16 + ix) * LDP;
                const float* wp = w_s + (ky * 5 + kx) * 32;
#pragma unroll
                for (int c4 = 0; c4 < 8; ++c4) {
                    const float4 x = *reinterpret_cast<const float4*>(ip + c4 * 4);
                    const float4 ww = *reinterpret_cast<const float4*>(wp + c4 * 4);
                    acc = fmaf(x.x, ww.x, acc); acc = fmaf(x.y, ww.y, acc);
                    acc = fmaf(x.z, ww.z, acc); acc = fmaf(x.w, ww.w, acc);
                }
            }
        }
        { const size_t ix = (size_t)n * 1024 + (2 * qy + py) * 32 + 2 * qx + px; a.out[ix] = conv_epilogue(acc, sc, sh, a.mode, true, a.yprev, ix); }
    }
}
// Forward form, "tap products first": deconv4 has ONE output channel, so every input pixel contributes 25 scalars
//     T[tap] = sum_c d3[pixel, c] * w[tap, c]
// to the outputs o = 2i + k - 1.  A lane owns a pixel (its 32 channels sit in registers, read once, coalesced from HBM),
// the weights of a tap are wave-uniform (scalar loads), and the products are accumulated into an fp32 LDS image of the
// sample with plain read-add-write: inside one instruction all lanes hold the SAME tap, so the addresses are distinct
// (deterministic).  One wave per sample; HBM-bound (the gather form above is LDS-read-bound: 3x slower).
__global__ __launch_bounds__(DS_WG) void k_deconv4_tp(ConvArgs a) {
    __shared__ float xacc[4 * 1024];
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int smp = blockIdx.x * 4 + w;
    DYN_N(a, n, blockIdx.x * 4)
    for (int i = tid; i < 4 * 1024; i += DS_WG) xacc[i] = 0.f;
    __syncthreads();
    float* xa = xacc + w * 1024;
    const float* __restrict__ wr = a.w_raw;                              // [25][32]
    if (smp < a.n) {
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
            const int p = j * 64 + lane, iy = p >> 4, ix = p & 15;
            float4 x[8];
            const float4* src = reinterpret_cast<const float4*>(a.in + ((size_t)smp * 256 + p) * 32);
#pragma unroll
            for (int c4 = 0; c4 < 8; ++c4) x[c4] = src[c4];
#pragma unroll 1
            for (int ky = 0; ky < 5; ++ky) {
                const int Y = 2 * iy + ky - 1;
#pragma unroll
                for (int kx = 0; kx < 5; ++kx) {
                    const float* wt = wr + (ky * 5 + kx) * 32;           // wave-uniform
                    float acc = 0.f;
#pragma unroll
                    for (int c4 = 0; c4 < 8; ++c4) {
                        acc = fmaf(x[c4].x, wt[c4 * 4], acc); acc = fmaf(x[c4].y, wt[c4 * 4 + 1], acc);
                        acc = fmaf(x[c4].z, wt[c4 * 4 + 2], acc); acc = fmaf(x[c4].w, wt[c4 * 4 + 3], acc);
                    }
                    const int X = 2 * ix + kx - 1;
                    if (Y >= 0 && Y < 32 && X >= 0 && X < 32) xa[Y * 32 + X] += acc;
                }
            }
        }
        const float sc = a.scale[0], sh = a.shift[0];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = i * 64 + lane;
            a.out[(size_t)smp * 1024 + o] = sigmoidf_(xa[o] * sc + sh);
        }
    }
}
void launch_deconv4(const ConvArgs& a, hipStream_t s) {
    if (a.mode == 0) hipLaunchKernelGGL(k_deconv4_tp, dim3((a.n + 3) / 4), dim3(DS_WG), 0, s, a);
    else hipLaunchKernelGGL(k_deconv4, dim3(a.n), dim3(DS_WG), 0, s, a);
}
