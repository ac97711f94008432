// kernels_bwd.hip -- backward pass of the sample-generation module (the reference computes tf.gradients of its
// cost, model/model.py:388, and never applies them; here the gradients exist and are applied).
// Structure mirrors the forward: persistent reverse-time recurrences per 32-row tile with the data-gradient
// contractions on the fp32 matrix pipe against TRANSPOSED weights (packed once per weight update), the gate
// gradients written to HBM, and every weight gradient computed afterwards as one big A^T.G reduction over all
// rows and steps (k_gemm_tn, slice partials + fixed-order reduce = deterministic).
#include "common.h"
#include "kernels.h"

__device__ __forceinline__ f32x16 splat16b(float v) {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = v;
    return z;
}
__device__ __forceinline__ void mma1b(f32x16& acc, const float* a_lane, const float4* __restrict__ b_lane, int G) {
    f32x16 t[1] = {acc};
    mma_groups<1>(t, a_lane, 0, b_lane, G);
    acc = t[0];
}

// ---- number of existing agents (id != 0 at the last observed frame), as float ------------------------------
__global__ void k_count_valid(const uint8_t* __restrict__ valid, int A, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int a = threadIdx.x; a < A; a += 256) s += valid[a] ? 1.f : 0.f;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = fmaxf(red[0], 1.f);
}
void launch_count_valid(const uint8_t* valid, int A, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_count_valid, dim3(1), dim3(256), 0, s, valid, A, out);
}

// ---- d L_sgm / d Y0:  lmask(a) present(a,t) / (N K nfut(a)) * (Y0 - gt) / ||Y0 - gt|| ------------------------------
__global__ void k_loss_grad_y(const float* __restrict__ Y, const float* __restrict__ fut, const uint8_t* __restrict__ valid,
                              const float* __restrict__ nfut, const float* __restrict__ nvalid, float* __restrict__ dY,
                              int n_scenes, int mno, int K, int T, float sx, float sy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long R = (long)n_scenes * K * mno;
    if (i >= R * T) return;
    const int t = i % T;
    const long r = i / T;
    const int slot = r % mno, sc = r / ((long)K * mno);
    const int a = sc * mno + slot;
    const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
    const float dx = Y[i * 2] - __fmul_rn(f[1], sx), dy = Y[i * 2 + 1] - __fmul_rn(f[2], sy);
    const float nrm = sqrtf(dx * dx + dy * dy);
    const float g = (valid[a] && f[0] != 0.f && nrm > 0.f) ? 1.0f / (nvalid[0] * (float)K * nfut[a] * nrm) : 0.f;
    dY[i * 2] = g * dx;
    dY[i * 2 + 1] = g * dy;
}
void launch_loss_grad_y(const float* Y, const float* fut, const uint8_t* valid, const float* nfut, const float* nvalid, float* dY,
                        int n_scenes, int mno, int K, int T, float sx, float sy, hipStream_t s) {
    const long n = (long)n_scenes * K * mno * T;
    hipLaunchKernelGGL(k_loss_grad_y, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Y, fut, valid, nfut, nvalid, dY, n_scenes,
                       mno, K, T, sx, sy);
}

// ------------------------------------------------------------------------------------------------------------------
// Decoder BPTT.  Tile = 32 rows, wave cb owns hidden columns [32cb, 32cb+32).  Per reverse step:
//   dh_t   = dh_{t}(from t+1) + dY0_t Wo^T
//   du = dh (h_{t-1} - c);  dc = dh (1-u);  dh_{t-1} = dh u
//   da_c = dc (1-c^2);      d(r h) = da_c Wc_h^T            (MFMA, K = H)
//   dr = d(rh) h_{t-1};     dh_{t-1} += d(rh) r
//   da_r = dr r(1-r); da_u = du u(1-u);  dh_{t-1} += [da_r|da_u] Wg_h^T     (MFMA, K = 2H)
// da_* and r*h_{t-1} go to HBM for the weight-gradient reductions; the constant-input sums dxg, dxc give dx_z.
// Both contractions run TRANSPOSED (mma SWAP: the packed weights are the A operand, the LDS rows the B operand): the accumulators
// hold, per lane, ONE row (lane & 31) and runs of FOUR consecutive hidden columns (32 cb + 8 q + 4 (lane >> 5) + 0..3, q = 0..3).
// Everything elementwise is layout-agnostic, so this only shapes the memory traffic: every saved stream is read, and every gradient
// stream written, as one 16-byte access per four elements (20 global loads + 24 stores per lane and step; the row-major form of
// rounds 1-3 issued 64 + 96 four-byte ones and spilled 63 dwords of stream offsets), and the operand tiles take 16-byte LDS writes
// (row stride = 4 mod 8 words: conflict-free for the b128 lane groups).  The bias column sums are reduced through LDS at the end.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mma1t(f32x16& acc, const float* a_lane, const float4* __restrict__ b_lane, int G) {
    f32x16 t[1] = {acc};
    const float* ap[1] = {a_lane};
    mma_groups_ptr<1, true>(t, ap, b_lane, G);
    acc = t[0];
}
__device__ __forceinline__ float f4get(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
template <int H, int NW = 2>
__global__ __launch_bounds__((H / 32) * 64, 2) void k_decoder_bwd(DecBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 32, NT = H / 32, NTHR = NT * 64, LD1 = H + 4, LD2 = 2 * H + 4, GH = H / 8, G2 = 2 * H / 8;
    float* A1 = smem;                    // [32][LD1]   da_c
    float* A2 = A1 + TM * LD1;           // [32][LD2]   da_r | da_u
    float* dy = A2 + TM * LD2;           // [32][NW]
    float* wo = dy + TM * NW;            // [NW][H] (transposed: a lane reads the head weights of its four columns as one float4)
    float* A3 = wo + NW * H;             // [32][LD1]   r * h_{t-1}: staged only to leave as whole rows
    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int row0 = blockIdx.x * TM;
    const int lr = lane & 31, hi = lane >> 5;
    const int c0 = cb * 32 + 4 * hi;                       // first column of run q: c0 + 8 q
    for (int i = tid; i < NW * H; i += NTHR) wo[(i % NW) * H + i / NW] = a.w_head[i];
    const float* a1_lane = A1 + lr * LD1 + 4 * hi;
    const float* a2_lane = A2 + lr * LD2 + 4 * hi;
    float* my1 = A1 + lr * LD1 + c0;
    float* my2 = A2 + lr * LD2 + c0;
    f32x16 dh = zero16(), sxr = zero16(), sxu = zero16(), sxc = zero16();
    const int nloc = min(TM, a.R - row0);
    const bool rok = lr < nloc;
    const int rcl = min(lr, nloc - 1);                     // rows past R read the tile's last row (their results are never stored)
    const size_t tb = (size_t)row0 * a.T;
    const float* svu = a.sv_u + tb * H; const float* svc = a.sv_c + tb * H; const float* svr = a.sv_r + tb * H; const float* svh = a.sv_h + tb * H;
    float* o_dac = a.dac + tb * H; float* o_rh = a.rh + tb * H; float* o_hp = a.hprev + tb * H; float* o_dag = a.dag + tb * 2 * H;
    if (a.dh_init) {                                       // encoders: the gradient arrives at the final state
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(a.dh_init + (size_t)(row0 + rcl) * a.ld_init + c0 + 8 * q);
            dh[4 * q] = v.x; dh[4 * q + 1] = v.y; dh[4 * q + 2] = v.z; dh[4 * q + 3] = v.w;
        }
    }
    for (int t = a.T - 1; t >= 0; --t) {
        int rc = rcl;
        asm volatile("v_mov_b32 %0, %1" : "=v"(rc) : "v"(rcl));     // opaque per step: the stream offsets are re-formed (two VALU ops), not hoisted and spilled
        const unsigned rt = (unsigned)(rc * a.T + t);
        __syncthreads();                                   // previous step's A2 / dy consumers are done
        if (tid < TM) {
            if constexpr (NW == 2) {
                float2 v = make_float2(0.f, 0.f);
                if (a.dY0) v = *reinterpret_cast<const float2*>(a.dY0 + ((size_t)min(row0 + tid, a.R - 1) * a.T + t) * 2);
                dy[tid * 2] = v.x; dy[tid * 2 + 1] = v.y;
            } else {
#pragma unroll
                for (int j = 0; j < NW; ++j) dy[tid * NW + j] = a.dY0[((size_t)min(row0 + tid, a.R - 1) * a.T + t) * NW + j];
            }
        }
        if (t == 0) {                                      // h_{-1} = Hx[agent] (decoder) or 0 (encoders): staged in A1, each run
            for (int i = tid; i < TM * (H >> 2); i += NTHR) {   // is read by its owner right before it is overwritten with da_c
                const int r = i / (H >> 2), c4 = i - r * (H >> 2);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.Hx) v = *reinterpret_cast<const float4*>(a.Hx + (size_t)agent_of_row(min(row0 + r, a.R - 1), a.K, a.mno) * a.ldhx + c4 * 4);
                *reinterpret_cast<float4*>(A1 + r * LD1 + c4 * 4) = v;
            }
        }
        __syncthreads();
        f32x16 dhp, rr, hp;
        float dyv[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) dyv[j] = dy[lr * NW + j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned ix = rt * H + c0 + 8 * q;
            const float4 u4 = *reinterpret_cast<const float4*>(svu + ix), c4 = *reinterpret_cast<const float4*>(svc + ix);
            const float4 r4 = *reinterpret_cast<const float4*>(svr + ix);
            const float4 h4 = (t > 0) ? *reinterpret_cast<const float4*>(svh + ix - H) : *reinterpret_cast<const float4*>(my1 + 8 * q);
            float dacv[4], dauv[4], rhv[4];
            float4 wj[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j) wj[j] = *reinterpret_cast<const float4*>(wo + j * H + c0 + 8 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q + e;
                const float u = f4get(u4, e), c = f4get(c4, e), r = f4get(r4, e), hprev = f4get(h4, e);
                float dht = dh[i] + dyv[0] * f4get(wj[0], e) + dyv[1] * f4get(wj[1], e);
#pragma unroll
                for (int j = 2; j < NW; ++j) dht += dyv[j] * f4get(wj[j], e);
                const float dau = dht * (hprev - c) * u * (1.0f - u);
                const float dc = dht * (1.0f - u);
                dhp[i] = dht * u;
                const float dac = dc * (1.0f - c * c);
                dacv[e] = dac; dauv[e] = dau; rhv[e] = r * hprev;
                sxc[i] += dac; sxu[i] += dau;
                rr[i] = r; hp[i] = hprev;
            }
            const float4 dac4 = make_float4(dacv[0], dacv[1], dacv[2], dacv[3]), dau4 = make_float4(dauv[0], dauv[1], dauv[2], dauv[3]);
            *reinterpret_cast<float4*>(my1 + 8 * q) = dac4;
            *reinterpret_cast<float4*>(my2 + H + 8 * q) = dau4;
            *reinterpret_cast<float4*>(A3 + lr * LD1 + c0 + 8 * q) = make_float4(rhv[0], rhv[1], rhv[2], rhv[3]);
        }
        __syncthreads();
        // da_c leaves through its LDS tile (the next contraction's operand anyway), every thread one 16-byte piece of a row's 512 contiguous
        // bytes: whole lines per store instruction.  Stored from the accumulator layout a lane writes 32 bytes of each of 32 different
        // rows per instruction, and the memory system moved 1.6x the streams' bytes (13.2 GB written for 8.4).
        // (r h_{t-1} the same way through a tile of its own; h_{t-1} -- the weight gradient's operand -- is a row-for-row copy of the saved states)
        for (int i = tid; i < nloc * (H >> 2); i += NTHR) {
            const int r = i / (H >> 2), c4 = i - r * (H >> 2);
            const size_t o = (size_t)(r * a.T + t) * H + c4 * 4;
            *reinterpret_cast<float4*>(o_dac + o) = *reinterpret_cast<const float4*>(A1 + r * LD1 + c4 * 4);
            *reinterpret_cast<float4*>(o_rh + o) = *reinterpret_cast<const float4*>(A3 + r * LD1 + c4 * 4);
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t > 0) hv = *reinterpret_cast<const float4*>(svh + o - H);
            else if (a.Hx) hv = *reinterpret_cast<const float4*>(a.Hx + (size_t)agent_of_row(row0 + r, a.K, a.mno) * a.ldhx + c4 * 4);
            *reinterpret_cast<float4*>(o_hp + o) = hv;
        }
        f32x16 drh = zero16();
        mma1t(drh, a1_lane, a.WcT_h + ((size_t)cb * GH) * 64 + lane, GH);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float darv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q + e;
                const float dr = drh[i] * hp[i];
                dhp[i] += drh[i] * rr[i];
                const float dar = dr * rr[i] * (1.0f - rr[i]);
                darv[e] = dar;
                sxr[i] += dar;
            }
            const float4 dar4 = make_float4(darv[0], darv[1], darv[2], darv[3]);
            *reinterpret_cast<float4*>(my2 + 8 * q) = dar4;
        }
        __syncthreads();
        for (int i = tid; i < nloc * (H >> 1); i += NTHR) {        // [da_r | da_u]: 2H contiguous floats per row
            const int r = i / (H >> 1), c4 = i - r * (H >> 1);
            *reinterpret_cast<float4*>(o_dag + (size_t)(r * a.T + t) * 2 * H + c4 * 4) = *reinterpret_cast<const float4*>(A2 + r * LD2 + c4 * 4);
        }
        f32x16 dhg = zero16();
        mma1t(dhg, a2_lane, a.WgT_h + ((size_t)cb * G2) * 64 + lane, G2);
#pragma unroll
        for (int i = 0; i < 16; ++i) dh[i] = dhp[i] + dhg[i];
    }
    __syncthreads();
    // the step sums of the gate gradients as tiles: operands of the constant-input contraction (decoder) and source of the bias column sums
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<float4*>(my2 + 8 * q) = rok ? make_float4(sxr[4 * q], sxr[4 * q + 1], sxr[4 * q + 2], sxr[4 * q + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(my2 + H + 8 * q) = rok ? make_float4(sxu[4 * q], sxu[4 * q + 1], sxu[4 * q + 2], sxu[4 * q + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(my1 + 8 * q) = rok ? make_float4(sxc[4 * q], sxc[4 * q + 1], sxc[4 * q + 2], sxc[4 * q + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    if (a.bias_part) {                                     // bias gradients: this tile's column sums (rows past R hold zeros), fixed row order
        float* part = a.bias_part + (size_t)blockIdx.x * 3 * H;
        for (int c = tid; c < 3 * H; c += NTHR) {
            const float* src = c < 2 * H ? A2 + c : A1 + (c - 2 * H);
            const int ld = c < 2 * H ? LD2 : LD1;
            float sum = 0.f;
            for (int r = 0; r < TM; ++r) sum += src[r * ld];
            part[c] = sum;
        }
    }
    if (!a.dxz) return;                                    // encoders: inputs are data, nothing upstream
    if (rok) {
        const size_t rg = (size_t)(row0 + lr);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<float4*>(a.dHx_rows + rg * H + c0 + 8 * q) = make_float4(dh[4 * q], dh[4 * q + 1], dh[4 * q + 2], dh[4 * q + 3]);
            *reinterpret_cast<float4*>(a.dxg + rg * 2 * H + c0 + 8 * q) = make_float4(sxr[4 * q], sxr[4 * q + 1], sxr[4 * q + 2], sxr[4 * q + 3]);
            *reinterpret_cast<float4*>(a.dxg + rg * 2 * H + H + c0 + 8 * q) = make_float4(sxu[4 * q], sxu[4 * q + 1], sxu[4 * q + 2], sxu[4 * q + 3]);
            *reinterpret_cast<float4*>(a.dxc + rg * H + c0 + 8 * q) = make_float4(sxc[4 * q], sxc[4 * q + 1], sxc[4 * q + 2], sxc[4 * q + 3]);
        }
    }
    f32x16 dxz = zero16();
    mma1t(dxz, a2_lane, a.WgT_x + ((size_t)cb * G2) * 64 + lane, G2);
    mma1t(dxz, a1_lane, a.WcT_x + ((size_t)cb * GH) * 64 + lane, GH);
    if (rok) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(a.dxz + (size_t)(row0 + lr) * H + c0 + 8 * q) = make_float4(dxz[4 * q], dxz[4 * q + 1], dxz[4 * q + 2], dxz[4 * q + 3]);
    }
}
void launch_decoder_bwd(const DecBwdArgs& a, hipStream_t s) {
    const int H = a.H;
    const size_t lds = (2 * 32 * (H + 4) + 32 * (2 * H + 4) + 32 * 5 + 5 * H) * sizeof(float);
    const dim3 grid((a.R + 31) / 32);
    if (a.nw == 5) {                                      // the X encoder with the Gaussian head's per-step gradient (desire_set_head_loss)
        if (H == 256) { allow_big_lds(k_decoder_bwd<256, 5>); hipLaunchKernelGGL((k_decoder_bwd<256, 5>), grid, dim3(512), lds, s, a); }
        else if (H == 128) { allow_big_lds(k_decoder_bwd<128, 5>); hipLaunchKernelGGL((k_decoder_bwd<128, 5>), grid, dim3(256), lds, s, a); }
        else hipLaunchKernelGGL((k_decoder_bwd<64, 5>), grid, dim3(128), lds, s, a);
        return;
    }
    if (H == 256) { allow_big_lds(k_decoder_bwd<256>); hipLaunchKernelGGL(k_decoder_bwd<256>, grid, dim3(512), lds, s, a); }
    else if (H == 128) { allow_big_lds(k_decoder_bwd<128>); hipLaunchKernelGGL(k_decoder_bwd<128>, grid, dim3(256), lds, s, a); }
    else hipLaunchKernelGGL(k_decoder_bwd<64>, grid, dim3(128), lds, s, a);
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradients:  out[Kd, N] (+)= sum_m A[m, Kd]^T G[m, N]   over M rows (up to R*T).
// Workgroup = one 64x64 output block x one slice of M; the A and G chunks (64 rows) go through LDS, the contraction
// index is m.  Slice partials are summed in slice order by k_reduce_slices (deterministic, no float atomics).
// ------------------------------------------------------------------------------------------------------------------
#define TN_MC 64
#define TN_LD 68
__global__ __launch_bounds__(256) void k_gemm_tn(TnArgs a) {
    __shared__ __attribute__((aligned(16))) float As[TN_MC * TN_LD];
    __shared__ __attribute__((aligned(16))) float Gs[TN_MC * TN_LD];
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int nbn = (a.N + 63) / 64;
    const int bk = (blockIdx.x / nbn) * 64, bn = (blockIdx.x % nbn) * 64;
    const int ti = w >> 1, tj = w & 1;
    const long mper = ((a.M + a.nslices - 1) / a.nslices + TN_MC - 1) / TN_MC * TN_MC;
    const long m_lo = (long)blockIdx.y * mper, m_hi = min(a.M, m_lo + mper);
    f32x16 acc = zero16();
    const int hi = lane >> 5, c = lane & 31;
    for (long m0 = m_lo; m0 < m_hi; m0 += TN_MC) {
        __syncthreads();
        for (int i = tid; i < TN_MC * 64; i += 256) {
            const int r = i >> 6, cc = i & 63;
            const long m = m0 + r;
            const bool ok = m < m_hi;
            float av = (ok && bk + cc < a.Kd) ? a.A[(size_t)m * a.lda + bk + cc] : 0.f;
            if (a.flags && ok && !((a.flags[m] >> ((bk + cc) / a.fcols)) & 1ull)) av = 0.f;     // block-sparse A: unwritten block
            As[r * TN_LD + cc] = av;
            Gs[r * TN_LD + cc] = (ok && bn + cc < a.N) ? a.G[(size_t)m * a.ldg + bn + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < TN_MC / 8; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 8 * g + 4 * hi + i;
                acc = mfma32(As[m * TN_LD + ti * 32 + c], Gs[m * TN_LD + tj * 32 + c], acc);
            }
        }
    }
    float* out = a.partial + (size_t)blockIdx.y * a.Kd * a.N;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = bk + ti * 32 + acc_row(i), n = bn + tj * 32 + c;
        if (k < a.Kd && n < a.N) out[(size_t)k * a.N + n] = acc[i];
    }
}
__global__ void k_reduce_slices(const float* __restrict__ partial, int nslices, int Kd, int N, float* __restrict__ out, int ldo,
                                int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Kd * N) return;
    float s = 0.f;
#pragma unroll 8
    for (int sl = 0; sl < nslices; ++sl) s += partial[(size_t)sl * Kd * N + i];
    float* o = out + (size_t)(i / N) * ldo + (i % N);
    *o = accumulate ? (*o + s) : s;
}
// Large blocks: (WK*64) x (WN*64) output tile per workgroup (WK*WN = 4 waves, each 2x2 32x32 tiles), 32-row chunks of
// m double-buffered in LDS, next chunk's global float4 loads issued before the current chunk's 64 MFMAs per wave.  A
// lane owns the column PAIR (2c, 2c+1) of its wave's 64-wide strip, so one ds_read_b64 per operand feeds two tiles
// (the output index map absorbs the interleave).  Needs 16-byte aligned rows (lda, ldg, Kd, N multiples of 4).
// CONV: the A operand is the im2col view of a convolution's large-grid tensor (ConvGather), column = tap*Cl + cl,
// row m = (sample, small-grid pixel) -- the weight gradient of a (transposed) convolution as ONE [25*Cl] x [Cs] GEMM.
// LISTS: the row-list form (TnArgs::rowlist) as an instantiation of its own -- as run-time branches inside the one kernel the list code made the
// compiler wait for every row load separately (vmcnt(0) per element), and the plain dense calls took twice their time
template <int WK, int WN, bool CONV, bool LISTS = false>
__global__ __launch_bounds__(256, 2) void k_gemm_tn2(TnArgs a, ConvGather cg) {
    constexpr int BK = WK * 64, BN = WN * 64, QA = BK / 4, QG = BN / 4, PA = 32 / (256 / QA), PG = 32 / (256 / QG);
    __shared__ __attribute__((aligned(16))) float As[2][32 * BK];
    __shared__ __attribute__((aligned(16))) float Gs[2][32 * BN];
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int nbn = (a.N + BN - 1) / BN;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int bk = (bx / nbn) * BK, bn = (bx % nbn) * BN;
    const int wk = w / WN, wn = w % WN;
    // slice y takes the 32-row chunks y, y + nslices, y + 2 nslices, ..: the workgroups in flight read NEIGHBOURING chunks (contiguous
    // ranges per slice put every stream a multiple of megabytes apart -- the same HBM channels at the same time)
    const int* const rl = LISTS ? a.rowlist + a.binbase[bk / a.fcols] : nullptr;       // this k-block's rows are list entries [0, bintotal[b])
    const long m_hi = LISTS ? (long)a.bintotal[bk / a.fcols] : a.M;
    const long step = (long)a.nslices * 32;
    const long m_lo = (long)by * 32;
    const int hi = lane >> 5, c = lane & 31;
    f32x16 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = zero16();
    const int qa = tid % QA, ra0 = tid / QA, qg = tid % QG, rg0 = tid / QG;
    const int kcol = bk + 4 * qa;
    const bool ka = kcol < a.Kd, na = bn + 4 * qg < a.N;
    int ky = 0, kx = 0, cl = 0;
    if (CONV) { const int tap = kcol / cg.Cl; cl = kcol - tap * cg.Cl; ky = tap / 5; kx = tap - ky * 5; }
    const int PP = cg.Ps * cg.Ps;
    const int ps_sh = (CONV && cg.Ps > 0 && (cg.Ps & (cg.Ps - 1)) == 0) ? __ffs(cg.Ps) - 1 : -1;
    float4 ra[PA], rg[PG];
    auto arow = [&](int j) { return CONV ? PA * ra0 + j : ra0 + (256 / QA) * j; };      // chunk row of this thread's j-th A row (convolutions: consecutive = one grid row)
    int ia[LISTS ? PA : 1], ig[LISTS ? PG : 1];            // list mode: the next chunk's row indices, fetched one gload ahead
    auto iload = [&](long m0) {
        if constexpr (LISTS) {
#pragma unroll
            for (int j = 0; j < PA; ++j) { const long m = m0 + arow(j); ia[j] = m < m_hi ? rl[m] : 0; }
#pragma unroll
            for (int j = 0; j < PG; ++j) { const long m = m0 + rg0 + (256 / QG) * j; ig[j] = m < m_hi ? rl[m] : 0; }
        }
    };
    iload(m_lo);
    auto gload = [&](long m0) {
        // convolution layers whose small grid is PA pixels wide: a thread's rows are one row of the grid (one sample, one py, px = j), the
        // gathers share a base address and differ by constant strides (kernels_bwd_x3.hip: k_gemm_tn2_xp)
        if (CONV && cg.Ps == PA && ps_sh >= 0) {
            const long m = m0 + PA * ra0;
            const long nn = m >> (2 * ps_sh);
            const int py = (int)(m >> ps_sh) & (PA - 1);
            const int qy = cg.stride * py + ky - cg.pad;
            const bool rowok = m < m_hi && ka && qy >= 0 && qy < cg.Pl;
            const float* base = a.A + (((size_t)nn * cg.Pl + (rowok ? qy : 0)) * cg.Pl) * cg.Cl + cl;
#pragma unroll
            for (int j = 0; j < PA; ++j) {
                const int qx = cg.stride * j + kx - cg.pad;
                ra[j] = (rowok && qx >= 0 && qx < cg.Pl) ? *reinterpret_cast<const float4*>(base + qx * cg.Cl) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const long m = m0 + arow(j);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_hi && ka) {
                if (CONV) {
                    long n; int p, py, px;                       // (sample, small-grid pixel) of row m
                    if (ps_sh >= 0) {                            // power-of-two grid side (every layer of this model): shifts, no divisions
                        n = m >> (2 * ps_sh); p = (int)(m & (PP - 1));
                        py = p >> ps_sh; px = p & (cg.Ps - 1);
                    } else {
                        n = m / PP; p = (int)(m - n * PP);
                        py = p / cg.Ps; px = p - py * cg.Ps;
                    }
                    const int qy = cg.stride * py + ky - cg.pad, qx = cg.stride * px + kx - cg.pad;
                    if (qy >= 0 && qy < cg.Pl && qx >= 0 && qx < cg.Pl)
                        v = *reinterpret_cast<const float4*>(a.A + (((size_t)n * cg.Pl + qy) * cg.Pl + qx) * cg.Cl + cl);
                } else {
                    long mr = m;
                    if constexpr (LISTS) mr = (long)ia[j];
                    v = *reinterpret_cast<const float4*>(a.A + (size_t)mr * a.lda + kcol);
                    // block-sparse A: a block whose flag is clear was never written by the producer (stale memory): read as zero
                    if (a.flags && !((a.flags[m] >> (kcol / a.fcols)) & 1ull)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const long m = m0 + rg0 + (256 / QG) * j;
            long mr = m;
            if constexpr (LISTS) mr = (long)ig[j];
            rg[j] = (m < m_hi && na) ? *reinterpret_cast<const float4*>(a.G + (size_t)mr * a.ldg + bn + 4 * qg) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        iload(m0 + step);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PA; ++j) *reinterpret_cast<float4*>(&As[buf][arow(j) * BK + 4 * qa]) = ra[j];
#pragma unroll
        for (int j = 0; j < PG; ++j) *reinterpret_cast<float4*>(&Gs[buf][(rg0 + (256 / QG) * j) * BN + 4 * qg]) = rg[j];
    };
    // block-sparse A (a.flags): only chunks with a set flag bit inside this workgroup's k-block are visited
    unsigned long long kmask = ~0ull;
    if (a.flags) {
        const int b_lo = bk / a.fcols, b_hi = min((bk + BK - 1) / a.fcols, 63);
        kmask = (b_hi - b_lo >= 63) ? ~0ull : (((1ull << (b_hi - b_lo + 1)) - 1ull) << b_lo);
    }
    auto next_live = [&](long m0) {                       // first chunk start >= m0 that has something in the k-block (uniform)
        if (!a.flags) return m0;
        for (; m0 < m_hi; m0 += step) {                  // one flag word per lane, OR-reduced over the wave (same value in every wave)
            unsigned long long f = (c < 32 && hi == 0 && m0 + c < m_hi) ? a.flags[m0 + c] & kmask : 0ull;
            unsigned lo32 = (unsigned)f | (unsigned)(f >> 32);
            const unsigned long long any = __ballot(lo32 != 0u);
            if (any) break;
        }
        return m0;
    };
    long m0 = next_live(m_lo);
    if (m0 < m_hi) { gload(m0); lstore(0); }
    __syncthreads();
    int buf = 0;
    while (m0 < m_hi) {
        const long m1 = next_live(m0 + step);
        const bool more = m1 < m_hi;
        if (more) gload(m1);
        const float* ap = &As[buf][hi * BK + wk * 64 + 2 * c];
        const float* gp = &Gs[buf][hi * BN + wn * 64 + 2 * c];
        if (bk + wk * 64 < a.Kd && bn + wn * 64 < a.N)      // a wave whose whole strip lies past Kd / N has only zeros to multiply
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const float2 av = *reinterpret_cast<const float2*>(ap + p * 2 * BK);
            const float2 gv = *reinterpret_cast<const float2*>(gp + p * 2 * BN);
            acc[0][0] = mfma32(av.x, gv.x, acc[0][0]);
            acc[0][1] = mfma32(av.x, gv.y, acc[0][1]);
            acc[1][0] = mfma32(av.y, gv.x, acc[1][0]);
            acc[1][1] = mfma32(av.y, gv.y, acc[1][1]);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
        m0 = m1;
    }
    float* out = a.partial + (size_t)by * a.Kd * a.N;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = bk + wk * 64 + 2 * acc_row(i) + u, n = bn + wn * 64 + 2 * c + v;
                if (k < a.Kd && n < a.N) out[(size_t)k * a.N + n] = acc[u][v][i];
            }
}
// same sum, 8 lanes per output striding the slices, then a fixed-order LDS combine (still deterministic)
__global__ __launch_bounds__(256) void k_reduce_slices8(const float* __restrict__ partial, int nslices, int Kd, int N, float* __restrict__ out,
                                                        int ldo, int accumulate) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float s = 0.f;
    if (i < Kd * N) {
#pragma unroll 8
        for (int sl = g; sl < nslices; sl += 8) s += partial[(size_t)sl * Kd * N + i];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && i < Kd * N) {
        const float t = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
        float* o = out + (size_t)(i / N) * ldo + (i % N);
        *o = accumulate ? (*o + t) : t;
    }
}
__global__ __launch_bounds__(256) void k_reduce_parts(const float* __restrict__ part, int nparts, int ld, int off, int N, float* __restrict__ out, int accumulate) {
    __shared__ float red[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float sacc = 0.f;
    if (n < N) {                                           // (unrolled: the loads of eight parts are in flight together -- one at a time the 2 560 tiles' parts took 0.13 ms per call)
#pragma unroll 8
        for (int pth = g; pth < nparts; pth += 8) sacc += part[(size_t)pth * ld + off + n];
    }
    red[g][c] = sacc;
    __syncthreads();
    if (g == 0 && n < N) {
        const float t = ((red[0][c] + red[1][c]) + (red[2][c] + red[3][c])) + ((red[4][c] + red[5][c]) + (red[6][c] + red[7][c]));
        out[n] = accumulate ? out[n] + t : t;
    }
}
void launch_reduce_parts(const float* part, int nparts, int ld, int off, int N, float* out, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(k_reduce_parts, dim3((N + 31) / 32), dim3(256), 0, s, part, nparts, ld, off, N, out, accumulate);
}
static void reduce_slices(const float* partial, int nslices, int Kd, int N, float* out, int ldo, int accumulate, hipStream_t s) {
    const int n = Kd * N;
    if (nslices >= 32 && n <= (1 << 18))
        hipLaunchKernelGGL(k_reduce_slices8, dim3((n + 31) / 32), dim3(256), 0, s, partial, nslices, Kd, N, out, ldo, accumulate);
    else
        hipLaunchKernelGGL(k_reduce_slices, dim3((n + 255) / 256), dim3(256), 0, s, partial, nslices, Kd, N, out, ldo, accumulate);
}

// Skinny weight gradients (one side <= 5 wide: the 2-d output head, the scalar score head, the 2-d velocity input, the 5-wide Gaussian head):
//   out[k, j] = sum_m W[m, k] * Nn[m, j],  W wide (KW = 4*VW columns, VW a power of two <= 256), Nn narrow (NN <= 5).
// HBM-bound streaming of W: thread owns one float4 column of W and every (256/VW)-th row.  transpose_out writes out[j, k].
template <int NN>
__global__ __launch_bounds__(256) void k_tn_skinny(const float* __restrict__ Wd, int ldw, int KW, const float* __restrict__ Nn, int ldn,
                                                   long M, int nslices, float* __restrict__ partial) {
    __shared__ float4 red[256];
    const int VW = KW >> 2, RP = 256 / VW;
    const int cv = threadIdx.x % VW, ro = threadIdx.x / VW;
    const long mper = (M + nslices - 1) / nslices;
    const long lo = (long)blockIdx.x * mper, hi = min(M, lo + mper);
    float4 acc[NN];
#pragma unroll
    for (int j = 0; j < NN; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* base = Wd + 4 * cv;
#pragma unroll 4
    for (long m = lo + ro; m < hi; m += RP) {
        const float4 x = *reinterpret_cast<const float4*>(base + (size_t)m * ldw);
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            const float g = Nn[(size_t)m * ldn + j];
            acc[j].x = fmaf(x.x, g, acc[j].x); acc[j].y = fmaf(x.y, g, acc[j].y);
            acc[j].z = fmaf(x.z, g, acc[j].z); acc[j].w = fmaf(x.w, g, acc[j].w);
        }
    }
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        __syncthreads();
        red[threadIdx.x] = acc[j];
        __syncthreads();
        if (ro == 0) {
            float4 t = red[cv];
            for (int q = 1; q < RP; ++q) { const float4 y = red[q * VW + cv]; t.x += y.x; t.y += y.y; t.z += y.z; t.w += y.w; }
            float* o = partial + (size_t)blockIdx.x * KW * NN;      // [k][j]
            o[(4 * cv + 0) * NN + j] = t.x; o[(4 * cv + 1) * NN + j] = t.y; o[(4 * cv + 2) * NN + j] = t.z; o[(4 * cv + 3) * NN + j] = t.w;
        }
    }
}
__global__ void k_reduce_slices_t(const float* __restrict__ partial, int nslices, int KW, int NN, float* __restrict__ out, int ldo, int accumulate) {
    // partial [slice][k][j] -> out[j][k] (ld = ldo)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KW * NN) return;
    float s = 0.f;
#pragma unroll 8
    for (int sl = 0; sl < nslices; ++sl) s += partial[(size_t)sl * KW * NN + i];
    float* o = out + (size_t)(i % NN) * ldo + (i / NN);
    *o = accumulate ? (*o + s) : s;
}
static bool skinny(const float* Wd, int ldw, int KW, const float* Nn, int ldn, int NN, long M, float* partial, float* out, int ldo,
                   int accumulate, bool transpose_out, hipStream_t s) {
    const int VW = KW >> 2;
    if ((KW & 3) || VW < 1 || VW > 256 || (VW & (VW - 1)) || (ldw & 3) || (reinterpret_cast<uintptr_t>(Wd) & 15) || NN < 1 || NN > 5) return false;
    long sl = M / 512; if (sl < 1) sl = 1; if (sl > 512) sl = 512;
    const int ns = (int)sl;
    switch (NN) {
        case 1: hipLaunchKernelGGL(k_tn_skinny<1>, dim3(ns), dim3(256), 0, s, Wd, ldw, KW, Nn, ldn, M, ns, partial); break;
        case 2: hipLaunchKernelGGL(k_tn_skinny<2>, dim3(ns), dim3(256), 0, s, Wd, ldw, KW, Nn, ldn, M, ns, partial); break;
        case 3: hipLaunchKernelGGL(k_tn_skinny<3>, dim3(ns), dim3(256), 0, s, Wd, ldw, KW, Nn, ldn, M, ns, partial); break;
        case 5: hipLaunchKernelGGL(k_tn_skinny<5>, dim3(ns), dim3(256), 0, s, Wd, ldw, KW, Nn, ldn, M, ns, partial); break;
        default: hipLaunchKernelGGL(k_tn_skinny<4>, dim3(ns), dim3(256), 0, s, Wd, ldw, KW, Nn, ldn, M, ns, partial); break;
    }
    if (transpose_out) hipLaunchKernelGGL(k_reduce_slices_t, dim3((KW * NN + 255) / 256), dim3(256), 0, s, partial, ns, KW, NN, out, ldo, accumulate);
    else reduce_slices(partial, ns, KW, NN, out, ldo, accumulate, s);
    return true;
}

// ---- per-bin row lists of a block-sparse A operand (the pooled tensor of the social-fc weight gradient) ------------------------------
// flags[m] bit b = block b of row m is non-zero.  Three passes: counts per (2048-row block, bin); one workgroup turns them into offsets and
// the bins' bases; the rows are written in row order (so the reduction order -- and the result -- does not depend on scheduling).
__global__ __launch_bounds__(256) void k_bin_count(const unsigned long long* __restrict__ flags, long M, int B, int* __restrict__ counts) {
    __shared__ int red[4];
    const long base = (long)blockIdx.x * 2048 + (long)threadIdx.x * 8;
    unsigned long long f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (base + k < M) ? flags[base + k] : 0ull;
    for (int b = 0; b < B; ++b) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c += (int)((f[k] >> b) & 1ull);
        for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) counts[(size_t)blockIdx.x * B + b] = (red[0] + red[1]) + (red[2] + red[3]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_bin_scan(int* __restrict__ counts, int nblk, int B, int* __restrict__ bintotal) {      // one workgroup per bin
    __shared__ int part[256];
    const int per = (nblk + 255) / 256, lo = threadIdx.x * per, hi = min(nblk, lo + per);
    const int b = blockIdx.x;
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[(size_t)i * B + b];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int v = part[i]; part[i] = run; run += v; }
        bintotal[b] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = counts[(size_t)i * B + b]; counts[(size_t)i * B + b] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_bin_fill(const unsigned long long* __restrict__ flags, long M, int B, const int* __restrict__ offs,
                                                  const int* __restrict__ bintotal, int* __restrict__ binbase, int* __restrict__ rowlist) {
    __shared__ int wsum[4];
    __shared__ int sbase[65];                              // the bins' bases: prefix sums of their totals (every workgroup forms them; workgroup 0 publishes them)
    if (threadIdx.x == 0) {
        int run = 0;
        for (int b = 0; b < B; ++b) { sbase[b] = run; run += bintotal[b]; }
        sbase[B] = run;
        if (blockIdx.x == 0) for (int b = 0; b <= B; ++b) binbase[b] = sbase[b];
    }
    __syncthreads();
    const long base = (long)blockIdx.x * 2048 + (long)threadIdx.x * 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (base + k < M) ? flags[base + k] : 0ull;
    for (int b = 0; b < B; ++b) {
        int c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c += (int)((f[k] >> b) & 1ull);
        int inc = c;                                        // inclusive scan over the wave
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int pos = sbase[b] + offs[(size_t)blockIdx.x * B + b] + (inc - c);
        for (int i = 0; i < w; ++i) pos += wsum[i];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((f[k] >> b) & 1ull) rowlist[pos++] = (int)(base + k);
        __syncthreads();
    }
}
void launch_bin_lists(const unsigned long long* flags, long M, int B, int* counts, int* binbase, int* bintotal, int* rowlist, hipStream_t s) {
    const int nblk = (int)((M + 2047) / 2048);
    hipLaunchKernelGGL(k_bin_count, dim3(nblk), dim3(256), 0, s, flags, M, B, counts);
    hipLaunchKernelGGL(k_bin_scan, dim3(B), dim3(256), 0, s, counts, nblk, B, bintotal);
    hipLaunchKernelGGL(k_bin_fill, dim3(nblk), dim3(256), 0, s, flags, M, B, static_cast<const int*>(counts), static_cast<const int*>(bintotal), binbase, rowlist);
}

static bool tn_big(const TnArgs& a) {
    return a.Kd >= 64 && a.N >= 64 && !(a.lda & 3) && !(a.ldg & 3) && !(a.Kd & 3) && !(a.N & 3) &&
           !(reinterpret_cast<uintptr_t>(a.A) & 15) && !(reinterpret_cast<uintptr_t>(a.G) & 15);
}
// output tiles (= workgroups per slice) of the form launch_gemm_tn picks; 0: the 64 x 64 tiles of k_gemm_tn or a skinny form
int gemm_tn_big_tiles(const TnArgs& a) {
    if (!tn_big(a)) return 0;
    return a.N <= 64 ? (a.Kd + 255) / 256 : ((a.Kd + 127) / 128) * ((a.N + 127) / 128);
}
void launch_gemm_tn(const TnArgs& a, float* out, int ldo, int accumulate, hipStream_t s) {
    if (a.N <= 5 && a.Kd >= 16 && skinny(a.A, a.lda, a.Kd, a.G, a.ldg, a.N, a.M, a.partial, out, ldo, accumulate, false, s)) return;
    if (a.Kd <= 4 && a.N >= 16 && skinny(a.G, a.ldg, a.N, a.A, a.lda, a.Kd, a.M, a.partial, out, ldo, accumulate, true, s)) return;
    const bool big = tn_big(a);
    if (big) {
        if (a.np == 2) launch_gemm_tn2_split(a, nullptr, a.N <= 64, s);
        else if (a.N <= 64) {
            const int nb = (a.Kd + 255) / 256;
            hipLaunchKernelGGL((k_gemm_tn2<4, 1, false>), dim3(nb, a.nslices), dim3(256), 0, s, a, ConvGather{});
        } else {
            const int nb = ((a.Kd + 127) / 128) * ((a.N + 127) / 128);
            if (a.rowlist) hipLaunchKernelGGL((k_gemm_tn2<2, 2, false, true>), dim3(nb, a.nslices), dim3(256), 0, s, a, ConvGather{});
            else hipLaunchKernelGGL((k_gemm_tn2<2, 2, false>), dim3(nb, a.nslices), dim3(256), 0, s, a, ConvGather{});
        }
    } else {
        const int nb = ((a.Kd + 63) / 64) * ((a.N + 63) / 64);
        hipLaunchKernelGGL(k_gemm_tn, dim3(nb, a.nslices), dim3(256), 0, s, a);
    }
    reduce_slices(a.partial, a.nslices, a.Kd, a.N, out, ldo, accumulate, s);
}

// ---- column sums: out[N] (+)= sum_m G[m, N] ---------------------------------------------------------------------------
__global__ void k_colsum(const float* __restrict__ G, int ldg, long M, int N, int nslices, float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const long mper = (M + nslices - 1) / nslices;
    const long lo = (long)blockIdx.y * mper, hi = min(M, lo + mper);
    float s = 0.f;
    if (n < N) for (long m = lo + q; m < hi; m += 4) s += G[(size_t)m * ldg + n];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && n < N) partial[(size_t)blockIdx.y * N + n] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// float4 variant for N = 4 * VN with VN a power of two <= 256: thread owns one float4 column and every (256/VN)-th row,
// four independent loads in flight; the workgroup covers all N columns, blockIdx.x = slice.
__global__ __launch_bounds__(256) void k_colsum4(const float* __restrict__ G, int ldg, long M, int N, int nslices, float* __restrict__ partial) {
    __shared__ float4 red[256];
    const int VN = N >> 2, RP = 256 / VN;
    const int cv = threadIdx.x % VN, ro = threadIdx.x / VN;
    // slice = every nslices-th group of 4 RP rows (neighbouring workgroups read neighbouring rows: contiguous ranges per slice start
    // megabytes apart and meet in the same HBM channels)
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    const float* base = G + 4 * cv;
    const long stride = (long)nslices * 4 * RP;
    long m = (long)blockIdx.x * 4 * RP + ro;
    for (; m + 3 * RP < M; m += stride) {
        const float4 x0 = *reinterpret_cast<const float4*>(base + (size_t)m * ldg);
        const float4 x1 = *reinterpret_cast<const float4*>(base + (size_t)(m + RP) * ldg);
        const float4 x2 = *reinterpret_cast<const float4*>(base + (size_t)(m + 2 * RP) * ldg);
        const float4 x3 = *reinterpret_cast<const float4*>(base + (size_t)(m + 3 * RP) * ldg);
        s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
        s1.x += x1.x; s1.y += x1.y; s1.z += x1.z; s1.w += x1.w;
        s2.x += x2.x; s2.y += x2.y; s2.z += x2.z; s2.w += x2.w;
        s3.x += x3.x; s3.y += x3.y; s3.z += x3.z; s3.w += x3.w;
    }
    for (int j = 0; j < 3; ++j)                                       // the ragged last group (at most one slice reaches it)
        if (m + j * RP < M) {
            const float4 x0 = *reinterpret_cast<const float4*>(base + (size_t)(m + j * RP) * ldg);
            s0.x += x0.x; s0.y += x0.y; s0.z += x0.z; s0.w += x0.w;
        }
    s0.x = (s0.x + s1.x) + (s2.x + s3.x); s0.y = (s0.y + s1.y) + (s2.y + s3.y);
    s0.z = (s0.z + s1.z) + (s2.z + s3.z); s0.w = (s0.w + s1.w) + (s2.w + s3.w);
    red[threadIdx.x] = s0;
    __syncthreads();
    if (ro == 0) {
        float4 t = red[cv];
        for (int j = 1; j < RP; ++j) { const float4 y = red[j * VN + cv]; t.x += y.x; t.y += y.y; t.z += y.z; t.w += y.w; }
        *reinterpret_cast<float4*>(partial + (size_t)blockIdx.x * N + 4 * cv) = t;
    }
}
__global__ __launch_bounds__(256) void k_sum_all(const float* __restrict__ G, long n4, float* __restrict__ partial) {
    __shared__ float red[256];
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    const float4* g = reinterpret_cast<const float4*>(G);
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long st = (long)gridDim.x * 256;
    for (; i + st < n4; i += 2 * st) {
        const float4 x = g[i], y = g[i + st];
        s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w;
        s1.x += y.x; s1.y += y.y; s1.z += y.z; s1.w += y.w;
    }
    if (i < n4) { const float4 x = g[i]; s0.x += x.x; s0.y += x.y; s0.z += x.z; s0.w += x.w; }
    red[threadIdx.x] = ((s0.x + s1.x) + (s0.y + s1.y)) + ((s0.z + s1.z) + (s0.w + s1.w));
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
void launch_colsum(const float* G, int ldg, long M, int N, int nslices, float* partial, float* out, int accumulate, hipStream_t s) {
    if (N == 1 && ldg == 1 && !(M & 3) && !(reinterpret_cast<uintptr_t>(G) & 15) && M >= 4096) {
        const int nb = 1024;
        hipLaunchKernelGGL(k_sum_all, dim3(nb), dim3(256), 0, s, G, M / 4, partial);
        reduce_slices(partial, nb, 1, 1, out, 1, accumulate, s);
        return;
    }
    if (N == 2 && ldg == 2 && !(M & 1) && !(reinterpret_cast<uintptr_t>(G) & 15) && M >= 4096) {
        // two-column sums (the 2-d head's bias gradient over rows x steps): two rows are one float4, and the four column sums of that view fold
        // pairwise when the partials are read as twice as many slices of width two.  (Through k_colsum two lanes of a wave did the work: 0.43 ms.)
        long sl = (M / 2) / 256; if (sl < 1) sl = 1; if (sl > 2048) sl = 2048;
        hipLaunchKernelGGL(k_colsum4, dim3((int)sl), dim3(256), 0, s, G, 4, M / 2, 4, (int)sl, partial);
        reduce_slices(partial, 2 * (int)sl, 1, 2, out, 2, accumulate, s);
        return;
    }
    const int VN = N >> 2;
    const bool vec = !(N & 3) && VN >= 1 && VN <= 256 && !(VN & (VN - 1)) && !(ldg & 3) && !(reinterpret_cast<uintptr_t>(G) & 15);
    if (vec) {
        long sl = M / 256; if (sl < 1) sl = 1; if (sl > 2048) sl = 2048;
        nslices = (int)sl;
        hipLaunchKernelGGL(k_colsum4, dim3(nslices), dim3(256), 0, s, G, ldg, M, N, nslices, partial);
    } else {
        hipLaunchKernelGGL(k_colsum, dim3((N + 63) / 64, nslices), dim3(256), 0, s, G, ldg, M, N, nslices, partial);
    }
    reduce_slices(partial, nslices, 1, N, out, N, accumulate, s);
}

// ------------------------------------------------------------------------------------------------------------------
// mask fc backward (forward: q = xhat Wm + bm, p = relu(q), beta = softmax(p), xz = beta * Hx):
//   dbeta = dxz * Hx;  dHx_rows += dxz * beta;  dp = beta * (dbeta - sum(beta dbeta));  dq = dp * (p > 0)
// 4 threads per row, like the forward softmax.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mask_bwd(const float* __restrict__ p, const float* __restrict__ dxz,
                                                  const float* __restrict__ Hx, int ldhx, float* __restrict__ dq,
                                                  float* __restrict__ dHx_rows, int R, int H, int Hl, int K, int mno) {
    // 4 threads per row, each owning every fourth float4 of it (a row's four threads read 64 contiguous bytes per instruction; the
    // earlier form walked 32 scalars per thread four times over: 0.78 of the kernel's time in the vector-memory path)
    const int r = blockIdx.x * 64 + (threadIdx.x >> 2), q4 = threadIdx.x & 3;
    const int row = min(r, R - 1);
    const int nv = H >> 4;                                    // float4 per thread (H = 64 / 128 / 256: 4 / 8 / 16)
    const float4* pr = reinterpret_cast<const float4*>(p + (size_t)row * H);
    const float4* gx = reinterpret_cast<const float4*>(dxz + (size_t)row * H);
    const float4* hx = reinterpret_cast<const float4*>(Hx + (size_t)agent_of_row(row, K, mno) * ldhx);
    float4 pv[16], gv[16], hv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < nv) { pv[j] = pr[4 * j + q4]; gv[j] = gx[4 * j + q4]; hv[j] = hx[4 * j + q4]; }
    float mx = -3.0e38f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < nv) mx = fmaxf(mx, fmaxf(fmaxf(pv[j].x, pv[j].y), fmaxf(pv[j].z, pv[j].w)));
    mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
    float sum = 0.f, dot = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if (j < nv) {
            const int c0 = 16 * j + 4 * q4;                   // first column of this float4
            float4 e;
            e.x = expf(pv[j].x - mx); e.y = expf(pv[j].y - mx); e.z = expf(pv[j].z - mx); e.w = expf(pv[j].w - mx);
            sum += (c0 < Hl ? e.x : 0.f) + (c0 + 1 < Hl ? e.y : 0.f) + (c0 + 2 < Hl ? e.z : 0.f) + (c0 + 3 < Hl ? e.w : 0.f);   // padded columns are not in the softmax
            pv[j].x = pv[j].x > 0.f ? 1.f : 0.f; pv[j].y = pv[j].y > 0.f ? 1.f : 0.f; pv[j].z = pv[j].z > 0.f ? 1.f : 0.f; pv[j].w = pv[j].w > 0.f ? 1.f : 0.f;   // relu'
            dot += e.x * gv[j].x * hv[j].x + e.y * gv[j].y * hv[j].y + e.z * gv[j].z * hv[j].z + e.w * gv[j].w * hv[j].w;
            hv[j].x *= gv[j].x; hv[j].y *= gv[j].y; hv[j].z *= gv[j].z; hv[j].w *= gv[j].w;                                     // dbeta = dxz * Hx
            gv[j].x *= e.x; gv[j].y *= e.y; gv[j].z *= e.z; gv[j].w *= e.w;                                                     // dxz * exp(p - max)
            // (pv = relu mask, hv = dbeta, gv = dxz * e, and e itself is gv / dxz: keep e in a register set of its own)
            pv[j].x *= e.x; pv[j].y *= e.y; pv[j].z *= e.z; pv[j].w *= e.w;                                                     // mask * e
        }
    sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
    dot += __shfl_xor(dot, 1); dot += __shfl_xor(dot, 2);
    const float inv = 1.0f / sum;
    dot *= inv;                                               // sum_c beta_c dbeta_c
    if (r < R) {
        float4* dq4 = reinterpret_cast<float4*>(dq + (size_t)row * H);
        float4* dh4 = reinterpret_cast<float4*>(dHx_rows + (size_t)row * H);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < nv) {
                // dq = beta * (dbeta - dot) * (p > 0) with beta = e / sum;  dHx_rows += dxz * beta
                float4 o;
                o.x = pv[j].x * inv * (hv[j].x - dot); o.y = pv[j].y * inv * (hv[j].y - dot);
                o.z = pv[j].z * inv * (hv[j].z - dot); o.w = pv[j].w * inv * (hv[j].w - dot);
                dq4[4 * j + q4] = o;
                float4 d = dh4[4 * j + q4];
                d.x += gv[j].x * inv; d.y += gv[j].y * inv; d.z += gv[j].z * inv; d.w += gv[j].w * inv;
                dh4[4 * j + q4] = d;
            }
    }
}
void launch_mask_bwd(const float* p, const float* dxz, const float* Hx, int ldhx, float* dq, float* dHx_rows, int R, int H,
                     int Hl, int K, int mno, hipStream_t s) {
    hipLaunchKernelGGL(k_mask_bwd, dim3((R + 63) / 64), dim3(256), 0, s, p, dxz, Hx, ldhx, dq, dHx_rows, R, H, Hl, K, mno);
}

// ------------------------------------------------------------------------------------------------------------------
// Convolution weight gradients (both directions of the CVAE stack).  With S the tensor on the SMALL pixel grid
// [n, Ps, Ps, Cs] and Lg the one on the LARGE grid [n, Pl, Pl, Cl], related by q = stride*p + k - pad per tap k:
//     dW[tap][cl][cs] = sum_{n, p} Lg[n, q(p, tap), cl] * S[n, p, cs]
// (transposed conv: S = layer input, Lg = d(conv output), weights [tap][co][ci];  forward conv: S = d(conv output),
//  Lg = layer input, weights [tap][ci][co] -- the same [tap][Cl][Cs] orientation.)
// grid = (taps fastest so the 25 workgroups of a slice share its activations in L2, slices); workgroup = 4 waves,
// wave w owns output tiles w, w+4 of the (Cl/32) x (Cs/32) tile grid; contraction chunks of 64 (n,p) pairs via LDS.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_wgrad(ConvWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int LDL = a.Cl + 4, LDS_ = a.Cs + 4;
    float* Ls = smem;                     // [64][LDL]
    float* Ss = smem + 64 * LDL;          // [64][LDS_]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int tap = blockIdx.x, ky = tap / 5, kx = tap - ky * 5;
    const int PP = a.Ps * a.Ps;
    const long Mtot = (long)a.n * PP;
    const long mper = ((Mtot + gridDim.y - 1) / gridDim.y + 63) / 64 * 64;
    const long m_lo = (long)blockIdx.y * mper, m_hi = min(Mtot, m_lo + mper);
    const int tjn = a.Cs / 32, ntiles = (a.Cl / 32) * tjn;
    f32x16 acc[2] = {zero16(), zero16()};
    const int hi = lane >> 5, c = lane & 31;
    for (long m0 = m_lo; m0 < m_hi; m0 += 64) {
        __syncthreads();
        for (int i = tid; i < 64 * (a.Cs / 4); i += 256) {
            const int r = i / (a.Cs / 4), c4 = i - r * (a.Cs / 4);
            const long m = m0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_hi) v = *reinterpret_cast<const float4*>(a.S + (size_t)m * a.Cs + c4 * 4);
            *reinterpret_cast<float4*>(Ss + r * LDS_ + c4 * 4) = v;
        }
        if (a.Cl >= 4) {
            for (int i = tid; i < 64 * (a.Cl / 4); i += 256) {
                const int r = i / (a.Cl / 4), c4 = i - r * (a.Cl / 4);
                const long m = m0 + r;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m < m_hi) {
                    const long n = m / PP; const int p = (int)(m - n * PP);
                    const int qy = a.stride * (p / a.Ps) + ky - a.pad, qx = a.stride * (p % a.Ps) + kx - a.pad;
                    if (qy >= 0 && qy < a.Pl && qx >= 0 && qx < a.Pl)
                        v = *reinterpret_cast<const float4*>(a.Lg + (((size_t)n * a.Pl + qy) * a.Pl + qx) * a.Cl + c4 * 4);
                }
                *reinterpret_cast<float4*>(Ls + r * LDL + c4 * 4) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int tile = w + 4 * tt;
            if (tile >= ntiles) continue;
            const int ti = tile / tjn, tj = tile - ti * tjn;
#pragma unroll
            for (int g = 0; g < 8; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 8 * g + 4 * hi + i;
                    acc[tt] = mfma32(Ls[m * LDL + ti * 32 + c], Ss[m * LDS_ + tj * 32 + c], acc[tt]);
                }
        }
    }
    float* out = a.partial + ((size_t)blockIdx.y * 25 + tap) * a.Cl * a.Cs;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int tile = w + 4 * tt;
        if (tile >= ntiles) continue;
        const int ti = tile / tjn, tj = tile - ti * tjn;
#pragma unroll
        for (int i = 0; i < 16; ++i) out[(size_t)(ti * 32 + acc_row(i)) * a.Cs + tj * 32 + c] = acc[tt][i];
    }
}
void launch_conv_wgrad(const ConvWgradArgs& a, int nslices, float* out, hipStream_t s) {
    if (a.Cl % 4 == 0 && a.Cs % 64 == 0) {
        TnArgs t{};
        t.A = a.Lg; t.lda = 0; t.G = a.S; t.ldg = a.Cs; t.M = (long)a.n * a.Ps * a.Ps; t.Kd = 25 * a.Cl; t.N = a.Cs;
        {
            const int nb = a.Cs <= 64 ? (t.Kd + 255) / 256 : ((t.Kd + 127) / 128) * ((t.N + 127) / 128);
            long sl = 1024 / nb; const long maxsl = (t.M + 255) / 256;
            if (sl > maxsl) sl = maxsl; if (sl < 1) sl = 1;
            while ((size_t)sl * t.Kd * t.N * sizeof(float) > ((size_t)96 << 20) && sl > 1) sl /= 2;
            nslices = (int)sl;
        }
        t.nslices = nslices; t.partial = a.partial;
        const ConvGather cg{a.Cl, a.Pl, a.Ps, a.stride, a.pad};
        t.np = a.np;
        if (a.np == 2) launch_gemm_tn2_split(t, &cg, a.Cs <= 64, s);
        else if (a.Cs <= 64) hipLaunchKernelGGL((k_gemm_tn2<4, 1, true>), dim3((t.Kd + 255) / 256, nslices), dim3(256), 0, s, t, cg);
        else hipLaunchKernelGGL((k_gemm_tn2<2, 2, true>), dim3(((t.Kd + 127) / 128) * ((t.N + 127) / 128), nslices), dim3(256), 0, s, t, cg);
        const int n = 25 * a.Cl * a.Cs;
        reduce_slices(a.partial, nslices, 25 * a.Cl, a.Cs, out, a.Cs, 0, s);
        return;
    }
    const size_t lds = 64 * (a.Cl + 4 + a.Cs + 4) * sizeof(float);
    hipLaunchKernelGGL(k_conv_wgrad, dim3(25, nslices), dim3(256), lds, s, a);
    const int n = 25 * a.Cl * a.Cs;
    reduce_slices(a.partial, nslices, 25 * a.Cl, a.Cs, out, a.Cs, 0, s);
}

// the two single-channel ends of the stack (deconv4: Lg = d(xhat-conv) [n,32,32], S = d3 [n,16,16,32];
// conv1: Lg = vae_in [n,32,32], S = d(conv1 output) [n,16,16,32]):  dW[tap][c] = sum_{n,p} Lg[n, 2p+k-1] * S[n,p,c]
__global__ __launch_bounds__(256) void k_w1ch_grad(const float* __restrict__ Lg, const float* __restrict__ S, int n, float* __restrict__ partial) {
    // the 32x32 image sits in LDS with a one-pixel zero border ([35][40], image at +1,+1): no bounds tests, and the five taps of
    // a row come from two aligned 16-byte reads instead of five scalar ones (the kernel is LDS-instruction bound)
    __shared__ __attribute__((aligned(16))) float lg[36 * 40];
    __shared__ float red[8][25 * 32 + 1];
    const int tid = threadIdx.x, c = tid & 31, pg = tid >> 5;
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    float acc[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) acc[k] = 0.f;
    for (int i = tid; i < 36 * 40; i += 256) lg[i] = 0.f;
    for (int smp = lo; smp < hi; ++smp) {
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) lg[((i >> 5) + 1) * 40 + (i & 31) + 1] = Lg[(size_t)smp * 1024 + i];
        __syncthreads();
        // a pixel PAIR (even px, px + 1) per pass: both read the same 8-float aligned window of a padded row -- taps base + 0..4 for the even
        // pixel, base + 2..6 for the odd one -- so two 16-byte reads serve ten products and nothing is selected per lane
        for (int q = pg; q < 128; q += 8) {
            const int py = q >> 3, px = 2 * (q & 7);
            const float* sp = S + ((size_t)smp * 256 + py * 16 + px) * 32 + c;
            const float s0 = sp[0], s1 = sp[32];
            const int base = 2 * px;                         // (even px: a multiple of 4)
#pragma unroll
            for (int ky = 0; ky < 5; ++ky) {
                const float* row = lg + (2 * py + ky) * 40 + base;
                const float4 v0 = *reinterpret_cast<const float4*>(row), v1 = *reinterpret_cast<const float4*>(row + 4);
                acc[ky * 5 + 0] = fmaf(v0.x, s0, acc[ky * 5 + 0]); acc[ky * 5 + 0] = fmaf(v0.z, s1, acc[ky * 5 + 0]);
                acc[ky * 5 + 1] = fmaf(v0.y, s0, acc[ky * 5 + 1]); acc[ky * 5 + 1] = fmaf(v0.w, s1, acc[ky * 5 + 1]);
                acc[ky * 5 + 2] = fmaf(v0.z, s0, acc[ky * 5 + 2]); acc[ky * 5 + 2] = fmaf(v1.x, s1, acc[ky * 5 + 2]);
                acc[ky * 5 + 3] = fmaf(v0.w, s0, acc[ky * 5 + 3]); acc[ky * 5 + 3] = fmaf(v1.y, s1, acc[ky * 5 + 3]);
                acc[ky * 5 + 4] = fmaf(v1.x, s0, acc[ky * 5 + 4]); acc[ky * 5 + 4] = fmaf(v1.z, s1, acc[ky * 5 + 4]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 25; ++k) red[pg][k * 32 + c] = acc[k];
    __syncthreads();
    for (int i = tid; i < 800; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) s += red[g][i];
        partial[(size_t)blockIdx.x * 800 + i] = s;
    }
}
void launch_w1ch_grad(const float* Lg, const float* S, int n, int nslices, float* partial, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_w1ch_grad, dim3(nslices), dim3(256), 0, s, Lg, S, n, partial);
    reduce_slices(partial, nslices, 25, 32, out, 32, 0, s);
}

// ---- reparameterisation + KLD backward:  dparams[a] = (dmu | dlogsig2) ---------------------------------------------
//   z = mu + sqrt(exp(ls)) eps   =>  dmu = sum_k dz ; dls = sum_k dz eps 0.5 sqrt(exp(ls))
//   kld = -0.5 sum(1 + ls - mu^2 - exp(ls)), weight valid/N  =>  dmu += w mu ; dls += w (-0.5)(1 - exp(ls))
__global__ void k_reparam_bwd(const float* __restrict__ dz, const float* __restrict__ eps, const float* __restrict__ params,
                              const uint8_t* __restrict__ valid, const float* __restrict__ nvalid, float* __restrict__ dparams,
                              int n_scenes, int mno, int K, int L, const int32_t* __restrict__ inv, int P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int A = n_scenes * mno;
    if (i >= A * L) return;
    const int a = i / L, l = i - a * L;
    const int sc = a / mno, slot = a - sc * mno;
    const float mu = params[(size_t)a * 2 * L + l], ls = params[(size_t)a * 2 * L + L + l];
    const float sd = sqrtf(expf(ls));
    float dmu = 0.f, dls = 0.f;
    // inv != nullptr: dz lives in the compact row order of kernels_compact.hip (r' = k*P + inv[a]; absent agents have no rows and no gradient)
    const int ip = inv ? inv[a] : 0;
    if (ip >= 0)
        for (int k = 0; k < K; ++k) {
            const size_t r = ((size_t)sc * K + k) * mno + slot;
            const float g = dz[(inv ? (size_t)k * P + ip : r) * L + l];
            dmu += g;
            dls += g * eps[r * L + l];
        }
    dls *= 0.5f * sd;
    const float wv = valid[a] ? 1.0f / nvalid[0] : 0.f;
    dparams[(size_t)a * 2 * L + l] = dmu + wv * mu;
    dparams[(size_t)a * 2 * L + L + l] = dls + wv * (-0.5f) * (1.0f - expf(ls));
}
void launch_reparam_bwd(const float* dz, const float* eps, const float* params, const uint8_t* valid, const float* nvalid,
                        float* dparams, int n_scenes, int mno, int K, int L, hipStream_t s, const int32_t* inv, int P) {
    const int n = n_scenes * mno * L;
    hipLaunchKernelGGL(k_reparam_bwd, dim3((n + 255) / 256), dim3(256), 0, s, dz, eps, params, valid, nvalid, dparams, n_scenes, mno, K, L, inv, P);
}

// ---- dHx[a] (+)= sum_k dHx_rows[(scene,k,slot)] --------------------------------------------------------------------------
__global__ void k_rows_to_agents(const float* __restrict__ rows, float* __restrict__ out, int ldo, int n_scenes, int mno, int K, int H, int gpt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_scenes * mno * H) return;
    const int a = i / H, c = i - a * H;
    const int sc = a / mno, slot = a - sc * mno;
    float s = 0.f;
    for (int k = 0; k < K; ++k) {
        size_t r = ((size_t)sc * K + k) * mno + slot;
        if (gpt) { const int G = sc * K + k; r = (size_t)(G / gpt) * 32 + (size_t)(G % gpt) * mno + slot; }      // padded tiles (kernels.h: IocArgs.gpt)
        s += rows[r * H + c];
    }
    out[(size_t)a * ldo + c] += s;
}
void launch_rows_to_agents(const float* rows, float* out, int ldo, int n_scenes, int mno, int K, int H, hipStream_t s, int gpt) {
    const int n = n_scenes * mno * H;
    hipLaunchKernelGGL(k_rows_to_agents, dim3((n + 255) / 256), dim3(256), 0, s, rows, out, ldo, n_scenes, mno, K, H, gpt);
}

// ------------------------------------------------------------------------------------------------------------------
// IOC loss gradients w.r.t. the scores:  CE(P, softmax_k(score)) with P = softmax_k(-max_t ||Y_gt - Y0_k||)
//   dscore_k = valid / N * (softmax_k(score) - P_k);  dscoreT[r, t] = dscore[r] (broadcast used by the tn reductions)
// ------------------------------------------------------------------------------------------------------------------
// (one thread per agent: the form for more than 64 samples per agent)
__global__ void k_score_grad_serial(const float* __restrict__ Y0, const float* __restrict__ fut, const float* __restrict__ score,
                             const uint8_t* __restrict__ valid, const float* __restrict__ nvalid, float* __restrict__ dscore,
                             float* __restrict__ dscoreT, int n_scenes, int mno, int K, int T, float sx, float sy) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_scenes * mno) return;
    const int sc = a / mno, slot = a - sc * mno;
    float m1 = -3.0e38f, m2 = -3.0e38f;
    for (int k = 0; k < K; ++k) {                       // pass 1: maxima of the two logit sets
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        float dm = 0.f;
        for (int t = 0; t < T; ++t) {
            const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
            if (f[0] == 0.f) continue;                   // frames without the object carry no ground truth
            const float dx = Y0[(r * T + t) * 2] - __fmul_rn(f[1], sx), dy = Y0[(r * T + t) * 2 + 1] - __fmul_rn(f[2], sy);
            dm = fmaxf(dm, sqrtf(dx * dx + dy * dy));
        }
        dscore[r] = dm;                                  // stash d_max
        m1 = fmaxf(m1, -dm); m2 = fmaxf(m2, score[r]);
    }
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < K; ++k) {
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        s1 += expf(-dscore[r] - m1); s2 += expf(score[r] - m2);
    }
    const float wv = valid[a] ? 1.0f / nvalid[0] : 0.f;
    for (int k = 0; k < K; ++k) {
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        const float g = wv * (expf(score[r] - m2) / s2 - expf(-dscore[r] - m1) / s1);
        dscore[r] = g;
        for (int t = 0; t < T; ++t) dscoreT[r * T + t] = g;
    }
}
// one thread per (agent, sample k): 32 k-lanes x 4 agents per workgroup; the softmax sums run over k in index order in every lane, so the
// result does not depend on the lane that forms it.  (One thread per agent walked K x T positions alone: 0.33 ms on 64 workgroups.)
__global__ __launch_bounds__(128) void k_score_grad(const float* __restrict__ Y0, const float* __restrict__ fut, const float* __restrict__ score,
                             const uint8_t* __restrict__ valid, const float* __restrict__ nvalid, float* __restrict__ dscore,
                             float* __restrict__ dscoreT, int n_scenes, int mno, int K, int T, float sx, float sy) {
    __shared__ float dmx[4][64], scr[4][64];
    const int ai = threadIdx.x >> 5, kl = threadIdx.x & 31;
    const int a = blockIdx.x * 4 + ai;
    const bool live = a < n_scenes * mno;
    const int sc = live ? a / mno : 0, slot = live ? a - sc * mno : 0;
    for (int k = kl; k < K && live; k += 32) {             // this lane's samples: d_max and the score
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        float dm = 0.f;
        for (int t = 0; t < T; ++t) {
            const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
            if (f[0] == 0.f) continue;                   // frames without the object carry no ground truth
            const float dx = Y0[(r * T + t) * 2] - __fmul_rn(f[1], sx), dy = Y0[(r * T + t) * 2 + 1] - __fmul_rn(f[2], sy);
            dm = fmaxf(dm, sqrtf(dx * dx + dy * dy));
        }
        dmx[ai][k] = dm; scr[ai][k] = score[r];
    }
    __syncthreads();
    if (!live) return;
    float m1 = -3.0e38f, m2 = -3.0e38f;
    for (int k = 0; k < K; ++k) { m1 = fmaxf(m1, -dmx[ai][k]); m2 = fmaxf(m2, scr[ai][k]); }
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < K; ++k) { s1 += expf(-dmx[ai][k] - m1); s2 += expf(scr[ai][k] - m2); }
    const float wv = valid[a] ? 1.0f / nvalid[0] : 0.f;
    for (int k = kl; k < K; k += 32) {
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        const float g = wv * (expf(scr[ai][k] - m2) / s2 - expf(-dmx[ai][k] - m1) / s1);
        dscore[r] = g;
        for (int t = 0; t < T; ++t) dscoreT[r * T + t] = g;
    }
}
void launch_score_grad(const float* Y0, const float* fut, const float* score, const uint8_t* valid, const float* nvalid,
                       float* dscore, float* dscoreT, int n_scenes, int mno, int K, int T, float sx, float sy, hipStream_t s) {
    const int A = n_scenes * mno;
    if (K > 64) {
        hipLaunchKernelGGL(k_score_grad_serial, dim3((A + 63) / 64), dim3(64), 0, s, Y0, fut, score, valid, nvalid, dscore, dscoreT,
                           n_scenes, mno, K, T, sx, sy);
        return;
    }
    hipLaunchKernelGGL(k_score_grad, dim3((A + 3) / 4), dim3(128), 0, s, Y0, fut, score, valid, nvalid, dscore, dscoreT,
                       n_scenes, mno, K, T, sx, sy);
}

// ------------------------------------------------------------------------------------------------------------------
// IOC BPTT.  Tile = 32 rows = whole (scene,k) groups (mno <= 32), wave cb owns hidden columns [32cb, 32cb+32).
// Reverse step t (forward: x = [e_v | e_s | e_r], e_r = relu(sum_b pool_b(h_{t-1}) W_b + b_s), GRU, score += h.w_s):
//   dh_t += dscore w_s;  GRU cell backward as in k_decoder_bwd, plus
//   de_r  = da_c Wc[e_r rows]^T + [da_r|da_u] Wg[e_r rows]^T ;  dpre_r = de_r (e_r > 0)
//   de_v  likewise (16 columns, wave 0);                         dpre_v = de_v (e_v > 0)
//   per bin b: dpool_b = dpre_r W_b^T (MFMA) -> LDS; every row j gathers sum_{i: j in bin b of i} dpool_b[i] into
//   registers (observer bit-masks), which after the 16 bins is added to dh_{t-1}: the pooling transpose without atomics.
//   pooled_b (rebuilt like the forward) is streamed to HBM for the social-fc weight gradient.
// ------------------------------------------------------------------------------------------------------------------
#ifndef IOC_BWD_OCC
#define IOC_BWD_OCC 2
#endif
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int TM> struct BwdMask { typedef unsigned long long type; };
template <> struct BwdMask<32> { typedef unsigned type; };
__device__ __forceinline__ int ffsm(unsigned m) { return __ffs((int)m); }
__device__ __forceinline__ int ffsm(unsigned long long m) { return __ffsll((long long)m); }
// TM = 32: whole groups of up to 32 agents per tile, 4 waves at H = 128, two workgroups per CU.  TM = 64: groups of 64 agents
// (one per tile), wave (mt, cb) owns rows [32mt, 32mt+32) x columns [32cb, 32cb+32), one workgroup per CU (144 KB of LDS).
// The 32 x 32 contractions run TRANSPOSED (mma SWAP, as in k_decoder_bwd): a lane holds ONE tile row and runs of four consecutive
// hidden columns -- 16-byte stream and LDS accesses and a quarter of the stream offsets (the row-major kernel of rounds 1-3 spilled 94
// dwords at H = 128 and took 7.4 ms longer per 81 920-row step); bias column sums by a butterfly over the rows (colsum16).  The packed
// 16 x 16 dpool contraction (CPB) writes its own LDS rows and is unaffected.
template <int H, int EV, int C, int TM, bool PAD = false>      // PAD: padded tiles (IocBwdArgs.gpt), 32-row form; a template parameter so that the packed form is unchanged
__global__ __launch_bounds__((H / 32) * (TM / 32) * 64, ((H / 32) * (TM / 32) <= 4) ? IOC_BWD_OCC : ((H / 32) * (TM / 32) <= 8 ? 2 : 1)) void k_ioc_bwd(IocBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT = H / 32, NTHR = NT * (TM / 32) * 64, TPR = NTHR / TM, NCH = H / (4 * TPR);
    typedef typename BwdMask<TM>::type mask_t;
    constexpr int E = EV + C + H, LD1 = H + 4, LD2 = 2 * H + 4, GH = H / 8, G2 = 2 * H / 8;
    const int B = a.G * a.G;
    const int KR = (2 * a.T + 7) / 8 * 8, LDR = KR + 4;                   // regression-head operand width
    // LDS (72 KB at H = 128, so two workgroups share a CU): three operand tiles, each reused once its MFMAs are done
    float* A1 = smem;                         // [32][LD1]  da_c;  then h_{t-1} (pooled rebuild);  then the neighbour gradient
    float* A2 = A1 + TM * LD1;                // [32][LD2]  da_r | da_u;  then dpool_b, double buffered
    float* A3 = A2 + 2 * TM * LD1;            // [32][LD1]  dpre_r  (2 LD1 > LD2: the two dpool tiles are the larger tenant)
    float* DP = A2, *HP = A1, *NB = A1;
    mask_t* masks = reinterpret_cast<mask_t*>(A3 + TM * LD1);         // [TM][B] neighbours of i in bin b (bit = slot)
    mask_t* obs = masks + TM * B;                                     // [TM][B] observers of j in bin b
    float* pc = reinterpret_cast<float*>(obs + TM * B);   // [32][2]
    float* dsc = pc + TM * 2;                 // [32]
    float* wsc = dsc + TM;                    // [H]
    unsigned char* vld = reinterpret_cast<unsigned char*>(wsc + H);   // [32]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + TM);            // [2] bins that hold a neighbour anywhere in the tile
    unsigned* rowbits = occ + 2;                                      // CPB: [B] rows of the tile with a neighbour in bin b
    unsigned char* rowlist = reinterpret_cast<unsigned char*>(rowbits + 36);   // CPB: per wave 32 bytes, packed row -> tile row
    // CPB (32-row tiles): dpool_b is only ever gathered from rows that HAVE a neighbour in bin b -- contract those rows only,
    // as 16-row v_mfma_f32_16x16x4_f32 tiles whose A rows are fetched through the row list (no packed copy needed)
    constexpr bool CPB = (TM == 32) && (H <= 128);
    float* DR = A2;                           // [32][LDR] regression-head operand (prologue only; 2T <= 2H assumed)

    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int cb = w % NT, mt = w / NT;
    const int row0 = blockIdx.x * TM;
    const int lr = lane & 31, hi = lane >> 5;
    const int rl = mt * 32 + lr;                           // this lane's tile row; its column runs start at c0 + 8 q
    const int c0 = cb * 32 + 4 * hi;
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int my_row = min(row0 + r8, a.R - 1);
    const int grp_base = (r8 / a.mno) * a.mno, my_slot = r8 - grp_base;
    const int gpt = PAD ? a.gpt : 0;                       // padded tiles (kernels.h: IocArgs.gpt): dead rows behind the tile's gpt groups
    const bool dead_row = gpt && (r8 / a.mno >= gpt || (int)blockIdx.x * gpt + r8 / a.mno >= a.ngrp);
    const int n_nb = dead_row ? 0 : a.mno;
    const float* a1_lane = A1 + (mt * 32 + (lane & 31)) * LD1 + 4 * (lane >> 5);
    const float* a2_lane = A2 + (mt * 32 + (lane & 31)) * LD2 + 4 * (lane >> 5);
    const float* a3_lane = A3 + (mt * 32 + (lane & 31)) * LD1 + 4 * (lane >> 5);
    // saved activations / gradient streams are addressed as (uniform tile base) + (32-bit offset inside the tile)
    const int nloc = min(TM, a.R - row0);
    const bool rok = rl < nloc;                            // rows past R read the tile's last row; nothing of theirs is stored or summed
    const int rcl = min(rl, nloc - 1);
    auto colsum16 = [&](const float (&x)[16]) {            // butterfly reduce-scatter over the 16 lanes sharing bits 0..3 (see k_ioc_bwd_x3)
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = x[i];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int n = 8 >> st, m = 1 << st;
            const bool up = (lane >> st) & 1;
#pragma unroll
            for (int j = 0; j < n; ++j) {
                const float send = up ? v[j] : v[j + n], keep = up ? v[j + n] : v[j];
                v[j] = keep + __shfl_xor(send, m);
            }
        }
        return v[0];
    };
    // this wave's 32 x 32 block of an fp32 LDS tile (rows 32 mt .., columns col_t + 32 cb ..) leaves as whole 128-byte lines of a stream (see
    // k_ioc_bwd_x3: from the accumulator layout a store instruction covers 32 bytes of 32 different rows, and the partial lines cost 1.6x the bytes)
    auto flush32 = [&](const float* tile, int ld, int col_t, float* out, int wout, int col_out, int t) {
        int ln;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));      // opaque per call: the addresses below are re-formed, not hoisted out of the time loop and spilled
        const int ch = ln & 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = mt * 32 + (ln >> 3) + 8 * k;
            const float4 v = *reinterpret_cast<const float4*>(tile + r * ld + col_t + cb * 32 + 4 * ch);
            if (r < nloc) *reinterpret_cast<float4*>(out + (size_t)(r * a.T + t) * wout + col_out + cb * 32 + 4 * ch) = v;
        }
    };
    const size_t tb = (size_t)row0 * a.T;
    const float* svu = a.sv_u + tb * H; const float* svc = a.sv_c + tb * H; const float* svr = a.sv_r + tb * H;
    const float* svx = a.sv_x + tb * E;
    float* o_dac = a.dac + tb * H; float* o_rh = a.rh + tb * H; float* o_hp = a.hprev + tb * H; float* o_dag = a.dag + tb * 2 * H;
    float* o_dpr = a.dpre_r + tb * H; float* o_dpv = a.dpre_v + tb * EV;

    for (int i = tid; i < H; i += NTHR) wsc[i] = a.w_score[i];
    if (tid < TM) {
        const int row = min(row0 + tid, a.R - 1);
        { const int ag = ioc_agent_of_row(row, a.K, a.mno, gpt, a.ngrp); vld[tid] = ag >= 0 ? a.valid[ag] : 0; }
        dsc[tid] = (row0 + tid < a.R) ? a.dscore[row] : 0.f;
    }
    for (int i = tid; i < TM * KR; i += NTHR) {
        const int r = i / KR, c = i - r * KR;
        DR[r * LDR + c] = (c < 2 * a.T && row0 + r < a.R) ? a.dYr[(size_t)(row0 + r) * 2 * a.T + c] : 0.f;
    }
    __syncthreads();
    int rl_s = rl, hi_s = hi, c0_s = c0;
    float cs_r = 0.f, cs_u = 0.f, cs_c = 0.f, cs_p = 0.f;   // this lane's column sums of da_r, da_u, da_c, dpre_r over its rows and all steps
    f32x16 dh = zero16();
    mma1t(dh, DR + rl * LDR + 4 * hi, a.WrT + ((size_t)cb * (KR / 8)) * 64 + lane, KR / 8);

    for (int t = a.T - 1; t >= 0; --t) {
        int rc = rcl;
        asm volatile("v_mov_b32 %0, %1" : "=v"(rc) : "v"(rcl));     // opaque per step: stream offsets are re-formed, not hoisted and spilled
        {   // the lane's coordinates as step-local values off an opaque lane id: the addresses built on them are re-formed where used (k_ioc_bwd_x3)
            int ls;
            asm volatile("v_mov_b32 %0, %1" : "=v"(ls) : "v"(lane));
            hi_s = ls >> 5; rl_s = mt * 32 + (ls & 31); c0_s = cb * 32 + 4 * hi_s;
        }
        const unsigned rt = (unsigned)(rc * a.T + t);
        __syncthreads();
        // ---- P0: positions, cleared masks, h_{t-1} tile ----
        if (tid < TM) {
            const int row = min(row0 + tid, a.R - 1);
            const float2 y = *reinterpret_cast<const float2*>(a.Y0 + ((size_t)row * a.T + t) * 2);
            pc[tid * 2] = y.x; pc[tid * 2 + 1] = y.y;
            float2 pv;
            if (t > 0) pv = *reinterpret_cast<const float2*>(a.Y0 + ((size_t)row * a.T + t - 1) * 2);
            else { const int ag = ioc_agent_of_row(row, a.K, a.mno, gpt, a.ngrp); pv = ag >= 0 ? make_float2(a.p_last[(size_t)ag * 2], a.p_last[(size_t)ag * 2 + 1]) : make_float2(0.f, 0.f); }
            if (row0 + tid < a.R) { a.vel[((size_t)row * a.T + t) * 2] = y.x - pv.x; a.vel[((size_t)row * a.T + t) * 2 + 1] = y.y - pv.y; }
        }
        for (int i = tid; i < 2 * TM * B; i += NTHR) masks[i] = 0;             // masks and obs are contiguous
        if (tid < 2) occ[tid] = 0;
        if (CPB && tid < B) rowbits[tid] = 0;
        auto load_hprev = [&](bool keep) {                                     // h_{t-1} tile -> A1's space (keep: and, as whole rows, to the weight gradient's operand stream)
            for (int i = tid; i < TM * (H >> 2); i += NTHR) {
                const int r = i / (H >> 2), c4 = i - r * (H >> 2);
                const int row = min(row0 + r, a.R - 1);
                const int ag0 = (t > 0) ? 0 : ioc_agent_of_row(row, a.K, a.mno, gpt, a.ngrp);
                const float* src = (t > 0) ? a.sv_h + ((size_t)row * a.T + t - 1) * H
                                           : a.Hx + (size_t)max(ag0, 0) * a.ldhx;
                const float4 hv = ag0 >= 0 ? *reinterpret_cast<const float4*>(src + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(HP + r * LD1 + c4 * 4) = hv;
                if (keep && r < nloc) *reinterpret_cast<float4*>(o_hp + (size_t)(r * a.T + t) * H + c4 * 4) = hv;
            }
        };
        load_hprev(true);                                // part 1 reads each element, then overwrites it with da_c
        __syncthreads();
        // ---- P1: neighbour / observer masks ----
        {
            const float px = pc[r8 * 2], py = pc[r8 * 2 + 1];
            float nbw, nbh;
            nb_opaque(a.nb_w, a.nb_h, nbw, nbh);
            const unsigned long long oc = nb_search<4>(pc, vld, grp_base, n_nb, q8, TPR, my_slot, px, py, nbw, nbh, a.G, a.bin_tab,
                                                      [&](int j, int b) {
                                                          atomicOr(&masks[r8 * B + b], (mask_t)1 << j);
                                                          atomicOr(&obs[(grp_base + j) * B + b], (mask_t)1 << my_slot);
                                                          if (CPB) atomicOr(&rowbits[b], 1u << r8);
                                                      });
            nb_publish_occ(oc, occ, B);
        }
        // ---- GRU cell backward, part 1 ----
        if (a.pool_flags && q8 == 0 && row0 + r8 < a.R) {      // which bins of this (row, t) hold a neighbour: the weight-gradient
            unsigned long long fl = 0ull;                       // GEMM skips the all-zero blocks of the pooled operand
            for (int b = 0; b < B; ++b) fl |= (unsigned long long)(masks[r8 * B + b] != 0) << b;
            a.pool_flags[(size_t)my_row * a.T + t] = fl;
        }
        f32x16 dhp, rr, hp;                          // what part 2 needs: r and h_{t-1} (everything else is stored at once)
        {
            float sc_c[16], sc_u[16];
            const float dscv = dsc[rl_s];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned ix = rt * H + c0_s + 8 * q;
                const float4 u4 = *reinterpret_cast<const float4*>(svu + ix), cc4 = *reinterpret_cast<const float4*>(svc + ix);
                const float4 r4 = *reinterpret_cast<const float4*>(svr + ix);
                const float4 h4 = *reinterpret_cast<const float4*>(A1 + rl_s * LD1 + c0_s + 8 * q);
                const float4 w4 = *reinterpret_cast<const float4*>(wsc + c0_s + 8 * q);
                float dacv[4], dauv[4], rhv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    const float u = f4get(u4, e), c = f4get(cc4, e), r = f4get(r4, e), hprev = f4get(h4, e);
                    const float dht = dh[i] + dscv * f4get(w4, e);
                    const float dau = dht * (hprev - c) * u * (1.0f - u);
                    const float dc = dht * (1.0f - u);
                    dhp[i] = dht * u;
                    const float dac = dc * (1.0f - c * c);
                    dacv[e] = dac; dauv[e] = dau; rhv[e] = r * hprev;
                    sc_c[i] = rok ? dac : 0.f; sc_u[i] = rok ? dau : 0.f;
                    rr[i] = r; hp[i] = hprev;
                }
                const float4 dac4 = make_float4(dacv[0], dacv[1], dacv[2], dacv[3]), dau4 = make_float4(dauv[0], dauv[1], dauv[2], dauv[3]);
                *reinterpret_cast<float4*>(A1 + rl_s * LD1 + c0_s + 8 * q) = dac4;
                *reinterpret_cast<float4*>(A2 + rl_s * LD2 + H + c0_s + 8 * q) = dau4;             // (the dpool tiles that share A2 were last read before the step's barrier)
                *reinterpret_cast<float4*>(A3 + rl_s * LD1 + c0_s + 8 * q) = make_float4(rhv[0], rhv[1], rhv[2], rhv[3]);    // r h_{t-1} borrows dpre_r's tile (written two phases on) on its way out
            }
            flush32(A1, LD1, 0, o_dac, H, 0, t);
            flush32(A2, LD2, H, o_dag, 2 * H, H, t);
            flush32(A3, LD1, 0, o_rh, H, 0, t);
            cs_c += colsum16(sc_c); cs_u += colsum16(sc_u);
        }
        __syncthreads();
        f32x16 drh = zero16(), der = zero16(), dev = zero16();
        mma1t(drh, a1_lane, a.WcT_h + ((size_t)cb * GH) * 64 + lane, GH);
        mma1t(der, a1_lane, a.WcT_er + ((size_t)cb * GH) * 64 + lane, GH);
        if (cb == 0) mma1t(dev, a1_lane, a.WcT_ev + lane, GH);
        {
            float sc_r[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float darv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    const float dr = drh[i] * hp[i];
                    dhp[i] += drh[i] * rr[i];
                    const float dar = dr * rr[i] * (1.0f - rr[i]);
                    darv[e] = dar;
                    sc_r[i] = rok ? dar : 0.f;
                }
                const float4 dar4 = make_float4(darv[0], darv[1], darv[2], darv[3]);
                *reinterpret_cast<float4*>(A2 + rl_s * LD2 + c0_s + 8 * q) = dar4;
            }
            flush32(A2, LD2, 0, o_dag, 2 * H, 0, t);
            cs_r += colsum16(sc_r);
        }
        __syncthreads();
        // da_c is consumed: its tile now takes h_{t-1} for the pooled rebuild (visible after the next barrier)
        load_hprev(false);
        {
            f32x16 dhg = zero16();
            mma1t(dhg, a2_lane, a.WgT_h + ((size_t)cb * G2) * 64 + lane, G2);
            mma1t(der, a2_lane, a.WgT_er + ((size_t)cb * G2) * 64 + lane, G2);
            if (cb == 0) mma1t(dev, a2_lane, a.WgT_ev + lane, G2);
            float sc_p[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 er4 = *reinterpret_cast<const float4*>(svx + (size_t)rt * E + EV + C + c0_s + 8 * q);
                float dprv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    dhp[i] += dhg[i];
                    const float dpr = f4get(er4, e) > 0.f ? der[i] : 0.f;
                    dprv[e] = dpr;
                    sc_p[i] = rok ? dpr : 0.f;
                }
                const float4 dpr4 = make_float4(dprv[0], dprv[1], dprv[2], dprv[3]);
                *reinterpret_cast<float4*>(A3 + rl_s * LD1 + c0_s + 8 * q) = dpr4;
                if (rok) {
                    if (cb == 0 && q < EV / 8) {                  // the e_v tile: columns 4 hi + 8 q + e < EV
                        const float4 ev4 = *reinterpret_cast<const float4*>(svx + (size_t)rt * E + 4 * hi_s + 8 * q);
                        *reinterpret_cast<float4*>(o_dpv + (size_t)rt * EV + 4 * hi_s + 8 * q) =
                            make_float4(ev4.x > 0.f ? dev[4 * q] : 0.f, ev4.y > 0.f ? dev[4 * q + 1] : 0.f, ev4.z > 0.f ? dev[4 * q + 2] : 0.f, ev4.w > 0.f ? dev[4 * q + 3] : 0.f);
                    }
                }
            }
            flush32(A3, LD1, 0, o_dpr, H, 0, t);
            cs_p += colsum16(sc_p);
        }
        __syncthreads();
        // ---- social pooling backward ----
        float4 nb[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) nb[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        // bins without a neighbour anywhere in the tile have dpool_b gathered by nobody: only their (zero) pooled rows are written
        unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
        om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
        int buf = 0;
        for (int b = 0; b < B; ++b) {
            const bool live = (om >> b) & 1ull;
            {   // pooled_b[i] = sum_{j in bin b of i} h_{t-1}[j]  -> HBM (operand of the social-fc weight gradient)
                float4 s[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                mask_t m2 = masks[r8 * B + b];
                while (m2) {
                    const int j = ffsm(m2) - 1;
                    m2 &= m2 - 1;
                    const float* src = HP + (grp_base + j) * LD1 + q8 * 4;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float4 v = *reinterpret_cast<const float4*>(src + c * 4 * TPR);
                        s[c].x += v.x; s[c].y += v.y; s[c].z += v.z; s[c].w += v.w;
                    }
                }
                if (row0 + r8 < a.R && (!a.pool_flags || masks[r8 * B + b] != 0)) {     // (flagged-empty blocks are never read)
                    float* dst = a.pooled + (((size_t)my_row * a.T + t) * B + b) * H + q8 * 4;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(dst + c * 4 * TPR) = s[c];
                }
            }
            if (!live) continue;
            float* dp = DP + buf * TM * LD1;
            buf ^= 1;
            unsigned rb = 0;
            if constexpr (CPB) {
                constexpr int T16 = H / 16;
                rb = (unsigned)__builtin_amdgcn_readfirstlane((int)rowbits[b]);
                const int n = __popc(rb);
                unsigned char* wrl = rowlist + w * 32;
                if (lane < TM && ((rb >> lane) & 1u)) wrl[__popc(rb & ((1u << lane) - 1u))] = (unsigned char)lane;
                const float4* w0 = a.WsT_c + ((size_t)(b * T16 + 2 * cb) * T16) * 64 + lane;
                for (int c16 = 0; c16 * 16 < n; ++c16) {
                    const int sl = 16 * c16 + (lane & 15);
                    const float* ap = A3 + (sl < n ? (int)wrl[sl] : 0) * LD1 + 4 * (lane >> 4);   // rows past n: any finite row, result unused
                    f32x4v a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < T16; ++g) {
                        const float4 av = *reinterpret_cast<const float4*>(ap + 16 * g);
                        const float4 b0 = w0[g * 64], b1 = w0[(T16 + g) * 64];
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b0.x, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b1.x, a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b0.y, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b1.y, a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b0.z, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b1.z, a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b0.w, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b1.w, a1, 0, 0, 0);
                    }
                    // packed rows of dpool_b: element i of a lane = packed row 16 c16 + 4 (lane>>4) + i, column (lane&15) of each tile
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float* d = dp + (16 * c16 + 4 * (lane >> 4) + i) * LD1 + cb * 32 + (lane & 15);
                        d[0] = a0[i]; d[16] = a1[i];
                    }
                }
            } else {
                f32x16 dpl = zero16();
                mma1t(dpl, a3_lane, a.WsT + ((size_t)(b * NT + cb) * GH) * 64 + lane, GH);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(dp + rl_s * LD1 + c0_s + 8 * q) = make_float4(dpl[4 * q], dpl[4 * q + 1], dpl[4 * q + 2], dpl[4 * q + 3]);
            }
            __syncthreads();
            mask_t m2 = obs[r8 * B + b];
            while (m2) {
                const int i2 = ffsm(m2) - 1;
                m2 &= m2 - 1;
                // dense form: row of agent i2; packed form: its rank among the rows that have a neighbour in this bin
                const int srow = CPB ? __popc(rb & ((1u << (grp_base + i2)) - 1u)) : grp_base + i2;
                const float* src = dp + srow * LD1 + q8 * 4;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float4 v = *reinterpret_cast<const float4*>(src + c * 4 * TPR);
                    nb[c].x += v.x; nb[c].y += v.y; nb[c].z += v.z; nb[c].w += v.w;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(NB + r8 * LD1 + q8 * 4 + c * 4 * TPR) = nb[c];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 n4 = *reinterpret_cast<const float4*>(NB + rl_s * LD1 + c0_s + 8 * q);
            dh[4 * q] = dhp[4 * q] + n4.x; dh[4 * q + 1] = dhp[4 * q + 1] + n4.y; dh[4 * q + 2] = dhp[4 * q + 2] + n4.z; dh[4 * q + 3] = dhp[4 * q + 3] + n4.w;
        }
    }
    if (a.bias_part) {                                     // one part per 32-row block: [da_r | da_u | da_c | dpre_r] column sums
        // after the butterfly a lane holds the sum over ITS 16-lane row group of one accumulator element; the other row group is lane ^ 16
        cs_r += __shfl_xor(cs_r, 16); cs_u += __shfl_xor(cs_u, 16); cs_c += __shfl_xor(cs_c, 16); cs_p += __shfl_xor(cs_p, 16);
        if (!(lane & 16)) {
            const int el = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);     // accumulator element this lane ended up with
            const int colb = c0 + 8 * (el >> 2) + (el & 3);
            float* part = a.bias_part + ((size_t)blockIdx.x * (TM / 32) + mt) * 4 * H;
            part[colb] = cs_r; part[H + colb] = cs_u; part[2 * H + colb] = cs_c; part[3 * H + colb] = cs_p;
        }
    }
    if (rok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4* d4 = reinterpret_cast<float4*>(a.dHx_rows + (size_t)(row0 + rl) * H + c0 + 8 * q);
            float4 v = *d4;
            v.x += dh[4 * q]; v.y += dh[4 * q + 1]; v.z += dh[4 * q + 2]; v.w += dh[4 * q + 3];
            *d4 = v;
        }
    }
}
static size_t ioc_bwd_lds(const IocBwdArgs& a, int TM) {
    const int H = a.H, LD1 = H + 4, B = a.G * a.G;
    size_t f = (size_t)TM * LD1 * 4 + (size_t)TM * B * (TM == 32 ? 2 : 4) + TM * 2 + TM + H;
    return f * sizeof(float) + TM + 64 + 512;               // + occupancy words, row bitmaps / row lists of the packed dpool
}
template <int H, int TM>
static void launch_ioc_bwd_t(const IocBwdArgs& a, hipStream_t s) {
    allow_big_lds(k_ioc_bwd<H, 16, 32, TM>);
    hipLaunchKernelGGL((k_ioc_bwd<H, 16, 32, TM>), dim3((a.R + TM - 1) / TM), dim3((H / 32) * (TM / 32) * 64), ioc_bwd_lds(a, TM), s, a);
}
// groups of up to 32 agents: 32-row tiles; 64 agents per scene (H <= 128): one 64-row tile per group
void launch_ioc_bwd(const IocBwdArgs& a, hipStream_t s) {
    if (a.mno > 32) {
        if (a.H == 128) launch_ioc_bwd_t<128, 64>(a, s); else launch_ioc_bwd_t<64, 64>(a, s);
        return;
    }
    if (a.gpt > 0 && a.H <= 128) {                          // padded tiles (slot classes that do not divide 32)
        if (a.H == 128) { allow_big_lds(k_ioc_bwd<128, 16, 32, 32, true>); hipLaunchKernelGGL((k_ioc_bwd<128, 16, 32, 32, true>), dim3((a.R + 31) / 32), dim3(256), ioc_bwd_lds(a, 32), s, a); }
        else { allow_big_lds(k_ioc_bwd<64, 16, 32, 32, true>); hipLaunchKernelGGL((k_ioc_bwd<64, 16, 32, 32, true>), dim3((a.R + 31) / 32), dim3(128), ioc_bwd_lds(a, 32), s, a); }
        return;
    }
    if (a.H == 256) launch_ioc_bwd_t<256, 32>(a, s);
    else if (a.H == 128) launch_ioc_bwd_t<128, 32>(a, s);
    else launch_ioc_bwd_t<64, 32>(a, s);
}
