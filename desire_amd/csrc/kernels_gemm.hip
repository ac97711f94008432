// kernels_gemm.hip -- row-tiled fp32-MFMA GEMMs with fused epilogues, reparameterisation and
// the mask fc (+softmax, *Hx).  Replaces, batched over all agents / all (agent,k) rows:
//   fc_c          model/model.py:243-251      relu([Hx,Hy] W + b)
//   vae_enc fc    model/model.py:488          flat W + b
//   deconv1       model/model.py:465          1x1 -> 4x4 transposed conv == GEMM [R,L]x[L,2048]
//   reparam       model/model.py:260-264      z = mu + sqrt(exp(logsig2)) * eps
//   mask fc       model/model.py:271-280      x_z = softmax(relu(xhat W + b)) * Hx
#include "common.h"
#include "kernels.h"

#define KC 256            // K-chunk staged in LDS per pass
#define LDA_C (KC + 4)    // 260 = 4*65: ds_read_b128 conflict-free

template <int EPI, int NTW, int MT = 2>      // MT 32-row tiles per workgroup
__global__ __launch_bounds__(DS_WG, 2) void k_gemm_rows(GemmArgs a) {
    constexpr int TMR = 32 * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int row0 = blockIdx.x * TMR;
    DYN_N(a, M, row0)
    f32x16 acc[NTW][MT];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[j][m] = zero16();
    const float* a_lane = smem + (lane & 31) * LDA_C + 4 * (lane >> 5);

    for (int k0 = 0; k0 < a.K; k0 += KC) {
        const int kc = min(KC, a.K - k0);
        const int q = kc >> 2;                      // float4 per row
        for (int i = tid; i < TMR * q; i += DS_WG) {
            const int r = i / q, c4 = i - r * q;
            const int row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < a.M) v = *reinterpret_cast<const float4*>(a.A + (size_t)row * a.lda + k0 + c4 * 4);
            *reinterpret_cast<float4*>(smem + r * LDA_C + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int nt = (blockIdx.y * NTW + j) * 4 + w;
            if (nt < a.NT)
                mma_groups<MT>(acc[j], a_lane, LDA_C, a.Bp + ((size_t)nt * a.G + (k0 >> 3)) * 64 + lane, kc >> 3);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = (blockIdx.y * NTW + j) * 4 + w;
        if (nt >= a.NT) continue;
        const int col = nt * 32 + (lane & 31);
        if (col >= a.N) continue;
        float p0 = 0.f, p1 = 0.f;
        if (EPI == EPI_SCALE_SHIFT_ELU) { const int ch = col % a.chmod; p0 = a.p0[ch]; p1 = a.p1[ch]; }
        else if (EPI == EPI_ELUGRAD || EPI == EPI_SIGGRAD) p0 = a.p0[col % a.chmod];
        else if (EPI != EPI_NONE) p0 = a.p0[col];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + m * 32 + acc_row(i);
                if (row >= a.M) continue;
                float v = acc[j][m][i];
                if (EPI == EPI_BIAS) v = v + p0;
                else if (EPI == EPI_BIAS_RELU) v = fmaxf(v + p0, 0.f);
                else if (EPI == EPI_SCALE_SHIFT_ELU) v = eluf_(v * p0 + p1);
                else if (EPI == EPI_ELUGRAD) { const float y = a.aux[(size_t)row * a.ldo + col]; v = v * (y > 0.f ? 1.0f : y + 1.0f) * p0; }
                else if (EPI == EPI_SIGGRAD) { const float y = a.aux[(size_t)row * a.ldo + col]; v = v * y * (1.0f - y) * p0; }
                a.out[(size_t)row * a.ldo + col] = v;
            }
    }
}

template <int NTW, int MT>
static void launch_gemm_rows_n(const GemmArgs& a, int epi, hipStream_t s) {
    constexpr int TMR = 32 * MT;
    dim3 grid((a.M + TMR - 1) / TMR, (a.NT + 4 * NTW - 1) / (4 * NTW));
    const size_t lds = TMR * LDA_C * sizeof(float);
    allow_big_lds(k_gemm_rows<EPI_BIAS, NTW, MT>); allow_big_lds(k_gemm_rows<EPI_BIAS_RELU, NTW, MT>);
    allow_big_lds(k_gemm_rows<EPI_SCALE_SHIFT_ELU, NTW, MT>); allow_big_lds(k_gemm_rows<EPI_NONE, NTW, MT>);
    allow_big_lds(k_gemm_rows<EPI_ELUGRAD, NTW, MT>); allow_big_lds(k_gemm_rows<EPI_SIGGRAD, NTW, MT>);
    if (epi == EPI_NONE) hipLaunchKernelGGL((k_gemm_rows<EPI_NONE, NTW, MT>), grid, dim3(DS_WG), lds, s, a);
    else if (epi == EPI_ELUGRAD) hipLaunchKernelGGL((k_gemm_rows<EPI_ELUGRAD, NTW, MT>), grid, dim3(DS_WG), lds, s, a);
    else if (epi == EPI_SIGGRAD) hipLaunchKernelGGL((k_gemm_rows<EPI_SIGGRAD, NTW, MT>), grid, dim3(DS_WG), lds, s, a);
    else if (epi == EPI_BIAS) hipLaunchKernelGGL((k_gemm_rows<EPI_BIAS, NTW, MT>), grid, dim3(DS_WG), lds, s, a);
    else if (epi == EPI_BIAS_RELU) hipLaunchKernelGGL((k_gemm_rows<EPI_BIAS_RELU, NTW, MT>), grid, dim3(DS_WG), lds, s, a);
    else hipLaunchKernelGGL((k_gemm_rows<EPI_SCALE_SHIFT_ELU, NTW, MT>), grid, dim3(DS_WG), lds, s, a);
}
// A workgroup normally holds a 64-row A tile and each wave runs four column tiles over it.  With few row tiles (a handful of
// windows per call) that leaves most CUs idle behind one long dependent MFMA chain, so small launches give every wave ONE
// 32 x 32 output tile instead (same k order per output: the results are bit-identical either way).
void launch_gemm_rows(const GemmArgs& a, int epi, hipStream_t s) {
    const int Msel = (a.dyn.cnt && a.M_hint > 0) ? a.M_hint : a.M;       // (device-side count: a.M is the worst case, which sizes the grid inside launch_gemm_rows_n)
    const long wgs = (long)((Msel + DS_TM - 1) / DS_TM) * ((a.NT + 15) / 16);
    if (wgs < 128) launch_gemm_rows_n<1, 1>(a, epi, s); else launch_gemm_rows_n<4, 2>(a, epi, s);
}

// ------------------------------------------------------------------------------------------------
// z = mu[agent] + sqrt(exp(logsig2[agent])) * eps[row]      (posterior)   |   z = eps   (prior)
// params [A, 2L] = (mu | logsig2) as written by the vae_enc fc.
// ------------------------------------------------------------------------------------------------
__global__ void k_reparam(const float* __restrict__ params, const float* __restrict__ eps,
                          float* __restrict__ z, int R, int L, int K, int mno, int posterior) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * L) return;
    const int r = i / L, l = i - r * L;
    float e = eps[i];
    if (posterior) {
        const int a = agent_of_row(r, K, mno);
        const float mu = params[(size_t)a * 2 * L + l], ls = params[(size_t)a * 2 * L + L + l];
        e = mu + sqrtf(expf(ls)) * e;
    }
    z[i] = e;
}

void launch_reparam(const float* params, const float* eps, float* z, int R, int L, int K, int mno,
                    int posterior, hipStream_t s) {
    const int n = R * L;
    hipLaunchKernelGGL(k_reparam, dim3((n + 255) / 256), dim3(256), 0, s, params, eps, z, R, L, K, mno, posterior);
}

// ------------------------------------------------------------------------------------------------
// mask fc: xz[r,:] = softmax(relu(xhat[r,:] @ Wm + bm)) * Hx[agent(r),:]
// one workgroup = 64 rows x all H (<=128) columns, K = V = 1024 in 4 chunks.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(DS_WG) void k_mask(MaskArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int row0 = blockIdx.x * DS_TM;
    DYN_P(a, row0)
    const int NT = a.H >> 5;                                 // 2, 4 or 8 column tiles; wave w takes w, w+4
    f32x16 acc[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
    const float* a_lane = smem + (lane & 31) * LDA_C + 4 * (lane >> 5);
    const int G = a.V >> 3;
    for (int k0 = 0; k0 < a.V; k0 += KC) {
        const int q = KC >> 2;
        for (int i = tid; i < DS_TM * q; i += DS_WG) {
            const int r = i / q, c4 = i - r * q;
            const int row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < a.R) v = *reinterpret_cast<const float4*>(a.xhat + (size_t)row * a.V + k0 + c4 * 4);
            *reinterpret_cast<float4*>(smem + r * LDA_C + c4 * 4) = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int nt = w + 4 * j;
            if (nt < NT) mma_groups<2>(acc[j], a_lane, LDA_C, a.Wp + ((size_t)nt * G + (k0 >> 3)) * 64 + lane, KC >> 3);
        }
        __syncthreads();
    }
    // relu(acc + b) -> LDS tile [64][H+4]
    const int LDT = a.H + 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nt = w + 4 * j;
        if (nt >= NT) continue;
        const int col = nt * 32 + (lane & 31);
        const float b = a.bias[col];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = fmaxf(acc[j][m][i] + b, 0.f);
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

void launch_mask(const MaskArgs& a, hipStream_t s) {
    const size_t lds = DS_TM * LDA_C * sizeof(float);
    allow_big_lds(k_mask);
    hipLaunchKernelGGL(k_mask, dim3((a.R + DS_TM - 1) / DS_TM), dim3(DS_WG), lds, s, a);
}
