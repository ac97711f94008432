// kernels_aux.hip -- the cold rows of the hot-path scope table: losses, the
// reference's temporal conv O1 / feature pooling O11 (A4, A11; no consumer in the reference, kept as
// ops), and the scene-context CNN rho(I) (A14, once per scene).  All VALU: none of them is on the
// per-sample critical path (0.35 GFLOP per scene image, O(A) scalars).
#include "common.h"
#include "kernels.h"

// ---- scene CNN: direct NHWC conv, SAME padding (TF rule), stride S, optional ReLU ----------------
// one thread per (pixel, 4 output channels)
__global__ void k_conv_direct(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ b,
                              float* __restrict__ out, int n, int Hi, int Wi, int Ci, int Co, int stride, int relu) {
    const int Ho = (Hi + stride - 1) / stride, Wo = (Wi + stride - 1) / stride;
    const int pad_t = max((Ho - 1) * stride + 5 - Hi, 0) / 2, pad_l = max((Wo - 1) * stride + 5 - Wi, 0) / 2;
    const int cq = Co >> 2;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * Ho * Wo * cq) return;
    const int c4 = idx % cq;
    const long pix = idx / cq;
    const int ox = pix % Wo, oy = (pix / Wo) % Ho, img = pix / ((long)Wo * Ho);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < 5; ++ky) {
        const int iy = oy * stride + ky - pad_t;
        if (iy < 0 || iy >= Hi) continue;
        for (int kx = 0; kx < 5; ++kx) {
            const int ix = ox * stride + kx - pad_l;
            if (ix < 0 || ix >= Wi) continue;
            const float* ip = in + (((size_t)img * Hi + iy) * Wi + ix) * Ci;
            const float* wp = w + ((size_t)(ky * 5 + kx) * Ci) * Co + c4 * 4;
            for (int ci = 0; ci < Ci; ++ci) {
                const float x = ip[ci];
                const float4 ww = *reinterpret_cast<const float4*>(wp + (size_t)ci * Co);
                acc.x = fmaf(x, ww.x, acc.x); acc.y = fmaf(x, ww.y, acc.y);
                acc.z = fmaf(x, ww.z, acc.z); acc.w = fmaf(x, ww.w, acc.w);
            }
        }
    }
    const float4 bb = *reinterpret_cast<const float4*>(b + c4 * 4);
    acc.x += bb.x; acc.y += bb.y; acc.z += bb.z; acc.w += bb.w;
    if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    *reinterpret_cast<float4*>(out + ((((size_t)img * Ho + oy) * Wo + ox) * Co) + c4 * 4) = acc;
}
void launch_conv_direct(const float* in, const float* w, const float* b, float* out, int n, int Hi, int Wi, int Ci,
                        int Co, int stride, int relu, hipStream_t s) {
    const int Ho = (Hi + stride - 1) / stride, Wo = (Wi + stride - 1) / stride;
    const long total = (long)n * Ho * Wo * (Co / 4);
    hipLaunchKernelGGL(k_conv_direct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, w, b, out, n, Hi, Wi,
                       Ci, Co, stride, relu);
}

// ---- O1 temporal conv: rho[a, c*100+q] = relu(sum_t X[a,t,c] * W[t,c,q] + b[c*100+q]),  c in {id, x} ----
__global__ void k_temporal_conv(const float* __restrict__ frames, const float* __restrict__ w, const float* __restrict__ b,
                                float* __restrict__ rho, int n_scenes, int T, int mno) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int A = n_scenes * mno;
    if (idx >= A * 200) return;
    const int a = idx / 200, o = idx - a * 200;
    const int c = o / 100, q = o - c * 100;
    const int sc = a / mno, slot = a - sc * mno;
    float acc = 0.f;
    for (int t = 0; t < T; ++t)
        acc = fmaf(frames[(((size_t)sc * T + t) * mno + slot) * 3 + c], w[(t * 2 + c) * 100 + q], acc);
    rho[idx] = fmaxf(acc + b[o], 0.f);
}
void launch_temporal_conv(const float* frames, const float* w, const float* b, float* rho, int n_scenes, int T, int mno,
                          hipStream_t s) {
    const int n = n_scenes * mno * 200;
    hipLaunchKernelGGL(k_temporal_conv, dim3((n + 255) / 256), dim3(256), 0, s, frames, w, b, rho, n_scenes, T, mno);
}

// ---- O11 feature pooling: f[r,t,:100] = y_x * rho[a,:100], f[r,t,100:] = y_y * rho[a,100:] ----
__global__ void k_feature_pooling(const float* __restrict__ Y, const float* __restrict__ rho, float* __restrict__ out,
                                  int R, int T, int K, int mno) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)R * T * 200) return;
    const int o = idx % 200;
    const long rt = idx / 200;
    const int r = rt / T;
    const int a = agent_of_row(r, K, mno);
    out[idx] = Y[rt * 2 + (o >= 100)] * rho[(size_t)a * 200 + o];
}
void launch_feature_pooling(const float* Y, const float* rho, float* out, int R, int T, int K, int mno, hipStream_t s) {
    const long n = (long)R * T * 200;
    hipLaunchKernelGGL(k_feature_pooling, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Y, rho, out, R, T, K, mno);
}

// ---- losses: one workgroup per agent; cost reduced by a single-block second kernel (deterministic) ----
__global__ void k_losses(const float* __restrict__ params, const float* __restrict__ Y, const float* __restrict__ fut,
                         const float* __restrict__ nfut, float* __restrict__ kld, float* __restrict__ recon, int n_scenes, int mno,
                         int K, int T, int L, float sx, float sy) {
    __shared__ float red[256];
    const int a = blockIdx.x, tid = threadIdx.x;
    const int sc = a / mno, slot = a - sc * mno;
    float s = 0.f;
    for (int l = tid; l < L; l += 256) {
        const float mu = params[(size_t)a * 2 * L + l], ls = params[(size_t)a * 2 * L + L + l];
        s += 1.0f + ls - mu * mu - expf(ls);
    }
    red[tid] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    if (tid == 0) kld[a] = -0.5f * red[0];
    __syncthreads();
    float d = 0.f;
    for (int i = tid; i < K * T; i += 256) {
        const int k = i / T, t = i - k * T;
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
        if (f[0] == 0.f) continue;                     // the object is not in this target frame (model/model.py:351-366)
        const float dx = Y[(r * T + t) * 2] - __fmul_rn(f[1], sx), dy = Y[(r * T + t) * 2 + 1] - __fmul_rn(f[2], sy);
        d += sqrtf(dx * dx + dy * dy);
    }
    red[tid] = d;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) red[tid] += red[tid + st]; __syncthreads(); }
    if (tid == 0) recon[a] = nfut[a] > 0.f ? red[0] / ((float)K * nfut[a]) : 0.f;
}
// Which agents and frames enter a loss (the reference's rule, model/model.py:351-366: an object that does not exist, or does not
// exist in the target frame, does not contribute): present(a, t) = id != 0 in future frame t; nfut[a] = number of present
// frames; lmask[a] = present at the last observed frame AND in at least one future frame.
__global__ void k_loss_mask(const uint8_t* __restrict__ valid, const float* __restrict__ fut, uint8_t* __restrict__ lmask,
                            float* __restrict__ nfut, int n_scenes, int mno, int T) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_scenes * mno) return;
    const int sc = a / mno, slot = a - sc * mno;
    int n = 0;
    for (int t = 0; t < T; ++t) n += fut[(((size_t)sc * T + t) * mno + slot) * 3] != 0.f ? 1 : 0;
    nfut[a] = (float)n;
    lmask[a] = (valid[a] && n > 0) ? 1 : 0;
}
void launch_loss_mask(const uint8_t* valid, const float* fut, uint8_t* lmask, float* nfut, int n_scenes, int mno, int T, hipStream_t s) {
    const int A = n_scenes * mno;
    hipLaunchKernelGGL(k_loss_mask, dim3((A + 127) / 128), dim3(128), 0, s, valid, fut, lmask, nfut, n_scenes, mno, T);
}
__global__ void k_cost(const float* __restrict__ kld, const float* __restrict__ recon, const uint8_t* __restrict__ valid,
                       float* __restrict__ cost, int A) {
    __shared__ float rs[256], rn[256];
    const int tid = threadIdx.x;
    float s = 0.f, n = 0.f;
    for (int a = tid; a < A; a += 256) if (valid[a]) { s += recon[a] + kld[a]; n += 1.f; }
    rs[tid] = s; rn[tid] = n;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) { if (tid < st) { rs[tid] += rs[tid + st]; rn[tid] += rn[tid + st]; } __syncthreads(); }
    if (tid == 0) { cost[0] = rs[0] / fmaxf(rn[0], 1.f); cost[1] = rn[0]; }
}
void launch_losses(const float* params, const float* Y, const float* fut, const uint8_t* valid, const float* nfut, float* kld,
                   float* recon, float* cost, int n_scenes, int mno, int K, int T, int L, float sx, float sy, hipStream_t s) {
    const int A = n_scenes * mno;                       // valid = the loss mask of k_loss_mask
    hipLaunchKernelGGL(k_losses, dim3(A), dim3(256), 0, s, params, Y, fut, nfut, kld, recon, n_scenes, mno, K, T, L, sx, sy);
    hipLaunchKernelGGL(k_cost, dim3(1), dim3(256), 0, s, kld, recon, valid, cost, A);
}

// ------------------------------------------------------------------------------------------------
// N1: window + slot builder (utils/data_loader.py:203-229 on the device).  One workgroup per window:
// the ids of the W = T_obs+T_pred frames go into a presence bitmap, slot = rank of the id among the
// window's sorted unique ids (0 counts when any padding row exists, exactly like np.unique), rows
// are copied.  Pure integer/copy work: bit-exact against DataLoader.window_to_slots.
// err[0] |= 1: id out of bitmap range, |= 2: more unique ids than slots (the reference's IndexError), |= 4: an id twice in
// one frame (the reference's ValueError).
// ------------------------------------------------------------------------------------------------
#define BW_WORDS 2048        // ids < 65536
__global__ __launch_bounds__(256) void k_build_windows(const float* __restrict__ frames, int F, int mno_in,
                                                       const int32_t* __restrict__ starts, int T_obs, int T_pred, int mno,
                                                       float* __restrict__ past, float* __restrict__ fut,
                                                       int32_t* __restrict__ err, int lookahead) {
    __shared__ unsigned int bits[BW_WORDS];
    __shared__ unsigned int pref[BW_WORDS];
    __shared__ unsigned int part[256];
    const int wdw = blockIdx.x, tid = threadIdx.x;
    const int W = T_obs + T_pred;
    const int f0 = starts[wdw];
    for (int i = tid; i < BW_WORDS; i += 256) bits[i] = 0u;
    for (int i = tid; i < T_obs * mno * 3; i += 256) past[(size_t)wdw * T_obs * mno * 3 + i] = 0.f;
    for (int i = tid; i < T_pred * mno * 3; i += 256) fut[(size_t)wdw * T_pred * mno * 3 + i] = 0.f;
    __syncthreads();
    const int n = W * mno_in;
    // lookahead: the loader ranks the ids of seq_length + 1 frames (its target is the window shifted by one frame,
    // utils/data_loader.py:203-209), so a DataLoader(seq_length = W) window takes its slots from frame f0 + W as well
    const int Wu = (lookahead && f0 + W < F) ? W + 1 : W;
    for (int i = tid; i < Wu * mno_in; i += 256) {
        const int t = i / mno_in, s = i - t * mno_in;
        const int f = min(f0 + t, F - 1);
        const float idf = frames[((size_t)f * mno_in + s) * 3];
        const unsigned id = (unsigned)idf;
        if (idf < 0.f || id >= BW_WORDS * 32u) { atomicOr(err, 1); continue; }
        atomicOr(&bits[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    // exclusive prefix of popcounts over the bitmap words: 8 words per thread, then a block scan
    unsigned local[8], sum = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { local[k] = sum; sum += __popc(bits[tid * 8 + k]); }
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned v = (tid >= off) ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const unsigned base = part[tid] - sum;
#pragma unroll
    for (int k = 0; k < 8; ++k) pref[tid * 8 + k] = base + local[k];
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int t = i / mno_in, s = i - t * mno_in;
        const int f = min(f0 + t, F - 1);
        const float* src = frames + ((size_t)f * mno_in + s) * 3;
        const float idf = src[0];
        if (idf == 0.f || idf < 0.f) continue;
        const unsigned id = (unsigned)idf;
        if (id >= BW_WORDS * 32u) continue;
        const unsigned slot = pref[id >> 5] + __popc(bits[id >> 5] & ((1u << (id & 31)) - 1u));
        if (slot >= (unsigned)mno) { atomicOr(err, 2); continue; }
        float* dst = (t < T_obs) ? past + (((size_t)wdw * T_obs + t) * mno + slot) * 3
                                 : fut + (((size_t)wdw * T_pred + (t - T_obs)) * mno + slot) * 3;
        // an id that occurs twice in one frame cannot go to one slot: the reference's ValueError (utils/data_loader.py:224-229)
        if (atomicExch(dst, idf) != 0.f) { atomicOr(err, 4); continue; }
        dst[1] = src[1]; dst[2] = src[2];
    }
    if (Wu > W) {        // the look-ahead frame is only ranked, not copied; the reference would still fail on it while filling its target
        __syncthreads();
        for (int s = tid; s < mno_in; s += 256) {
            const float idf = frames[((size_t)(f0 + W) * mno_in + s) * 3];
            if (!(idf > 0.f)) continue;
            const unsigned id = (unsigned)idf;
            if (id >= BW_WORDS * 32u) continue;
            const unsigned slot = pref[id >> 5] + __popc(bits[id >> 5] & ((1u << (id & 31)) - 1u));
            if (slot >= (unsigned)mno) atomicOr(err, 2);
            for (int s2 = 0; s2 < s; ++s2)
                if (frames[((size_t)(f0 + W) * mno_in + s2) * 3] == idf) { atomicOr(err, 4); break; }
        }
    }
}
void launch_build_windows(const float* frames, int F, int mno_in, const int32_t* starts, int n, int T_obs, int T_pred,
                          int mno, float* past, float* fut, int32_t* err, int lookahead, hipStream_t s) {
    hipLaunchKernelGGL(k_build_windows, dim3(n), dim3(256), 0, s, frames, F, mno_in, starts, T_obs, T_pred, mno, past, fut, err, lookahead);
}

// ---- N3: bivariate-Gaussian head of sample() (model/model.py:552-565,595-611,661-669) ----------------
// params [n,5] = (mux, muy, log sx, log sy, atanh-space corr); normals [n,2] ~ N(0,1) supplied by the caller;
// out [n,2] = sample, clipped to <= 1.0 like the reference.  Cholesky form of the covariance at :608.
__global__ void k_gaussian_sample(const float* __restrict__ p, const float* __restrict__ nrm, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float mux = p[i * 5], muy = p[i * 5 + 1];
    const float sx = expf(p[i * 5 + 2]), sy = expf(p[i * 5 + 3]), rho = tanhf(p[i * 5 + 4]);
    const float n0 = nrm[i * 2], n1 = nrm[i * 2 + 1];
    const float x = mux + sx * n0;
    const float y = muy + sy * (rho * n0 + sqrtf(fmaxf(1.0f - rho * rho, 0.f)) * n1);
    out[i * 2] = fminf(x, 1.0f);
    out[i * 2 + 1] = fminf(y, 1.0f);
}
void launch_gaussian_sample(const float* p, const float* nrm, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(k_gaussian_sample, dim3((n + 255) / 256), dim3(256), 0, s, p, nrm, out, n);
}

// ---- N4: ADE / FDE evaluation, best-of-K and mean-of-K, per agent ------------------------------------
// out [A,4] = (ade_mean, fde_mean, ade_min, fde_min) in the units of Y (normalised).
__global__ void k_ade_fde(const float* __restrict__ Y, const float* __restrict__ fut, float* __restrict__ out, int n_scenes,
                          int mno, int K, int T, float sx, float sy) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_scenes * mno) return;
    const int sc = a / mno, slot = a - sc * mno;
    float am = 0.f, fm = 0.f, amin = 3.0e38f, fmin_ = 3.0e38f;
    for (int k = 0; k < K; ++k) {
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        float s = 0.f, last = 0.f;
        int np = 0;
        for (int t = 0; t < T; ++t) {                  // frames the object is absent from carry no ground truth: skipped
            const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
            if (f[0] == 0.f) continue;
            const float dx = Y[(r * T + t) * 2] - __fmul_rn(f[1], sx), dy = Y[(r * T + t) * 2 + 1] - __fmul_rn(f[2], sy);
            last = sqrtf(dx * dx + dy * dy);           // FDE = the error at the last frame the object is present in
            s += last; ++np;
        }
        s = np ? s / (float)np : 0.f;
        am += s; fm += last;
        amin = fminf(amin, s); fmin_ = fminf(fmin_, last);
    }
    out[a * 4] = am / (float)K; out[a * 4 + 1] = fm / (float)K; out[a * 4 + 2] = amin; out[a * 4 + 3] = fmin_;
}
void launch_ade_fde(const float* Y, const float* fut, float* out, int n_scenes, int mno, int K, int T, float sx, float sy,
                    hipStream_t s) {
    const int A = n_scenes * mno;
    hipLaunchKernelGGL(k_ade_fde, dim3((A + 127) / 128), dim3(128), 0, s, Y, fut, out, n_scenes, mno, K, T, sx, sy);
}


// ------------------------------------------------------------------------------------------------------------------
// dims.bn_mode = 1, "per-object" batch normalisation: the reference runs its conv stacks one object at a time inside
// defaults_scope(batch_normalize=True, phase=train) (model/model.py:453-462,471-481), i.e. batch statistics over a batch
// of ONE -- per-sample, per-channel moments over the layer's pixels.  The conv kernels then run with a linear epilogue
// (the conv bias cancels against the mean) and this kernel normalises one sample per workgroup in place:
//     y = act( (x - mean_c) * gamma_c / sqrt(var_c + 1e-3) + beta_c ),  var biased (tf.nn.moments), act = ELU or sigmoid.
// x [n, P, C] (NHWC), P*C <= 8192.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_instnorm_act(const float* x, float* y, int n, int P, int C, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int sig) {
    __shared__ float xs[8192];
    __shared__ float part[256];
    __shared__ float mean_s[128], rstd_s[128];
    const int tid = threadIdx.x, smp = blockIdx.x;
    const int N = P * C;
    const float* xp = x + (size_t)smp * N;
    float* yp = y + (size_t)smp * N;                // y may be x (in place: inference) or another buffer (training keeps the pre-norm x)
    for (int i = tid; i < N; i += 256) xs[i] = xp[i];
    __syncthreads();
    const int G = 256 / C;                       // threads per channel (C in {1, 32, 64, 128} -> 256, 8, 4, 2)
    const int c = tid % C, g = tid / C;
    float s = 0.f;
    for (int p = g; p < P; p += G) s += xs[p * C + c];
    part[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int j = 0; j < G; ++j) t += part[j * C + tid];
        mean_s[tid] = t / (float)P;
    }
    __syncthreads();
    const float m = mean_s[c];
    s = 0.f;
    for (int p = g; p < P; p += G) { const float dlt = xs[p * C + c] - m; s += dlt * dlt; }
    part[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int j = 0; j < G; ++j) t += part[j * C + tid];
        rstd_s[tid] = gamma[tid] / sqrtf(t / (float)P + 1e-3f);
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int ch = i % C;
        const float v = (xs[i] - mean_s[ch]) * rstd_s[ch] + beta[ch];
        yp[i] = sig ? sigmoidf_(v) : eluf_(v);
    }
}
void launch_instnorm_act(float* x, int n, int P, int C, const float* gamma, const float* beta, int sig, hipStream_t s) {
    hipLaunchKernelGGL(k_instnorm_act, dim3(n), dim3(256), 0, s, (const float*)x, x, n, P, C, gamma, beta, sig);
}
void launch_instnorm_act_oop(const float* x, float* y, int n, int P, int C, const float* gamma, const float* beta, int sig, hipStream_t s) {
    hipLaunchKernelGGL(k_instnorm_act, dim3(n), dim3(256), 0, s, x, y, n, P, C, gamma, beta, sig);
}

// Backward of y = act(gamma * xh + beta), xh = (x - mean) * rstd with per-sample per-channel moments over the P pixels:
//     g = dy * act'(y);   dx = gamma * rstd * ( g - mean_p(g) - xh * mean_p(g * xh) )
// One workgroup per sample; x (pre-norm, kept by the training-mode forward) is staged in LDS, the moments are recomputed exactly
// as the forward took them.  dy -> dx in place.  gamma / beta are constants of the training spec (no gradient).
__global__ __launch_bounds__(256) void k_instnorm_act_bwd(float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                          int n, int P, int C, const float* __restrict__ gamma, int sig) {
    __shared__ float xs[8192];
    __shared__ float part[256], part2[256];
    __shared__ float mean_s[128], rstd_s[128], mg_s[128], mgx_s[128];
    const int tid = threadIdx.x, smp = blockIdx.x;
    const int N = P * C;
    const float* xp = x + (size_t)smp * N;
    const float* yp = y + (size_t)smp * N;
    float* gp = dy + (size_t)smp * N;
    for (int i = tid; i < N; i += 256) xs[i] = xp[i];
    __syncthreads();
    const int G = 256 / C;
    const int c = tid % C, g = tid / C;
    float s = 0.f;
    for (int p = g; p < P; p += G) s += xs[p * C + c];
    part[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int j = 0; j < G; ++j) t += part[j * C + tid];
        mean_s[tid] = t / (float)P;
    }
    __syncthreads();
    const float m = mean_s[c];
    s = 0.f;
    for (int p = g; p < P; p += G) { const float dlt = xs[p * C + c] - m; s += dlt * dlt; }
    part[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int j = 0; j < G; ++j) t += part[j * C + tid];
        rstd_s[tid] = 1.0f / sqrtf(t / (float)P + 1e-3f);
    }
    __syncthreads();
    const float rs = rstd_s[c];
    float sg = 0.f, sgx = 0.f;
    for (int p = g; p < P; p += G) {
        const int i = p * C + c;
        const float yv = yp[i];
        const float gv = gp[i] * (sig ? yv * (1.0f - yv) : (yv > 0.f ? 1.0f : yv + 1.0f));
        sg += gv; sgx += gv * (xs[i] - m) * rs;
    }
    part[tid] = sg; part2[tid] = sgx;
    __syncthreads();
    if (tid < C) {
        float t = 0.f, t2 = 0.f;
        for (int j = 0; j < G; ++j) { t += part[j * C + tid]; t2 += part2[j * C + tid]; }
        mg_s[tid] = t / (float)P; mgx_s[tid] = t2 / (float)P;
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int ch = i % C;
        const float yv = yp[i];
        const float gv = gp[i] * (sig ? yv * (1.0f - yv) : (yv > 0.f ? 1.0f : yv + 1.0f));
        const float xh = (xs[i] - mean_s[ch]) * rstd_s[ch];
        gp[i] = gamma[ch] * rstd_s[ch] * (gv - mg_s[ch] - xh * mgx_s[ch]);
    }
}
void launch_instnorm_act_bwd(float* dy, const float* x, const float* y, int n, int P, int C, const float* gamma, int sig, hipStream_t s) {
    hipLaunchKernelGGL(k_instnorm_act_bwd, dim3(n), dim3(256), 0, s, dy, x, y, n, P, C, gamma, sig);
}


// ------------------------------------------------------------------------------------------------------------------
// dims.bn_mode = 2, WHOLE-BATCH statistics: prettytensor batch_normalize in phase=train (model/model.py:453,459-461,471) over
// everything one call batches -- per-channel moments over all n samples and their P pixels (tf.nn.moments: biased variance,
// mean first, then the centred second moment), eps 1e-3.  Deterministic: fixed per-block partial sums, reduced in block order.
// x [n, P, C]; part [BN_BLOCKS, C] scratch, stat [2, C] = (mean | gamma * rstd).
// ------------------------------------------------------------------------------------------------------------------
#define BN_BLOCKS 512
__global__ __launch_bounds__(256) void k_bn_partial(const float* __restrict__ x, size_t rows, int C, const float* __restrict__ mean,
                                                    float* __restrict__ part) {
    __shared__ float red[256];
    const int tid = threadIdx.x, c = tid % C, g = tid / C, G = 256 / C;
    const size_t per = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    const float m = mean ? mean[c] : 0.f;
    float s = 0.f;
    for (size_t r = r0 + g; r < r1; r += G) { const float v = x[r * C + c] - m; s += mean ? v * v : v; }
    red[tid] = s;
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
        for (int j = 0; j < G; ++j) t += red[j * C + tid];
        part[(size_t)blockIdx.x * C + tid] = t;
    }
}
__global__ void k_bn_finish(const float* __restrict__ part, int nb, int C, float inv_n, const float* __restrict__ gamma, float* __restrict__ stat,
                            int second) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float t = 0.f;
    for (int b = 0; b < nb; ++b) t += part[(size_t)b * C + c];
    if (!second) stat[c] = t * inv_n;
    else if (second == 1) stat[C + c] = gamma[c] / sqrtf(t * inv_n + 1e-3f);
    else stat[C + c] = 1.0f / sqrtf(t * inv_n + 1e-3f);              // (backward: the plain reciprocal deviation)
}
__global__ void k_bn_apply(float* __restrict__ x, size_t n, int C, const float* __restrict__ stat, const float* __restrict__ beta, int sig) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % C);
        const float v = (x[i] - stat[ch]) * stat[C + ch] + beta[ch];
        x[i] = sig ? sigmoidf_(v) : eluf_(v);
    }
}
void launch_batchnorm_act(float* x, size_t n, int P, int C, const float* gamma, const float* beta, int sig, float* part, float* stat, hipStream_t s) {
    const size_t rows = n * P;                                  // NHWC: a "row" = one pixel's C channels
    const int nb = rows < BN_BLOCKS ? (int)rows : BN_BLOCKS;
    hipLaunchKernelGGL(k_bn_partial, dim3(nb), dim3(256), 0, s, x, rows, C, (const float*)nullptr, part);
    hipLaunchKernelGGL(k_bn_finish, dim3((C + 63) / 64), dim3(64), 0, s, part, nb, C, 1.0f / (float)rows, gamma, stat, 0);
    hipLaunchKernelGGL(k_bn_partial, dim3(nb), dim3(256), 0, s, x, rows, C, stat, part);
    hipLaunchKernelGGL(k_bn_finish, dim3((C + 63) / 64), dim3(64), 0, s, part, nb, C, 1.0f / (float)rows, gamma, stat, 1);
    const size_t tot = rows * C, nbk = (tot + 255) / 256;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)(nbk < 4096 ? nbk : 4096)), dim3(256), 0, s, x, tot, C, stat, beta, sig);
}

// Backward of the whole-batch form, y = act(gamma * xh + beta), xh = (x - mean) * rstd with per-channel moments over ALL rows (every
// sample's every pixel) of the call -- the reference's literal phase=train batch-norm when objects are batched (model/model.py:453,459-461,
// 471):   g = dy * act'(y);   dx = gamma * rstd * ( g - mean_rows(g) - xh * mean_rows(g * xh) ).
// The moments are recomputed from the kept pre-norm tensor exactly as the forward took them (same per-block partition, same order);
// the two gradient means use the same deterministic two-stage reduction.  dy -> dx in place; gamma / beta are constants of the spec.
__global__ __launch_bounds__(256) void k_bn_bwd_partial(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                        size_t rows, int C, const float* __restrict__ stat, int sig, float* __restrict__ part) {
    __shared__ float red[256], red2[256];
    const int tid = threadIdx.x, c = tid % C, g = tid / C, G = 256 / C;
    const size_t per = (rows + gridDim.x - 1) / gridDim.x;
    const size_t r0 = (size_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    const float m = stat[c], rs = stat[C + c];
    float sg = 0.f, sgx = 0.f;
    for (size_t r = r0 + g; r < r1; r += G) {
        const size_t i = r * C + c;
        const float yv = y[i];
        const float gv = dy[i] * (sig ? yv * (1.0f - yv) : (yv > 0.f ? 1.0f : yv + 1.0f));
        sg += gv; sgx += gv * (x[i] - m) * rs;
    }
    red[tid] = sg; red2[tid] = sgx;
    __syncthreads();
    if (tid < C) {
        float t = 0.f, t2 = 0.f;
        for (int j = 0; j < G; ++j) { t += red[j * C + tid]; t2 += red2[j * C + tid]; }
        part[(size_t)blockIdx.x * 2 * C + tid] = t;
        part[(size_t)blockIdx.x * 2 * C + C + tid] = t2;
    }
}
__global__ void k_bn_bwd_finish(const float* __restrict__ part, int nb, int C, float inv_n, float* __restrict__ stat2) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * C) return;
    float t = 0.f;
    for (int b = 0; b < nb; ++b) t += part[(size_t)b * 2 * C + c];
    stat2[c] = t * inv_n;
}
__global__ void k_bn_bwd_apply(float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y, size_t n, int C,
                               const float* __restrict__ stat, const float* __restrict__ stat2, const float* __restrict__ gamma, int sig) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % C);
        const float yv = y[i];
        const float gv = dy[i] * (sig ? yv * (1.0f - yv) : (yv > 0.f ? 1.0f : yv + 1.0f));
        const float xh = (x[i] - stat[ch]) * stat[C + ch];
        dy[i] = gamma[ch] * stat[C + ch] * (gv - stat2[ch] - xh * stat2[C + ch]);
    }
}
// part: [BN_BLOCKS][2C] scratch, stat [2][C] = (mean | rstd), stat2 [2][C] = (mean g | mean g xh)
void launch_batchnorm_act_bwd(float* dy, const float* x, const float* y, size_t n, int P, int C, const float* gamma, int sig,
                              float* part, float* stat, float* stat2, hipStream_t s) {
    const size_t rows = n * P;
    const int nb = rows < BN_BLOCKS ? (int)rows : BN_BLOCKS;
    hipLaunchKernelGGL(k_bn_partial, dim3(nb), dim3(256), 0, s, x, rows, C, (const float*)nullptr, part);
    hipLaunchKernelGGL(k_bn_finish, dim3((C + 63) / 64), dim3(64), 0, s, part, nb, C, 1.0f / (float)rows, gamma, stat, 0);
    hipLaunchKernelGGL(k_bn_partial, dim3(nb), dim3(256), 0, s, x, rows, C, stat, part);
    hipLaunchKernelGGL(k_bn_finish, dim3((C + 63) / 64), dim3(64), 0, s, part, nb, C, 1.0f / (float)rows, gamma, stat, 2);
    hipLaunchKernelGGL(k_bn_bwd_partial, dim3(nb), dim3(256), 0, s, (const float*)dy, x, y, rows, C, (const float*)stat, sig, part);
    hipLaunchKernelGGL(k_bn_bwd_finish, dim3((2 * C + 63) / 64), dim3(64), 0, s, (const float*)part, nb, C, 1.0f / (float)rows, stat2);
    const size_t tot = rows * C, nbk = (tot + 255) / 256;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)(nbk < 4096 ? nbk : 4096)), dim3(256), 0, s, dy, x, y, tot, C, (const float*)stat,
                       (const float*)stat2, gamma, sig);
}

// Stream-ordered fill / copy as KERNELS: the hot sequences stay kernel-only, which keeps them capturable into a hipGraph
// (memset / memcpy nodes of a captured stream were observed to run out of order on repeated launches of the same exec).
__global__ void k_fill_f32(float* __restrict__ dst, size_t n, float v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void k_copy_f32(float* __restrict__ dst, const float* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_fill_f32(float* dst, size_t n, float v, hipStream_t s) {
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)(nb < 2048 ? (nb ? nb : 1) : 2048)), dim3(256), 0, s, dst, n, v);
}
void launch_copy_f32(float* dst, const float* src, size_t n, hipStream_t s) {
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(k_copy_f32, dim3((unsigned)(nb < 2048 ? (nb ? nb : 1) : 2048)), dim3(256), 0, s, dst, src, n);
}

// dst[r, c] = src[r * ld + c] for c < cols: the logical columns of a (zero-padded) row-major tensor
__global__ void k_copy_cols(float* __restrict__ dst, const float* __restrict__ src, size_t rows, int cols, int ld) {
    const size_t n = rows * cols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / cols;
        dst[i] = src[r * ld + (i - r * cols)];
    }
}
void launch_copy_cols(float* dst, const float* src, size_t rows, int cols, int ld, hipStream_t s) {
    const size_t nb = (rows * cols + 255) / 256;
    hipLaunchKernelGGL(k_copy_cols, dim3((unsigned)(nb < 2048 ? (nb ? nb : 1) : 2048)), dim3(256), 0, s, dst, src, rows, cols, ld);
}
