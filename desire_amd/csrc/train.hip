// train.hip -- training half of the C ABI: training-mode buffers, backward orchestration, gradient access.
// The reference computes tf.gradients(cost) and an Adam update but never runs them (model/model.py:388-403,
// train.py:181); here they run.  Gradients live in ONE flat fp32 buffer in natural (TF) layouts, so a multi-GPU
// caller all-reduces a single tensor (RCCL through torch.distributed) between desire_backward and desire_adam_step.
#include "ctx.h"

#include <cmath>
#include <cstdio>
#include <cstring>

namespace {

int ensure(desire_ctx* h, const char* name, size_t bytes) {
    if (h->ws.count(name) && h->ws[name].bytes >= bytes) return 0;
    if (h->ws.count(name)) h->ws[name].release();
    return h->ws[name].alloc(bytes);
}

float* G(desire_ctx* h, const std::string& name) { return W(h, "Gflat") + h->slots.at(name).off; }

// weight gradient block: out[Kd, N] = A^T G over M rows, written into a [.., ldo] matrix
void tn(desire_ctx* h, const float* A, int lda, const float* Gm, int ldg, long M, int Kd, int N, float* out, int ldo,
        int accumulate, hipStream_t s, const unsigned long long* flags = nullptr, int fcols = 0,
        const int* rowlist = nullptr, const int* binbase = nullptr, const int* bintotal = nullptr) {
    TnArgs a{};
    a.A = A; a.lda = lda; a.G = Gm; a.ldg = ldg; a.M = M; a.Kd = Kd; a.N = N; a.flags = flags; a.fcols = fcols;
    a.rowlist = rowlist; a.binbase = binbase; a.bintotal = bintotal;
    // row lists serve the 128 x 128 tile form only (one tile row = one flag block of 128 columns); anything else keeps the flag words
    if (rowlist && !(fcols == 128 && N > 64 && gemm_tn_big_tiles(a) > 0)) a.rowlist = nullptr;
    if (a.rowlist) a.flags = nullptr;
    a.np = (h->d.bf16 == 2 && (train_x3_mask(h) & 1)) ? 2 : 0;
    // slices.  Split operands: the large forms keep two workgroups per CU and a workgroup's time per chunk does not depend on its MFMA count
    // (it waits for its operands), so ONE full round of 512 workgroups is best -- 680 took 2.56 ms where 512 take 1.87.  fp32 operands: the
    // kernel is bound by the matrix pipe, tiles that hang over Kd / N finish early, and more workgroups than slots balance that (4.16 vs 5.22 ms)
    const long big_tiles = a.np == 2 ? gemm_tn_big_tiles(a) : 0;
    const long blocks = ((Kd + 63) / 64) * ((N + 63) / 64);
    long sl = big_tiles ? 512 / big_tiles : 2048 / blocks;
    if (sl < 1) sl = 1; if (sl > (big_tiles ? 512 : 256)) sl = big_tiles ? 512 : 256;
    const long maxsl = (M + 63) / 64; if (sl > maxsl) sl = maxsl;
    while ((size_t)sl * Kd * N * sizeof(float) > h->ws["tn_partial"].bytes && sl > 1) sl /= 2;
    a.nslices = (int)sl; a.partial = W(h, "tn_partial");
    launch_gemm_tn(a, out, ldo, accumulate, s);
}
void colsum(desire_ctx* h, const float* Gm, int ldg, long M, int N, float* out, int accumulate, hipStream_t s) {
    long sl = 256; const long maxsl = (M + 3) / 4; if (sl > maxsl) sl = maxsl; if (sl < 1) sl = 1;
    launch_colsum(Gm, ldg, M, N, (int)sl, W(h, "tn_partial"), out, accumulate, s);
}

// ---- optimiser state lives next to the gradients: Wflat (master weights), Mflat, Vflat, all in Gflat's layout ----
__global__ void k_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       size_t n, float lr_t, float b1, float b2, float eps) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// global-norm clipping (tf.clip_by_global_norm, model/model.py:390): g *= min(1, clip / ||g||), all on the device
__global__ void k_sqsum(const float* __restrict__ g, size_t n, float* __restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += g[i] * g[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ void k_clip_scale(float* __restrict__ g, size_t n, const float* __restrict__ partial, int np, float clip, float* norm_out) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < np; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    const float norm = sqrtf(red[0]);
    if (blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;
    const float sc = norm > clip ? clip / norm : 1.f;
    if (sc == 1.f) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] *= sc;
}

// every packed / raw device operand is a gather of the master weights: dst[i] = idx[i] ? Wflat[idx[i]-1] : 0
struct Seg { float* dst; unsigned long long idx_off; unsigned long long n; };
__global__ void k_repack(const Seg* __restrict__ segs, const uint32_t* __restrict__ idx, const float* __restrict__ w) {
    const Seg sg = segs[blockIdx.y];
    const uint32_t* ix = idx + sg.idx_off;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < sg.n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t j = ix[i];
        sg.dst[i] = j ? w[j - 1] : 0.f;
    }
}

// split [hi | lo] bf16 packs of kernels_x3.hip (dims.bf16 = 2): dst[i] = bf16(w), dst[n + i] = bf16(w - hi), w = Wflat[idx[i]-1]
struct Seg16 { uint16_t* dst; unsigned long long idx_off; unsigned long long n; unsigned long long np; };     // np = pieces (2: "#x3" packs, 3: "#x6")
__device__ __forceinline__ uint16_t bf16_rne_dev(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__global__ void k_repack_split(const Seg16* __restrict__ segs, const uint32_t* __restrict__ idx, const float* __restrict__ w) {
    const Seg16 sg = segs[blockIdx.y];
    const uint32_t* ix = idx + sg.idx_off;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < sg.n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t j = ix[i];
        float v = j ? w[j - 1] : 0.f;
        for (unsigned long long pc = 0; pc < sg.np; ++pc) {               // piece pc = bf16 of what the earlier pieces left (each subtraction exact)
            const uint16_t b = bf16_rne_dev(v);
            sg.dst[pc * sg.n + i] = b;
            v -= __uint_as_float((uint32_t)b << 16);
        }
    }
}

// folded batch-norm shift follows the (trainable) conv bias: shift = beta + scale * (b - mean); scale is frozen
__global__ void k_refold(const float* __restrict__ w, float* __restrict__ shift, const float* __restrict__ scale, size_t off_b,
                         size_t off_beta, size_t off_mean, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) shift[c] = w[off_beta + c] + scale[c] * (w[off_b + c] - w[off_mean + c]);
}

// ------------------------------------------------------------------------------------------------------------------
// The reference's loss for its 5-wide output layer (model/model.py:315-366: output_w / output_b on the state, get_coef :552-565,
// -log(max(N(next position), 1e-20)) :494-550, the id == 0 masking :351-366, the mean :374-376), teacher-forced over the observed
// frames of the X encoder.  One wave per (agent, observed frame t): o = h_t W5 + b5; target = the position in frame t + 1 (the next
// observed frame, or the first future frame for t = T_obs - 1); the pair counts when the object exists in both frames.  Log form
// (z / (2 (1 - rho^2)) + log(2 pi sx sy sqrt(1 - rho^2)), clamped at -log 1e-20 with no gradient beyond, like the reference's max):
// the pdf itself underflows fp32 long before its logarithm matters.  Writes nll[a,t] (0 when not counted), cnt[a,t] and the raw
// gradient dO[a,t,5] = d nll / d o (Graves 2013, eq. 25-28); k_head_sum / k_head_scale turn them into the mean and its gradient.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head_nll(const float* __restrict__ sv_h, const float* __restrict__ sv_x, const float* __restrict__ past,
                                                  const float* __restrict__ fut, const float* __restrict__ W5, const float* __restrict__ b5,
                                                  int A, int T, int T_pred, int H, int mno, float sx_, float sy_, float* __restrict__ nll,
                                                  float* __restrict__ cnt, float* __restrict__ dO) {
    const int wv = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wv >= A * T) return;
    const int a = wv / T, t = wv - a * T, scene = a / mno, slot = a - scene * mno;
    float o[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const float* hrow = sv_h + (size_t)wv * H;
    for (int c = lane; c < H; c += 64) {
        const float hv = hrow[c];
#pragma unroll
        for (int j = 0; j < 5; ++j) o[j] = fmaf(hv, W5[c * 5 + j], o[j]);
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) o[j] += __shfl_xor(o[j], m);
        o[j] += b5[j];
    }
    if (lane) return;
    const float* now = past + (((size_t)scene * T + t) * mno + slot) * 3;
    const float* nxt = (t + 1 < T) ? now + (size_t)mno * 3 : fut + ((size_t)scene * T_pred * mno + slot) * 3;
    float x, y;
    if (t + 1 < T) { x = sv_x[((size_t)a * T + t + 1) * 2]; y = sv_x[((size_t)a * T + t + 1) * 2 + 1]; }
    else { x = __fmul_rn(nxt[1], sx_); y = __fmul_rn(nxt[2], sy_); }
    const bool counted = now[0] != 0.f && nxt[0] != 0.f;
    float L = 0.f, g[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (counted) {
        const float sx = __expf(o[2]), sy = __expf(o[3]), rho = tanhf(o[4]);
        const float nx = (x - o[0]) / sx, ny = (y - o[1]) / sy;
        const float neg = fmaxf(1.0f - rho * rho, 1e-12f);
        const float z = nx * nx + ny * ny - 2.0f * rho * nx * ny;
        L = z / (2.0f * neg) + 1.8378770664093453f + o[2] + o[3] + 0.5f * __logf(neg);      // log(2 pi) + log sx + log sy + log sqrt(1 - rho^2)
        if (L < 46.051701859880914f) {                   // -log(1e-20): beyond it the reference's max() pins the value and kills the gradient
            g[0] = -(nx - rho * ny) / (neg * sx);
            g[1] = -(ny - rho * nx) / (neg * sy);
            g[2] = 1.0f - nx * (nx - rho * ny) / neg;
            g[3] = 1.0f - ny * (ny - rho * nx) / neg;
            g[4] = -nx * ny + rho * z / neg - rho;
        } else L = 46.051701859880914f;
    }
    nll[wv] = L; cnt[wv] = counted ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 5; ++j) dO[(size_t)wv * 5 + j] = g[j];
}
// loss_out[5] = weight * mean nll over the counted pairs, loss_out[7] = their number (one block, fixed order: deterministic)
__global__ void k_head_sum(const float* __restrict__ nll, const float* __restrict__ cnt, int n, float weight, float* __restrict__ loss_out) {
    __shared__ float rs[256], rc[256];
    float s = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { s += nll[i]; c += cnt[i]; }
    rs[threadIdx.x] = s; rc[threadIdx.x] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tc = 0.f;
        for (int i = 0; i < 256; ++i) { ts += rs[i]; tc += rc[i]; }
        loss_out[5] = weight * ts / fmaxf(tc, 1.f);
        loss_out[7] = tc;
    }
}
__global__ void k_head_scale(float* __restrict__ dO, int n5, float weight, const float* __restrict__ loss_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n5) dO[i] *= weight / fmaxf(loss_out[7], 1.f);
}

// per-agent loss terms of DESIGN.md section 8: out[a] = {recon, kld, ce, reg} (0 for absent agents)
__global__ void k_train_loss(const float* __restrict__ Y0, const float* __restrict__ Yr, const float* __restrict__ fut,
                             const float* __restrict__ score, const float* __restrict__ params, const uint8_t* __restrict__ valid,
                             const float* __restrict__ nfut, float* __restrict__ out, int n_scenes, int mno, int K, int T, int L,
                             float sx, float sy) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_scenes * mno) return;
    const int sc = a / mno, slot = a - sc * mno;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid[a]) {
        float m1 = -3.0e38f, m2 = -3.0e38f, e0s = 0.f, e1s = 0.f;
        for (int k = 0; k < K; ++k) {
            const size_t r = ((size_t)sc * K + k) * mno + slot;
            float dm = 0.f;
            for (int t = 0; t < T; ++t) {
                const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
                if (f[0] == 0.f) continue;               // absent in this target frame (model/model.py:351-366)
                const float gx = __fmul_rn(f[1], sx), gy = __fmul_rn(f[2], sy);
                float dx = Y0[(r * T + t) * 2] - gx, dy = Y0[(r * T + t) * 2 + 1] - gy;
                const float e0 = sqrtf(dx * dx + dy * dy);
                dx = Yr[(r * T + t) * 2] - gx; dy = Yr[(r * T + t) * 2 + 1] - gy;
                e0s += e0; e1s += sqrtf(dx * dx + dy * dy);
                dm = fmaxf(dm, e0);
            }
            m1 = fmaxf(m1, -dm); m2 = fmaxf(m2, score[r]);
        }
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < K; ++k) {
            const size_t r = ((size_t)sc * K + k) * mno + slot;
            float dm = 0.f;
            for (int t = 0; t < T; ++t) {
                const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
                if (f[0] == 0.f) continue;
                const float dx = Y0[(r * T + t) * 2] - __fmul_rn(f[1], sx), dy = Y0[(r * T + t) * 2 + 1] - __fmul_rn(f[2], sy);
                dm = fmaxf(dm, sqrtf(dx * dx + dy * dy));
            }
            s1 += expf(-dm - m1); s2 += expf(score[r] - m2);
        }
        float ce = 0.f;
        for (int k = 0; k < K; ++k) {
            const size_t r = ((size_t)sc * K + k) * mno + slot;
            float dm = 0.f;
            for (int t = 0; t < T; ++t) {
                const float* f = fut + (((size_t)sc * T + t) * mno + slot) * 3;
                if (f[0] == 0.f) continue;
                const float dx = Y0[(r * T + t) * 2] - __fmul_rn(f[1], sx), dy = Y0[(r * T + t) * 2 + 1] - __fmul_rn(f[2], sy);
                dm = fmaxf(dm, sqrtf(dx * dx + dy * dy));
            }
            ce -= expf(-dm - m1) / s1 * (score[r] - m2 - logf(s2));
        }
        float kl = 0.f;
        for (int j = 0; j < L; ++j) {
            const float mu = params[(size_t)a * 2 * L + j], ls = params[(size_t)a * 2 * L + L + j];
            kl += 1.f + ls - mu * mu - expf(ls);
        }
        o = make_float4(e0s / ((float)K * nfut[a]), -0.5f * kl, ce, e1s / ((float)K * nfut[a]));
    }
    reinterpret_cast<float4*>(out)[a] = o;
}
__global__ void k_sum_loss(const float* __restrict__ per_agent, const uint8_t* __restrict__ valid, int A, float* __restrict__ out) {
    __shared__ float red[4][256];
    float s[4] = {0.f, 0.f, 0.f, 0.f}; float nv = 0.f;
    for (int a = threadIdx.x; a < A; a += 256) {
        for (int j = 0; j < 4; ++j) s[j] += per_agent[(size_t)a * 4 + j];
        nv += valid[a] ? 1.f : 0.f;
    }
    __shared__ float rnv[256];
    for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = s[j];
    rnv[threadIdx.x] = nv;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t[4] = {0.f, 0.f, 0.f, 0.f}; float n = 0.f;
        for (int i = 0; i < 256; ++i) { for (int j = 0; j < 4; ++j) t[j] += red[j][i]; n += rnv[i]; }
        n = fmaxf(n, 1.f);
        for (int j = 0; j < 4; ++j) out[j] = t[j] / n;
        out[4] = n;
    }
}

// Builds the gather maps: the packing code (desire_pack_all) is run a second time over weights whose VALUES are their
// own 1-based flat indices (as bit patterns); whatever it would have uploaded is then the index map of that operand.
// Each map is verified against the real device operand, so an operand that is not a pure gather cannot slip through.
int build_repack_maps(desire_ctx* h) {
    std::vector<float> flat(h->n_params, 0.f);
    auto real = h->host_w;
    for (auto& kv : h->slots) {
        const auto& src = real.at(kv.first);
        std::memcpy(flat.data() + kv.second.off, src.data(), kv.second.n * sizeof(float));
        std::vector<float> coded(kv.second.n);
        for (size_t i = 0; i < kv.second.n; ++i) {
            const uint32_t u = (uint32_t)(kv.second.off + i + 1);
            std::memcpy(&coded[i], &u, 4);
        }
        h->host_w[kv.first] = std::move(coded);
    }
    h->pack_mode = 1; h->captured.clear();
    const int rc = desire_pack_all(h);
    h->pack_mode = 0; h->host_w = std::move(real);
    if (rc) return rc;
    if (ensure(h, "Wflat", h->n_params * sizeof(float)) || ensure(h, "Mflat", h->n_params * sizeof(float)) ||
        ensure(h, "Vflat", h->n_params * sizeof(float)))
        return fail(DESIRE_ERR_HIP, "hipMalloc failed for the optimiser state");
    HIPCHK(hipMemcpy(W(h, "Wflat"), flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(W(h, "Mflat"), 0, h->n_params * sizeof(float)));
    HIPCHK(hipMemset(W(h, "Vflat"), 0, h->n_params * sizeof(float)));
    std::vector<uint32_t> all_idx; std::vector<Seg> segs; std::vector<Seg16> segs16;
    std::vector<float> devcopy;
    for (auto& kv : h->captured) {
        const std::string& name = kv.first; const auto& v = kv.second;
        const bool is3 = name.size() > 3 && name.compare(name.size() - 3, 3, "#x3") == 0, is6 = name.size() > 3 && name.compare(name.size() - 3, 3, "#x6") == 0;
        if (is3 || is6) {      // split [hi | lo] (or [p0 | p1 | p2]) pack: v = index code per bf16 slot
            const size_t np = is6 ? 3 : 2;
            const std::string real_name = name.substr(0, name.size() - 3);
            auto it = h->dev.find(real_name);
            if (it == h->dev.end() || it->second.bytes != (np * v.size() + (np * v.size() & 1)) * 2)
                return fail(DESIRE_ERR_STATE, "repack map: split operand " + real_name + " changed shape");
            std::vector<uint16_t> dev16(np * v.size() + (np * v.size() & 1));
            HIPCHK(hipMemcpy(dev16.data(), it->second.p, it->second.bytes, hipMemcpyDeviceToHost));
            std::vector<uint32_t> ix(v.size());
            for (size_t i = 0; i < v.size(); ++i) {
                uint32_t u; std::memcpy(&u, &v[i], 4);
                if (u > h->n_params) return fail(DESIRE_ERR_STATE, "repack map: split operand " + real_name + " is not a gather of the weights");
                ix[i] = u;
                float want = u ? flat[u - 1] : 0.f;
                for (size_t pc = 0; pc < np; ++pc) {
                    const uint16_t b = bf16_rne(want);
                    if (dev16[pc * v.size() + i] != b)
                        return fail(DESIRE_ERR_STATE, "repack map: split operand " + real_name + " does not match its map");
                    want -= bf16_to_f32(b);
                }
            }
            segs16.push_back(Seg16{static_cast<uint16_t*>(it->second.p), (unsigned long long)all_idx.size(), (unsigned long long)v.size(), (unsigned long long)np});
            all_idx.insert(all_idx.end(), ix.begin(), ix.end());
            while (all_idx.size() % 4) all_idx.push_back(0);
            continue;
        }
        auto it = h->dev.find(name);
        if (it == h->dev.end() || it->second.bytes != v.size() * sizeof(float))
            return fail(DESIRE_ERR_STATE, "repack map: operand " + name + " changed shape");
        devcopy.resize(v.size());
        HIPCHK(hipMemcpy(devcopy.data(), it->second.p, it->second.bytes, hipMemcpyDeviceToHost));
        bool gather = true;
        std::vector<uint32_t> ix(v.size());
        for (size_t i = 0; i < v.size() && gather; ++i) {
            uint32_t u; std::memcpy(&u, &v[i], 4);
            if (u > h->n_params) { gather = false; break; }
            ix[i] = u;
            const float want = u ? flat[u - 1] : 0.f;
            if (std::memcmp(&want, &devcopy[i], 4) != 0) gather = false;
        }
        const bool folded = name.size() > 6 && (name.rfind("/scale") == name.size() - 6 || name.rfind("/shift") == name.size() - 6);
        if (!gather) {
            if (folded) continue;                              // handled by k_refold
            return fail(DESIRE_ERR_STATE, "repack map: operand " + name + " is not a gather of the weights");
        }
        if (folded) continue;
        segs.push_back(Seg{it->second.f(), (unsigned long long)all_idx.size(), (unsigned long long)v.size()});
        all_idx.insert(all_idx.end(), ix.begin(), ix.end());
        while (all_idx.size() % 4) all_idx.push_back(0);
    }
    h->captured.clear();
    if (ensure(h, "repack_idx", all_idx.size() * sizeof(uint32_t)) || ensure(h, "repack_segs", segs.size() * sizeof(Seg)) ||
        ensure(h, "repack_segs16", std::max<size_t>(1, segs16.size()) * sizeof(Seg16)))
        return fail(DESIRE_ERR_HIP, "hipMalloc failed for the repack maps");
    if (!segs16.empty()) HIPCHK(hipMemcpy(h->ws["repack_segs16"].p, segs16.data(), segs16.size() * sizeof(Seg16), hipMemcpyHostToDevice));
    h->n_seg16 = (int)segs16.size();
    HIPCHK(hipMemcpy(h->ws["repack_idx"].p, all_idx.data(), all_idx.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->ws["repack_segs"].p, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice));
    h->n_seg = (int)segs.size();
    return DESIRE_OK;
}

int repack(desire_ctx* h, hipStream_t s) {
    hipLaunchKernelGGL(k_repack, dim3(64, h->n_seg), dim3(256), 0, s, static_cast<const Seg*>(h->ws["repack_segs"].p),
                       static_cast<const uint32_t*>(h->ws["repack_idx"].p), W(h, "Wflat"));
    if (h->n_seg16)
        hipLaunchKernelGGL(k_repack_split, dim3(64, h->n_seg16), dim3(256), 0, s, static_cast<const Seg16*>(h->ws["repack_segs16"].p),
                           static_cast<const uint32_t*>(h->ws["repack_idx"].p), W(h, "Wflat"));
    for (const char* n : {"vae_enc/conv1", "vae_enc/conv2", "vae_enc/conv3", "vae_dec/deconv1", "vae_dec/deconv2",
                          "vae_dec/deconv3", "vae_dec/deconv4"}) {
        const std::string p(n);
        const int C = (int)h->slots.at(p + "/b").n;
        hipLaunchKernelGGL(k_refold, dim3((C + 63) / 64), dim3(64), 0, s, W(h, "Wflat"), h->dev.at(p + "/shift").f(),
                           h->dev.at(p + "/scale").f(), h->slots.at(p + "/b").off, h->slots.at(p + "/bn/beta").off,
                           h->slots.at(p + "/bn/moving_mean").off, C);
    }
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

}  // namespace

extern "C" int desire_set_training(desire_handle* h, int enable) {
    if (int rc = desire_ready(h)) return rc;
    if ((enable != 0) != h->training) { h->cp_pending = false; h->cp_enc = false; }      // compaction maps: inference and training learn the counts differently (a new desire_encode comes first)
    if (!enable) {
        if (h->training) {          // the trained master copy becomes the handle's weights: desire_get_weight and a later
            std::vector<float> flat(h->n_params);          // desire_set_training(h, 1) start from it (Adam moments restart at zero)
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemcpy(flat.data(), W(h, "Wflat"), flat.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (auto& kv : h->slots) h->host_w[kv.first].assign(flat.begin() + kv.second.off, flat.begin() + kv.second.off + kv.second.n);
        }
        h->training = false;
        return DESIRE_OK;
    }
    const desire_dims& d = h->d;
    if (!d.posterior) return fail(DESIRE_ERR_STATE, "training needs the posterior path (dims.posterior = 1)");
    if (d.bf16 == 1 || d.bf16 == 3) return fail(DESIRE_ERR_STATE, "training runs with dims.bf16 = 0 (fp32 operands) or 2 (split-bf16 operands where a kernel has that form, fp32 kernels elsewhere); 1 and 3 are inference-only");
    if (d.ref_compat) return fail(DESIRE_ERR_STATE, "ref_compat is forward-only: the reference never defines a runnable cost (model/model.py:342)");
    if (d.mno > 128) return fail(DESIRE_ERR_STATE, "training supports up to 128 agents per scene (160 .. 256 run the step-wise IOC: inference)");
    if (ioc_uses_cluster(d.mno, d.H, d.grid_size * d.grid_size, 0) && (d.H > 128 || d.grid_size > 4))
        return fail(DESIRE_ERR_STATE, "training of groups larger than one workgroup tile (64 / 96 / 128 agents: cluster-form BPTT) needs H <= 128 and grid_size <= 4");
    if (d.iters > 4) return fail(DESIRE_ERR_STATE, "training keeps the activations of every IOC refinement pass: iters <= 4");
    if (d.T_pred > d.H) return fail(DESIRE_ERR_STATE, "training needs T_pred <= H");
    if (d.grid_size > 4 && d.mno > 32)
        return fail(DESIRE_ERR_STATE, "training with more than 16 social bins needs mno <= 32 (LDS budget of the 64-row IOC backward tile)");
    if (h->slots.empty()) {
        size_t off = 0;
        for (auto& kv : h->want) { h->slots[kv.first] = WSlot{off, kv.second}; off += (kv.second + 3) / 4 * 4; }
        h->n_params = off;
    }
    const size_t R = h->R, T = d.T_pred, H = d.H, f = sizeof(float);
    const size_t RS = R + 128;                                  // IOC buffers: rows + slack for the partial padded tiles of the slot classes (DESIRE_FLAG_COMPACT_IOC)
    const size_t Tm = d.T_pred > d.T_obs ? d.T_pred : d.T_obs;
    const size_t NP = d.iters;                                  // IOC passes: each keeps its own saves
    struct B { const char* n; size_t bytes; };
    const B bufs[] = {
        {"Gflat", h->n_params * f}, {"nvalid", 4 * f}, {"tn_partial", (size_t)96 << 20},
        {"dec_sv_r", R * T * H * f}, {"dec_sv_u", R * T * H * f}, {"dec_sv_c", R * T * H * f}, {"dec_sv_h", R * T * H * f},
        {"dY0", R * T * 2 * f}, {"dec_dag", R * T * 2 * H * f}, {"dec_dac", R * T * H * f}, {"dec_rh", R * T * H * f},
        {"dec_hprev", R * T * H * f}, {"dec_dxg", R * 2 * H * f}, {"dec_dxc", R * H * f}, {"dxz", R * H * f},
        {"dHx_rows", R * H * f}, {"mask_sv_p", R * H * f}, {"dq_mask", R * H * f},
        {"dconv4", R * 1024 * f}, {"dconv3", R * 8192 * f}, {"dconv2", R * 4096 * f}, {"dconv1", R * 2048 * f},
        {"dz", R * d.L * f}, {"dparams", (size_t)h->A * 2 * d.L * f}, {"dconvE3", (size_t)h->A * 2048 * f},
        {"dconvE2", (size_t)h->A * 4096 * f}, {"dconvE1", (size_t)h->A * 8192 * f}, {"dq_c", (size_t)h->A * h->V * f},
        {"dHxHy", (size_t)h->A * 2 * H * f},
        {"ioc_sv_x", NP * RS * T * (size_t)h->E * f}, {"ioc_sv_r", NP * RS * T * H * f}, {"ioc_sv_u", NP * RS * T * H * f}, {"ioc_sv_c", NP * RS * T * H * f},
        {"ioc_sv_h", NP * RS * T * H * f}, {"ioc_Yin", NP * RS * T * 2 * f}, {"dscore0", RS * f},
        {"Y_ref", R * T * 2 * f}, {"score_sv", R * f}, {"dYr", R * T * 2 * f}, {"dscore", R * f},
        {"dscoreT", R * T * f}, {"ioc_dag", RS * T * 2 * H * f}, {"ioc_dac", RS * T * H * f}, {"ioc_rh", RS * T * H * f},
        {"ioc_hprev", RS * T * H * f}, {"ioc_dpre_r", RS * T * H * f}, {"ioc_dpre_v", RS * T * d.E_v * f}, {"ioc_vel", RS * T * 2 * f},
        {"ioc_pooled", RS * T * (size_t)h->B * H * f}, {"ioc_pool_flags", RS * T * sizeof(unsigned long long)},
        {"bin_counts", ((RS * T + 2047) / 2048) * (size_t)h->B * sizeof(int)}, {"bin_base", ((size_t)h->B + 1) * sizeof(int)},
        {"bin_total", (size_t)h->B * sizeof(int)}, {"bin_list", H == 128 ? RS * T * (size_t)h->B * sizeof(int) : 4},
        {"enc_dag", (size_t)h->A * Tm * 2 * H * f}, {"enc_dac", (size_t)h->A * Tm * H * f}, {"enc_rh", (size_t)h->A * Tm * H * f},
        {"enc_hprev", (size_t)h->A * Tm * H * f},
        {"ex_sv_r", (size_t)h->A * d.T_obs * H * f}, {"ex_sv_u", (size_t)h->A * d.T_obs * H * f}, {"ex_sv_c", (size_t)h->A * d.T_obs * H * f},
        {"ex_sv_h", (size_t)h->A * d.T_obs * H * f}, {"ex_sv_x", (size_t)h->A * d.T_obs * 2 * f},
        {"ey_sv_r", (size_t)h->A * T * H * f}, {"ey_sv_u", (size_t)h->A * T * H * f}, {"ey_sv_c", (size_t)h->A * T * H * f},
        {"ey_sv_h", (size_t)h->A * T * H * f}, {"ey_sv_x", (size_t)h->A * T * 2 * f},
    };
    for (const B& b : bufs)
        if (ensure(h, b.n, b.bytes)) return fail(DESIRE_ERR_HIP, std::string("hipMalloc failed for training buffer ") + b.n);
    if (d.bn_mode != 0) {          // pre-norm copies of the seven conv layers (instance-norm / batch-norm backward)
        const B pre[] = {{"conv1_pre", (size_t)h->A * 8192 * f}, {"conv2_pre", (size_t)h->A * 4096 * f}, {"conv3_pre", (size_t)h->A * 2048 * f},
                         {"deconv1_pre", R * 2048 * f}, {"deconv2_pre", R * 4096 * f}, {"deconv3_pre", R * 8192 * f}, {"deconv4_pre", R * 1024 * f}};
        for (const B& b : pre)
            if (ensure(h, b.n, b.bytes)) return fail(DESIRE_ERR_HIP, std::string("hipMalloc failed for training buffer ") + b.n);
    }
    if (d.bn_mode == 2 && (ensure(h, "bn_part2", (size_t)512 * 256 * f) || ensure(h, "bn_stat2", (size_t)2 * 128 * f) || ensure(h, "bn_statb", (size_t)2 * 128 * f)))
        return fail(DESIRE_ERR_HIP, "hipMalloc failed for the batch-norm backward scratch");
    if (ensure(h, "head_nll", (size_t)h->A * d.T_obs * f) || ensure(h, "head_cnt", (size_t)h->A * d.T_obs * f) || ensure(h, "head_dO", (size_t)h->A * d.T_obs * 5 * f))
        return fail(DESIRE_ERR_HIP, "hipMalloc failed for the Gaussian-head loss buffers");
    if (ensure(h, "loss_pa", (size_t)h->A * 4 * f) || ensure(h, "loss_out", 8 * f) || ensure(h, "bias_part", ((RS + 31) / 32 + 1) * 4 * H * f))
        return fail(DESIRE_ERR_HIP, "hipMalloc failed for the loss / bias-gradient buffers");
    HIPCHK(hipMemset(h->ws["loss_out"].p, 0, 8 * f));
    if (int rc = build_repack_maps(h)) return rc;
    h->adam_t = 0;
    h->training = true;
    return DESIRE_OK;
}

extern "C" int desire_backward(desire_handle* h, const float* dev_past, const float* dev_fut, const float* dev_eps, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!h->training) return fail(DESIRE_ERR_STATE, "desire_set_training(h, 1) and a training-mode desire_forward come first");
    if (!dev_past || !dev_fut || !dev_eps) return fail(DESIRE_ERR_ARG, "null argument");
    const desire_dims& d = h->d;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = d.H, T = d.T_pred;
    const long R = h->R;
    // DESIRE_FLAG_COMPACT_ROWS: the training-mode forward ran the per-row sample-generation stages on the K*P rows of the P present agents
    // (compact row order r' = k*P + a', one pseudo-scene of P slots: kernels_compact.hip) and left every save of those stages in that order;
    // their whole backward runs on the same rows.  Rs / mnos / ns_s are the (rows, slots, scenes) those stages see; the IOC module and the
    // encoders keep the caller's layout.  A gradient enters the compact domain once (dY0) and leaves it twice (dparams, dHx).
    const bool compact = h->cp_last;
    const int P = compact ? h->cp_P : 0;
    const long Rs = compact ? (long)P * d.K : R;
    const int mnos = compact ? P : d.mno, ns_s = compact ? 1 : d.n_scenes;
    if (compact && (ensure(h, "cp_dY0", (size_t)R * T * 2 * sizeof(float)) || ensure(h, "cp_dHx_rows", (size_t)R * H * sizeof(float)) ||
                    ensure(h, "cp_dHx", (size_t)h->A * H * sizeof(float))))
        return fail(DESIRE_ERR_HIP, "hipMalloc failed for the compact-row gradient buffers");
    const int32_t* amap = compact ? static_cast<const int32_t*>(h->ws["cp_amap"].p) : nullptr;
    const bool enc_c = compact && h->cp_enc;                // the encoder stack ran on the present agents as well (desire_encode)
    const int Ae = enc_c ? P : h->A;
    if (enc_c && (ensure(h, "cp_dparams", (size_t)h->A * 2 * d.L * sizeof(float)) || ensure(h, "cp_dHxHy", (size_t)h->A * 2 * H * sizeof(float)) ||
                  ensure(h, "dHxHy_ioc", (size_t)h->A * H * sizeof(float))))
        return fail(DESIRE_ERR_HIP, "hipMalloc failed for the compact encoder gradient buffers");
    const float* HxE = enc_c ? W(h, "cp_HxHy") : W(h, "HxHy");
    float* dHE = enc_c ? W(h, "cp_dHxHy") : W(h, "dHxHy");
    const float* HxS = compact ? W(h, "cp_HxHy") : W(h, "HxHy");
    float* dHxS = compact ? W(h, "cp_dHx_rows") : W(h, "dHx_rows");
    launch_fill_f32(W(h, "Gflat"), h->n_params, 0.f, s);
    // loss mask: present at the last observed frame and in at least one target frame; every loss term below is masked per
    // target frame (model/model.py:351-366).  `valid` (presence at the last observed frame) stays what social pooling uses.
    const uint8_t* valid = static_cast<const uint8_t*>(h->ws["lmask"].p);
    launch_loss_mask(static_cast<const uint8_t*>(h->ws["valid"].p), dev_fut, static_cast<uint8_t*>(h->ws["lmask"].p), W(h, "nfut"),
                     d.n_scenes, d.mno, T, s);
    launch_count_valid(valid, h->A, W(h, "nvalid"), s);
    // ---- sample-generation module ----
    launch_loss_grad_y(W(h, "Y0"), dev_fut, valid, W(h, "nfut"), W(h, "nvalid"), W(h, "dY0"), d.n_scenes, d.mno, d.K, T, d.sx, d.sy, s);
    if (compact) launch_gather_rows(W(h, "dY0"), W(h, "cp_dY0"), amap, P, d.K, d.mno, T * 2, s);
    const float* dY0s = compact ? W(h, "cp_dY0") : W(h, "dY0");
    const int n_tiles32 = (int)((R + 31) / 32), n_tiles32s = (int)((Rs + 31) / 32);
    if (Rs > 0) {
    DecBwdArgs b{};
    b.dY0 = dY0s; b.sv_r = W(h, "dec_sv_r"); b.sv_u = W(h, "dec_sv_u"); b.sv_c = W(h, "dec_sv_c"); b.sv_h = W(h, "dec_sv_h");
    b.Hx = HxS; b.ldhx = 2 * H; b.w_head = D(h, "head/w");
    b.WcT_h = D4(h, "dec/WcT_h"); b.WgT_h = D4(h, "dec/WgT_h"); b.WgT_x = D4(h, "dec/WgT_x"); b.WcT_x = D4(h, "dec/WcT_x");
    b.R = (int)Rs; b.K = d.K; b.mno = mnos; b.T = T; b.H = H;
    b.dag = W(h, "dec_dag"); b.dac = W(h, "dec_dac"); b.rh = W(h, "dec_rh"); b.hprev = W(h, "dec_hprev");
    b.dxg = W(h, "dec_dxg"); b.dxc = W(h, "dec_dxc"); b.dxz = W(h, "dxz"); b.dHx_rows = dHxS;
    // bias gradients = column sums of the gate-gradient streams: summed per tile inside the BPTT kernels (no further pass over the streams)
    b.bias_part = W(h, "bias_part");                            // (allocated by desire_set_training: no hipMalloc inside a call that may be under stream capture)
    { Timer t(h, s, "bwd_decoder"); launch_decoder_bwd(b, s); }
    launch_reduce_parts(b.bias_part, n_tiles32s, 3 * H, 0, 2 * H, G(h, "dec/gates/bias"), 0, s);
    launch_reduce_parts(b.bias_part, n_tiles32s, 3 * H, 2 * H, H, G(h, "dec/candidate/bias"), 0, s);
    {
        Timer t(h, s, "bwd_decoder_wgrad");
        tn(h, W(h, "dec_sv_h"), H, dY0s, 2, Rs * T, H, 2, G(h, "head/w"), 2, 0, s);
        colsum(h, dY0s, 2, Rs * T, 2, G(h, "head/b"), 0, s);
        float* gk = G(h, "dec/gates/kernel");        // [(H+H), 2H]
        tn(h, W(h, "xz"), H, W(h, "dec_dxg"), 2 * H, Rs, H, 2 * H, gk, 2 * H, 0, s);
        tn(h, W(h, "dec_hprev"), H, W(h, "dec_dag"), 2 * H, Rs * T, H, 2 * H, gk + (size_t)H * 2 * H, 2 * H, 0, s);
        float* ck = G(h, "dec/candidate/kernel");    // [(H+H), H]
        tn(h, W(h, "xz"), H, W(h, "dec_dxc"), H, Rs, H, H, ck, H, 0, s);
        tn(h, W(h, "dec_rh"), H, W(h, "dec_dac"), H, Rs * T, H, H, ck + (size_t)H * H, H, 0, s);
    }
    }       // Rs > 0
    if (compact) launch_fill_f32(W(h, "dHx_rows"), (size_t)R * H, 0.f, s);      // the IOC module accumulates into it; the decoder no longer initialises it
    const int V = h->V, L = d.L, A = h->A;
    const bool bn1 = d.bn_mode != 0;                       // batch statistics -- per object (mode 1, the reference graph's batch of one) or over
                                                           // the whole batch (mode 2): the conv data-gradient kernels run with a linear epilogue and
                                                           // norm_bwd takes the gradient through activation + normalisation
    auto norm_bwd = [&](float* dy, const float* pre, const float* y, int n, int P, int C, const float* gamma, int sig) {
        if (d.bn_mode == 2) launch_batchnorm_act_bwd(dy, pre, y, (size_t)n, P, C, gamma, sig, W(h, "bn_part2"), W(h, "bn_statb"), W(h, "bn_stat2"), s);
        else launch_instnorm_act_bwd(dy, pre, y, n, P, C, gamma, sig, s);
    };
    // (under per-object statistics a conv bias cancels against the mean: its gradient is exactly zero, so the column sums of the
    //  post-norm gradients -- pure rounding noise -- are NOT fed to Adam; the Gflat slots of vae_*/b stay at the fill value 0)
    // ---- ranking / refinement module (trajectories detached: its only path into the rest is dHx) ----
    {
        Timer t(h, s, "bwd_ioc");
        const int E = h->E, B = h->B;
        launch_loss_grad_y(W(h, "Y_ref"), dev_fut, valid, W(h, "nfut"), W(h, "nvalid"), W(h, "dYr"), d.n_scenes, d.mno, d.K, T, d.sx, d.sy, s);
        launch_score_grad(W(h, "Y0"), dev_fut, W(h, "score_sv"), valid, W(h, "nvalid"), W(h, "dscore"), W(h, "dscoreT"), d.n_scenes,
                          d.mno, d.K, T, d.sx, d.sy, s);
        // one BPTT per refinement pass, last pass first: Y_final = Y0 + sum_p dY_p, so every pass's regression head sees the same
        // dL/dY_final; only the last pass's scores enter the loss.  Weight gradients of the passes accumulate.
        // DESIRE_FLAG_COMPACT_IOC: the forward ran one launch sequence per slot class (api.hip: IocView) and left each class's saves at its row
        // offset; the BPTT and every weight-gradient reduction below run per class on the same views, accumulating.  Otherwise: one view, the
        // handle's own shape.
        struct BView { long R; int mno, n_scenes; const float* Hx; const float* p_last; const uint8_t* valid; size_t row_off; const int32_t* cmap; int gpt, ngrp; };
        std::vector<BView> views;
        if (h->ci_last) {
            if (ensure(h, "ci_dYr", (size_t)(R + 128) * T * 2 * sizeof(float)) || ensure(h, "ci_dscore", (size_t)(R + 128) * sizeof(float)) || ensure(h, "ci_dscoreT", (size_t)(R + 128) * T * sizeof(float)) ||
                ensure(h, "ci_dHx_rows", (size_t)(R + 128) * H * sizeof(float)) || ensure(h, "ci_dHx", (size_t)h->A * H * sizeof(float)) || ensure(h, "dHxHy_ioc", (size_t)h->A * H * sizeof(float)))
                return fail(DESIRE_ERR_HIP, "hipMalloc failed for the slot-class gradient buffers");
            launch_fill_f32(W(h, "dHxHy_ioc"), (size_t)h->A * H, 0.f, s);
            int m4[4];
            compact_classes(h, m4);
            size_t aoff = 0, roff = 0;
            for (int i = 0; i < h->ci_n; ++i) {
                const int c = h->ci_cls[i], n_c = h->ci_cnt[i], m_c = m4[c];
                const int gpt = (m_c <= 32 && 32 % m_c) ? 32 / m_c : 0, ngrp = n_c * d.K;          // padded tiles: as desire_ioc_refine seated the class
                const long R_c = gpt ? (long)((ngrp + gpt - 1) / gpt) * 32 : (long)n_c * d.K * m_c;
                views.push_back(BView{R_c, m_c, n_c, W(h, "ci_Hx") + aoff * 2 * H, W(h, "ci_pl") + aoff * 2,
                                      static_cast<const uint8_t*>(h->ws["ci_valid"].p) + aoff, roff, static_cast<const int32_t*>(h->ws["ci_map"].p) + (size_t)c * h->A, gpt, ngrp});
                aoff += (size_t)n_c * m_c; roff += (size_t)R_c;
            }
        } else
            views.push_back(BView{R, d.mno, d.n_scenes, W(h, "HxHy"), W(h, "p_last"), static_cast<const uint8_t*>(h->ws["valid"].p), 0, nullptr, 0, 0});
        const long RTf = (R + 128) * T;                    // stride of a refinement pass's saves (desire_set_training: rows + slack)
        launch_fill_f32(W(h, "dscore0"), (size_t)R, 0.f, s);
        bool first = true;                          // the first launch sequence writes the weight gradients, the others accumulate
        for (size_t vi = 0; vi < views.size(); ++vi) {
        const BView& v = views[vi];
        const long Rv = v.R, RT = Rv * T;
        const int n_tiles32v = (int)((Rv + 31) / 32);
        float* dYr_v = W(h, "dYr"); float* dscore_v = W(h, "dscore"); float* dscoreT_v = W(h, "dscoreT"); float* dHx_v = W(h, "dHx_rows");
        if (v.cmap) {          // the class's rows of the loss gradients; its own d loss / d Hx rows
            dYr_v = W(h, "ci_dYr"); dscore_v = W(h, "ci_dscore"); dscoreT_v = W(h, "ci_dscoreT"); dHx_v = W(h, "ci_dHx_rows");
            launch_cls_rows(W(h, "dYr"), dYr_v, v.cmap, v.n_scenes, v.mno, d.K, d.mno, 2 * T, 0, s, v.gpt);
            launch_cls_rows(W(h, "dscore"), dscore_v, v.cmap, v.n_scenes, v.mno, d.K, d.mno, 1, 0, s, v.gpt);
            launch_cls_rows(W(h, "dscoreT"), dscoreT_v, v.cmap, v.n_scenes, v.mno, d.K, d.mno, T, 0, s, v.gpt);
            launch_fill_f32(dHx_v, (size_t)Rv * H, 0.f, s);
        }
        for (int p = d.iters - 1; p >= 0; --p) {
            const bool last_pass = p == d.iters - 1;
            const int acc = first ? 0 : 1;
            const int acc_s = vi > 0 ? 1 : 0;         // (score weights: the last pass of every view)
            first = false;
            const size_t po = (size_t)p * RTf + v.row_off * T;
            const float* sv_h = W(h, "ioc_sv_h") + po * H;
            const float* sv_x = W(h, "ioc_sv_x") + po * E;
            IocBwdArgs q{};
            q.Y0 = W(h, "ioc_Yin") + po * 2; q.p_last = v.p_last; q.valid = v.valid; q.Hx = v.Hx; q.ldhx = 2 * H;
            q.dYr = dYr_v; q.dscore = last_pass ? dscore_v : W(h, "dscore0");
            q.sv_x = sv_x; q.sv_r = W(h, "ioc_sv_r") + po * H; q.sv_u = W(h, "ioc_sv_u") + po * H; q.sv_c = W(h, "ioc_sv_c") + po * H; q.sv_h = sv_h;
            q.w_score = D(h, "ioc/score_w");
            q.R = (int)Rv; q.K = d.K; q.mno = v.mno; q.T = T; q.H = H; q.G = d.grid_size; q.nb_w = d.nb_w; q.nb_h = d.nb_h;
            q.gpt = v.gpt; q.ngrp = v.ngrp;
            q.WrT = D4(h, "ioc/WrT"); q.WcT_h = D4(h, "ioc/WcT_h"); q.WcT_er = D4(h, "ioc/WcT_er"); q.WcT_ev = D4(h, "ioc/WcT_ev");
            q.WgT_h = D4(h, "ioc/WgT_h"); q.WgT_er = D4(h, "ioc/WgT_er"); q.WgT_ev = D4(h, "ioc/WgT_ev"); q.WsT = D4(h, "ioc/WsT"); q.WsT_c = D4(h, "ioc/WsT_c");
            q.dag = W(h, "ioc_dag"); q.dac = W(h, "ioc_dac"); q.rh = W(h, "ioc_rh"); q.hprev = W(h, "ioc_hprev");
            q.dpre_r = W(h, "ioc_dpre_r"); q.dpre_v = W(h, "ioc_dpre_v"); q.vel = W(h, "ioc_vel"); q.pooled = W(h, "ioc_pooled");
            q.pool_flags = static_cast<unsigned long long*>(h->ws["ioc_pool_flags"].p);
            q.dHx_rows = dHx_v;
            q.bin_tab = d.bin_mode == 1 ? W(h, "bin_tab") : nullptr;
            const bool cl_bwd = ioc_uses_cluster(v.mno, d.H, d.grid_size * d.grid_size, 0);
            q.bias_part = cl_bwd ? nullptr : W(h, "bias_part");           // (the cluster form keeps the separate column-sum passes)
            if (cl_bwd) {
                const size_t n_groups = (size_t)Rv / v.mno;
                HIPCHK(hipMemsetAsync(h->ws["grp_cnt"].p, 0, n_groups * sizeof(int), s));
                if (vi == 0 && last_pass) HIPCHK(hipMemsetAsync(h->ws["ioc_err"].p, 0, sizeof(int), s));
                if (launch_ioc_bwd_cluster(q, static_cast<int*>(h->ws["grp_cnt"].p), static_cast<int*>(h->ws["ioc_err"].p), s))
                    return fail(DESIRE_ERR_STATE, "cluster-form IOC backward does not serve this shape");
            } else if (d.bf16 == 2 && (train_x3_mask(h) & 4) && ioc_bwd_x3_supported(v.mno, H)) {      // split-bf16 operands in the data-gradient contractions
                q.WcT_h = D4(h, "ioc/WcT16"); q.WgT_h = D4(h, "ioc/WgT16"); q.WsT = D4(h, "ioc/WsT16");
#ifdef DESIRE_IOC_TIMING
                if (!h->ws.count("dbgb")) { h->ws["dbgb"].alloc(12 * sizeof(long long)); }
                q.dbg = static_cast<long long*>(h->ws["dbgb"].p);
#endif
                launch_ioc_bwd_x3(q, s);
#ifdef DESIRE_IOC_TIMING
                {
                    long long host[12];
                    (void)hipStreamSynchronize(s);
                    (void)hipMemcpy(host, q.dbg, sizeof(host), hipMemcpyDeviceToHost);
                    const char* nm[12] = {"loop tail (dh)", "bar top", "P0 pos/clear/load h", "bar P0", "P1 masks + part 1 (loads, stores, images)", "barriers after parts",
                                          "t2 mma + dar", "gates mma + dpr", "pooled rebuild + store", "dpool mma + tile write", "bin barrier", "gather / NB"};
                    long long tot = 0; for (int k = 0; k < 12; ++k) tot += host[k];
                    fprintf(stderr, "k_ioc_bwd_x3 block 7 wave 0: total %lld cycles\n", tot);
                    for (int k = 0; k < 12; ++k) fprintf(stderr, "  %-45s %12lld  %5.1f %%\n", nm[k], host[k], 100.0 * host[k] / (double)tot);
                }
#endif
            } else
            launch_ioc_bwd(q, s);
            tn(h, sv_h + (size_t)(T - 1) * H, T * H, dYr_v, 2 * T, Rv, H, 2 * T, G(h, "ioc/reg/w"), 2 * T, acc, s);
            colsum(h, dYr_v, 2 * T, Rv, 2 * T, G(h, "ioc/reg/b"), acc, s);
            if (last_pass) {
                tn(h, sv_h, H, dscoreT_v, 1, RT, H, 1, G(h, "ioc/score/w"), 1, acc_s, s);
                colsum(h, dscoreT_v, 1, RT, 1, G(h, "ioc/score/b"), acc_s, s);
            }
            float* gk = G(h, "ioc/gates/kernel");            // [(E+H), 2H]
            tn(h, sv_x, E, W(h, "ioc_dag"), 2 * H, RT, E, 2 * H, gk, 2 * H, acc, s);
            tn(h, W(h, "ioc_hprev"), H, W(h, "ioc_dag"), 2 * H, RT, H, 2 * H, gk + (size_t)E * 2 * H, 2 * H, acc, s);
            if (cl_bwd) colsum(h, W(h, "ioc_dag"), 2 * H, RT, 2 * H, G(h, "ioc/gates/bias"), acc, s);
            else launch_reduce_parts(q.bias_part, n_tiles32v, 4 * H, 0, 2 * H, G(h, "ioc/gates/bias"), acc, s);
            float* ck = G(h, "ioc/candidate/kernel");        // [(E+H), H]
            tn(h, sv_x, E, W(h, "ioc_dac"), H, RT, E, H, ck, H, acc, s);
            tn(h, W(h, "ioc_rh"), H, W(h, "ioc_dac"), H, RT, H, H, ck + (size_t)E * H, H, acc, s);
            if (cl_bwd) colsum(h, W(h, "ioc_dac"), H, RT, H, G(h, "ioc/candidate/bias"), acc, s);
            else launch_reduce_parts(q.bias_part, n_tiles32v, 4 * H, 2 * H, H, G(h, "ioc/candidate/bias"), acc, s);
            const unsigned long long* pflags = static_cast<const unsigned long long*>(h->ws["ioc_pool_flags"].p);
            if (H == 128 && (size_t)RT * (size_t)B < ((size_t)1 << 31)) {      // (list positions are ints)
                // one output tile row = one bin (128 columns): each contracts only the (row, t) pairs that hold a neighbour in ITS bin, from
                // per-bin row lists built out of the flags -- 23 % of the rows at the bench's density, where skipping whole 32-row chunks
                // by their OR-ed flags still visited about half of them, zero rows and all
                int* bl_counts = static_cast<int*>(h->ws["bin_counts"].p); int* bl_base = static_cast<int*>(h->ws["bin_base"].p);
                int* bl_total = static_cast<int*>(h->ws["bin_total"].p); int* bl_list = static_cast<int*>(h->ws["bin_list"].p);
                launch_bin_lists(pflags, RT, B, bl_counts, bl_base, bl_total, bl_list, s);
                tn(h, W(h, "ioc_pooled"), B * H, W(h, "ioc_dpre_r"), H, RT, B * H, H, G(h, "ioc/social_fc/w"), H, acc, s, pflags, H, bl_list, bl_base, bl_total);
            } else
                tn(h, W(h, "ioc_pooled"), B * H, W(h, "ioc_dpre_r"), H, RT, B * H, H, G(h, "ioc/social_fc/w"), H, acc, s, pflags, H);   // empty (row, t, bin) blocks are skipped
            if (cl_bwd) colsum(h, W(h, "ioc_dpre_r"), H, RT, H, G(h, "ioc/social_fc/b"), acc, s);
            else launch_reduce_parts(q.bias_part, n_tiles32v, 4 * H, 3 * H, H, G(h, "ioc/social_fc/b"), acc, s);
            tn(h, W(h, "ioc_vel"), 2, W(h, "ioc_dpre_v"), d.E_v, RT, 2, d.E_v, G(h, "ioc/vel_fc/w"), d.E_v, acc, s);
            colsum(h, W(h, "ioc_dpre_v"), d.E_v, RT, d.E_v, G(h, "ioc/vel_fc/b"), acc, s);
        }
        if (v.cmap) {          // the class's share of d loss / d Hx: rows -> class agents -> agents (padding slots dropped)
            launch_fill_f32(W(h, "ci_dHx"), (size_t)v.n_scenes * v.mno * H, 0.f, s);
            launch_rows_to_agents(dHx_v, W(h, "ci_dHx"), H, v.n_scenes, v.mno, d.K, H, s, v.gpt);
            launch_cls_scatter_add_agents(W(h, "ci_dHx"), H, W(h, "dHxHy_ioc"), H, v.cmap, v.n_scenes * v.mno, H, s);
        }
        }       // views
    }
    // ---- mask fc ----
    if (Rs > 0) {
        Timer t(h, s, "bwd_mask");
        launch_mask_bwd(W(h, "mask_sv_p"), W(h, "dxz"), HxS, 2 * H, W(h, "dq_mask"), dHxS, (int)Rs, H, h->Hl, d.K, mnos, s);
        colsum(h, W(h, "dq_mask"), H, Rs, H, G(h, "mask_fc/b"), 0, s);
        tn(h, W(h, "xhat"), V, W(h, "dq_mask"), H, Rs, V, H, G(h, "mask_fc/w"), H, 0, s);
        GemmArgs g{};
        g.A = W(h, "dq_mask"); g.lda = H; g.M = (int)Rs; g.K = H; g.Bp = D4(h, "mask/WT"); g.G = H / 8; g.NT = V / 32;
        g.out = W(h, "dconv4"); g.ldo = V; g.N = V; g.p0 = D(h, "vae_dec/deconv4/scale"); g.chmod = 1; g.aux = W(h, "xhat");
        if (bn1) {          // per-object batch-norm: gradient w.r.t. the layer OUTPUT first, then through activation + instance norm
            launch_gemm_rows(g, EPI_NONE, s);
            norm_bwd(W(h, "dconv4"), W(h, "deconv4_pre"), W(h, "xhat"), (int)Rs, 1024, 1, D(h, "vae_dec/deconv4/gamma"), 1);
        } else
        launch_gemm_rows(g, EPI_SIGGRAD, s);
    }
    // ---- CVAE decoder (each data gradient = the forward kernel of the mirrored layer with a gradient epilogue) ----
    if (Rs > 0) {
        Timer t(h, s, "bwd_cvae_dec");
        const int NSL = 78;
        launch_w1ch_grad(W(h, "dconv4"), W(h, "d3"), (int)Rs, Rs < 2048 ? (int)Rs : 2048, W(h, "tn_partial"), G(h, "vae_dec/deconv4/w"), s);
        if (!bn1) colsum(h, W(h, "dconv4"), 1, Rs * 1024, 1, G(h, "vae_dec/deconv4/b"), 0, s);
        ConvArgs c{};
        c.n = (int)Rs;
        c.in = W(h, "dconv4"); c.out = W(h, "dconv3"); c.w_raw = D(h, "vae_dec/deconv4/raw");
        c.scale = D(h, "vae_dec/deconv3/scale"); c.shift = c.scale; c.mode = bn1 ? 3 : 1; c.yprev = W(h, "d3");
        launch_conv1(c, s);
        if (bn1) norm_bwd(W(h, "dconv3"), W(h, "deconv3_pre"), W(h, "d3"), (int)Rs, 256, 32, D(h, "vae_dec/deconv3/gamma"), 0);
        ConvWgradArgs wg{};
        wg.np = (h->d.bf16 == 2 && (train_x3_mask(h) & 1)) ? 2 : 0;
        wg.S = W(h, "d2"); wg.Cs = 64; wg.Ps = 8; wg.Lg = W(h, "dconv3"); wg.Cl = 32; wg.Pl = 16; wg.stride = 2; wg.pad = 1;
        wg.n = (int)Rs; wg.partial = W(h, "tn_partial");
        launch_conv_wgrad(wg, NSL, G(h, "vae_dec/deconv3/w"), s);
        if (!bn1) colsum(h, W(h, "dconv3"), 32, Rs * 256, 32, G(h, "vae_dec/deconv3/b"), 0, s);
        const bool x3 = h->d.bf16 == 2 && (train_x3_mask(h) & 2);                  // split-bf16 operands in the two large data-gradient convolutions
        c.in = W(h, "dconv3"); c.out = W(h, "dconv2"); c.Wp = D4(h, x3 ? "vae_dec/deconv3/Wbwd16" : "vae_dec/deconv3/Wbwd");
        c.scale = D(h, "vae_dec/deconv2/scale"); c.shift = c.scale; c.yprev = W(h, "d2");
        if (x3) launch_conv2_x3(c, s); else launch_conv2(c, s);
        if (bn1) norm_bwd(W(h, "dconv2"), W(h, "deconv2_pre"), W(h, "d2"), (int)Rs, 64, 64, D(h, "vae_dec/deconv2/gamma"), 0);
        wg.S = W(h, "d1"); wg.Cs = 128; wg.Ps = 4; wg.Lg = W(h, "dconv2"); wg.Cl = 64; wg.Pl = 8; wg.stride = 1; wg.pad = 0;
        launch_conv_wgrad(wg, NSL, G(h, "vae_dec/deconv2/w"), s);
        if (!bn1) colsum(h, W(h, "dconv2"), 64, Rs * 64, 64, G(h, "vae_dec/deconv2/b"), 0, s);
        c.in = W(h, "dconv2"); c.out = W(h, "dconv1"); c.Wp = D4(h, x3 ? "vae_dec/deconv2/Wbwd16" : "vae_dec/deconv2/Wbwd");
        c.scale = D(h, "vae_dec/deconv1/scale"); c.shift = c.scale; c.yprev = W(h, "d1");
        if (x3) launch_conv3_x3(c, s); else launch_conv3(c, s);
        if (bn1) norm_bwd(W(h, "dconv1"), W(h, "deconv1_pre"), W(h, "d1"), (int)Rs, 16, 128, D(h, "vae_dec/deconv1/gamma"), 0);
        tn(h, W(h, "dconv1"), 2048, W(h, "z"), L, Rs, 2048, L, G(h, "vae_dec/deconv1/w"), L, 0, s);
        if (!bn1) colsum(h, W(h, "dconv1"), 128, Rs * 16, 128, G(h, "vae_dec/deconv1/b"), 0, s);
        GemmArgs g{};
        g.A = W(h, "dconv1"); g.lda = 2048; g.M = (int)Rs; g.K = 2048; g.Bp = D4(h, "vae_dec/deconv1/WT"); g.G = 2048 / 8;
        g.NT = (L + 31) / 32; g.out = W(h, "dz"); g.ldo = L; g.N = L;
        launch_gemm_rows(g, EPI_NONE, s);
    }
    // ---- latent + CVAE encoder + fc_c ----
    {
        Timer t(h, s, "bwd_cvae_enc");
        launch_reparam_bwd(W(h, "dz"), dev_eps, W(h, "params"), valid, W(h, "nvalid"), W(h, "dparams"), d.n_scenes, d.mno, d.K, L, s,
                           compact ? static_cast<const int32_t*>(h->ws["cp_inv"].p) : nullptr, P);
        // the encoder stack ran on the P present agents (desire_encode, DESIRE_FLAG_COMPACT_ROWS): its saves are in compact agent order and its
        // backward runs on the same agents -- dparams gathered in, d loss / d (Hx | Hy) assembled in the compact order (dHE)
        if (enc_c) launch_gather_agents(W(h, "dparams"), W(h, "cp_dparams"), amap, P, 2 * L, s);
        const float* dparE = enc_c ? W(h, "cp_dparams") : W(h, "dparams");
        if (Ae > 0) {
        const int A = Ae;                                  // (shadows the handle's agent count inside this block)
        tn(h, W(h, "c3"), 2048, dparE, 2 * L, A, 2048, 2 * L, G(h, "vae_enc/fc/w"), 2 * L, 0, s);
        colsum(h, dparE, 2 * L, A, 2 * L, G(h, "vae_enc/fc/b"), 0, s);
        GemmArgs g{};
        g.A = dparE; g.lda = 2 * L; g.M = A; g.K = 2 * L; g.Bp = D4(h, "vae_enc/fc/WT"); g.G = 2 * L / 8; g.NT = 64;
        g.out = W(h, "dconvE3"); g.ldo = 2048; g.N = 2048; g.p0 = D(h, "vae_enc/conv3/scale"); g.chmod = 128; g.aux = W(h, "c3");
        if (bn1) {
            launch_gemm_rows(g, EPI_NONE, s);
            norm_bwd(W(h, "dconvE3"), W(h, "conv3_pre"), W(h, "c3"), A, 16, 128, D(h, "vae_enc/conv3/gamma"), 0);
        } else
        launch_gemm_rows(g, EPI_ELUGRAD, s);
        const int NSL = A >= 2048 ? 64 : (A >= 256 ? 16 : 4);
        ConvWgradArgs wg{};
        wg.np = (h->d.bf16 == 2 && (train_x3_mask(h) & 1)) ? 2 : 0;
        wg.n = A; wg.partial = W(h, "tn_partial");
        wg.S = W(h, "dconvE3"); wg.Cs = 128; wg.Ps = 4; wg.Lg = W(h, "c2"); wg.Cl = 64; wg.Pl = 8; wg.stride = 1; wg.pad = 0;
        launch_conv_wgrad(wg, NSL, G(h, "vae_enc/conv3/w"), s);
        if (!bn1) colsum(h, W(h, "dconvE3"), 128, (long)A * 16, 128, G(h, "vae_enc/conv3/b"), 0, s);
        ConvArgs c{};
        c.n = A; c.mode = bn1 ? 3 : 1;
        c.in = W(h, "dconvE3"); c.out = W(h, "dconvE2"); c.Wp = D4(h, "vae_enc/conv3/Wbwd");
        c.scale = D(h, "vae_enc/conv2/scale"); c.shift = c.scale; c.yprev = W(h, "c2");
        launch_deconv2(c, s);
        if (bn1) norm_bwd(W(h, "dconvE2"), W(h, "conv2_pre"), W(h, "c2"), A, 64, 64, D(h, "vae_enc/conv2/gamma"), 0);
        wg.S = W(h, "dconvE2"); wg.Cs = 64; wg.Ps = 8; wg.Lg = W(h, "c1"); wg.Cl = 32; wg.Pl = 16; wg.stride = 2; wg.pad = 1;
        launch_conv_wgrad(wg, NSL, G(h, "vae_enc/conv2/w"), s);
        if (!bn1) colsum(h, W(h, "dconvE2"), 64, (long)A * 64, 64, G(h, "vae_enc/conv2/b"), 0, s);
        c.in = W(h, "dconvE2"); c.out = W(h, "dconvE1"); c.Wp = D4(h, "vae_enc/conv2/Wbwd");
        c.scale = D(h, "vae_enc/conv1/scale"); c.shift = c.scale; c.yprev = W(h, "c1");
        launch_deconv3(c, s);
        if (bn1) norm_bwd(W(h, "dconvE1"), W(h, "conv1_pre"), W(h, "c1"), A, 256, 32, D(h, "vae_enc/conv1/gamma"), 0);
        launch_w1ch_grad(W(h, "vae_in"), W(h, "dconvE1"), A, A < 1024 ? A : 1024, W(h, "tn_partial"), G(h, "vae_enc/conv1/w"), s);
        if (!bn1) colsum(h, W(h, "dconvE1"), 32, (long)A * 256, 32, G(h, "vae_enc/conv1/b"), 0, s);
        c.in = W(h, "dconvE1"); c.out = W(h, "dq_c"); c.w_raw = D(h, "vae_enc/conv1/raw"); c.mode = 2; c.yprev = W(h, "vae_in");
        launch_deconv4(c, s);
        tn(h, HxE, 2 * H, W(h, "dq_c"), V, A, 2 * H, V, G(h, "fc_c/w"), V, 0, s);
        colsum(h, W(h, "dq_c"), V, A, V, G(h, "fc_c/b"), 0, s);
        g = GemmArgs{};
        g.A = W(h, "dq_c"); g.lda = V; g.M = A; g.K = V; g.Bp = D4(h, "fc_c/WT"); g.G = V / 8; g.NT = 2 * H / 32;
        g.out = dHE; g.ldo = 2 * H; g.N = 2 * H;
        launch_gemm_rows(g, EPI_NONE, s);
        }       // Ae > 0
        if (enc_c) {
            if (P > 0) {
                launch_rows_to_agents(dHxS, dHE, 2 * H, 1, P, d.K, H, s);                      // decoder + mask share: compact rows -> compact agents
                if (!h->ci_last) {                                                                // IOC share: the caller's rows -> agents -> compact agents
                    launch_fill_f32(W(h, "dHxHy_ioc"), (size_t)h->A * H, 0.f, s);
                    launch_rows_to_agents(W(h, "dHx_rows"), W(h, "dHxHy_ioc"), H, d.n_scenes, d.mno, d.K, H, s);
                }
                launch_gather_add_agents(W(h, "dHxHy_ioc"), H, dHE, 2 * H, amap, P, H, s);
            }
        } else {
            launch_rows_to_agents(W(h, "dHx_rows"), W(h, "dHxHy"), 2 * H, d.n_scenes, d.mno, d.K, H, s);
            if (h->ci_last) launch_rows_to_agents(W(h, "dHxHy_ioc"), W(h, "dHxHy"), 2 * H, h->A, 1, 1, H, s);       // + the slot classes' share (one "row" per agent)
            if (compact && P > 0) {         // the compact stages' share of d loss / d Hx: rows -> compact agents -> agents
                launch_fill_f32(W(h, "cp_dHx"), (size_t)P * H, 0.f, s);
                launch_rows_to_agents(dHxS, W(h, "cp_dHx"), H, 1, P, d.K, H, s);
                launch_scatter_add_agents(W(h, "cp_dHx"), H, W(h, "dHxHy"), 2 * H, amap, P, H, s);
            }
        }
    }
    // ---- encoders: BPTT from the final state (Hx / Hy), zero initial state ----
    const bool head_loss = h->head_loss_w > 0.f;
    if (Ae > 0) {
    const int A = Ae;                                      // encoders: the agents the stack ran on
    const float* pastE = enc_c ? W(h, "cp_past") : dev_past; const float* futE = enc_c ? W(h, "cp_fut") : dev_fut;
    const int mnoE = enc_c ? Ae : d.mno;
    if (head_loss) {
        // Gaussian-head term (desire_set_head_loss): nll and d nll / d o per (agent, observed frame), the mean, its gradient; then the
        // head's own weight gradients.  The gradient w.r.t. the encoder states enters the X-encoder BPTT below, step by step.
        const int n = A * d.T_obs;
        Timer t(h, s, "bwd_head_nll");
        hipLaunchKernelGGL(k_head_nll, dim3((n * 64 + 255) / 256), dim3(256), 0, s, W(h, "ex_sv_h"), W(h, "ex_sv_x"), pastE, futE,
                           D(h, "gauss_head/w"), D(h, "gauss_head/b"), A, d.T_obs, d.T_pred, H, mnoE, d.sx, d.sy, W(h, "head_nll"), W(h, "head_cnt"),
                           W(h, "head_dO"));
        hipLaunchKernelGGL(k_head_sum, dim3(1), dim3(256), 0, s, W(h, "head_nll"), W(h, "head_cnt"), n, h->head_loss_w, W(h, "loss_out"));
        hipLaunchKernelGGL(k_head_scale, dim3((n * 5 + 255) / 256), dim3(256), 0, s, W(h, "head_dO"), n * 5, h->head_loss_w, W(h, "loss_out"));
        tn(h, W(h, "ex_sv_h"), H, W(h, "head_dO"), 5, (long)n, H, 5, G(h, "gauss_head/w"), 5, 0, s);
        colsum(h, W(h, "head_dO"), 5, (long)n, 5, G(h, "gauss_head/b"), 0, s);
    }
    auto enc_bwd = [&](const std::string& p, const char* sv, int Te, int col0) {
        DecBwdArgs e{};
        e.sv_r = W(h, (std::string(sv) + "_sv_r").c_str()); e.sv_u = W(h, (std::string(sv) + "_sv_u").c_str());
        e.sv_c = W(h, (std::string(sv) + "_sv_c").c_str()); e.sv_h = W(h, (std::string(sv) + "_sv_h").c_str());
        e.w_head = D(h, "head/w");
        if (head_loss && p == "enc_x") { e.dY0 = W(h, "head_dO"); e.w_head = D(h, "gauss_head/w"); e.nw = 5; }     // d L_head / d h_t = dO_t W5^T, every step
        e.WcT_h = D4(h, (p + "/WcT_h").c_str()); e.WgT_h = D4(h, (p + "/WgT_h").c_str());
        e.R = A; e.K = 1; e.mno = mnoE; e.T = Te; e.H = H;
        e.dag = W(h, "enc_dag"); e.dac = W(h, "enc_dac"); e.rh = W(h, "enc_rh"); e.hprev = W(h, "enc_hprev");
        e.dh_init = dHE + col0; e.ld_init = 2 * H;
        launch_decoder_bwd(e, s);
        const float* xs = W(h, (std::string(sv) + "_sv_x").c_str());
        float* gk = G(h, p + "/gates/kernel");           // [(2+H), 2H]
        tn(h, xs, 2, W(h, "enc_dag"), 2 * H, (long)A * Te, 2, 2 * H, gk, 2 * H, 0, s);
        tn(h, W(h, "enc_hprev"), H, W(h, "enc_dag"), 2 * H, (long)A * Te, H, 2 * H, gk + (size_t)2 * 2 * H, 2 * H, 0, s);
        colsum(h, W(h, "enc_dag"), 2 * H, (long)A * Te, 2 * H, G(h, p + "/gates/bias"), 0, s);
        float* ck = G(h, p + "/candidate/kernel");       // [(2+H), H]
        tn(h, xs, 2, W(h, "enc_dac"), H, (long)A * Te, 2, H, ck, H, 0, s);
        tn(h, W(h, "enc_rh"), H, W(h, "enc_dac"), H, (long)A * Te, H, H, ck + (size_t)2 * H, H, 0, s);
        colsum(h, W(h, "enc_dac"), H, (long)A * Te, H, G(h, p + "/candidate/bias"), 0, s);
    };
    { Timer t(h, s, "bwd_encoder_y"); enc_bwd("enc_y", "ey", d.T_pred, H); }
    { Timer t(h, s, "bwd_encoder_x"); enc_bwd("enc_x", "ex", d.T_obs, 0); }
    }       // Ae > 0
    HIPCHK(hipGetLastError());
    if (ioc_uses_cluster(d.mno, d.H, d.grid_size * d.grid_size, 0)) {
        int e = 0;
        HIPCHK(hipMemcpyAsync(&e, h->ws["ioc_err"].p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (e) return fail(DESIRE_ERR_HIP, "IOC cluster backward: hand-off timed out (workgroups of a group were not co-resident)");
    }
    return DESIRE_OK;
}

extern "C" int desire_set_head_loss(desire_handle* h, float weight) {
    if (!h) return fail(DESIRE_ERR_ARG, "null argument");
    if (!(weight >= 0.f)) return fail(DESIRE_ERR_ARG, "weight must be >= 0");
    if (h->d.ref_compat) return fail(DESIRE_ERR_STATE, "ref_compat is forward-only");
    h->head_loss_w = weight;
    if (h->training && weight == 0.f) {              // the reported term goes back to zero with the switch
        const float z[3] = {0.f, 0.f, 0.f};
        HIPCHK(hipMemcpy(W(h, "loss_out") + 5, z, sizeof(float), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(W(h, "loss_out") + 7, z, sizeof(float), hipMemcpyHostToDevice));
    }
    return DESIRE_OK;
}

extern "C" int desire_get_grad(desire_handle* h, const char* name, float* host_out, size_t n, void* stream) {
    if (!h || !name || !host_out) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    auto it = h->slots.find(name);
    if (it == h->slots.end()) return fail(DESIRE_ERR_ARG, std::string("unknown weight: ") + name);
    const size_t nu = h->want_user.at(name);
    if (nu != n) return fail(DESIRE_ERR_ARG, std::string(name) + ": expected " + std::to_string(nu) + " values");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIPCHK(hipStreamSynchronize(s));
    std::vector<float> phys(it->second.n);
    HIPCHK(hipMemcpy(phys.data(), W(h, "Gflat") + it->second.off, phys.size() * sizeof(float), hipMemcpyDeviceToHost));
    desire_extract(h, name, phys.data(), host_out);
    return DESIRE_OK;
}

extern "C" int desire_grad_buffer(desire_handle* h, float** dev_ptr, size_t* n) {
    if (!h || !dev_ptr || !n) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    *dev_ptr = W(h, "Gflat");
    *n = h->n_params;
    return DESIRE_OK;
}

static int train_loss_enqueue(desire_handle* h, const float* dev_fut, hipStream_t s) {
    const desire_dims& d = h->d;
    const uint8_t* valid = static_cast<const uint8_t*>(h->ws["lmask"].p);
    launch_loss_mask(static_cast<const uint8_t*>(h->ws["valid"].p), dev_fut, static_cast<uint8_t*>(h->ws["lmask"].p), W(h, "nfut"),
                     d.n_scenes, d.mno, d.T_pred, s);
    hipLaunchKernelGGL(k_train_loss, dim3((h->A + 63) / 64), dim3(64), 0, s, W(h, "Y0"), W(h, "Y_ref"), dev_fut, W(h, "score_sv"),
                       W(h, "params"), valid, W(h, "nfut"), W(h, "loss_pa"), d.n_scenes, d.mno, d.K, d.T_pred, d.L, d.sx, d.sy);
    hipLaunchKernelGGL(k_sum_loss, dim3(1), dim3(256), 0, s, W(h, "loss_pa"), valid, h->A, W(h, "loss_out"));
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_train_loss(desire_handle* h, const float* dev_fut, float* host_out5, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    if (!dev_fut || !host_out5) return fail(DESIRE_ERR_ARG, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = train_loss_enqueue(h, dev_fut, s)) return rc;
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(host_out5, W(h, "loss_out"), 5 * sizeof(float), hipMemcpyDeviceToHost));
    return DESIRE_OK;
}

// The same terms (the 8 floats of loss_out: the five above, then the Gaussian-head terms of desire_set_head_loss) into a DEVICE buffer, stream-ordered, no synchronisation: the training loop reads them one step late (a pinned
// copy + event), so the host never waits for the step it has just enqueued and the loader thread keeps running ahead.
extern "C" int desire_train_loss_async(desire_handle* h, const float* dev_fut, float* dev_out8, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    if (!dev_fut || !dev_out8) return fail(DESIRE_ERR_ARG, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = train_loss_enqueue(h, dev_fut, s)) return rc;
    launch_copy_f32(dev_out8, W(h, "loss_out"), 8, s);
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_adam_step(desire_handle* h, float lr, float beta1, float beta2, float eps, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return fail(DESIRE_ERR_ARG, "bad Adam hyper-parameters");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int t = ++h->adam_t;
    const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, t)) / (1.0 - std::pow((double)beta1, t));
    Timer tm(h, s, "bwd_adam_repack");
    hipLaunchKernelGGL(k_adam, dim3(1024), dim3(256), 0, s, W(h, "Wflat"), W(h, "Gflat"), W(h, "Mflat"), W(h, "Vflat"), h->n_params,
                       (float)lr_t, beta1, beta2, eps);
    return repack(h, s);
}

// Optimiser state for checkpoints: the Adam moments are the workspace tensors "Mflat" / "Vflat" (desire_device_buffer; same flat
// layout as the gradient buffer), the step counter t of the bias correction is read (set = 0) or written (set = 1) here.
extern "C" int desire_adam_state(desire_handle* h, int32_t* step, int set) {
    if (!h || !step) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    if (set) { if (*step < 0) return fail(DESIRE_ERR_ARG, "step must be >= 0"); h->adam_t = *step; }
    else *step = h->adam_t;
    return DESIRE_OK;
}

extern "C" int desire_get_weight(desire_handle* h, const char* name, float* host_out, size_t n, void* stream) {
    if (!h || !name || !host_out) return fail(DESIRE_ERR_ARG, "null argument");
    auto w = h->want_user.find(name);
    if (w == h->want_user.end()) return fail(DESIRE_ERR_ARG, std::string("unknown weight: ") + name);
    if (w->second != n) return fail(DESIRE_ERR_ARG, std::string(name) + ": expected " + std::to_string(w->second) + " values");
    if (h->training) {
        hipStream_t s = static_cast<hipStream_t>(stream);
        HIPCHK(hipStreamSynchronize(s));
        const WSlot& sl = h->slots.at(name);
        std::vector<float> phys(sl.n);
        HIPCHK(hipMemcpy(phys.data(), W(h, "Wflat") + sl.off, sl.n * sizeof(float), hipMemcpyDeviceToHost));
        desire_extract(h, name, phys.data(), host_out);
        return DESIRE_OK;
    }
    auto it = h->host_w.find(name);
    if (it == h->host_w.end()) return fail(DESIRE_ERR_STATE, std::string("weight not set: ") + name);
    desire_extract(h, name, it->second.data(), host_out);
    return DESIRE_OK;
}

extern "C" int desire_clip_grads(desire_handle* h, float max_norm, float* host_norm_out, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    if (!(max_norm > 0.f)) return fail(DESIRE_ERR_ARG, "max_norm must be positive");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* part = W(h, "tn_partial");
    hipLaunchKernelGGL(k_sqsum, dim3(512), dim3(256), 0, s, W(h, "Gflat"), h->n_params, part);
    hipLaunchKernelGGL(k_clip_scale, dim3(512), dim3(256), 0, s, W(h, "Gflat"), h->n_params, part, 512, max_norm, W(h, "loss_out") + 6);
    HIPCHK(hipGetLastError());
    if (host_norm_out) {
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(host_norm_out, W(h, "loss_out") + 6, sizeof(float), hipMemcpyDeviceToHost));
    }
    return DESIRE_OK;
}
