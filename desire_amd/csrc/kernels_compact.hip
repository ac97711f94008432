// kernels_compact.hip -- present-row compaction of the per-row sample-generation stages (dims.flags & DESIRE_FLAG_COMPACT_ROWS).
//
// The loader pads every window to max_num_obj slots (utils/data_loader.py:209-229: absent slots are zero rows) and the reference masks
// id-0 objects in the cost only (model/model.py:351-366).  On real SDD windows most slots are padding (bookstore/video6: 9 of 32), and
// reparameterisation -> deconv1..4 -> mask fc -> GRU decoder are independent per (agent, k) row, so they need not run on padding at all.
//
// Compact domain: the P agents present at the last observed frame, in agent order, as ONE pseudo-scene of P slots:
//     compact agent a' in [0, P)           <->  full agent   amap[a'] = scene*mno + slot
//     compact row   r' = k*P + a'          <->  full row     r = (scene*K + k)*mno + slot
// so every per-row kernel runs unchanged with (n_scenes, mno, R) = (1, P, K*P): agent_of_row(r', K, P) = a'.
// These kernels build the map, gather the agent-level inputs, and scatter / gather rows between the two domains.  All VALU, all HBM-bound
// and small (O(A) or O(R * T * 2) floats).
#include "common.h"
#include "kernels.h"

// ---- the map: one workgroup, ordered scan over valid[A]; amap[a'] = a; *count_dev = *count_host = P --------------------------------
__global__ __launch_bounds__(1024) void k_present_scan(const uint8_t* __restrict__ valid, int A, int32_t* __restrict__ amap,
                                                       int32_t* __restrict__ inv, int32_t* __restrict__ count_dev, volatile int32_t* count_host) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int a0 = 0; a0 < A; a0 += 1024) {
        const int a = a0 + tid;
        const int v = (a < A && valid[a]) ? 1 : 0;
        const unsigned long long m = __ballot(v);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wv] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (v) amap[off + before] = a;
        if (a < A) inv[a] = v ? off + before : -1;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base_s += t; }
        __syncthreads();
    }
    if (tid == 0) {
        *count_dev = base_s;
        *count_host = base_s;
        __threadfence_system();
    }
}
void launch_present_scan(const uint8_t* valid, int A, int32_t* amap, int32_t* inv, int32_t* count_dev, int32_t* count_host, hipStream_t s) {
    hipLaunchKernelGGL(k_present_scan, dim3(1), dim3(1024), 0, s, valid, A, amap, inv, count_dev, count_host);
}

// ---- agent-level gathers: out[a', :] = in[amap[a'], :]  (HxHy, p_last, params) ------------------------------------------------------
// grids of the element-wise kernels that may carry a device-side count: capped, the kernels stride over their elements -- with the worst case as the
// launch size most of an uncapped grid would be workgroups that exit at once (0.1 ms of dead dispatches per step on 512 SDD windows)
static inline unsigned cp_grid(long n) { const long b = (n + 255) / 256; return (unsigned)(b < 8192 ? (b < 1 ? 1 : b) : 8192); }
#define CP_FOR(i, total) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)(total); i += (long)gridDim.x * blockDim.x)
// (dynP, here and below: the count on the DEVICE -- the host passes the worst case as P for the grid and the kernel reads the real one; kernels.h: DynCount)
__global__ void k_gather_agents(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ amap, int P, int ld, const int32_t* __restrict__ dynP) {
    if (dynP) P = dynP[0];
    CP_FOR(i, (long)P * ld) {
        const int ap = (int)(i / ld), c = (int)(i - (long)ap * ld);
        out[i] = in[(size_t)amap[ap] * ld + c];
    }
}
void launch_gather_agents(const float* in, float* out, const int32_t* amap, int P, int ld, hipStream_t s, const int32_t* dynP) {
    const long n = (long)P * ld;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_agents, dim3(cp_grid(n)), dim3(256), 0, s, in, out, amap, P, ld, dynP);
}
// out[amap[a'], c0 + c] += in[a', c]  (the compact domain's share of d loss / d Hx back onto the agents; one writer per element)
__global__ void k_scatter_add_agents(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, const int32_t* __restrict__ amap, int P, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * n) return;
    const int ap = (int)(i / n), c = (int)(i - (long)ap * n);
    out[(size_t)amap[ap] * ldo + c] += in[(size_t)ap * ldi + c];
}
void launch_scatter_add_agents(const float* in, int ldi, float* out, int ldo, const int32_t* amap, int P, int n, hipStream_t s) {
    const long t = (long)P * n;
    if (t <= 0) return;
    hipLaunchKernelGGL(k_scatter_add_agents, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, in, ldi, out, ldo, amap, P, n);
}

__device__ __forceinline__ size_t full_row_of(int rp, int P, int K, int mno, const int32_t* __restrict__ amap) {
    const int k = rp / P, ap = rp - k * P;
    const int a = amap[ap];
    const int sc = a / mno, slot = a - sc * mno;
    return ((size_t)sc * K + k) * mno + slot;
}

// ---- z[r', :] from eps[r, :] (k_reparam of kernels_gemm.hip on the compact rows; params_c is the GATHERED [P, 2L]) -----------------
__global__ void k_reparam_c(const float* __restrict__ params_c, const float* __restrict__ eps, float* __restrict__ z,
                            const int32_t* __restrict__ amap, int P, int K, int mno, int L, int posterior, const int32_t* __restrict__ dynP) {
    if (dynP) P = dynP[0];
    CP_FOR(i, (long)P * K * L) {
        const int rp = (int)(i / L), l = (int)(i - (long)rp * L);
        const size_t r = full_row_of(rp, P, K, mno, amap);
        float e = eps[r * L + l];
        if (posterior) {
            const int ap = rp % P;
            const float mu = params_c[(size_t)ap * 2 * L + l], ls = params_c[(size_t)ap * 2 * L + L + l];
            e = mu + sqrtf(expf(ls)) * e;
        }
        z[i] = e;
    }
}
void launch_reparam_c(const float* params_c, const float* eps, float* z, const int32_t* amap, int P, int K, int mno, int L, int posterior, hipStream_t s, const int32_t* dynP) {
    const long n = (long)P * K * L;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_reparam_c, dim3(cp_grid(n)), dim3(256), 0, s, params_c, eps, z, amap, P, K, mno, L, posterior, dynP);
}

// ---- rows between the domains: n floats per row ---------------------------------------------------------------------------------------
// scatter: full[r, :] = compact[r', :] (to one or two destinations; rows of absent agents are NOT touched: the caller zero-fills first)
__global__ void k_scatter_rows(const float* __restrict__ comp, float* __restrict__ full0, float* __restrict__ full1,
                               const int32_t* __restrict__ amap, int P, int K, int mno, int n, const int32_t* __restrict__ dynP) {
    if (dynP) P = dynP[0];
    CP_FOR(i, (long)P * K * n) {
        const int rp = (int)(i / n), c = (int)(i - (long)rp * n);
        const size_t r = full_row_of(rp, P, K, mno, amap);
        const float v = comp[i];
        full0[r * n + c] = v;
        if (full1) full1[r * n + c] = v;
    }
}
void launch_scatter_rows(const float* comp, float* full0, float* full1, const int32_t* amap, int P, int K, int mno, int n, hipStream_t s, const int32_t* dynP) {
    const long t = (long)P * K * n;
    if (t <= 0) return;
    hipLaunchKernelGGL(k_scatter_rows, dim3(cp_grid(t)), dim3(256), 0, s, comp, full0, full1, amap, P, K, mno, n, dynP);
}
// gather: compact[r', :] = full[r, :]
__global__ void k_gather_rows(const float* __restrict__ full, float* __restrict__ comp, const int32_t* __restrict__ amap, int P, int K, int mno, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * K * n) return;
    const int rp = (int)(i / n), c = (int)(i - (long)rp * n);
    comp[i] = full[full_row_of(rp, P, K, mno, amap) * n + c];
}
void launch_gather_rows(const float* full, float* comp, const int32_t* amap, int P, int K, int mno, int n, hipStream_t s) {
    const long t = (long)P * K * n;
    if (t <= 0) return;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, full, comp, amap, P, K, mno, n);
}

// =====================================================================================================================================
// IOC class repacking (dims.flags & DESIRE_FLAG_COMPACT_IOC).  The IOC kernels tile rows by whole (scene, k) groups of mno slots; a window
// with c present agents still occupies mno rows per sample.  Here every window is re-seated in the smallest slot class m in {8, 16, 32, ...,
// mno} that holds its present agents (compacted to the front, order kept; padding slots absent), windows of one class form a pseudo-batch
// (n_c windows x m_c slots) that the SAME kernels run on, and windows without any present agent are not run at all.
//     class agent  i = w'*m_c + j   <->  full agent cmap[i] (or -1: padding);   class row (w'*K + k)*m_c + j  <->  full row of (cmap[i], k)
// =====================================================================================================================================
// one workgroup: windows in order; cnt[c] windows per class (device + mapped host copy), cls_win[c][w'] = window, cmap[c][w'*m_c + j].
// A class whose launch would be smaller than min_rows rows (K * m_c * its windows) is folded into the next larger one: a launch of the
// persistent kernel costs T steps of latency whatever its size, more than the padding rows saved.
__global__ __launch_bounds__(1024) void k_class_scan(const uint8_t* __restrict__ valid, int n_scenes, int mno, int n_cls, int4 msz, int K, int min_rows,
                                                     int32_t* __restrict__ cls_win, int32_t* __restrict__ cmap, int32_t* __restrict__ cnt_dev,
                                                     volatile int32_t* cnt_host) {
    __shared__ int wsum[4][16];
    __shared__ int base_s[4];
    __shared__ int tot_s[4];
    __shared__ int remap_s[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int A = n_scenes * mno;
    const int m[4] = {msz.x, msz.y, msz.z, msz.w};
    auto class_of = [&](int w) {
        int c = 0;
        for (int j = 0; j < mno; ++j) c += valid[(size_t)w * mno + j] ? 1 : 0;
        if (c == 0) return -1;
        int cls = 0;
        while (cls < n_cls - 1 && c > m[cls]) ++cls;
        return cls;
    };
    if (tid < 4) { base_s[tid] = 0; tot_s[tid] = 0; }
    __syncthreads();
    for (int w = tid; w < n_scenes; w += 1024) { const int cls = class_of(w); if (cls >= 0) atomicAdd(&tot_s[cls], 1); }
    __syncthreads();
    if (tid == 0) {
        int carry = 0;
        for (int q = 0; q < 4; ++q) remap_s[q] = q;
        for (int q = 0; q < n_cls - 1; ++q) {
            const int n = tot_s[q] + carry;
            if (n > 0 && (long)n * K * m[q] < (long)min_rows) { carry = n; for (int x = 0; x <= q; ++x) if (remap_s[x] == q) remap_s[x] = q + 1; }
            else carry = 0;
        }
    }
    __syncthreads();
    for (int w0 = 0; w0 < n_scenes; w0 += 1024) {
        const int w = w0 + tid;
        int cls = w < n_scenes ? class_of(w) : -1;
        if (cls >= 0) cls = remap_s[cls];
        int pos = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned long long b = __ballot(cls == q);
            if (cls == q) pos = __popcll(b & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[q][wv] = __popcll(b);
        }
        __syncthreads();
        if (cls >= 0) {
            int off = base_s[cls];
            for (int x = 0; x < wv; ++x) off += wsum[cls][x];
            const int wp = off + pos;
            cls_win[(size_t)cls * n_scenes + wp] = w;
            int32_t* dst = cmap + (size_t)cls * A + (size_t)wp * m[cls];
            int j2 = 0;
            for (int j = 0; j < mno; ++j) if (valid[(size_t)w * mno + j]) dst[j2++] = w * mno + j;
            for (; j2 < m[cls]; ++j2) dst[j2] = -1;
        }
        __syncthreads();
        if (tid < 4) { int t = 0; for (int x = 0; x < 16; ++x) t += wsum[tid][x]; base_s[tid] += t; }
        __syncthreads();
    }
    if (tid < 4) { cnt_dev[tid] = base_s[tid]; cnt_host[tid] = base_s[tid]; }
    __threadfence_system();
}
void launch_class_scan(const uint8_t* valid, int n_scenes, int mno, int n_cls, const int* m4, int K, int min_rows, int32_t* cls_win, int32_t* cmap,
                       int32_t* cnt_dev, int32_t* cnt_host, hipStream_t s) {
    hipLaunchKernelGGL(k_class_scan, dim3(1), dim3(1024), 0, s, valid, n_scenes, mno, n_cls, make_int4(m4[0], m4[1], m4[2], m4[3]), K, min_rows, cls_win, cmap,
                       cnt_dev, cnt_host);
}

// agent-level inputs of one class: Hx (ld floats per agent, zeros for padding), p_last, valid, grid of the window
__global__ void k_cls_gather_agents(const float* __restrict__ Hx, int ld, const float* __restrict__ p_last, const int32_t* __restrict__ gos,
                                    const int32_t* __restrict__ cmap, const int32_t* __restrict__ win, int n_c, int m_c,
                                    float* __restrict__ Hx_c, float* __restrict__ p_c, uint8_t* __restrict__ valid_c, int32_t* __restrict__ gos_c,
                                    const int32_t* __restrict__ dynN) {
    if (dynN) n_c = dynN[0];
    const long NA = (long)n_c * m_c;
    CP_FOR(i, NA * ld) {
        if (i < n_c) gos_c[i] = gos[win[i]];
        if (i < NA) {
            const int a = cmap[i];
            valid_c[i] = a >= 0 ? 1 : 0;
            p_c[2 * i] = a >= 0 ? p_last[2 * (size_t)a] : 0.f;
            p_c[2 * i + 1] = a >= 0 ? p_last[2 * (size_t)a + 1] : 0.f;
        }
        const long ia = i / ld; const int c = (int)(i - ia * ld);
        const int a = cmap[ia];
        Hx_c[i] = a >= 0 ? Hx[(size_t)a * ld + c] : 0.f;
    }
}
void launch_cls_gather_agents(const float* Hx, int ld, const float* p_last, const int32_t* gos, const int32_t* cmap, const int32_t* win, int n_c, int m_c,
                              float* Hx_c, float* p_c, uint8_t* valid_c, int32_t* gos_c, hipStream_t s, const int32_t* dynN) {
    const long n = (long)n_c * m_c * ld;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_cls_gather_agents, dim3(cp_grid(n)), dim3(256), 0, s, Hx, ld, p_last, gos, cmap, win, n_c, m_c, Hx_c, p_c, valid_c, gos_c, dynN);
}
// rows: dir = 0 gather  comp[class row of (w', k, j), :] = full[row(cmap, k), :] (zeros for padding);  dir = 1 scatter back (padding rows dropped).
// Class row of (w', k, j): packed (gpt = 0) (w'*K + k)*m_c + j; padded tiles (gpt > 0) tile*32 + (G % gpt)*m_c + j with G = w'*K + k, tile = G / gpt.
__global__ void k_cls_rows(float* __restrict__ full, float* __restrict__ comp, const int32_t* __restrict__ cmap, int n_c, int m_c, int K, int mno, int n, int dir,
                           int gpt, long rows, const int32_t* __restrict__ dynN) {
    if (dynN) {
        n_c = dynN[0];
        const long ngrp = (long)n_c * K;
        rows = gpt ? ((ngrp + gpt - 1) / gpt) * 32 : ngrp * m_c;
    }
    CP_FOR(i, rows * n) {
        const long rc = i / n; const int c = (int)(i - rc * n);
        int j; long g;
        if (gpt) {
            const int il = (int)(rc & 31), gi = il / m_c;
            j = il - gi * m_c; g = (rc >> 5) * gpt + gi;
            if (gi >= gpt || g >= (long)n_c * K) { if (!dir) comp[i] = 0.f; continue; }
        } else { j = (int)(rc % m_c); g = rc / m_c; }
        const int k = (int)(g % K); const int wp = (int)(g / K);
        const int a = cmap[(size_t)wp * m_c + j];
        if (a < 0) { if (!dir) comp[i] = 0.f; continue; }
        const int sc = a / mno, slot = a - sc * mno;
        const size_t r = ((size_t)sc * K + k) * mno + slot;
        if (dir) full[r * n + c] = comp[i]; else comp[i] = full[r * n + c];
    }
}
void launch_cls_rows(float* full, float* comp, const int32_t* cmap, int n_c, int m_c, int K, int mno, int n, int dir, hipStream_t s, int gpt, const int32_t* dynN) {
    const long ngrp = (long)n_c * K;
    const long rows = gpt ? ((ngrp + gpt - 1) / gpt) * 32 : ngrp * m_c;
    const long t = rows * n;
    if (t <= 0) return;
    hipLaunchKernelGGL(k_cls_rows, dim3(cp_grid(t)), dim3(256), 0, s, full, comp, cmap, n_c, m_c, K, mno, n, dir, gpt, rows, dynN);
}
// out[cmap[i], c] += in[i, c] for the seated agents of a class (cmap[i] >= 0; one writer per element: an agent sits in exactly one class slot)
__global__ void k_cls_scatter_add_agents(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, const int32_t* __restrict__ cmap, int NA, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)NA * n) return;
    const int ia = (int)(i / n), c = (int)(i - (long)ia * n);
    const int a = cmap[ia];
    if (a >= 0) out[(size_t)a * ldo + c] += in[(size_t)ia * ldi + c];
}
void launch_cls_scatter_add_agents(const float* in, int ldi, float* out, int ldo, const int32_t* cmap, int NA, int n, hipStream_t s) {
    const long t = (long)NA * n;
    if (t <= 0) return;
    hipLaunchKernelGGL(k_cls_scatter_add_agents, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, in, ldi, out, ldo, cmap, NA, n);
}

// =====================================================================================================================================
// Encoder-stage compaction (DESIRE_FLAG_COMPACT_ROWS, round 5b): the GRU encoders and the CVAE encoder are per-agent too.  `valid` is read off
// the last observed frame BEFORE the encoders, the windows of the present agents are gathered into one pseudo-scene [1, T, P, 3], and the encoder
// stack runs on P agents; HxHy / p_last / params are scattered back for the stages that keep the caller's layout (IOC, losses).
// =====================================================================================================================================
__global__ void k_valid_from_frames(const float* __restrict__ past, int n_scenes, int T, int mno, uint8_t* __restrict__ valid) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_scenes * mno) return;
    const int sc = a / mno, slot = a - sc * mno;
    valid[a] = past[(((size_t)sc * T + (T - 1)) * mno + slot) * 3] != 0.f ? 1 : 0;          // id != 0 at the last observed frame (k_encoder's rule)
}
void launch_valid_from_frames(const float* past, int n_scenes, int T, int mno, uint8_t* valid, hipStream_t s) {
    const int A = n_scenes * mno;
    hipLaunchKernelGGL(k_valid_from_frames, dim3((A + 255) / 256), dim3(256), 0, s, past, n_scenes, T, mno, valid);
}
// frames_c[0, t, a', :] = frames[scene, t, slot, :]
__global__ void k_gather_frames(const float* __restrict__ frames, float* __restrict__ out, const int32_t* __restrict__ amap, int P, int T, int mno, const int32_t* __restrict__ dynP) {
    if (dynP) P = dynP[0];
    CP_FOR(i, (long)T * P) {
        const int t = (int)(i / P), ap = (int)(i - (long)t * P);
        const int a = amap[ap];
        const int sc = a / mno, slot = a - sc * mno;
        const float* src = frames + (((size_t)sc * T + t) * mno + slot) * 3;
        float* dst = out + (size_t)i * 3;
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    }
}
void launch_gather_frames(const float* frames, float* out, const int32_t* amap, int P, int T, int mno, hipStream_t s, const int32_t* dynP) {
    const long n = (long)T * P;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_frames, dim3(cp_grid(n)), dim3(256), 0, s, frames, out, amap, P, T, mno, dynP);
}
// out[amap[a'], c] = in[a', c]   (rows of absent agents untouched: the caller zero-fills)
__global__ void k_scatter_agents(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ amap, int P, int ld, const int32_t* __restrict__ dynP) {
    if (dynP) P = dynP[0];
    CP_FOR(i, (long)P * ld) {
        const int ap = (int)(i / ld), c = (int)(i - (long)ap * ld);
        out[(size_t)amap[ap] * ld + c] = in[i];
    }
}
void launch_scatter_agents(const float* in, float* out, const int32_t* amap, int P, int ld, hipStream_t s, const int32_t* dynP) {
    const long n = (long)P * ld;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_scatter_agents, dim3(cp_grid(n)), dim3(256), 0, s, in, out, amap, P, ld, dynP);
}
// out[a', c] += in[amap[a'], c]   (n columns; agent-level gradient from the caller's layout into the compact one)
__global__ void k_gather_add_agents(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, const int32_t* __restrict__ amap, int P, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * n) return;
    const int ap = (int)(i / n), c = (int)(i - (long)ap * n);
    out[(size_t)ap * ldo + c] += in[(size_t)amap[ap] * ldi + c];
}
void launch_gather_add_agents(const float* in, int ldi, float* out, int ldo, const int32_t* amap, int P, int n, hipStream_t s) {
    const long t = (long)P * n;
    if (t <= 0) return;
    hipLaunchKernelGGL(k_gather_add_agents, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, in, ldi, out, ldo, amap, P, n);
}
