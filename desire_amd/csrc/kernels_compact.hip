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
__global__ void k_gather_agents(const float* __restrict__ in, float* __restrict__ out, const int32_t* __restrict__ amap, int P, int ld) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * ld) return;
    const int ap = (int)(i / ld), c = (int)(i - (long)ap * ld);
    out[i] = in[(size_t)amap[ap] * ld + c];
}
void launch_gather_agents(const float* in, float* out, const int32_t* amap, int P, int ld, hipStream_t s) {
    const long n = (long)P * ld;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_gather_agents, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, amap, P, ld);
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
                            const int32_t* __restrict__ amap, int P, int K, int mno, int L, int posterior) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * K * L) return;
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
void launch_reparam_c(const float* params_c, const float* eps, float* z, const int32_t* amap, int P, int K, int mno, int L, int posterior, hipStream_t s) {
    const long n = (long)P * K * L;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_reparam_c, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, params_c, eps, z, amap, P, K, mno, L, posterior);
}

// ---- rows between the domains: n floats per row ---------------------------------------------------------------------------------------
// scatter: full[r, :] = compact[r', :] (to one or two destinations; rows of absent agents are NOT touched: the caller zero-fills first)
__global__ void k_scatter_rows(const float* __restrict__ comp, float* __restrict__ full0, float* __restrict__ full1,
                               const int32_t* __restrict__ amap, int P, int K, int mno, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)P * K * n) return;
    const int rp = (int)(i / n), c = (int)(i - (long)rp * n);
    const size_t r = full_row_of(rp, P, K, mno, amap);
    const float v = comp[i];
    full0[r * n + c] = v;
    if (full1) full1[r * n + c] = v;
}
void launch_scatter_rows(const float* comp, float* full0, float* full1, const int32_t* amap, int P, int K, int mno, int n, hipStream_t s) {
    const long t = (long)P * K * n;
    if (t <= 0) return;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, s, comp, full0, full1, amap, P, K, mno, n);
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
