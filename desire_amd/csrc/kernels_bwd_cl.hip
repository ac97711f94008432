// kernels_bwd_cl.hip -- BPTT of the IOC module for (scene, k) groups of 64 / 96 / 128 agents: the cluster form of k_ioc_bwd
// (kernels_bwd.hip).  A group spans tpg = mno/32 workgroups of 32 rows, like the forward k_ioc_cl (kernels_rnn.hip); what they
// exchange per reverse step is d(pre-activation of the social embedding) = dpre_r [32, H] -- and that tensor is streamed to
// HBM anyway (it is the G operand of the social-fc weight gradient), so the hand-off costs one release / acquire per step and
// no extra traffic: a member stores its dpre_r(t), publishes (cluster.h), and after the group has arrived reads the other
// members' rows of a.dpre_r.
//
// The transpose of social pooling is taken in "pool, then contract" order (the forward's order):
//       dh_{t-1}[j] += sum_b ( sum_{i : j in bin b of i} dpre_r[i] ) . W_b^T
// i.e. per bin the dpre_r of the OBSERVERS of row j are summed (VALU, bit-masks over the whole group, double-buffered LDS
// operand) and one K = H contraction against W_b^T accumulates into registers -- no redundant work between the members
// (k_ioc_bwd contracts first and gathers afterwards, which in a cluster would make every member contract every row).
// Everything else (GRU cell backward, regression / score heads, streams for the weight-gradient GEMMs) is k_ioc_bwd's.
#include "cluster.h"
#include "kernels.h"

#define BCLMAXM 128

__device__ __forceinline__ void mma1c(f32x16& acc, const float* a_lane, const float4* __restrict__ b_lane, int G) {
    f32x16 t[1] = {acc};
    mma_groups<1>(t, a_lane, 0, b_lane, G);
    acc = t[0];
}

template <int H, int EV, int C>
__global__ __launch_bounds__((H / 32) * 64, 1) void k_ioc_bwd_cl(IocBwdArgs a, int* __restrict__ grp_cnt, int* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 32, NT = H / 32, NTHR = NT * 64, TPR = NTHR / TM, NCH = H / (4 * TPR);
    constexpr int E = EV + C + H, LD1 = H + 4, LD2 = 2 * H + 4, GH = H / 8, G2 = 2 * H / 8;
    const int B = a.G * a.G;
    const int KR = (2 * a.T + 7) / 8 * 8, LDR = KR + 4;
    const int tpg = a.mno / 32;
    float* A1 = smem;                         // [32][LD1]  da_c
    float* A2 = A1 + TM * LD1;                // [32][LD2]  da_r | da_u;  then the observer-pooled operand, double buffered [2][32][LD1]
    float* GD = A2 + 2 * TM * LD1;            // [mno][LD1] dpre_r(t) of the WHOLE group
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(GD + BCLMAXM * LD1);   // [32][B][2] neighbours of my row i in bin b
    unsigned long long* obs = masks + TM * B * 2;                                            // [32][B][2] observers of my row j in bin b
    float* pg = reinterpret_cast<float*>(obs + TM * B * 2);     // [mno][2] positions of the group at step t
    float* dsc = pg + BCLMAXM * 2;            // [32]
    float* wsc = dsc + TM;                    // [H]
    unsigned char* vld = reinterpret_cast<unsigned char*>(wsc + H);   // [mno]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + BCLMAXM);        // [2] bins in which one of my rows is observed
    float* DR = A2;                           // [32][LDR] regression-head operand (prologue only)

    const int lane = lane_id(), cb = wave_id(), tid = threadIdx.x;
    const int col = cb * 32 + (lane & 31);
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int tile_pos = blockIdx.x % tpg;
    const int n_tiles = a.R / TM;
    const int my_slot = tile_pos * TM + r8;
    const float* a1_lane = A1 + (lane & 31) * LD1 + 4 * (lane >> 5);
    const float* a2_lane = A2 + (lane & 31) * LD2 + 4 * (lane >> 5);
    const int rofs = 4 * (lane >> 5);         // + (i&3) + 8*(i>>2) = local row of accumulator element i
    for (int i = tid; i < H; i += NTHR) wsc[i] = a.w_score[i];

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int row0 = tile * TM;
        const int grow0 = row0 - tile_pos * TM;
        int* cnt = grp_cnt + grow0 / a.mno;
        const size_t tb = (size_t)row0 * a.T;
        const float* svu = a.sv_u + tb * H; const float* svc = a.sv_c + tb * H; const float* svr = a.sv_r + tb * H;
        const float* svx = a.sv_x + tb * E;
        float* o_dac = a.dac + tb * H; float* o_rh = a.rh + tb * H; float* o_hp = a.hprev + tb * H; float* o_dag = a.dag + tb * 2 * H;
        float* o_dpr = a.dpre_r + tb * H; float* o_dpv = a.dpre_v + tb * EV;
        auto tl = [&](int i, int t) { return (unsigned)((rofs + (i & 3) + 8 * (i >> 2)) * a.T + t); };
        __syncthreads();
        for (int i = tid; i < a.mno; i += NTHR) vld[i] = a.valid[agent_of_row(grow0 + i, a.K, a.mno)];
        if (tid < TM) dsc[tid] = a.dscore[row0 + tid];
        for (int i = tid; i < TM * KR; i += NTHR) {
            const int r = i / KR, c = i - r * KR;
            DR[r * LDR + c] = (c < 2 * a.T) ? a.dYr[(size_t)(row0 + r) * 2 * a.T + c] : 0.f;
        }
        __syncthreads();
        f32x16 dh = zero16();
        mma1c(dh, DR + (lane & 31) * LDR + 4 * (lane >> 5), a.WrT + ((size_t)cb * (KR / 8)) * 64 + lane, KR / 8);
        int published = 0;

        for (int t = a.T - 1; t >= 0; --t) {
            __syncthreads();
            // ---- P0: positions of the whole group, velocity of my rows, cleared masks ----
            for (int i = tid; i < a.mno; i += NTHR) {
                const float2 y = *reinterpret_cast<const float2*>(a.Y0 + ((size_t)(grow0 + i) * a.T + t) * 2);
                pg[i * 2] = y.x; pg[i * 2 + 1] = y.y;
            }
            if (tid < TM) {
                const int row = row0 + tid;
                const float2 y = *reinterpret_cast<const float2*>(a.Y0 + ((size_t)row * a.T + t) * 2);
                float2 pv;
                if (t > 0) pv = *reinterpret_cast<const float2*>(a.Y0 + ((size_t)row * a.T + t - 1) * 2);
                else { const int ag = agent_of_row(row, a.K, a.mno); pv = make_float2(a.p_last[(size_t)ag * 2], a.p_last[(size_t)ag * 2 + 1]); }
                a.vel[((size_t)row * a.T + t) * 2] = y.x - pv.x; a.vel[((size_t)row * a.T + t) * 2 + 1] = y.y - pv.y;
            }
            for (int i = tid; i < 4 * TM * B; i += NTHR) masks[i] = 0ull;        // masks and obs are contiguous
            if (tid < 2) occ[tid] = 0;
            __syncthreads();
            // ---- P1: neighbours of my rows (pooled rebuild, flags) and observers of my rows (gradient gather) ----
            {
                const float px = pg[my_slot * 2], py = pg[my_slot * 2 + 1];
                const bool me_valid = vld[my_slot] != 0;
                for (int j = q8; j < a.mno; j += TPR) {
                    if (j == my_slot) continue;
                    const float qx = pg[j * 2], qy = pg[j * 2 + 1];
                    if (vld[j]) {
                        const int b = neighbor_bin_dev(px, py, qx, qy, a.nb_w, a.nb_h, a.G, a.bin_tab);
                        if (b >= 0) atomicOr(&masks[(r8 * B + b) * 2 + (j >> 6)], 1ull << (j & 63));
                    }
                    if (me_valid) {                                     // row j pools MY hidden state in bin b2 of ITS window
                        const int b2 = neighbor_bin_dev(qx, qy, px, py, a.nb_w, a.nb_h, a.G, a.bin_tab);
                        if (b2 >= 0) { atomicOr(&obs[(r8 * B + b2) * 2 + (j >> 6)], 1ull << (j & 63)); atomicOr(&occ[b2 >> 5], 1u << (b2 & 31)); }
                    }
                }
            }
            // ---- GRU cell backward, part 1 (k_ioc_bwd) ----
            f32x16 dhp, rr, hp;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = rofs + (i & 3) + 8 * (i >> 2);
                const unsigned ix = tl(i, t) * H + col;
                const float u = svu[ix], c = svc[ix], r = svr[ix];
                const float hprev = (t > 0) ? a.sv_h[((size_t)(row0 + rl) * a.T + t - 1) * H + col]
                                            : a.Hx[(size_t)agent_of_row(row0 + rl, a.K, a.mno) * a.ldhx + col];
                const float dht = dh[i] + dsc[rl] * wsc[col];
                const float dau = dht * (hprev - c) * u * (1.0f - u);
                const float dc = dht * (1.0f - u);
                dhp[i] = dht * u;
                const float dac = dc * (1.0f - c * c);
                A1[rl * LD1 + col] = dac;
                A2[rl * LD2 + H + col] = dau;
                o_dac[ix] = dac; o_rh[ix] = r * hprev; o_hp[ix] = hprev;
                o_dag[tl(i, t) * 2 * H + H + col] = dau;
                rr[i] = r; hp[i] = hprev;
            }
            __syncthreads();
            if (a.pool_flags && q8 == 0) {
                unsigned long long fl = 0ull;
                for (int b = 0; b < B; ++b) fl |= (unsigned long long)((masks[(r8 * B + b) * 2] | masks[(r8 * B + b) * 2 + 1]) != 0ull) << b;
                a.pool_flags[(size_t)(row0 + r8) * a.T + t] = fl;
            }
            f32x16 drh = zero16(), der = zero16(), dev = zero16();
            mma1c(drh, a1_lane, a.WcT_h + ((size_t)cb * GH) * 64 + lane, GH);
            mma1c(der, a1_lane, a.WcT_er + ((size_t)cb * GH) * 64 + lane, GH);
            if (cb == 0) mma1c(dev, a1_lane, a.WcT_ev + lane, GH);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = rofs + (i & 3) + 8 * (i >> 2);
                const float dr = drh[i] * hp[i];
                dhp[i] += drh[i] * rr[i];
                const float dar = dr * rr[i] * (1.0f - rr[i]);
                A2[rl * LD2 + col] = dar;
                o_dag[tl(i, t) * 2 * H + col] = dar;
            }
            __syncthreads();
            {
                f32x16 dhg = zero16();
                mma1c(dhg, a2_lane, a.WgT_h + ((size_t)cb * G2) * 64 + lane, G2);
                mma1c(der, a2_lane, a.WgT_er + ((size_t)cb * G2) * 64 + lane, G2);
                if (cb == 0) mma1c(dev, a2_lane, a.WgT_ev + lane, G2);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rl = rofs + (i & 3) + 8 * (i >> 2);
                    dhp[i] += dhg[i];
                    const unsigned ixx = tl(i, t) * E;
                    const float er = svx[ixx + EV + C + col];
                    const float dpr = er > 0.f ? der[i] : 0.f;
                    GD[(tile_pos * TM + rl) * LD1 + col] = dpr;                 // my rows of the group tile (streamed out below)
                    if (cb == 0 && (lane & 31) < EV) {
                        const float ev = svx[ixx + (lane & 31)];
                        o_dpv[tl(i, t) * EV + (lane & 31)] = ev > 0.f ? dev[i] : 0.f;
                    }
                }
            }
            __syncthreads();
            // dpre_r(t) of my rows -> HBM (operand of the social-fc weight gradient AND what the other members read): row-major copy with
            // 8-byte write-through stores, then the fence-free arrival (cluster.h: *_wt)
            for (int i = tid; i < TM * (H >> 1); i += NTHR) {
                const int r = i / (H >> 1), c2 = i - r * (H >> 1);
                const float2 v = *reinterpret_cast<const float2*>(GD + (tile_pos * TM + r) * LD1 + 2 * c2);
                st_agent_u64(o_dpr + ((size_t)r * a.T + t) * H + 2 * c2, make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)));
            }
            group_publish_wt(cnt);                                // (includes a workgroup barrier)
            ++published;
            // ---- while the others arrive: pooled_b[i] = sum_{j in bin b of i} h_{t-1}[j] -> HBM (operand of the social-fc weight
            //      gradient); the neighbours' h_{t-1} come from the forward's saves (or Hx at t = 0), any member of the group ----
            for (int b = 0; b < B; ++b) {
                const unsigned long long m0 = masks[(r8 * B + b) * 2], m1 = masks[(r8 * B + b) * 2 + 1];
                if (a.pool_flags && !(m0 | m1)) continue;                        // flagged-empty blocks are never read
                float4 s[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int wd = 0; wd < 2; ++wd) {
                    unsigned long long m2 = wd ? m1 : m0;
                    while (m2) {
                        const int j = wd * 64 + __ffsll((long long)m2) - 1;
                        m2 &= m2 - 1;
                        const float* src = (t > 0) ? a.sv_h + ((size_t)(grow0 + j) * a.T + t - 1) * H
                                                   : a.Hx + (size_t)agent_of_row(grow0 + j, a.K, a.mno) * a.ldhx;
#pragma unroll
                        for (int c = 0; c < NCH; ++c) {
                            const float4 v = *reinterpret_cast<const float4*>(src + q8 * 4 + c * 4 * TPR);
                            s[c].x += v.x; s[c].y += v.y; s[c].z += v.z; s[c].w += v.w;
                        }
                    }
                }
                float* dst = a.pooled + (((size_t)(row0 + r8) * a.T + t) * B + b) * H + q8 * 4;
#pragma unroll
                for (int c = 0; c < NCH; ++c) *reinterpret_cast<float4*>(dst + c * 4 * TPR) = s[c];
            }
            // ---- the other members' dpre_r(t) ----
            group_wait_wt(cnt, tpg * published, err);
            for (int tp = 0; tp < tpg; ++tp) {
                if (tp == tile_pos) continue;
                for (int i = tid; i < TM * (H >> 1); i += NTHR) {
                    const int r = i / (H >> 1), c2 = i - r * (H >> 1);
                    const uint2 v = ld_agent_u64(a.dpre_r + ((size_t)(grow0 + tp * TM + r) * a.T + t) * H + 2 * c2);
                    *reinterpret_cast<float2*>(GD + (tp * TM + r) * LD1 + 2 * c2) = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                }
            }
            __syncthreads();
            // ---- social pooling backward: pool the observers' dpre_r per bin, contract with W_b^T into registers ----
            auto build = [&](int b, int buf) {
                float* ab = A2 + buf * TM * LD1 + r8 * LD1;
                float4 s[NCH];
#pragma unroll
                for (int c = 0; c < NCH; ++c) s[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int wd = 0; wd < 2; ++wd) {
                    unsigned long long m2 = obs[(r8 * B + b) * 2 + wd];
                    while (m2) {
                        const int i2 = wd * 64 + __ffsll((long long)m2) - 1;
                        m2 &= m2 - 1;
                        const float* src = GD + i2 * LD1 + q8 * 4;
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
            f32x16 nb = zero16();
            unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
            om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
            int buf = 0;
            if (om) build(__ffsll((long long)om) - 1, 0);
            __syncthreads();
            while (om) {
                const int b = __ffsll((long long)om) - 1;
                om &= om - 1;
                if (om) build(__ffsll((long long)om) - 1, buf ^ 1);
                mma1c(nb, A2 + buf * TM * LD1 + (lane & 31) * LD1 + 4 * (lane >> 5), a.WsT + ((size_t)(b * NT + cb) * GH) * 64 + lane, GH);
                __syncthreads();
                buf ^= 1;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dh[i] = dhp[i] + nb[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) a.dHx_rows[(size_t)(row0 + rofs + (i & 3) + 8 * (i >> 2)) * H + col] += dh[i];
    }
}

static size_t ioc_bwd_cl_lds(const IocBwdArgs& a) {
    const int H = a.H, LD1 = H + 4, B = a.G * a.G, TM = 32;
    size_t f = (size_t)TM * LD1 + 2 * TM * LD1 + (size_t)BCLMAXM * LD1 + BCLMAXM * 2 + TM + H;
    return f * sizeof(float) + (size_t)4 * TM * B * 8 + BCLMAXM + 8 + 64;
}
template <int H>
static int launch_t(const IocBwdArgs& a, int* grp_cnt, int* err, hipStream_t s) {
    auto kern = k_ioc_bwd_cl<H, 16, 32>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const size_t lds = ioc_bwd_cl_lds(a);
    if (lds > 160 * 1024) return -1;
    const int tpg = a.mno / 32, n_tiles = a.R / 32;
    int grid = n_tiles < 256 ? n_tiles : 256;                // one workgroup per CU: all of them resident
    grid -= grid % tpg;
    if (grid < tpg) return -1;
    hipLaunchKernelGGL(kern, dim3(grid), dim3((H / 32) * 64), lds, s, a, grp_cnt, err);
    return 0;
}
// groups of 64 / 96 / 128 agents (mno a multiple of 32), H <= 128, at most 16 social bins (LDS budget); != 0 otherwise
int launch_ioc_bwd_cluster(const IocBwdArgs& a, int* grp_cnt, int* err, hipStream_t s) {
    if (a.mno % 32 || a.mno > BCLMAXM || a.R % 32) return -1;
    if (a.H == 128) return launch_t<128>(a, grp_cnt, err, s);
    if (a.H == 64) return launch_t<64>(a, grp_cnt, err, s);
    return -1;
}
