// kernels_bwd_x3.hip -- weight-gradient reductions with split-bf16 operands (dims.bf16 = 2 while training).
//
// Every weight gradient of the model is one A^T.G reduction over all rows and steps (kernels_bwd.hip: k_gemm_tn2), on the fp32
// matrix pipe 37 of the training step's 126 ms.  The same reduction with every fp32 product as three bf16 MFMAs (split.h:
// x = hi + lo, hi.hi + lo.hi + hi.lo, fp32 accumulation, ~2^-16 relative per product) costs 3/16 of the matrix time.
//
// Layout: the contraction runs over m (rows x steps), which is the SLOW index of both operands in HBM, while a bf16 MFMA lane
// wants 8 consecutive contraction indices of one output row.  The staging through LDS does the transposition: a thread loads
// the SAME four columns of P consecutive m-rows (coalesced float4 per row), splits them, and writes per column one P-element run
// of the [column][32 m] bf16 image of each piece -- so fragments are single 16-byte LDS reads.  Rows of the image are 64 bytes:
// the four 16-byte slots of a row are XOR-swizzled and the row order is column-major over the thread's four columns, which keeps
// both the 8-byte writes (lanes = consecutive float4 columns) and the 16-byte fragment reads (lanes = consecutive columns) spread
// over all banks.
#include "common.h"
#include "kernels.h"

#include "split.h"
#include <cstdio>

namespace {

template <int BC> __device__ __forceinline__ int img_row(int col) { return (col & 3) * (BC / 4) + (col >> 2); }
__device__ __forceinline__ int img_swz(int col) { return ((col >> 3) ^ col) & 3; }

// P rows x 4 columns (r[j] = columns col0..col0+3 of chunk row P*r0 + j) -> the NP piece images
template <int BC, int P, int NP>
__device__ __forceinline__ void put_rows(u16* img, int piece_stride, int col0, int r0, const float4 (&r)[P]) {
    const int slot = (P * r0) >> 3, sub = ((P * r0) & 7) * 2;
#pragma unroll
    for (int comp = 0; comp < 4; ++comp) {
        unsigned pc[NP][P / 2];
#pragma unroll
        for (int jj = 0; jj < P / 2; ++jj) {
            const float4 x0 = r[2 * jj], x1 = r[2 * jj + 1];
            const float v0 = comp == 0 ? x0.x : comp == 1 ? x0.y : comp == 2 ? x0.z : x0.w;
            const float v1 = comp == 0 ? x1.x : comp == 1 ? x1.y : comp == 2 ? x1.z : x1.w;
            unsigned t[NP];
            splitp<NP>(v0, v1, t);
#pragma unroll
            for (int i = 0; i < NP; ++i) pc[i][jj] = t[i];
        }
        const int col = col0 + comp;
        char* dst = reinterpret_cast<char*>(img) + img_row<BC>(col) * 64 + ((slot ^ img_swz(col)) << 4) + sub;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            char* d = dst + (size_t)i * piece_stride * 2;
            if (P == 8) *reinterpret_cast<uint4*>(d) = make_uint4(pc[i][0], pc[i][1], pc[i][2], pc[i][3]);
            else if (P == 4) *reinterpret_cast<uint2*>(d) = make_uint2(pc[i][0], pc[i][1]);
            else *reinterpret_cast<unsigned*>(d) = pc[i][0];
        }
    }
}
template <int BC>
__device__ __forceinline__ uint4 get_frag(const u16* img, int col, int slot) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(img) + img_row<BC>(col) * 64 + ((slot ^ img_swz(col)) << 4));
}

#ifdef DESIRE_IOC_TIMING
__device__ long long g_tn_ticks[8];
#define TICKT(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#else
#define TICKT(k)
#endif
// (WK*64) x (WN*64) output tile per workgroup, wave = 2x2 tiles of 32x32, 32-row chunks of m double-buffered in LDS as piece images
template <int WK, int WN, bool CONV, int NP, bool LISTS = false>      // LISTS: the row-list form (TnArgs::rowlist), an instantiation of its own
__global__ __launch_bounds__(256, 2) void k_gemm_tn2_xp(TnArgs a, ConvGather cg) {
#ifdef DESIRE_IOC_TIMING
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    constexpr int BK = WK * 64, BN = WN * 64, QA = BK / 4, QG = BN / 4, PA = 32 / (256 / QA), PG = 32 / (256 / QG);
    constexpr int IA = BK * 32, IG = BN * 32;                       // bf16 elements of one piece image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x[];
    u16* As = reinterpret_cast<u16*>(smem_x);                       // [2][NP][BK][32]
    u16* Gs = As + 2 * NP * IA;                                     // [2][NP][BN][32]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int nbn = (a.N + BN - 1) / BN;
    const int bx = blockIdx.x, by = blockIdx.y;
    const int bk = (bx / nbn) * BK, bn = (bx % nbn) * BN;
    const int wk = w / WN, wn = w % WN;
    // slice y takes the 32-row chunks y, y + nslices, y + 2 nslices, ..: the workgroups in flight read NEIGHBOURING chunks (contiguous
    // ranges per slice put every stream a multiple of megabytes apart -- the same HBM channel at the same time)
    // row lists (TnArgs::rowlist): this k-block's rows are list entries [0, bintotal[b]) instead of all M rows
    const int* const rl = LISTS ? a.rowlist + a.binbase[bk / a.fcols] : nullptr;
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
    // two register stages: the chunk after next is already in flight while the current one is multiplied (the contraction is short
    // now -- 24 bf16 MFMAs per wave and chunk -- so one chunk in flight per workgroup left the kernel waiting on HBM latency)
    float4 raA[PA], rgA[PG], raB[PA], rgB[PG];
    // list mode: the row indices of the chunk the NEXT gload takes (chunks follow each other at a fixed stride there) are fetched one gload ahead,
    // so a chunk's row loads do not wait for their own indices
    // Two index sets, tied to the two register stages: a gload FIRST requests the indices of the next gload's chunk (into the other set), THEN its
    // own rows -- so the wait for those indices (vmcnt = this chunk's row loads) leaves the rows in flight.  Requested after the rows, the indices
    // were the youngest loads and waiting for them drained the whole prefetch (3.8 instead of 2.1 ms for the social-fc gradient).
    constexpr int NI = LISTS ? PA : 1, NJ = LISTS ? PG : 1;
    int iaA[NI], igA[NJ], iaB[NI], igB[NJ];
    auto iload = [&](int (&ia)[NI], int (&ig)[NJ], long m0) {
        if constexpr (LISTS) {
#pragma unroll
            for (int j = 0; j < PA; ++j) { const long m = m0 + PA * ra0 + j; ia[j] = m < m_hi ? rl[m] : 0; }
#pragma unroll
            for (int j = 0; j < PG; ++j) { const long m = m0 + PG * rg0 + j; ig[j] = m < m_hi ? rl[m] : 0; }
        }
    };
    iload(iaA, igA, m_lo);
    auto gload = [&](float4 (&ra)[PA], float4 (&rg)[PG], long m0, int (&ia)[NI], int (&ig)[NJ], int (&ian)[NI], int (&ign)[NJ]) {
        iload(ian, ign, m0 + step);
        // convolution layers whose small grid is PA pixels wide (every large one here: 8 x 8 with eight rows per thread, 4 x 4 with four): a thread's
        // rows are ONE row of the grid -- one sample, one py, px = j -- so the eight gathers share a base address and differ by constant strides
        // (the general form below forms eight 64-bit addresses out of shifts and masks and sat on the register limit: a spill inside this loop
        // waits for vmcnt(0), i.e. for both prefetched chunks, and the deconv3 weight gradient swung between 5.2 and 8.3 ms with unrelated edits)
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
            const long m = m0 + PA * ra0 + j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_hi && ka) {
                if (CONV) {
                    long n; int p, py, px;                       // (sample, small-grid pixel) of row m
                    if (ps_sh >= 0) { n = m >> (2 * ps_sh); p = (int)(m & (PP - 1)); py = p >> ps_sh; px = p & (cg.Ps - 1); }
                    else { n = m / PP; p = (int)(m - n * PP); py = p / cg.Ps; px = p - py * cg.Ps; }
                    const int qy = cg.stride * py + ky - cg.pad, qx = cg.stride * px + kx - cg.pad;
                    if (qy >= 0 && qy < cg.Pl && qx >= 0 && qx < cg.Pl)
                        v = *reinterpret_cast<const float4*>(a.A + (((size_t)n * cg.Pl + qy) * cg.Pl + qx) * cg.Cl + cl);
                } else {
                    long mr = m;
                    if constexpr (LISTS) mr = (long)ia[j];
                    v = *reinterpret_cast<const float4*>(a.A + (size_t)mr * a.lda + kcol);
                    if (a.flags && !((a.flags[m] >> (kcol / a.fcols)) & 1ull)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const long m = m0 + PG * rg0 + j;
            long mr = m;
            if constexpr (LISTS) mr = (long)ig[j];
            rg[j] = (m < m_hi && na) ? *reinterpret_cast<const float4*>(a.G + (size_t)mr * a.ldg + bn + 4 * qg) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf, const float4 (&ra)[PA], const float4 (&rg)[PG]) {
        put_rows<BK, PA, NP>(As + buf * NP * IA, IA, 4 * qa, ra0, ra);
        put_rows<BN, PG, NP>(Gs + buf * NP * IG, IG, 4 * qg, rg0, rg);
    };
    // block-sparse A (a.flags): only chunks with a set flag bit inside this workgroup's k-block are visited.  The scan looks 64 chunks
    // ahead at a time (lane = chunk: the OR of its 32 flag words), so its memory latency is paid once per 2 MB of operands
    unsigned long long kmask = ~0ull;
    if (a.flags) {
        const int b_lo = bk / a.fcols, b_hi = min((bk + BK - 1) / a.fcols, 63);
        kmask = (b_hi - b_lo >= 63) ? ~0ull : (((1ull << (b_hi - b_lo + 1)) - 1ull) << b_lo);
    }
    long grp = -1;                                         // 64-chunk group (of this slice's sequence) the live mask belongs to
    unsigned long long live = 0ull;
    auto next_live = [&](long m0) {                        // first chunk of this slice at or after m0 with something in the k-block
        if (!a.flags) return m0;
        while (m0 < m_hi) {
            const long q = (m0 - m_lo) / step;             // position in the slice's sequence
            if ((q >> 6) != grp) {
                grp = q >> 6;
                const long mb = m_lo + ((grp << 6) + lane) * step;
                unsigned long long f = 0ull;
                if (mb + 32 <= a.M) {
                    const uint4* fp = reinterpret_cast<const uint4*>(a.flags + mb);
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const uint4 v = fp[i]; f |= ((unsigned long long)(v.y | v.w) << 32) | (v.x | v.z); }
                } else {
                    for (long m = mb; m < a.M; ++m) f |= a.flags[m];
                }
                f &= kmask;
                live = __ballot(((unsigned)f | (unsigned)(f >> 32)) != 0u);
            }
            const unsigned long long rest = live >> (q & 63);
            if (rest) return m0 + step * (long)(__ffsll((long long)rest) - 1);
            m0 = m_lo + ((grp + 1) << 6) * step;
        }
        return m0;
    };
    const bool mine = bk + wk * 64 < a.Kd && bn + wn * 64 < a.N;       // a wave whose whole strip lies past Kd / N has only zeros to multiply
    auto compute = [&](int buf) {
        if (!mine) return;
        const u16* ab = As + buf * NP * IA;
        const u16* gb = Gs + buf * NP * IG;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            uint4 gf[2][NP];                                    // (the A fragments one 32-row block at a time: 8 registers fewer in flight)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int i = 0; i < NP; ++i) gf[v][i] = get_frag<BN>(gb + i * IG, wn * 64 + 32 * v + c, 2 * kb + hi);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                uint4 af[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) af[i] = get_frag<BK>(ab + i * IA, wk * 64 + 32 * u + c, 2 * kb + hi);
#pragma unroll
                for (int v = 0; v < 2; ++v) acc[u][v] = mfma_xp<NP>(af, gf[v], acc[u][v]);
            }
        }
    };
    // at the loop top: LDS[buf] holds chunk mA, stage B holds chunk mB, stage A holds chunk mC (each only if < m_hi)
    long mA = next_live(m_lo);
    if (mA < m_hi) gload(raA, rgA, mA, iaA, igA, iaB, igB);
    long mB = mA < m_hi ? next_live(mA + step) : m_hi;
    if (mB < m_hi) gload(raB, rgB, mB, iaB, igB, iaA, igA);
    if (mA < m_hi) lstore(0, raA, rgA);
    long mC = mB < m_hi ? next_live(mB + step) : m_hi;
    if (mC < m_hi) gload(raA, rgA, mC, iaA, igA, iaB, igB);
    __syncthreads();
    int buf = 0;
    TICKT(0)
    while (mA < m_hi) {
        compute(buf);
        TICKT(1)
        if (mB < m_hi) lstore(buf ^ 1, raB, rgB);
        TICKT(2)
        __syncthreads();
        TICKT(3)
        buf ^= 1;
        long mD = mC < m_hi ? next_live(mC + step) : m_hi;
        if (mD < m_hi) gload(raB, rgB, mD, iaB, igB, iaA, igA);
        TICKT(4)
        mA = mB; mB = mC; mC = mD;
        if (!(mA < m_hi)) break;
        compute(buf);
        TICKT(1)
        if (mB < m_hi) lstore(buf ^ 1, raA, rgA);
        TICKT(2)
        __syncthreads();
        TICKT(3)
        buf ^= 1;
        mD = mC < m_hi ? next_live(mC + step) : m_hi;
        if (mD < m_hi) gload(raA, rgA, mD, iaA, igA, iaB, igB);
        TICKT(4)
        mA = mB; mB = mC; mC = mD;
    }
#ifdef DESIRE_IOC_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 3 && tid == 0)
        for (int k = 0; k < 8; ++k) g_tn_ticks[k] = tacc[k];
#endif
    float* out = a.partial + (size_t)by * a.Kd * a.N;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = bk + wk * 64 + 32 * u + acc_row(i), n = bn + wn * 64 + 32 * v + c;
                if (k < a.Kd && n < a.N) out[(size_t)k * a.N + n] = acc[u][v][i];
            }
}

template <int WK, int WN, bool CONV, bool LISTS = false>
void launch_t(const TnArgs& a, const ConvGather& cg, int nblocks, hipStream_t s) {
    constexpr int NP = 2;
    const size_t lds = (size_t)2 * NP * (WK + WN) * 64 * 32 * sizeof(u16);
    allow_big_lds(k_gemm_tn2_xp<WK, WN, CONV, NP, LISTS>);
    hipLaunchKernelGGL((k_gemm_tn2_xp<WK, WN, CONV, NP, LISTS>), dim3(nblocks, a.nslices), dim3(256), lds, s, a, cg);
#ifdef DESIRE_IOC_TIMING
    if (a.M > 1000000 && !CONV && !a.flags) {
        long long host[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tn_ticks), sizeof(host));
        const char* nm[5] = {"prologue", "compute (frag reads + mfma)", "lstore (split + LDS writes; waits for its loads)", "barrier", "next_live + gload issue"};
        long long tot = 0; for (int k = 0; k < 5; ++k) tot += host[k];
        fprintf(stderr, "k_gemm_tn2_xp<%d,%d> Kd=%d N=%d slices=%d: total %lld cycles\n", WK, WN, a.Kd, a.N, a.nslices, tot);
        for (int k = 0; k < 5; ++k) fprintf(stderr, "  %-50s %12lld  %5.1f %%\n", nm[k], host[k], 100.0 * host[k] / (double)tot);
    }
#endif
}

}  // namespace

// the partial-tile stage of launch_gemm_tn / launch_conv_wgrad's "big" forms with split operands (the slice reduction stays the caller's)
void launch_gemm_tn2_split(const TnArgs& a, const ConvGather* cg, bool narrow_n, hipStream_t s) {
    if (narrow_n) {
        const int nb = (a.Kd + 255) / 256;
        if (cg) launch_t<4, 1, true>(a, *cg, nb, s); else launch_t<4, 1, false>(a, ConvGather{}, nb, s);
    } else {
        const int nb = ((a.Kd + 127) / 128) * ((a.N + 127) / 128);
        if (cg) launch_t<2, 2, true>(a, *cg, nb, s);
        else if (a.rowlist) launch_t<2, 2, false, true>(a, ConvGather{}, nb, s);
        else launch_t<2, 2, false>(a, ConvGather{}, nb, s);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Data-gradient convolutions of the CVAE decoder (k_conv_gather with the layers' roles swapped) with split operands: the
// input image is split ONCE when it is staged ([pixel][CI] bf16 images of hi and lo in LDS), so every A fragment is a single
// 16-byte LDS read per piece; weights are [hi | lo] packs in bf16-MFMA fragment order per tap ("*/Wbwd16").  A wave owns MT
// row blocks of 32 output pixels x one 32-channel n-tile; every weight fragment feeds all MT blocks.
// ------------------------------------------------------------------------------------------------------------------
namespace {
template <int CI, int IW, int OW, int STRIDE, int PAD, int CO, int SPW>
__global__ __launch_bounds__(256, 2) void k_conv_gather_x3(ConvArgs a) {
    constexpr int PIX = OW * OW, ROWS = SPW * PIX, LDB = CI + 8, NT = CO / 32, G16 = CI / 16;
    constexpr int WM = 4 / NT, MT = ROWS / (32 * WM);                 // waves along m; row blocks per wave
    static_assert(NT == 2 || NT == 4, "n-tiles per workgroup");
    static_assert(ROWS % (32 * WM) == 0, "rows per workgroup");
    constexpr int NPX = SPW * IW * IW, IMG = (NPX + 1) * LDB;         // + one zero pixel for taps outside the image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];
    u16* img = reinterpret_cast<u16*>(smem_c);                        // [2 pieces][NPX + 1][LDB]
    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int s0 = blockIdx.x * SPW;
    // position of pixel (y, x) of a sample's image.  Stride 2: the four parity classes are stored as separate (IW/2)^2 sub-images -- a tap
    // reads ONE class, so the lanes of a fragment read (ox, oy)-consecutive pixels LDB apart (80 / 144 bytes: conflict-free 16-byte
    // reads) instead of every second pixel (160 bytes apart: 8 lanes per bank group)
    auto pos = [](int y, int x) {
        if (STRIDE == 2) return (((y & 1) * 2 + (x & 1)) * (IW / 2) + (y >> 1)) * (IW / 2) + (x >> 1);
        return y * IW + x;
    };
    for (int i = tid; i < LDB; i += 256) { img[NPX * LDB + i] = 0; img[IMG + NPX * LDB + i] = 0; }
    constexpr int Q = CI / 4;
    for (int i = tid; i < NPX * Q; i += 256) {
        const int pix = i / Q, c4 = i - pix * Q;
        const int ls = pix / (IW * IW), pp = pix - ls * IW * IW;
        const int smp = s0 + ls;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (smp < a.n) v = *reinterpret_cast<const float4*>(a.in + ((size_t)s0 * IW * IW + pix) * CI + c4 * 4);
        unsigned p0[2], p1[2];
        splitp<2>(v.x, v.y, p0); splitp<2>(v.z, v.w, p1);
        const int at = (ls * IW * IW + pos(pp / IW, pp % IW)) * LDB + c4 * 4;
        *reinterpret_cast<uint2*>(img + at) = make_uint2(p0[0], p1[0]);
        *reinterpret_cast<uint2*>(img + IMG + at) = make_uint2(p0[1], p1[1]);
    }
    __syncthreads();
    const int nt = w % NT, mt0 = (w / NT) * MT;
    const int hi = lane >> 5;
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
    const uint4* Wp = reinterpret_cast<const uint4*>(a.Wp);
    constexpr size_t PLO = (size_t)25 * NT * G16 * 64;                 // uint4 from the hi pack to the lo pack
    const uint4* bl = Wp + ((size_t)nt * G16) * 64 + lane;
    // The 25 * G16 fragment pairs of this wave's n-tile are walked as ONE stream (tap-major): a ring of RD pairs in flight.  With one pair
    // ahead -- MT * 3 MFMAs, 100 to 200 matrix cycles, against ~1000 cycles for a fragment to arrive from L2 -- every k-group waited for its
    // fragments: 3.76 / 2.58 ms for the two layers where the MFMAs are 0.5 ms.
    constexpr int NG = 25 * G16, RD = 10;
    static_assert(NG % RD == 0 && (G16 & (G16 - 1)) == 0, "whole rings; power-of-two k-groups per tap");
    uint4 b[RD][2];
    auto frag_at = [&](int idx) { return bl + ((size_t)(idx / G16) * NT * G16 + (idx & (G16 - 1))) * 64; };     // packs are [tap][n-tile][k-group]
#pragma unroll
    for (int j = 0; j < RD; ++j) { const uint4* q = frag_at(j); b[j][0] = q[0]; b[j][1] = q[PLO]; }
    __builtin_amdgcn_sched_barrier(0);
    const u16* ap[MT];
#pragma unroll 1
    for (int c0 = 0; c0 < NG; c0 += RD) {
#pragma unroll
        for (int j = 0; j < RD; ++j) {
            const int idx = c0 + j, tap = idx / G16, g = idx & (G16 - 1);
            if (j == 0 || g == 0) {                            // (g == 0 is a compile-time test only when RD is a multiple of G16; cheap either way)
                const int ky = tap / 5, kx = tap - ky * 5;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int iy = oy[m] * STRIDE + ky - PAD, ix = ox[m] * STRIDE + kx - PAD;
                    const bool ok = iy >= 0 && iy < IW && ix >= 0 && ix < IW;
                    ap[m] = img + (ok ? sm[m] * IW * IW + pos(iy, ix) : NPX) * LDB + 8 * hi;
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                uint4 av[2];
                av[0] = *reinterpret_cast<const uint4*>(ap[m] + g * 16);
                av[1] = *reinterpret_cast<const uint4*>(ap[m] + IMG + g * 16);
                acc[m] = mfma_xp<2>(av, b[j], acc[m]);
            }
            if (idx + RD < NG) { const uint4* q = frag_at(idx + RD); b[j][0] = q[0]; b[j][1] = q[PLO]; }
            __builtin_amdgcn_sched_barrier(0);
        }
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
template <int CI, int IW, int OW, int STRIDE, int PAD, int CO, int SPW>
void launch_cg(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * (SPW * IW * IW + 1) * (CI + 8) * sizeof(u16);
    allow_big_lds(k_conv_gather_x3<CI, IW, OW, STRIDE, PAD, CO, SPW>);
    hipLaunchKernelGGL((k_conv_gather_x3<CI, IW, OW, STRIDE, PAD, CO, SPW>), dim3((a.n + SPW - 1) / SPW), dim3(256), lds, s, a);
}
}  // namespace
// a.Wp = the [hi | lo] pack ("vae_dec/deconv3/Wbwd16", "vae_dec/deconv2/Wbwd16")
void launch_conv2_x3(const ConvArgs& a, hipStream_t s) { launch_cg<32, 16, 8, 2, 1, 64, 1>(a, s); }     // [n,16,16,32] -> [n,8,8,64]
void launch_conv3_x3(const ConvArgs& a, hipStream_t s) { launch_cg<64, 8, 4, 1, 0, 128, 4>(a, s); }     // [n,8,8,64]  -> [n,4,4,128]

// ------------------------------------------------------------------------------------------------------------------
// IOC BPTT with split operands (k_ioc_bwd of kernels_bwd.hip, groups of up to 32 agents): the gate-gradient tiles da_c, da_r | da_u and
// dpre_r are written as [hi | lo] bf16 images by the lanes that produce them (same LDS bytes as the fp32 tiles they replace), so every
// data-gradient contraction reads single 16-byte fragments per piece and runs as three bf16 MFMAs per fp32 product.  The row-compacted
// dpool of the fp32 kernel is gone: the dense 32-row contraction per bin costs 24 bf16 MFMAs per wave, less than one packed chunk did.
// ------------------------------------------------------------------------------------------------------------------
typedef float f32x4v_x __attribute__((ext_vector_type(4)));
template <int TM> struct BwdMask { typedef unsigned long long type; };
template <> struct BwdMask<32> { typedef unsigned type; };
__device__ __forceinline__ int ffsm(unsigned m) { return __ffs((int)m); }
__device__ __forceinline__ void mma1b(f32x16& acc, const float* a_lane, const float4* __restrict__ b_lane, int G) {
    f32x16 t[1] = {acc};
    mma_groups<1>(t, a_lane, 0, b_lane, G);
    acc = t[0];
}
namespace {
__device__ __forceinline__ float f4e(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }
// ------------------------------------------------------------------------------------------------------------------
// Every contraction runs TRANSPOSED (mmax_groups SWAP: the packs are the A operand, the images the B operand): a lane holds ONE tile
// row and runs of four consecutive hidden columns, so the saved streams are read and the gradient streams written 16 bytes at a time,
// the piece images take 8-byte writes, and the per-element stream offsets shrink to four per stream (the row-major kernel of round 3
// spilled 54 dwords of them and took 4.3 ms longer per 81 920-row step).  Bias column sums by a butterfly over the rows (colsum16).
// ------------------------------------------------------------------------------------------------------------------
// gate / candidate contractions of the BPTT step: one k-group of pack fragments ahead (a deeper ring was measured in round 5 and was not faster:
// docs/DESIGN_DETAIL.md section 13 item 3)
#define BWDX3_MM(NB) mmax_groups<NB, 2, true>
#ifdef DESIRE_IOC_TIMING
#define TICKB(k) { const long long now_ = clock64(); tacc[k] += now_ - tprev; tprev = now_; }
#else
#define TICKB(k)
#endif
template <int H, int EV, int C, bool PAD = false>      // PAD: padded tiles (IocBwdArgs.gpt); a template parameter so that the packed form is unchanged
__global__ __launch_bounds__((H / 32) * 64, 2) void k_ioc_bwd_x3(IocBwdArgs a) {
#ifdef DESIRE_IOC_TIMING
    long long tacc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev = clock64();
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TM = 32;
    constexpr int NT = H / 32, NTHR = NT * (TM / 32) * 64, TPR = NTHR / TM, NCH = H / (4 * TPR);
    typedef typename BwdMask<TM>::type mask_t;
    constexpr int E = EV + C + H, LD1 = H + 4, LDB1 = H + 8, LDB2 = 2 * H + 8, G16 = H / 16, G32 = 2 * H / 16;
    constexpr int ILO1 = TM * LDB1, ILO2 = TM * LDB2;                      // bf16 elements from the hi image to the lo image
    const int B = a.G * a.G;
    const int KR = (2 * a.T + 7) / 8 * 8, LDR = KR + 4;                   // regression-head operand width
    // LDS (73 KB at H = 128, so two workgroups share a CU).  The MFMA operands live as [hi | lo] bf16 images (same bytes as an fp32 tile):
    float* A1 = smem;                         // [32][LD1]  fp32: h_{t-1} (part 1, pooled rebuild);  then the neighbour gradient
    float* A2 = A1 + TM * LD1;                // images [2][32][LDB2] of da_r | da_u;  then dpool_b (fp32 [32][LD1]), double buffered
    u16* I2 = reinterpret_cast<u16*>(A2);
    u16* I3 = reinterpret_cast<u16*>(A2 + 2 * TM * LD1);              // images [2][32][LDB1]: da_c (part 1), then dpre_r (part 2, bins)
    float* DP = A2, *HP = A1, *NB = A1;
    mask_t* masks = reinterpret_cast<mask_t*>(I3 + 2 * ILO1);         // [TM][B] neighbours of i in bin b (bit = slot)
    mask_t* obs = masks + TM * B;                                     // [TM][B] observers of j in bin b
    float* pc = reinterpret_cast<float*>(obs + TM * B);   // [32][2]
    float* dsc = pc + TM * 2;                 // [32]
    float* wsc = dsc + TM;                    // [H]
    unsigned char* vld = reinterpret_cast<unsigned char*>(wsc + H);   // [32]
    unsigned* occ = reinterpret_cast<unsigned*>(vld + TM);            // [2] bins that hold a neighbour anywhere in the tile
    uint2* lut = reinterpret_cast<uint2*>(occ + 2);                   // [16] nibble -> 4 bf16 (0.0 / 1.0): A fragments of the scatter MFMAs
    float* DR = A2;                           // [32][LDR] regression-head operand (prologue only; 2T <= 2H assumed)

    const int lane = lane_id(), w = wave_id(), tid = threadIdx.x;
    const int cb = w % NT, mt = w / NT;
    const int row0 = blockIdx.x * TM;
    const int lr = lane & 31, hi = lane >> 5;              // this lane's tile row; its column runs start at c0 + 8 q
    const int c0 = cb * 32 + 4 * hi;
    const int r8 = tid / TPR, q8 = tid % TPR;
    const int my_row = min(row0 + r8, a.R - 1);
    const int grp_base = (r8 / a.mno) * a.mno, my_slot = r8 - grp_base;
    const int gpt = PAD ? a.gpt : 0;                       // padded tiles (kernels.h: IocArgs.gpt): dead rows behind the tile's gpt groups
    const bool dead_row = gpt && (r8 / a.mno >= gpt || (int)blockIdx.x * gpt + r8 / a.mno >= a.ngrp);
    const int n_nb = dead_row ? 0 : a.mno;
    const u16* a2_lane = I2 + (lane & 31) * LDB2 + 8 * (lane >> 5);
    const u16* a3_lane = I3 + (lane & 31) * LDB1 + 8 * (lane >> 5);
    // four consecutive columns c..c+3 of this lane's row -> both piece images of a row-major bf16 tile: one 8-byte LDS write per piece
    // step-local copies of the lane's coordinates, re-derived from an opaque copy of the lane id at the top of every step: the LDS / stream
    // addresses built on them are then re-formed where they are used (a few VALU operations) instead of sitting, hoisted, in spilled registers
    int lr_s = lr, hi_s = hi, c0_s = c0;
    auto put4 = [&](u16* img, int ld, int ilo, int c, float v0, float v1, float v2, float v3) {
        unsigned pa[2], pb[2];
        splitp<2>(v0, v1, pa);
        splitp<2>(v2, v3, pb);
        u16* x = img + lr_s * ld + c;
        *reinterpret_cast<uint2*>(x) = make_uint2(pa[0], pb[0]);
        *reinterpret_cast<uint2*>(x + ilo) = make_uint2(pa[1], pb[1]);
    };
    // This wave's 32 x 32 block of a [hi | lo] image pair leaves as WHOLE LINES of an fp32 stream: eight lanes per row read it back (its own
    // columns, just written: LDS is in order within a wave, no barrier) and store 128 contiguous bytes per row.  Stored from the accumulator
    // layout a lane writes 32 bytes of each of 32 rows per instruction, and the memory system then moves 1.6x the streams' bytes (partial
    // lines are written back and fetched again: k_decoder_bwd 13.2 -> 8.4 GB written, 5.5 -> 3.6 GB fetched when it stopped doing that).
    // The stream holds hi + lo, i.e. the value the split weight-gradient kernels would split it into anyway (exact to 2^-17).
    auto flush32 = [&](const u16* img, int ld, int ilo, int col_img, float* out, int wout, int col_out, int t, int nloc) {
        int ln;
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));      // opaque per call: the twenty-odd addresses below are re-formed, not hoisted out of the time loop and spilled
        const int ch = ln & 7;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = (ln >> 3) + 8 * k;
            const u16* p = img + r * ld + col_img + cb * 32 + 4 * ch;
            const uint2 vh = *reinterpret_cast<const uint2*>(p), vl = *reinterpret_cast<const uint2*>(p + ilo);
            float4 v;
            v.x = __uint_as_float(vh.x << 16) + __uint_as_float(vl.x << 16);
            v.y = __uint_as_float(vh.x & 0xffff0000u) + __uint_as_float(vl.x & 0xffff0000u);
            v.z = __uint_as_float(vh.y << 16) + __uint_as_float(vl.y << 16);
            v.w = __uint_as_float(vh.y & 0xffff0000u) + __uint_as_float(vl.y & 0xffff0000u);
            if (r < nloc) *reinterpret_cast<float4*>(out + (size_t)(r * a.T + t) * wout + col_out + cb * 32 + 4 * ch) = v;
        }
    };
    // column sums over the tile's rows without 16 accumulators per tensor: a butterfly reduce-scatter over the 16 lanes that share
    // bits 0..3 of the row (15 exchanges): lane bits (b0 b1 b2 b3) end up with the sum of accumulator element 8 b0 + 4 b1 + 2 b2 + b3
    auto colsum16 = [&](const float (&x)[16]) {
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
    // weight packs: [hi | lo], n-tiles [h columns | e_r columns | e_v tile] (api.hip: "ioc/WcT16", "ioc/WgT16"), per-bin blocks ("ioc/WsT16")
    const uint4* WcT = reinterpret_cast<const uint4*>(a.WcT_h);
    const uint4* WgT = reinterpret_cast<const uint4*>(a.WgT_h);
    const uint4* WsT = reinterpret_cast<const uint4*>(a.WsT);
    const uint4* bc2[2] = {WcT + ((size_t)cb * G16) * 64 + lane, WcT + ((size_t)(NT + cb) * G16) * 64 + lane};
    const uint4* bcv[1] = {WcT + ((size_t)(2 * NT) * G16) * 64 + lane};
    const uint4* bg2[2] = {WgT + ((size_t)cb * G32) * 64 + lane, WgT + ((size_t)(NT + cb) * G32) * 64 + lane};
    const uint4* bgv[1] = {WgT + ((size_t)(2 * NT) * G32) * 64 + lane};
    constexpr size_t PLC = (size_t)(2 * NT + 1) * G16 * 64, PLG = (size_t)(2 * NT + 1) * G32 * 64;
    const size_t PLS = (size_t)B * NT * G16 * 64;
    // saved activations / gradient streams are addressed as (uniform tile base) + (32-bit offset inside the tile)
    const int nloc = min(TM, a.R - row0);
    const bool rok = lr < nloc;                            // rows past R read the tile's last row; nothing of theirs is stored or summed
    const int rcl = min(lr, nloc - 1);
    const size_t tb = (size_t)row0 * a.T;
    const float* svu = a.sv_u + tb * H; const float* svc = a.sv_c + tb * H; const float* svr = a.sv_r + tb * H;
    const float* svx = a.sv_x + tb * E;
    float* o_dac = a.dac + tb * H; float* o_rh = a.rh + tb * H; float* o_hp = a.hprev + tb * H; float* o_dag = a.dag + tb * 2 * H;
    float* o_dpr = a.dpre_r + tb * H; float* o_dpv = a.dpre_v + tb * EV;

    if (tid < 16) {
        const unsigned lo = ((tid & 1) ? 0x3F80u : 0u) | ((tid & 2) ? 0x3F800000u : 0u);
        const unsigned hi2 = ((tid & 4) ? 0x3F80u : 0u) | ((tid & 8) ? 0x3F800000u : 0u);
        lut[tid] = make_uint2(lo, hi2);
    }
    const int gb31 = (lr / a.mno) * a.mno;                 // first tile row of the group this lane's row belongs to (obs bits are slots of the group)
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
    float cs_r = 0.f, cs_u = 0.f, cs_c = 0.f, cs_p = 0.f;   // this lane's column sums of da_r, da_u, da_c, dpre_r over its rows and all steps
    f32x16 dh = zero16();
    {
        f32x16 tt[1] = {dh};
        const float* apr[1] = {DR + lr * LDR + 4 * hi};
        mma_groups_ptr<1, true>(tt, apr, a.WrT + ((size_t)cb * (KR / 8)) * 64 + lane, KR / 8);
        dh = tt[0];
    }

    for (int t = a.T - 1; t >= 0; --t) {
        int rc = rcl;
        asm volatile("v_mov_b32 %0, %1" : "=v"(rc) : "v"(rcl));     // opaque per step: stream offsets are re-formed, not hoisted and spilled
        {
            int ls;
            asm volatile("v_mov_b32 %0, %1" : "=v"(ls) : "v"(lane));
            lr_s = ls & 31; hi_s = ls >> 5; c0_s = cb * 32 + 4 * hi_s;
        }
        const unsigned rt = (unsigned)(rc * a.T + t);
        // this step's gate values are requested before anything else: they arrive while P0 / the mask search run (asked for where part 1
        // uses them, behind two barriers the compiler does not move loads across, their HBM latency was exposed every step)
        float4 pu[4], pcx[4], pr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned ix = rt * H + c0_s + 8 * q;
            pu[q] = *reinterpret_cast<const float4*>(svu + ix); pcx[q] = *reinterpret_cast<const float4*>(svc + ix); pr[q] = *reinterpret_cast<const float4*>(svr + ix);
        }
        TICKB(0)
        __syncthreads();
        TICKB(1)
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
        auto load_hprev = [&]() {                                              // h_{t-1} tile -> A1's space
            for (int i = tid; i < TM * (H >> 2); i += NTHR) {
                const int r = i / (H >> 2), c4 = i - r * (H >> 2);
                const int row = min(row0 + r, a.R - 1);
                const int ag0 = (t > 0) ? 0 : ioc_agent_of_row(row, a.K, a.mno, gpt, a.ngrp);
                const float* src = (t > 0) ? a.sv_h + ((size_t)row * a.T + t - 1) * H
                                           : a.Hx + (size_t)max(ag0, 0) * a.ldhx;
                const float4 hv = ag0 >= 0 ? *reinterpret_cast<const float4*>(src + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(HP + r * LD1 + c4 * 4) = hv;
                if (r < nloc) *reinterpret_cast<float4*>(o_hp + (size_t)(r * a.T + t) * H + c4 * 4) = hv;      // the weight gradient's h_{t-1} operand, whole rows
            }
        };
        load_hprev();                                    // read by part 1 and by the pooled rebuild of this step
        TICKB(2)
        __syncthreads();
        TICKB(3)
        // ---- P1: neighbour / observer masks ----
        {
            const float px = pc[r8 * 2], py = pc[r8 * 2 + 1];
            float nbw, nbh;
            nb_opaque(a.nb_w, a.nb_h, nbw, nbh);
            const unsigned long long oc = nb_search<4>(pc, vld, grp_base, n_nb, q8, TPR, my_slot, px, py, nbw, nbh, a.G, a.bin_tab,
                                                      [&](int j, int b) {
                                                          atomicOr(&masks[r8 * B + b], (mask_t)1 << j);
                                                          atomicOr(&obs[(grp_base + j) * B + b], (mask_t)1 << my_slot);
                                                      });
            nb_publish_occ(oc, occ, B);
        }
        // ---- GRU cell backward, part 1 ----
        if (a.pool_flags && q8 == 0 && row0 + r8 < a.R) {      // which bins of this (row, t) hold a neighbour: the weight-gradient
            unsigned long long fl = 0ull;                       // GEMM skips the all-zero blocks of the pooled operand
            for (int b = 0; b < B; ++b) fl |= (unsigned long long)(masks[r8 * B + b] != 0) << b;
            a.pool_flags[(size_t)my_row * a.T + t] = fl;
        }
        f32x16 dhp, rr;                              // what part 2 needs: r (h_{t-1} is re-read from its tile; everything else is stored at once)
        float sc_c[16], sc_u[16];
        const float dscv = dsc[lr_s];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned ix = rt * H + c0_s + 8 * q;
            const float4 u4 = pu[q], cc4 = pcx[q], r4 = pr[q];
            const float4 h4 = *reinterpret_cast<const float4*>(A1 + lr_s * LD1 + c0_s + 8 * q);
            const float4 w4 = *reinterpret_cast<const float4*>(wsc + c0_s + 8 * q);
            float dacv[4], dauv[4], rhv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * q + e;
                const float u = f4e(u4, e), c = f4e(cc4, e), r = f4e(r4, e), hprev = f4e(h4, e);
                const float dht = dh[i] + dscv * f4e(w4, e);
                const float dau = dht * (hprev - c) * u * (1.0f - u);
                const float dc = dht * (1.0f - u);
                dhp[i] = dht * u;
                const float dac = dc * (1.0f - c * c);
                dacv[e] = dac; dauv[e] = dau; rhv[e] = r * hprev;
                sc_c[i] = rok ? dac : 0.f; sc_u[i] = rok ? dau : 0.f;
                rr[i] = r;
            }
            put4(I3, LDB1, ILO1, c0_s + 8 * q, dacv[0], dacv[1], dacv[2], dacv[3]);
            put4(I2, LDB2, ILO2, H + c0_s + 8 * q, dauv[0], dauv[1], dauv[2], dauv[3]);
            put4(I2, LDB2, ILO2, c0_s + 8 * q, rhv[0], rhv[1], rhv[2], rhv[3]);              // r h_{t-1} borrows da_r's slot (written after the next contraction) on its way out
        }
        flush32(I3, LDB1, ILO1, 0, o_dac, H, 0, t, nloc);
        flush32(I2, LDB2, ILO2, H, o_dag, 2 * H, H, t, nloc);
        flush32(I2, LDB2, ILO2, 0, o_rh, H, 0, t, nloc);
        cs_c += colsum16(sc_c); cs_u += colsum16(sc_u);
        TICKB(4)
        __syncthreads();
        TICKB(5)
        __builtin_amdgcn_s_setprio(3);          // (wave priority by phase, as in k_ioc: 3 / 2 / 1 for the candidate / gate / pooling contractions -- IOC backward 21.5 -> 21.2 ms)
        f32x16 dev = zero16(), der;
        {
            f32x16 t2[2] = {zero16(), zero16()};                  // drh | de_r, one pass over the da_c fragments
            if (cb == 0) {                                        // (the e_v tile rides on the same fragment stream: one latency chain for this wave, not two)
                f32x16 t3[3] = {t2[0], t2[1], dev};
                const uint4* b3[3] = {bc2[0], bc2[1], bcv[0]};
                BWDX3_MM(3)(t3, a3_lane, ILO1, b3, PLC, G16);
                t2[0] = t3[0]; t2[1] = t3[1]; dev = t3[2];
            } else
                BWDX3_MM(2)(t2, a3_lane, ILO1, bc2, PLC, G16);
            der = t2[1];
            float sc_r[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float darv[4];
                const float4 h4 = *reinterpret_cast<const float4*>(A1 + lr_s * LD1 + c0_s + 8 * q);      // h_{t-1}, still in its tile (not carried in registers across the contraction)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    const float dr = t2[0][i] * f4e(h4, e);
                    dhp[i] += t2[0][i] * rr[i];
                    const float dar = dr * rr[i] * (1.0f - rr[i]);
                    darv[e] = dar;
                    sc_r[i] = rok ? dar : 0.f;
                }
                put4(I2, LDB2, ILO2, c0_s + 8 * q, darv[0], darv[1], darv[2], darv[3]);
            }
            flush32(I2, LDB2, ILO2, 0, o_dag, 2 * H, 0, t, nloc);
            cs_r += colsum16(sc_r);
        }
        __builtin_amdgcn_s_setprio(0);
        TICKB(6)
        __syncthreads();
        TICKB(5)
        // (h_{t-1} is still in its tile -- da_c went to the images, not over it as in the fp32 kernel -- so the pooled rebuild needs no reload);
        // da_c is consumed, its images take dpre_r
        __builtin_amdgcn_s_setprio(2);
        {
            f32x16 t2[2] = {zero16(), der};                       // dh (gates) | de_r
            if (cb == 0) {
                f32x16 t3[3] = {t2[0], t2[1], dev};
                const uint4* b3[3] = {bg2[0], bg2[1], bgv[0]};
                BWDX3_MM(3)(t3, a2_lane, ILO2, b3, PLG, G32);
                t2[0] = t3[0]; t2[1] = t3[1]; dev = t3[2];
            } else
                BWDX3_MM(2)(t2, a2_lane, ILO2, bg2, PLG, G32);
            float sc_p[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 er4 = *reinterpret_cast<const float4*>(svx + (size_t)rt * E + EV + C + c0_s + 8 * q);
                float dprv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * q + e;
                    dhp[i] += t2[0][i];
                    const float dpr = f4e(er4, e) > 0.f ? t2[1][i] : 0.f;
                    dprv[e] = dpr;
                    sc_p[i] = rok ? dpr : 0.f;
                }
                if (rok) {
                    if (cb == 0 && q < EV / 8) {                  // the e_v tile: columns 4 hi + 8 q + e < EV
                        const float4 ev4 = *reinterpret_cast<const float4*>(svx + (size_t)rt * E + 4 * hi_s + 8 * q);
                        *reinterpret_cast<float4*>(o_dpv + (size_t)rt * EV + 4 * hi_s + 8 * q) =
                            make_float4(ev4.x > 0.f ? dev[4 * q] : 0.f, ev4.y > 0.f ? dev[4 * q + 1] : 0.f, ev4.z > 0.f ? dev[4 * q + 2] : 0.f, ev4.w > 0.f ? dev[4 * q + 3] : 0.f);
                    }
                }
                put4(I3, LDB1, ILO1, c0_s + 8 * q, dprv[0], dprv[1], dprv[2], dprv[3]);
            }
            flush32(I3, LDB1, ILO1, 0, o_dpr, H, 0, t, nloc);
            cs_p += colsum16(sc_p);
        }
        __builtin_amdgcn_s_setprio(0);
        TICKB(7)
        __syncthreads();
        TICKB(5)
        // ---- social pooling backward ----
        // dh_j += sum_b sum_{i : j in bin b of i} dpool_b[i] = sum_b (M_b^T dpool_b)[j]: dpool_b comes out of its contraction UNtransposed (lane = hidden
        // column, registers = rows), which is the B-operand layout of a second MFMA whose A operand is the 0/1 observer matrix (exact in bf16, built from
        // the obs bit words through the nibble table): the scatter costs 4 MFMAs per bin and wave and no LDS tile, no barrier and no gather loop --
        // the bin loop runs barrier-free, every wave on its own column block.  (dpool_b enters as its two bf16 pieces, like every operand of this kernel.)
        f32x16 nbacc = zero16();
        // bins without a neighbour anywhere in the tile have dpool_b gathered by nobody: only their (zero) pooled rows are written
        unsigned long long om = (unsigned long long)__builtin_amdgcn_readfirstlane((int)occ[0]) & 0xffffffffull;
        om |= (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)occ[1]) << 32;
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
            TICKB(8)
            if (!live) continue;
            __builtin_amdgcn_s_setprio(1);
            f32x16 dpl[1] = {zero16()};
            const unsigned ts[1] = {(unsigned)((b * NT + cb) * G16 * 64)};
            mmax_ring<1, 2, false, G16, G16>(dpl, a3_lane, ILO1, WsT, ts, (unsigned)PLS);
            TICKB(9)
            // accumulator elements 0..7 = rows 4 hi + {0..3, 8..11}, elements 8..15 = the same + 16: the k order of the two scatter MFMAs
            const FragP<2> p0 = split8<2>(dpl[0][0], dpl[0][1], dpl[0][2], dpl[0][3], dpl[0][4], dpl[0][5], dpl[0][6], dpl[0][7]);
            const FragP<2> p1 = split8<2>(dpl[0][8], dpl[0][9], dpl[0][10], dpl[0][11], dpl[0][12], dpl[0][13], dpl[0][14], dpl[0][15]);
            const unsigned mo = ((unsigned)obs[lr_s * B + b] << gb31) >> (4 * hi_s);        // observers of row j = lr as tile rows, this half-wave's rows first
            const uint2 l0 = lut[mo & 15u], l1 = lut[(mo >> 8) & 15u], l2 = lut[(mo >> 16) & 15u], l3 = lut[(mo >> 24) & 15u];
            const uint4 m0 = make_uint4(l0.x, l0.y, l1.x, l1.y), m1 = make_uint4(l2.x, l2.y, l3.x, l3.y);
            nbacc = mfma16(m0, p0.p[1], nbacc); nbacc = mfma16(m1, p1.p[1], nbacc);
            nbacc = mfma16(m0, p0.p[0], nbacc); nbacc = mfma16(m1, p1.p[0], nbacc);
            TICKB(11)
            __builtin_amdgcn_s_setprio(0);
        }
        TICKB(10)
        __syncthreads();                                   // every wave is done with h_{t-1} (pooled rebuilds): its tile takes the neighbour gradient
        TICKB(5)
#pragma unroll
        for (int i = 0; i < 16; ++i) NB[(4 * hi_s + 8 * (i >> 2) + (i & 3)) * LD1 + cb * 32 + lr_s] = nbacc[i];
        TICKB(11)
        __syncthreads();
        TICKB(5)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 n4 = *reinterpret_cast<const float4*>(NB + lr_s * LD1 + c0_s + 8 * q);
            dh[4 * q] = dhp[4 * q] + n4.x; dh[4 * q + 1] = dhp[4 * q + 1] + n4.y; dh[4 * q + 2] = dhp[4 * q + 2] + n4.z; dh[4 * q + 3] = dhp[4 * q + 3] + n4.w;
        }
    }
    if (a.bias_part) {                                     // one part per tile: [da_r | da_u | da_c | dpre_r] column sums
        // after the butterfly a lane holds the sum over ITS 16-lane row group of one accumulator element; the other row group is lane ^ 16
        cs_r += __shfl_xor(cs_r, 16); cs_u += __shfl_xor(cs_u, 16); cs_c += __shfl_xor(cs_c, 16); cs_p += __shfl_xor(cs_p, 16);
        if (!(lane & 16)) {
            const int el = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);     // accumulator element this lane ended up with
            const int colb = c0 + 8 * (el >> 2) + (el & 3);
            float* part = a.bias_part + (size_t)blockIdx.x * 4 * H;
            part[colb] = cs_r; part[H + colb] = cs_u; part[2 * H + colb] = cs_c; part[3 * H + colb] = cs_p;
        }
    }
#ifdef DESIRE_IOC_TIMING
    if (a.dbg && blockIdx.x == 7 && tid == 0)
        for (int k = 0; k < 12; ++k) a.dbg[k] = tacc[k];
#endif
    if (rok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4* d4 = reinterpret_cast<float4*>(a.dHx_rows + (size_t)(row0 + lr) * H + c0 + 8 * q);
            float4 v = *d4;
            v.x += dh[4 * q]; v.y += dh[4 * q + 1]; v.z += dh[4 * q + 2]; v.w += dh[4 * q + 3];
            *d4 = v;
        }
    }
}

template <int H>
void launch_ioc_bwd_x3_t(const IocBwdArgs& a, hipStream_t s) {
    const int B = a.G * a.G;
    const size_t lds = (size_t)(32 * (H + 4) * 3) * sizeof(float) + (size_t)2 * 32 * (H + 8) * sizeof(u16) + (size_t)2 * 32 * B * sizeof(unsigned)
                       + (size_t)(32 * 2 + 32 + H) * sizeof(float) + 32 + 64 + 16 * sizeof(uint2);
    if (a.gpt > 0) {                                        // padded tiles
        allow_big_lds(k_ioc_bwd_x3<H, 16, 32, true>);
        hipLaunchKernelGGL((k_ioc_bwd_x3<H, 16, 32, true>), dim3((a.R + 31) / 32), dim3((H / 32) * 64), lds, s, a);
        return;
    }
    allow_big_lds(k_ioc_bwd_x3<H, 16, 32>);
    hipLaunchKernelGGL((k_ioc_bwd_x3<H, 16, 32>), dim3((a.R + 31) / 32), dim3((H / 32) * 64), lds, s, a);
}
}  // namespace
bool ioc_bwd_x3_supported(int mno, int H) { return mno <= 32 && (H == 128 || H == 64); }
// a.WcT_h / a.WgT_h / a.WsT = the [hi | lo] packs "ioc/WcT16" / "ioc/WgT16" / "ioc/WsT16"; a.WrT stays the fp32 pack (prologue)
void launch_ioc_bwd_x3(const IocBwdArgs& a, hipStream_t s) {
    if (a.H == 128) launch_ioc_bwd_x3_t<128>(a, s); else launch_ioc_bwd_x3_t<64>(a, s);
}
