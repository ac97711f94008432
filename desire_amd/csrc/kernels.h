// kernels.h -- host-visible launchers and argument blocks (internal to libdesire_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// dynamic LDS above 64 KiB must be opted into once per kernel
template <class F>
inline void allow_big_lds(F* f) {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(f), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
}

// A launch whose row count is known on the DEVICE only (DESIRE_FLAG_COMPACT_*, inference: the present agents of a batch / the windows seated in a
// slot class, counted by kernels_compact.hip's scans).  The host sizes the grid -- and picks the kernel variant -- for the worst case and the kernel
// replaces its count by cnt[0] * mul in its first instructions; workgroups beyond it exit before they touch memory.  No read-back, no host wait, the
// call is hipGraph-capturable.  cnt == nullptr (every other launch): the count in the argument block stands.
// hint: a GUESS of cnt[0] (the previous call's count, read from the scans' mapped word without waiting; 0 = none).  It only ever shrinks a GRID: launchers of
// kernels that stride over their tiles (k_encoder_pair, k_deconv2/3 and their six-product forms) size the grid for hint * 1.25 + slack instead of the worst case,
// and a count above that is still served -- more slowly -- by the stride loop.
struct DynCount { const int32_t* cnt; int mul; int hint; };
// units (rows / samples / agents) a strided launch is sized for: the worst case in the arguments, or the hinted count with a quarter of slack
inline int dyn_units(int worst, const DynCount& d) {
    if (!d.cnt || d.hint <= 0) return worst;
    const long g = (long)d.hint * d.mul, want = g + g / 4 + 256;
    return (int)(want < (long)worst ? want : (long)worst);
}

enum { EPI_BIAS = 0, EPI_BIAS_RELU = 1, EPI_SCALE_SHIFT_ELU = 2, EPI_NONE = 3, EPI_ELUGRAD = 4, EPI_SIGGRAD = 5 };

// out[M, N] = epi(A[M, K] @ W[K, N]); W in packed fragment order [NT][G][64] float4.
struct GemmArgs {
    const float* A; int lda; int M; int K;
    const float4* Bp; int G; int NT;
    float* out; int ldo; int N;
    const float* p0; const float* p1; int chmod;
    const float* aux;                              // [M, ldo] saved activation for the gradient epilogues
    DynCount dyn;                                  // M = cnt[0] * mul on the device (see DynCount)
    int M_hint;                                    // with dyn: a GUESS of the device-side M (the previous call's count) -- picks the kernel variant, never a bound
};
void launch_gemm_rows(const GemmArgs& a, int epi, hipStream_t s);

void launch_reparam(const float* params, const float* eps, float* z, int R, int L, int K, int mno,
                    int posterior, hipStream_t s);

struct MaskArgs {
    const float* xhat; int R; int V; int H; int K; int mno;
    int Hl;                                        // logical width: the softmax runs over columns [0, Hl) (Hl < H: zero-padded tile)
    const float4* Wp; const float* bias; const float* Hx; int ldhx; float* xz;
    float* sv_p;                                   // optional [R,H]: relu(xhat W + b) kept for the backward softmax
    DynCount dyn;                                  // one pseudo-scene of P = cnt[0] present agents: mno = P, R = P * K (DynCount)
};
void launch_mask(const MaskArgs& a, hipStream_t s);
// six-product forms of deconv1 and the mask fc (kernels_x6.hip; weight pointers = three-piece packs)
bool rows_x6_supported(int K, int NT);
void launch_deconv1_x6(const GemmArgs& a, hipStream_t s);
void launch_mask_x6(const MaskArgs& a, hipStream_t s);

// ---- CVAE conv / deconv stack (kernels_conv.hip) ----
struct ConvArgs {
    const float* in; float* out; int n;            // n = samples (agents or rows)
    const float4* Wp; const float* w_raw;          // packed taps / raw weights (VALU kernels)
    const float* scale; const float* shift;        // folded bias + frozen batch-norm
    int mode; const float* yprev;                  // epilogue mode (common.h:conv_epilogue); saved activation of the
                                                   // destination layer for the backward modes
    DynCount dyn;                                  // n = cnt[0] * mul on the device (see DynCount)
};
void launch_conv1(const ConvArgs& a, hipStream_t s);     // [n,32,32,1]  -> [n,16,16,32]
void launch_conv2(const ConvArgs& a, hipStream_t s);     // [n,16,16,32] -> [n,8,8,64]
void launch_conv2_x3(const ConvArgs& a, hipStream_t s);  // same with split-bf16 operands (kernels_bwd_x3.hip; a.Wp = [hi | lo] pack)
void launch_conv3_x3(const ConvArgs& a, hipStream_t s);  // [n,8,8,64] -> [n,4,4,128]
void launch_conv3(const ConvArgs& a, hipStream_t s);     // [n,8,8,64]   -> [n,4,4,128]
void launch_deconv2(const ConvArgs& a, hipStream_t s);   // [n,4,4,128]  -> [n,8,8,64]
void launch_deconv3(const ConvArgs& a, hipStream_t s);   // [n,8,8,64]   -> [n,16,16,32]
void launch_deconv4(const ConvArgs& a, hipStream_t s);   // [n,16,16,32] -> [n,32,32] (+sigmoid)

// ---- recurrent kernels (kernels_rnn.hip) ----
struct EncArgs {
    const float* frames; int n_scenes; int T; int mno;     // [n_scenes, T, mno, 3]
    float sx, sy; int H;
    const float* wx_g; const float* b_g;                   // x rows of gates kernel [2, 2H], bias
    const float* wx_c; const float* b_c;                   // x rows of candidate kernel [2, H]
    const float4* Whg; const float4* Whc;                  // packed h rows [H,2H], [H,H]
    float* out; int ldo;                                   // final state -> out[a*ldo + c]
    float* p_last;                                         // optional [A,2]: normalised last pos
    uint8_t* valid;                                        // optional [A]: id != 0 at last frame
    float* sv_r; float* sv_u; float* sv_c; float* sv_h; float* sv_x;   // optional training saves [A,T,H] x4, [A,T,2]
    // autoregressive rollout after the T observed frames (sample(), model/model.py:643-681): n_roll more steps whose input is
    // the position drawn from the bivariate Gaussian the 5-wide head reads off the current state
    int n_roll; const float* w5; const float* b5;          // head [H,5], [5]
    const float* normals; float* roll_out;                 // [n_roll, A, 2] N(0,1) draws in, sampled positions (clipped <= 1) out
    DynCount dyn;                                          // one pseudo-scene of mno = cnt[0] present agents (n_scenes = 1; DynCount)
};
void launch_encoder(const EncArgs& a, hipStream_t s);
void launch_encoder_pair(const EncArgs& past, const EncArgs& fut, hipStream_t s);   // both encoders, one launch (same H)
void launch_encoder_bf16(const EncArgs& a, hipStream_t s);    // kernels_bf16.hip; Whg / Whc = bf16 packs

struct DecArgs {
    const float* xz; const float* Hx; int ldhx; const float* p_last;
    int R; int K; int mno; int H; int T;
    const float4* Wxg; const float4* Wxc; const float4* Whg; const float4* Whc;
    const float* b_g; const float* b_c;
    const float* w_head; const float* b_head;              // [H,2], [2]
    float* Y;                                              // [R, T, 2]
    float* hdump;                                          // optional [R, T, H] hidden states h_t
    float* sv_r; float* sv_u; float* sv_c;                 // optional [R, T, H] gate values (training mode)
    DynCount dyn;                                          // one pseudo-scene of P = cnt[0] present agents: mno = P, R = P * K (DynCount)
};
void launch_decoder(const DecArgs& a, hipStream_t s);
void launch_decoder_bf16(const DecArgs& a, hipStream_t s);    // kernels_bf16.hip; Whg / Whc = bf16 packs

struct IocArgs {
    float* Y; float* score;                                // [R,T,2] in/out, [R]
    const float* Hx; int ldhx; const float* p_last; const uint8_t* valid;
    int R; int K; int mno; int H; int T; int iters;
    int C; int Gh; int Gw; int E_v; int G; float nb_w, nb_h;   // G = social grid side
    const float* grids; const int32_t* grid_of_scene;
    const float* w_vel; const float* b_vel;                // [2,E_v], [E_v]
    const float4* Wsoc; const float* b_soc;                // packed per bin: [B][NT][H/8][64]
    const float4* Wsoc_c;                                  // same weights in 16x16x4 fragment order (row-compacted pooling, variant 8)
    const float4* Wg; const float4* Wc; const float* b_g; const float* b_c;   // K = E+H
    const float* w_score; const float* b_score;            // [H], [1]
    const float4* Wreg; const float* b_reg; int NTreg;     // [H, 2T] packed
    int variant;                                           // dims.ioc_form (DESIRE_IOC_*, include/desire_hip.h)
    long long* dbg;                                        // per-phase cycle counters (DESIRE_IOC_TIMING builds)
    float* hex; int* grp_cnt; int* err;                    // cluster form: exchange buffer [2][R][H], group counters, error word
    int nspl;                                              // > 1: bin-split form of k_ioc (hex = [tiles][2][nspl][32 H], grp_cnt per tile)
    float* sv_x; float* sv_r; float* sv_u; float* sv_c; float* sv_h;   // training saves: [R,T,E], [R,T,H] x4 (32-row form only)
    const float* bin_tab;                                  // log-polar bin table (common.h:neighbor_bin_dev) or nullptr = rectangular grid
    // PADDED TILES (slot classes that do not divide 32, DESIRE_FLAG_COMPACT_IOC; k_ioc<.., TM = 32> and k_ioc_x3, inference): gpt > 0 = a 32-row tile holds
    // gpt whole groups of mno slots (gpt * mno <= 32) followed by dead rows; group G = tile * gpt + (row & 31) / mno, ngrp real groups; R = tiles * 32.
    // gpt = 0: rows are packed, r = (scene * K + k) * mno + slot (mno divides 32).
    int gpt; int ngrp;
    // windows of this launch counted on the device (a slot class of DESIRE_FLAG_COMPACT_IOC, inference): n_c = dyn.cnt[0] windows of mno slots ->
    // ngrp = n_c * K, R = gpt ? ceil(ngrp / gpt) * 32 : ngrp * mno (ioc_dyn_rows, common.h); the arguments hold the worst case (every window in this class)
    DynCount dyn;
};
void launch_ioc(const IocArgs& a, hipStream_t s);
// Which IOC form serves (mno, H, bins): the cluster form (32-row tiles exchanging hidden states through global memory) takes every
// group that does not fit ONE workgroup's LDS tile -- more than 64 agents, 64 agents at H = 256, or 64 agents with so many
// social bins (> 25 at H = 128) that the 64-row tile's neighbour masks push it past 160 KB.
inline bool ioc_uses_cluster(int mno, int H, int bins, int variant) {
    if (mno > 64 || (mno == 64 && H == 256) || (variant == 4 && mno >= 64)) return true;
    if (mno == 64) {
        const size_t tile = ((size_t)65 * (2 * H + 52) + 2 * 64 * (H + 4) + 64 * 4 + 48 + (H / 32) * 64) * 4 + (size_t)64 * bins * 8 + 128;
        return tile > 160 * 1024;
    }
    return false;
}
// Few tiles (a handful of windows): how many workgroups share one 32-row tile's social bins (k_ioc NSPL), so that the launch covers
// up to 256 CUs instead of one per (scene, k) group.  1 = the plain form.
inline int ioc_bin_split(int R, int mno, int H, int bins, int iters) {
    if (mno > 32 || H > 128 || iters != 1) return 1;
    const int tiles = (R + 31) / 32;
    for (int n = 4; n >= 2; --n)          // (8 per tile measured no faster than 4: what is left of a step is the part every member repeats)
        if (tiles * n <= 256 && bins >= n) return n;
    return 1;
}
int ioc_bin_split_capacity(const IocArgs& a, int n);           // resident workgroups of the n-member bin-split kernel on this device (kernels_rnn.hip)
void launch_ioc_bf16(const IocArgs& a, hipStream_t s);
// bf16 cluster form (kernels_bf16_cl.hip): groups of 64 / 96 / 128 agents over mno/32 workgroups; returns != 0 when the
// persistent grid cannot be made resident
int launch_ioc_bf16_cluster(const IocArgs& a, hipStream_t s);
// split-bf16 form (kernels_x3.hip): fp32-equivalent results from three bf16 MFMAs per product; weight pointers = [hi | lo] packs
bool ioc_x3_supported(int mno, int H, int bins);
void launch_ioc_x3(const IocArgs& a, hipStream_t s);
void launch_ioc_x6(const IocArgs& a, hipStream_t s);
bool ioc_x6r2_supported(int mno, int H, int bins);             // ... on 64-row tiles, two row blocks per wave (kernels_x6r2.hip)
void launch_ioc_x6r2(const IocArgs& a, hipStream_t s);
void launch_ioc_x3r2(const IocArgs& a, hipStream_t s);             // ... with two-piece operands (dims.bf16 = 2)
// sample generation with three-piece operands (kernels_x6.hip, dims.bf16 = 3)
bool decoder_x6_supported(int H);
void launch_decoder_x6(const DecArgs& a, hipStream_t s, int np = 3);      // np = 2 (training-mode form only): two-piece operands
void launch_deconv2_x6(const ConvArgs& a, hipStream_t s, int np = 3);      // np = 2: the first two pieces of the packs, three products (training-mode forward)
void launch_deconv3_x6(const ConvArgs& a, hipStream_t s, int np = 3);
            // three bf16 pieces per operand, six products (dims.bf16 = 3)
// agent-sharded IOC, one step per launch (kernels_rnn.hip: k_ioc_step)
struct IocStepArgs {
    int t; int rank; int nranks; int m_loc; int n_scenes; int K; int R;       // R = local rows = n_scenes * K * m_loc
    int H; int T; int Gh; int Gw; int G; float nb_w, nb_h;
    const float* Yall; const float* plast_all; const uint8_t* valid_all; const float* Hall;
    // peer form (desire_ioc_peer_pass): per-rank base pointers instead of one gathered array each -- device tables of nranks pointers to
    // the rank's OWN block in the gathered layouts above (its exchange region, mapped through hipIpc when it lives in another process /
    // on another GPU).  nullptr = the gathered arrays.
    // (BY VALUE in the kernel arguments, not a table in device memory: a table rewritten by hipMemcpy for a new handle at a recycled
    //  address was read stale by the next kernels -- the kernarg segment is fresh per dispatch)
    int peer; const float* Yp[8]; const float* plp[8]; const uint8_t* vp[8]; const float* Hp[8];
    const float* st_h; float* st_h_out; float* st_score; float* st_h_copy;
    const float* grids; const int32_t* grid_of_scene;
    const float* w_vel; const float* b_vel; const float4* Wsoc; const float* b_soc;
    const float4* Wg; const float4* Wc; const float* b_g; const float* b_c; const float* w_score;
    const float* bin_tab;
    int np; size_t plo_soc, plo_g, plo_c;                  // np = 2 / 3: Wsoc / Wg / Wc are split packs (uint4 units between pieces), see k_ioc_step
};
void launch_ioc_step(const IocStepArgs& a, hipStream_t s);
// peer exchange (kernels_rnn.hip): progress counters in the ranks' exchange regions, written / polled with system-scope atomics
struct PeerFlags { const unsigned* f[8]; };
void launch_peer_wait(const PeerFlags& flags, int nranks, const unsigned* epoch, unsigned per_pass, unsigned stage, int* err, hipStream_t s);
void launch_peer_set(unsigned* flag, const unsigned* epoch, unsigned per_pass, unsigned stage, hipStream_t s);
void launch_peer_epoch(unsigned* epoch, hipStream_t s);
void launch_hx_rows(float* out, const float* HxHy, int ldhx, int n_scenes, int K, int mno, int H, hipStream_t s);
// own block of the exchange region for one pass: presence flags, last observed positions, decoded positions, h_{-1} = Hx per row
void launch_peer_publish(const uint8_t* valid, const float* p_last, const float* Y, const float* HxHy, int ldhx, uint8_t* o_valid,
                         float* o_plast, float* o_Y, float* o_H, int n_scenes, int K, int mno, int T, int H, hipStream_t s);
void launch_ioc_finish(float* Y, const float* dY, const float* st_score, const float* b_score, float* score, int R, int T, hipStream_t s);
struct ConvArgs;
void launch_deconv2_bf16(const ConvArgs& a, hipStream_t s);
void launch_conv2_bf16(const ConvArgs& a, hipStream_t s);
void launch_conv3_bf16(const ConvArgs& a, hipStream_t s);
void launch_deconv1_bf16(const GemmArgs& a, hipStream_t s);     // a.Bp = bf16 pack [L, 2048]
void launch_mask_bf16(const MaskArgs& a, hipStream_t s);        // a.Wp = bf16 pack [1024, H]
void launch_deconv3_bf16(const ConvArgs& a, hipStream_t s);
void launch_deconv34_bf16(const ConvArgs& a, const float* sc4, const float* sh4, hipStream_t s);   // a.Wp = W3 pack, a.w_raw = W4 chain pack    // kernels_bf16.hip; weight pointers = bf16 packs

void launch_neighbor_bins(const float* pos, const uint8_t* valid, int32_t* bins, int n_groups, int mno,
                          float nb_w, float nb_h, int G, const float* bin_tab, hipStream_t s);
void launch_scene_cells(const float* pos, int32_t* cells, int n, int Gh, int Gw, hipStream_t s);

// ---- cold rows (kernels_aux.hip) ----
void launch_fill_f32(float* dst, size_t n, float v, hipStream_t s);
void launch_copy_f32(float* dst, const float* src, size_t n, hipStream_t s);
void launch_copy_cols(float* dst, const float* src, size_t rows, int cols, int ld, hipStream_t s);
void launch_instnorm_act(float* x, int n, int P, int C, const float* gamma, const float* beta, int sig, hipStream_t s);
void launch_instnorm_act_oop(const float* x, float* y, int n, int P, int C, const float* gamma, const float* beta, int sig, hipStream_t s);
void launch_instnorm_act_bwd(float* dy, const float* x, const float* y, int n, int P, int C, const float* gamma, int sig, hipStream_t s);
void launch_batchnorm_act(float* x, size_t n, int P, int C, const float* gamma, const float* beta, int sig, float* part, float* stat, hipStream_t s);
void launch_batchnorm_act_bwd(float* dy, const float* x, const float* y, size_t n, int P, int C, const float* gamma, int sig,
                              float* part, float* stat, float* stat2, hipStream_t s);
void launch_conv_direct(const float* in, const float* w, const float* b, float* out, int n, int Hi, int Wi, int Ci,
                        int Co, int stride, int relu, hipStream_t s);
void launch_temporal_conv(const float* frames, const float* w, const float* b, float* rho, int n_scenes, int T, int mno,
                          hipStream_t s);
void launch_feature_pooling(const float* Y, const float* rho, float* out, int R, int T, int K, int mno, hipStream_t s);
void launch_losses(const float* params, const float* Y, const float* fut, const uint8_t* lmask, const float* nfut, float* kld,
                   float* recon, float* cost, int n_scenes, int mno, int K, int T, int L, float sx, float sy, hipStream_t s);
void launch_loss_mask(const uint8_t* valid, const float* fut, uint8_t* lmask, float* nfut, int n_scenes, int mno, int T, hipStream_t s);
void launch_build_windows(const float* frames, int F, int mno_in, const int32_t* starts, int n, int T_obs, int T_pred,
                          int mno, float* past, float* fut, int32_t* err, int lookahead, hipStream_t s);
void launch_gaussian_sample(const float* p, const float* nrm, float* out, int n, hipStream_t s);
void launch_ade_fde(const float* Y, const float* fut, float* out, int n_scenes, int mno, int K, int T, float sx, float sy,
                    hipStream_t s);

// ---- backward (kernels_bwd.hip) ----
void launch_count_valid(const uint8_t* valid, int A, float* out, hipStream_t s);
void launch_loss_grad_y(const float* Y, const float* fut, const uint8_t* lmask, const float* nfut, const float* nvalid, float* dY,
                        int n_scenes, int mno, int K, int T, float sx, float sy, hipStream_t s);
struct DecBwdArgs {
    const float* dY0;                                      // [R,T,nw]: per-step gradient w.r.t. the head's outputs (nullptr: none)
    int nw;                                                // width of that head: 0 / 2 = the decoder's (x, y) head [H,2]; 5 = the Gaussian head [H,5]
    const float* sv_r; const float* sv_u; const float* sv_c; const float* sv_h;   // [R,T,H] from the training-mode forward
    const float* Hx; int ldhx; const float* w_head;
    const float4* WcT_h; const float4* WgT_h; const float4* WgT_x; const float4* WcT_x;   // transposed, packed
    int R, K, mno, T, H;
    float* dag; float* dac; float* rh; float* hprev;       // [R,T,2H], [R,T,H] x3: gate gradients, r*h_{t-1}, h_{t-1}
    float* dxg; float* dxc; float* dxz; float* dHx_rows;   // [R,2H], [R,H], [R,H], [R,H]  (dxz == null: encoder use)
    const float* dh_init; int ld_init;                     // optional gradient w.r.t. the FINAL state (encoders)
    float* bias_part;                                      // optional [tiles][3H]: per-tile column sums of da_r | da_u | da_c over rows and steps
};                                                         // (the bias gradients, without another pass over the gate-gradient streams)
// out[n] (+)= sum_p part[p * ld + off + n], n < N: fixed order over p (deterministic)
void launch_reduce_parts(const float* part, int nparts, int ld, int off, int N, float* out, int accumulate, hipStream_t s);
void launch_decoder_bwd(const DecBwdArgs& a, hipStream_t s);
struct TnArgs { const float* A; int lda; const float* G; int ldg; long M; int Kd; int N; int nslices; float* partial;
                // optional block-sparsity of A: flags[m] bit b set <=> columns [b*fcols, (b+1)*fcols) of row m can be non-zero.  A 32-row
                // chunk whose flags are clear over the workgroup's whole k-block is skipped (no loads, no MFMAs).
                const unsigned long long* flags; int fcols;
                int np;                                     // 2: split-bf16 operands (hi + lo, three bf16 MFMAs per product) in the large forms; 0: fp32
                // optional ROW LISTS per column block of A (launch_bin_lists): the k-block b = bk / fcols (one output tile row = one block, fcols = 128)
                // contracts only the rows rowlist[binbase[b] .. + bintotal[b]) -- those whose block b is non-zero -- instead of all M
                const int* rowlist; const int* binbase; const int* bintotal; };
// im2col view of a convolution's large-grid tensor as the A operand: column = tap*Cl + cl, row m = (sample, small-grid pixel)
struct ConvGather { int Cl, Pl, Ps, stride, pad; };
void launch_gemm_tn2_split(const TnArgs& a, const ConvGather* cg, bool narrow_n, hipStream_t s);       // kernels_bwd_x3.hip
void launch_gemm_tn(const TnArgs& a, float* out, int ldo, int accumulate, hipStream_t s);
// per-bin lists of the rows whose flag bit is set, in row order (deterministic): counts = scratch [ceil(M/2048)][B]; binbase [B+1]; bintotal [B]; rowlist [<= M*B]
void launch_bin_lists(const unsigned long long* flags, long M, int B, int* counts, int* binbase, int* bintotal, int* rowlist, hipStream_t s);
int gemm_tn_big_tiles(const TnArgs& a);                      // workgroups per slice of the 128 x 128 / 256 x 64 forms (0: another form)
void launch_colsum(const float* G, int ldg, long M, int N, int nslices, float* partial, float* out, int accumulate, hipStream_t s);
void launch_mask_bwd(const float* p, const float* dxz, const float* Hx, int ldhx, float* dq, float* dHx_rows, int R, int H,
                     int Hl, int K, int mno, hipStream_t s);
struct ConvWgradArgs { const float* S; int Cs; int Ps; const float* Lg; int Cl; int Pl; int stride; int pad; int n; float* partial; int np; };
void launch_conv_wgrad(const ConvWgradArgs& a, int nslices, float* out, hipStream_t s);
void launch_w1ch_grad(const float* Lg, const float* S, int n, int nslices, float* partial, float* out, hipStream_t s);
void launch_reparam_bwd(const float* dz, const float* eps, const float* params, const uint8_t* valid, const float* nvalid,
                        float* dparams, int n_scenes, int mno, int K, int L, hipStream_t s, const int32_t* inv = nullptr, int P = 0);
void launch_rows_to_agents(const float* rows, float* out, int ldo, int n_scenes, int mno, int K, int H, hipStream_t s, int gpt = 0);
void launch_score_grad(const float* Y0, const float* fut, const float* score, const uint8_t* valid, const float* nvalid,
                       float* dscore, float* dscoreT, int n_scenes, int mno, int K, int T, float sx, float sy, hipStream_t s);
struct IocBwdArgs {
    const float* Y0; const float* p_last; const uint8_t* valid; const float* Hx; int ldhx;
    const float* dYr; const float* dscore;
    const float* sv_x; const float* sv_r; const float* sv_u; const float* sv_c; const float* sv_h;
    const float* w_score;
    int R, K, mno, T, H, G; float nb_w, nb_h;
    const float4* WrT; const float4* WcT_h; const float4* WcT_er; const float4* WcT_ev;
    const float4* WgT_h; const float4* WgT_er; const float4* WgT_ev; const float4* WsT;
    const float4* WsT_c;                                   // WsT in 16x16x4 fragment order (row-compacted dpool, 32-row tiles)
    unsigned long long* pool_flags;                        // [R, T] out: bit b = bin b of this (row, t) holds a neighbour (nullptr: not wanted)
    float* dag; float* dac; float* rh; float* hprev; float* dpre_r; float* dpre_v; float* vel; float* pooled;
    float* dHx_rows;
    const float* bin_tab;
    float* bias_part;                                      // optional [32-row blocks][4H]: column sums of da_r | da_u | da_c | dpre_r (32-row forms only)
    long long* dbg;                                        // per-phase cycle counters (DESIRE_IOC_TIMING builds)
    int gpt; int ngrp;                                     // padded tiles (see IocArgs.gpt): 32-row forms of k_ioc_bwd / k_ioc_bwd_x3
};
void launch_ioc_bwd(const IocBwdArgs& a, hipStream_t s);
bool ioc_bwd_x3_supported(int mno, int H);                       // kernels_bwd_x3.hip: groups of up to 32 agents, H = 64 / 128
void launch_ioc_bwd_x3(const IocBwdArgs& a, hipStream_t s);      // a.WcT_h / a.WgT_h / a.WsT = the [hi | lo] packs "ioc/W?T16"
// cluster form (kernels_bwd_cl.hip): groups of 64 / 96 / 128 agents, H <= 128, <= 16 bins; grp_cnt zeroed per launch; != 0: shape not served
int launch_ioc_bwd_cluster(const IocBwdArgs& a, int* grp_cnt, int* err, hipStream_t s);

// ---- present-row compaction (kernels_compact.hip; DESIRE_FLAG_COMPACT_ROWS) ----
void launch_present_scan(const uint8_t* valid, int A, int32_t* amap, int32_t* inv, int32_t* count_dev, int32_t* count_host, hipStream_t s);
void launch_gather_agents(const float* in, float* out, const int32_t* amap, int P, int ld, hipStream_t s, const int32_t* dynP = nullptr);     // dynP / dynN: the count on the device (DynCount)
void launch_scatter_add_agents(const float* in, int ldi, float* out, int ldo, const int32_t* amap, int P, int n, hipStream_t s);
void launch_reparam_c(const float* params_c, const float* eps, float* z, const int32_t* amap, int P, int K, int mno, int L, int posterior, hipStream_t s, const int32_t* dynP = nullptr);
void launch_scatter_rows(const float* comp, float* full0, float* full1, const int32_t* amap, int P, int K, int mno, int n, hipStream_t s, const int32_t* dynP = nullptr);
void launch_gather_rows(const float* full, float* comp, const int32_t* amap, int P, int K, int mno, int n, hipStream_t s);
// IOC class repacking (DESIRE_FLAG_COMPACT_IOC)
void launch_class_scan(const uint8_t* valid, int n_scenes, int mno, int n_cls, const int* m4, int K, int min_rows, int32_t* cls_win, int32_t* cmap,
                       int32_t* cnt_dev, int32_t* cnt_host, hipStream_t s);
void launch_cls_gather_agents(const float* Hx, int ld, const float* p_last, const int32_t* gos, const int32_t* cmap, const int32_t* win, int n_c, int m_c,
                              float* Hx_c, float* p_c, uint8_t* valid_c, int32_t* gos_c, hipStream_t s, const int32_t* dynN = nullptr);
void launch_cls_rows(float* full, float* comp, const int32_t* cmap, int n_c, int m_c, int K, int mno, int n, int dir, hipStream_t s, int gpt = 0, const int32_t* dynN = nullptr);
void launch_cls_scatter_add_agents(const float* in, int ldi, float* out, int ldo, const int32_t* cmap, int NA, int n, hipStream_t s);
// encoder-stage compaction
void launch_valid_from_frames(const float* past, int n_scenes, int T, int mno, uint8_t* valid, hipStream_t s);
void launch_gather_frames(const float* frames, float* out, const int32_t* amap, int P, int T, int mno, hipStream_t s, const int32_t* dynP = nullptr);
void launch_scatter_agents(const float* in, float* out, const int32_t* amap, int P, int ld, hipStream_t s, const int32_t* dynP = nullptr);
void launch_gather_add_agents(const float* in, int ldi, float* out, int ldo, const int32_t* amap, int P, int n, hipStream_t s);
