// train.hip -- training half of the C ABI: training-mode buffers, backward orchestration, gradient access.
// The reference computes tf.gradients(cost) and an Adam update but never runs them (model/model.py:388-403,
// train.py:181); here they run.  Gradients live in ONE flat fp32 buffer in natural (TF) layouts, so a multi-GPU
// caller all-reduces a single tensor (RCCL through torch.distributed) between desire_backward and desire_adam_step.
#include "ctx.h"

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
        int accumulate, hipStream_t s) {
    TnArgs a{};
    a.A = A; a.lda = lda; a.G = Gm; a.ldg = ldg; a.M = M; a.Kd = Kd; a.N = N;
    const long blocks = ((Kd + 63) / 64) * ((N + 63) / 64);
    long sl = 2048 / blocks; if (sl < 1) sl = 1; if (sl > 256) sl = 256;
    const long maxsl = (M + 63) / 64; if (sl > maxsl) sl = maxsl;
    while ((size_t)sl * Kd * N * sizeof(float) > h->ws["tn_partial"].bytes && sl > 1) sl /= 2;
    a.nslices = (int)sl; a.partial = W(h, "tn_partial");
    launch_gemm_tn(a, out, ldo, accumulate, s);
}
void colsum(desire_ctx* h, const float* Gm, int ldg, long M, int N, float* out, int accumulate, hipStream_t s) {
    long sl = 256; const long maxsl = (M + 3) / 4; if (sl > maxsl) sl = maxsl; if (sl < 1) sl = 1;
    launch_colsum(Gm, ldg, M, N, (int)sl, W(h, "tn_partial"), out, accumulate, s);
}

}  // namespace

extern "C" int desire_set_training(desire_handle* h, int enable) {
    if (int rc = desire_ready(h)) return rc;
    if (!enable) { h->training = false; return DESIRE_OK; }
    const desire_dims& d = h->d;
    if (!d.posterior) return fail(DESIRE_ERR_STATE, "training needs the posterior path (dims.posterior = 1)");
    if (d.mno > 32) return fail(DESIRE_ERR_STATE, "training supports mno <= 32 in this round");
    if (d.iters != 1) return fail(DESIRE_ERR_STATE, "training supports one IOC refinement pass (iters = 1)");
    if (d.T_pred > d.H) return fail(DESIRE_ERR_STATE, "training needs T_pred <= H");
    if (h->slots.empty()) {
        size_t off = 0;
        for (auto& kv : h->want) { h->slots[kv.first] = WSlot{off, kv.second}; off += (kv.second + 3) / 4 * 4; }
        h->n_params = off;
    }
    const size_t R = h->R, T = d.T_pred, H = d.H, f = sizeof(float);
    const size_t Tm = d.T_pred > d.T_obs ? d.T_pred : d.T_obs;
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
        {"ioc_sv_x", R * T * (size_t)h->E * f}, {"ioc_sv_r", R * T * H * f}, {"ioc_sv_u", R * T * H * f}, {"ioc_sv_c", R * T * H * f},
        {"ioc_sv_h", R * T * H * f}, {"Y_ref", R * T * 2 * f}, {"score_sv", R * f}, {"dYr", R * T * 2 * f}, {"dscore", R * f},
        {"dscoreT", R * T * f}, {"ioc_dag", R * T * 2 * H * f}, {"ioc_dac", R * T * H * f}, {"ioc_rh", R * T * H * f},
        {"ioc_hprev", R * T * H * f}, {"ioc_dpre_r", R * T * H * f}, {"ioc_dpre_v", R * T * d.E_v * f}, {"ioc_vel", R * T * 2 * f},
        {"ioc_pooled", R * T * (size_t)h->B * H * f},
        {"enc_dag", (size_t)h->A * Tm * 2 * H * f}, {"enc_dac", (size_t)h->A * Tm * H * f}, {"enc_rh", (size_t)h->A * Tm * H * f},
        {"enc_hprev", (size_t)h->A * Tm * H * f},
        {"ex_sv_r", (size_t)h->A * d.T_obs * H * f}, {"ex_sv_u", (size_t)h->A * d.T_obs * H * f}, {"ex_sv_c", (size_t)h->A * d.T_obs * H * f},
        {"ex_sv_h", (size_t)h->A * d.T_obs * H * f}, {"ex_sv_x", (size_t)h->A * d.T_obs * 2 * f},
        {"ey_sv_r", (size_t)h->A * T * H * f}, {"ey_sv_u", (size_t)h->A * T * H * f}, {"ey_sv_c", (size_t)h->A * T * H * f},
        {"ey_sv_h", (size_t)h->A * T * H * f}, {"ey_sv_x", (size_t)h->A * T * 2 * f},
    };
    for (const B& b : bufs)
        if (ensure(h, b.n, b.bytes)) return fail(DESIRE_ERR_HIP, std::string("hipMalloc failed for training buffer ") + b.n);
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
    HIPCHK(hipMemsetAsync(W(h, "Gflat"), 0, h->n_params * sizeof(float), s));
    const uint8_t* valid = static_cast<const uint8_t*>(h->ws["valid"].p);
    launch_count_valid(valid, h->A, W(h, "nvalid"), s);
    // ---- sample-generation module ----
    launch_loss_grad_y(W(h, "Y0"), dev_fut, valid, W(h, "nvalid"), W(h, "dY0"), d.n_scenes, d.mno, d.K, T, d.sx, d.sy, s);
    DecBwdArgs b{};
    b.dY0 = W(h, "dY0"); b.sv_r = W(h, "dec_sv_r"); b.sv_u = W(h, "dec_sv_u"); b.sv_c = W(h, "dec_sv_c"); b.sv_h = W(h, "dec_sv_h");
    b.Hx = W(h, "HxHy"); b.ldhx = 2 * H; b.w_head = D(h, "head/w");
    b.WcT_h = D4(h, "dec/WcT_h"); b.WgT_h = D4(h, "dec/WgT_h"); b.WgT_x = D4(h, "dec/WgT_x"); b.WcT_x = D4(h, "dec/WcT_x");
    b.R = (int)R; b.K = d.K; b.mno = d.mno; b.T = T; b.H = H;
    b.dag = W(h, "dec_dag"); b.dac = W(h, "dec_dac"); b.rh = W(h, "dec_rh"); b.hprev = W(h, "dec_hprev");
    b.dxg = W(h, "dec_dxg"); b.dxc = W(h, "dec_dxc"); b.dxz = W(h, "dxz"); b.dHx_rows = W(h, "dHx_rows");
    { Timer t(h, s, "bwd_decoder"); launch_decoder_bwd(b, s); }
    {
        Timer t(h, s, "bwd_decoder_wgrad");
        tn(h, W(h, "dec_sv_h"), H, W(h, "dY0"), 2, R * T, H, 2, G(h, "head/w"), 2, 0, s);
        colsum(h, W(h, "dY0"), 2, R * T, 2, G(h, "head/b"), 0, s);
        float* gk = G(h, "dec/gates/kernel");        // [(H+H), 2H]
        tn(h, W(h, "xz"), H, W(h, "dec_dxg"), 2 * H, R, H, 2 * H, gk, 2 * H, 0, s);
        tn(h, W(h, "dec_hprev"), H, W(h, "dec_dag"), 2 * H, R * T, H, 2 * H, gk + (size_t)H * 2 * H, 2 * H, 0, s);
        colsum(h, W(h, "dec_dag"), 2 * H, R * T, 2 * H, G(h, "dec/gates/bias"), 0, s);
        float* ck = G(h, "dec/candidate/kernel");    // [(H+H), H]
        tn(h, W(h, "xz"), H, W(h, "dec_dxc"), H, R, H, H, ck, H, 0, s);
        tn(h, W(h, "dec_rh"), H, W(h, "dec_dac"), H, R * T, H, H, ck + (size_t)H * H, H, 0, s);
        colsum(h, W(h, "dec_dac"), H, R * T, H, G(h, "dec/candidate/bias"), 0, s);
    }
    const int V = h->V, L = d.L, A = h->A;
    // ---- ranking / refinement module (trajectories detached: its only path into the rest is dHx) ----
    {
        Timer t(h, s, "bwd_ioc");
        const int E = h->E, B = h->B;
        launch_loss_grad_y(W(h, "Y_ref"), dev_fut, valid, W(h, "nvalid"), W(h, "dYr"), d.n_scenes, d.mno, d.K, T, d.sx, d.sy, s);
        launch_score_grad(W(h, "Y0"), dev_fut, W(h, "score_sv"), valid, W(h, "nvalid"), W(h, "dscore"), W(h, "dscoreT"), d.n_scenes,
                          d.mno, d.K, T, d.sx, d.sy, s);
        IocBwdArgs q{};
        q.Y0 = W(h, "Y0"); q.p_last = W(h, "p_last"); q.valid = valid; q.Hx = W(h, "HxHy"); q.ldhx = 2 * H;
        q.dYr = W(h, "dYr"); q.dscore = W(h, "dscore");
        q.sv_x = W(h, "ioc_sv_x"); q.sv_r = W(h, "ioc_sv_r"); q.sv_u = W(h, "ioc_sv_u"); q.sv_c = W(h, "ioc_sv_c"); q.sv_h = W(h, "ioc_sv_h");
        q.w_score = D(h, "ioc/score_w");
        q.R = (int)R; q.K = d.K; q.mno = d.mno; q.T = T; q.H = H; q.G = d.grid_size; q.nb_w = d.nb_w; q.nb_h = d.nb_h;
        q.WrT = D4(h, "ioc/WrT"); q.WcT_h = D4(h, "ioc/WcT_h"); q.WcT_er = D4(h, "ioc/WcT_er"); q.WcT_ev = D4(h, "ioc/WcT_ev");
        q.WgT_h = D4(h, "ioc/WgT_h"); q.WgT_er = D4(h, "ioc/WgT_er"); q.WgT_ev = D4(h, "ioc/WgT_ev"); q.WsT = D4(h, "ioc/WsT");
        q.dag = W(h, "ioc_dag"); q.dac = W(h, "ioc_dac"); q.rh = W(h, "ioc_rh"); q.hprev = W(h, "ioc_hprev");
        q.dpre_r = W(h, "ioc_dpre_r"); q.dpre_v = W(h, "ioc_dpre_v"); q.vel = W(h, "ioc_vel"); q.pooled = W(h, "ioc_pooled");
        q.dHx_rows = W(h, "dHx_rows");
        launch_ioc_bwd(q, s);
        const long RT = R * T;
        tn(h, W(h, "ioc_sv_h") + (size_t)(T - 1) * H, T * H, W(h, "dYr"), 2 * T, R, H, 2 * T, G(h, "ioc/reg/w"), 2 * T, 0, s);
        colsum(h, W(h, "dYr"), 2 * T, R, 2 * T, G(h, "ioc/reg/b"), 0, s);
        tn(h, W(h, "ioc_sv_h"), H, W(h, "dscoreT"), 1, RT, H, 1, G(h, "ioc/score/w"), 1, 0, s);
        colsum(h, W(h, "dscoreT"), 1, RT, 1, G(h, "ioc/score/b"), 0, s);
        float* gk = G(h, "ioc/gates/kernel");            // [(E+H), 2H]
        tn(h, W(h, "ioc_sv_x"), E, W(h, "ioc_dag"), 2 * H, RT, E, 2 * H, gk, 2 * H, 0, s);
        tn(h, W(h, "ioc_hprev"), H, W(h, "ioc_dag"), 2 * H, RT, H, 2 * H, gk + (size_t)E * 2 * H, 2 * H, 0, s);
        colsum(h, W(h, "ioc_dag"), 2 * H, RT, 2 * H, G(h, "ioc/gates/bias"), 0, s);
        float* ck = G(h, "ioc/candidate/kernel");        // [(E+H), H]
        tn(h, W(h, "ioc_sv_x"), E, W(h, "ioc_dac"), H, RT, E, H, ck, H, 0, s);
        tn(h, W(h, "ioc_rh"), H, W(h, "ioc_dac"), H, RT, H, H, ck + (size_t)E * H, H, 0, s);
        colsum(h, W(h, "ioc_dac"), H, RT, H, G(h, "ioc/candidate/bias"), 0, s);
        tn(h, W(h, "ioc_pooled"), B * H, W(h, "ioc_dpre_r"), H, RT, B * H, H, G(h, "ioc/social_fc/w"), H, 0, s);
        colsum(h, W(h, "ioc_dpre_r"), H, RT, H, G(h, "ioc/social_fc/b"), 0, s);
        tn(h, W(h, "ioc_vel"), 2, W(h, "ioc_dpre_v"), d.E_v, RT, 2, d.E_v, G(h, "ioc/vel_fc/w"), d.E_v, 0, s);
        colsum(h, W(h, "ioc_dpre_v"), d.E_v, RT, d.E_v, G(h, "ioc/vel_fc/b"), 0, s);
    }
    // ---- mask fc ----
    {
        Timer t(h, s, "bwd_mask");
        launch_mask_bwd(W(h, "mask_sv_p"), W(h, "dxz"), W(h, "HxHy"), 2 * H, W(h, "dq_mask"), W(h, "dHx_rows"), (int)R, H, d.K, d.mno, s);
        colsum(h, W(h, "dq_mask"), H, R, H, G(h, "mask_fc/b"), 0, s);
        tn(h, W(h, "xhat"), V, W(h, "dq_mask"), H, R, V, H, G(h, "mask_fc/w"), H, 0, s);
        GemmArgs g{};
        g.A = W(h, "dq_mask"); g.lda = H; g.M = (int)R; g.K = H; g.Bp = D4(h, "mask/WT"); g.G = H / 8; g.NT = V / 32;
        g.out = W(h, "dconv4"); g.ldo = V; g.N = V; g.p0 = D(h, "vae_dec/deconv4/scale"); g.chmod = 1; g.aux = W(h, "xhat");
        launch_gemm_rows(g, EPI_SIGGRAD, s);
    }
    // ---- CVAE decoder (each data gradient = the forward kernel of the mirrored layer with a gradient epilogue) ----
    {
        Timer t(h, s, "bwd_cvae_dec");
        const int NSL = 40;
        launch_w1ch_grad(W(h, "dconv4"), W(h, "d3"), (int)R, 256, W(h, "tn_partial"), G(h, "vae_dec/deconv4/w"), s);
        colsum(h, W(h, "dconv4"), 1, R * 1024, 1, G(h, "vae_dec/deconv4/b"), 0, s);
        ConvArgs c{};
        c.n = (int)R;
        c.in = W(h, "dconv4"); c.out = W(h, "dconv3"); c.w_raw = D(h, "vae_dec/deconv4/raw");
        c.scale = D(h, "vae_dec/deconv3/scale"); c.shift = c.scale; c.mode = 1; c.yprev = W(h, "d3");
        launch_conv1(c, s);
        ConvWgradArgs wg{};
        wg.S = W(h, "d2"); wg.Cs = 64; wg.Ps = 8; wg.Lg = W(h, "dconv3"); wg.Cl = 32; wg.Pl = 16; wg.stride = 2; wg.pad = 1;
        wg.n = (int)R; wg.partial = W(h, "tn_partial");
        launch_conv_wgrad(wg, NSL, G(h, "vae_dec/deconv3/w"), s);
        colsum(h, W(h, "dconv3"), 32, R * 256, 32, G(h, "vae_dec/deconv3/b"), 0, s);
        c.in = W(h, "dconv3"); c.out = W(h, "dconv2"); c.Wp = D4(h, "vae_dec/deconv3/Wbwd");
        c.scale = D(h, "vae_dec/deconv2/scale"); c.shift = c.scale; c.yprev = W(h, "d2");
        launch_conv2(c, s);
        wg.S = W(h, "d1"); wg.Cs = 128; wg.Ps = 4; wg.Lg = W(h, "dconv2"); wg.Cl = 64; wg.Pl = 8; wg.stride = 1; wg.pad = 0;
        launch_conv_wgrad(wg, NSL, G(h, "vae_dec/deconv2/w"), s);
        colsum(h, W(h, "dconv2"), 64, R * 64, 64, G(h, "vae_dec/deconv2/b"), 0, s);
        c.in = W(h, "dconv2"); c.out = W(h, "dconv1"); c.Wp = D4(h, "vae_dec/deconv2/Wbwd");
        c.scale = D(h, "vae_dec/deconv1/scale"); c.shift = c.scale; c.yprev = W(h, "d1");
        launch_conv3(c, s);
        tn(h, W(h, "dconv1"), 2048, W(h, "z"), L, R, 2048, L, G(h, "vae_dec/deconv1/w"), L, 0, s);
        colsum(h, W(h, "dconv1"), 128, R * 16, 128, G(h, "vae_dec/deconv1/b"), 0, s);
        GemmArgs g{};
        g.A = W(h, "dconv1"); g.lda = 2048; g.M = (int)R; g.K = 2048; g.Bp = D4(h, "vae_dec/deconv1/WT"); g.G = 2048 / 8;
        g.NT = (L + 31) / 32; g.out = W(h, "dz"); g.ldo = L; g.N = L;
        launch_gemm_rows(g, EPI_NONE, s);
    }
    // ---- latent + CVAE encoder + fc_c ----
    {
        Timer t(h, s, "bwd_cvae_enc");
        launch_reparam_bwd(W(h, "dz"), dev_eps, W(h, "params"), valid, W(h, "nvalid"), W(h, "dparams"), d.n_scenes, d.mno, d.K, L, s);
        tn(h, W(h, "c3"), 2048, W(h, "dparams"), 2 * L, A, 2048, 2 * L, G(h, "vae_enc/fc/w"), 2 * L, 0, s);
        colsum(h, W(h, "dparams"), 2 * L, A, 2 * L, G(h, "vae_enc/fc/b"), 0, s);
        GemmArgs g{};
        g.A = W(h, "dparams"); g.lda = 2 * L; g.M = A; g.K = 2 * L; g.Bp = D4(h, "vae_enc/fc/WT"); g.G = 2 * L / 8; g.NT = 64;
        g.out = W(h, "dconvE3"); g.ldo = 2048; g.N = 2048; g.p0 = D(h, "vae_enc/conv3/scale"); g.chmod = 128; g.aux = W(h, "c3");
        launch_gemm_rows(g, EPI_ELUGRAD, s);
        const int NSL = 8;
        ConvWgradArgs wg{};
        wg.n = A; wg.partial = W(h, "tn_partial");
        wg.S = W(h, "dconvE3"); wg.Cs = 128; wg.Ps = 4; wg.Lg = W(h, "c2"); wg.Cl = 64; wg.Pl = 8; wg.stride = 1; wg.pad = 0;
        launch_conv_wgrad(wg, NSL, G(h, "vae_enc/conv3/w"), s);
        colsum(h, W(h, "dconvE3"), 128, (long)A * 16, 128, G(h, "vae_enc/conv3/b"), 0, s);
        ConvArgs c{};
        c.n = A; c.mode = 1;
        c.in = W(h, "dconvE3"); c.out = W(h, "dconvE2"); c.Wp = D4(h, "vae_enc/conv3/Wbwd");
        c.scale = D(h, "vae_enc/conv2/scale"); c.shift = c.scale; c.yprev = W(h, "c2");
        launch_deconv2(c, s);
        wg.S = W(h, "dconvE2"); wg.Cs = 64; wg.Ps = 8; wg.Lg = W(h, "c1"); wg.Cl = 32; wg.Pl = 16; wg.stride = 2; wg.pad = 1;
        launch_conv_wgrad(wg, NSL, G(h, "vae_enc/conv2/w"), s);
        colsum(h, W(h, "dconvE2"), 64, (long)A * 64, 64, G(h, "vae_enc/conv2/b"), 0, s);
        c.in = W(h, "dconvE2"); c.out = W(h, "dconvE1"); c.Wp = D4(h, "vae_enc/conv2/Wbwd");
        c.scale = D(h, "vae_enc/conv1/scale"); c.shift = c.scale; c.yprev = W(h, "c1");
        launch_deconv3(c, s);
        launch_w1ch_grad(W(h, "vae_in"), W(h, "dconvE1"), A, A < 64 ? A : 64, W(h, "tn_partial"), G(h, "vae_enc/conv1/w"), s);
        colsum(h, W(h, "dconvE1"), 32, (long)A * 256, 32, G(h, "vae_enc/conv1/b"), 0, s);
        c.in = W(h, "dconvE1"); c.out = W(h, "dq_c"); c.w_raw = D(h, "vae_enc/conv1/raw"); c.mode = 2; c.yprev = W(h, "vae_in");
        launch_deconv4(c, s);
        tn(h, W(h, "HxHy"), 2 * H, W(h, "dq_c"), V, A, 2 * H, V, G(h, "fc_c/w"), V, 0, s);
        colsum(h, W(h, "dq_c"), V, A, V, G(h, "fc_c/b"), 0, s);
        g = GemmArgs{};
        g.A = W(h, "dq_c"); g.lda = V; g.M = A; g.K = V; g.Bp = D4(h, "fc_c/WT"); g.G = V / 8; g.NT = 2 * H / 32;
        g.out = W(h, "dHxHy"); g.ldo = 2 * H; g.N = 2 * H;
        launch_gemm_rows(g, EPI_NONE, s);
        launch_rows_to_agents(W(h, "dHx_rows"), W(h, "dHxHy"), 2 * H, d.n_scenes, d.mno, d.K, H, s);
    }
    // ---- encoders: BPTT from the final state (Hx / Hy), zero initial state ----
    auto enc_bwd = [&](const std::string& p, const char* sv, int Te, int col0) {
        DecBwdArgs e{};
        e.sv_r = W(h, (std::string(sv) + "_sv_r").c_str()); e.sv_u = W(h, (std::string(sv) + "_sv_u").c_str());
        e.sv_c = W(h, (std::string(sv) + "_sv_c").c_str()); e.sv_h = W(h, (std::string(sv) + "_sv_h").c_str());
        e.w_head = D(h, "head/w");
        e.WcT_h = D4(h, (p + "/WcT_h").c_str()); e.WgT_h = D4(h, (p + "/WgT_h").c_str());
        e.R = A; e.K = 1; e.mno = d.mno; e.T = Te; e.H = H;
        e.dag = W(h, "enc_dag"); e.dac = W(h, "enc_dac"); e.rh = W(h, "enc_rh"); e.hprev = W(h, "enc_hprev");
        e.dh_init = W(h, "dHxHy") + col0; e.ld_init = 2 * H;
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
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_get_grad(desire_handle* h, const char* name, float* host_out, size_t n, void* stream) {
    if (!h || !name || !host_out) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    auto it = h->slots.find(name);
    if (it == h->slots.end()) return fail(DESIRE_ERR_ARG, std::string("unknown weight: ") + name);
    if (it->second.n != n) return fail(DESIRE_ERR_ARG, std::string(name) + ": expected " + std::to_string(it->second.n) + " values");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(host_out, W(h, "Gflat") + it->second.off, n * sizeof(float), hipMemcpyDeviceToHost));
    return DESIRE_OK;
}

extern "C" int desire_grad_buffer(desire_handle* h, float** dev_ptr, size_t* n) {
    if (!h || !dev_ptr || !n) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->training) return fail(DESIRE_ERR_STATE, "not in training mode");
    *dev_ptr = W(h, "Gflat");
    *n = h->n_params;
    return DESIRE_OK;
}
