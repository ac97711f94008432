// api_forward.hip -- the launch sequence of the hot path behind desire_encode / desire_sample / desire_ioc_refine / desire_forward, including the
// present-row compaction and the IOC slot classes (DESIRE_FLAG_COMPACT_*).  Host code only; split out of api.hip in round 5.
#include "ctx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// DESIRE_FLAG_COMPACT_ROWS: the per-row sample-generation stages run on the rows of present agents only (kernels_compact.hip)
bool compact_rows(const desire_ctx* h) { return (h->d.flags & DESIRE_FLAG_COMPACT_ROWS) != 0; }
// DESIRE_FLAG_COMPACT_IOC: windows re-seated in the smallest slot class that holds their present agents (kernels_compact.hip).  Shapes served by the
// step-wise IOC (more than 128 slots, or split operands at H = 256) keep their own layout.
bool compact_ioc(const desire_ctx* h) {
    const desire_dims& d = h->d;
    if (!(d.flags & DESIRE_FLAG_COMPACT_IOC) || d.mno > 128) return false;
    const int B_ = d.grid_size * d.grid_size;
    const bool split_mode = (d.bf16 == 2 || d.bf16 == 3) && !h->training;
    const bool split_served = ioc_x3_supported(d.mno, d.H, B_) || (d.mno == 64 && ioc_x6r2_supported(d.mno, d.H, B_));
    return !(split_mode && !split_served && d.H == 256 && d.ioc_form == DESIRE_IOC_AUTO);
}
// slot classes that do not divide 32 (padded tiles: k_ioc<TM = 32> and k_ioc_x3 have that form, inference, groups of <= 32 slots, H <= 128)
bool compact_padded_ok(const desire_ctx* h) {
    const desire_dims& d = h->d;
    if (h->training && !((d.bf16 == 0 || (d.bf16 == 2 && (train_x3_mask(h) & 4))) && d.ioc_form == DESIRE_IOC_AUTO)) return false;      // (BPTT: k_ioc_bwd / k_ioc_bwd_x3)
    return (d.bf16 == 0 || d.bf16 == 2) && d.H <= 128 && (d.ioc_form == DESIRE_IOC_AUTO);
}
int compact_classes(const desire_ctx* h, int* m4) {         // slot classes: the three largest candidates below the handle's own mno, then mno itself
    // candidates: 8, 16, 32, 64, 96; where the padded-tile kernels serve the handle also 10 (three groups of <= 10 slots per 32-row tile: a window with
    // 9 present agents -- the typical SDD bookstore window -- runs in 10 rows per sample instead of 16)
    int cand[6], nc = 0;
    const bool pad = compact_padded_ok(h);
    for (int m : {8, 10, 16, 32, 64, 96}) if (m < h->d.mno && (m != 10 || pad)) cand[nc++] = m;
    int n = 0;
    for (int i = nc > 3 ? nc - 3 : 0; i < nc; ++i) m4[n++] = cand[i];
    m4[n++] = h->d.mno;
    for (int i = n; i < 4; ++i) m4[i] = h->d.mno;
    return n;
}
// DEVICE-SIDE COUNTS (round 6): in inference with frozen batch-norm the host never learns how many agents are present -- every compacted launch is
// sized for the worst case and reads its count from the scans' device words (kernels.h: DynCount), so a compacted call has no host wait and can be
// captured in a hipGraph.  Training keeps the read-back (its backward sizes two dozen reductions from P), and so do per-sample batch statistics
// (bn_mode 1: the normalisation kernels are not count-aware) and desire_set_option("compact_host_counts", 1) -- the A/B switch.
bool compact_dyn(const desire_ctx* h) { return !h->training && h->d.bn_mode == 0 && !h->cp_host_counts; }
// worst case of one slot class: every window of the batch seated in it
static size_t class_rows_worst(const desire_ctx* h, int m_c) {
    const int gpt = (m_c <= 32 && 32 % m_c) ? 32 / m_c : 0;
    const size_t ngrp = (size_t)h->d.n_scenes * h->d.K;
    return gpt ? ((ngrp + gpt - 1) / gpt) * 32 : ngrp * m_c;
}
int compact_setup(desire_ctx* h) {
    const desire_dims& d = h->d;
    const size_t A = h->A, R = h->R, f = sizeof(float);
    // slot-class buffers: with device-side counts a class's region starts at a STATIC offset (the sum of the worst cases of the classes before it)
    size_t Ac = A, Rc = R + 128, Wc = (size_t)d.n_scenes;
    if (d.flags & DESIRE_FLAG_COMPACT_IOC) {
        int m4[4];
        const int n_cls = compact_classes(h, m4);
        Ac = 0; Rc = 0; Wc = 0;
        for (int c = 0; c < n_cls; ++c) { Ac += (size_t)d.n_scenes * m4[c]; Rc += class_rows_worst(h, m4[c]); Wc += (size_t)d.n_scenes; }
        Ac = std::max(Ac, A); Rc = std::max(Rc, R + 128);
    }
    struct WS { const char* n; size_t bytes; };
    const WS list[] = {{"cp_amap", A * sizeof(int32_t)}, {"cp_inv", A * sizeof(int32_t)}, {"cp_count", 8 * sizeof(int32_t)}, {"cp_HxHy", A * 2 * d.H * f},
                       {"cp_plast", A * 2 * f}, {"cp_params", A * 2 * d.L * f}, {"cp_Y0", R * (size_t)d.T_pred * 2 * f},
                       {"cp_past", A * (size_t)d.T_obs * 3 * f}, {"cp_fut", A * (size_t)d.T_pred * 3 * f}, {"cp_valid2", A}};
    const WS list_ioc[] = {{"ci_win", 4 * (size_t)d.n_scenes * sizeof(int32_t)}, {"ci_map", 4 * A * sizeof(int32_t)}, {"ci_Hx", Ac * 2 * d.H * f}, {"ci_pl", Ac * 2 * f},
                           {"ci_valid", Ac}, {"ci_gos", Wc * sizeof(int32_t)}, {"ci_Y", Rc * (size_t)d.T_pred * 2 * f}, {"ci_score", Rc * f}};      // (+ a partial padded tile per class)
    for (const WS& w : list)
        if (!h->ws[w.n].p && h->ws[w.n].alloc(w.bytes)) return fail(DESIRE_ERR_HIP, std::string("hipMalloc failed for ") + w.n);
    if (h->d.flags & DESIRE_FLAG_COMPACT_IOC)
        for (const WS& w : list_ioc)
            if (!h->ws[w.n].p && h->ws[w.n].alloc(w.bytes)) return fail(DESIRE_ERR_HIP, std::string("hipMalloc failed for ") + w.n);
    if (!h->cp_ev) HIPCHK(hipEventCreateWithFlags(&h->cp_ev, hipEventDisableTiming));
    if (!h->cp_host) {
        int32_t* p = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&p), 8 * sizeof(int32_t), hipHostMallocMapped) != hipSuccess || !p)
            return fail(DESIRE_ERR_HIP, "hipHostMalloc failed for the present-agent count words");
        for (int i = 0; i < 8; ++i) p[i] = 0;
        h->cp_host = p;
    }
    return DESIRE_OK;
}
// the present-agent scan (+ the slot-class scan) over `valid`, and the event desire_sample / desire_ioc_refine wait on
static int compact_scans(desire_ctx* h, hipStream_t s) {
    const desire_dims& d = h->d;
    launch_present_scan(static_cast<const uint8_t*>(h->ws["valid"].p), h->A, static_cast<int32_t*>(h->ws["cp_amap"].p), static_cast<int32_t*>(h->ws["cp_inv"].p),
                        static_cast<int32_t*>(h->ws["cp_count"].p), h->cp_host, s);
    if (compact_ioc(h)) {
        int m4[4];
        const int n_cls = compact_classes(h, m4);
        launch_class_scan(static_cast<const uint8_t*>(h->ws["valid"].p), d.n_scenes, d.mno, n_cls, m4, d.K, h->ci_min_rows, static_cast<int32_t*>(h->ws["ci_win"].p),
                          static_cast<int32_t*>(h->ws["ci_map"].p), static_cast<int32_t*>(h->ws["cp_count"].p) + 4, h->cp_host + 4, s);
    }
    if (!compact_dyn(h)) HIPCHK(hipEventRecord(h->cp_ev, s));             // (device-side counts: nobody waits, and the call stays capturable)
    h->cp_pending = true;
    return DESIRE_OK;
}
// waits (once per desire_encode) for the scans' counts to reach the host
static int compact_wait(desire_ctx* h, hipStream_t s) {
    if (!h->cp_pending) return fail(DESIRE_ERR_STATE, "DESIRE_FLAG_COMPACT_*: desire_encode comes first (it builds the present-agent maps)");
    if (compact_dyn(h)) return DESIRE_OK;                                // the kernels read the counts themselves
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (s && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
        return fail(DESIRE_ERR_STATE, "DESIRE_FLAG_COMPACT_* read the present-agent counts back: not capturable in a hipGraph");
    HIPCHK(hipEventSynchronize(h->cp_ev));
    return DESIRE_OK;
}

extern "C" int desire_encode(desire_handle* h, const float* dev_past, const float* dev_fut, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    const desire_dims& d = h->d;
    if (!dev_past) return fail(DESIRE_ERR_ARG, "dev_past is null");
    if (d.posterior && !dev_fut) return fail(DESIRE_ERR_ARG, "dims.posterior=1 needs dev_fut");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = d.H, A = h->A;
    // DESIRE_FLAG_COMPACT_ROWS: the encoder stack is per-agent as well.  `valid` is read off the last observed frame first, the scans run, the host
    // learns P, and the GRU encoders + CVAE encoder run on the P present agents as ONE pseudo-scene of P slots (frames gathered to [1, T, P, 3]);
    // HxHy / p_last / params are scattered back for the stages that keep the caller's layout (IOC, losses).  Ae = agents the stack runs on.
    // (not while the Gaussian-head loss is on: that term counts every (object, observed frame) pair, including objects that have left by the last
    //  observed frame, which the present-agent map does not hold)
    const bool enc_c = compact_rows(h) && !(h->training && h->head_loss_w > 0.f);
    const bool dyn = enc_c && compact_dyn(h);
    const int32_t* dynP = nullptr;                                       // the present-agent count on the device (cp_count[0]) when `dyn`
    int hintP = 0;
    h->cp_enc = false;
    int Ae = A;
    const float* pastE = dev_past; const float* futE = dev_fut;
    float* HxE = W(h, "HxHy"); float* plE = W(h, "p_last"); uint8_t* validE = static_cast<uint8_t*>(h->ws["valid"].p);
    float* paramsE = W(h, "params");
    const int32_t* amap = nullptr;
    if (enc_c) {
        if (int rc = compact_setup(h)) return rc;
        if (dyn) {
            dynP = static_cast<const int32_t*>(h->ws["cp_count"].p);
            // a GUESS of P for choices that are about speed only (which variant of a row GEMM): whatever the mapped word holds -- the previous call's count, or
            // this one's if the scan has already run.  Never a bound: the grids are the worst case's and the kernels read the real count.
            hintP = *static_cast<volatile int32_t*>(h->cp_host);
            if (hintP < 0 || hintP > A) hintP = 0;
        }
        launch_valid_from_frames(dev_past, d.n_scenes, d.T_obs, d.mno, validE, s);
        if (int rc = compact_scans(h, s)) return rc;
        if (int rc = compact_wait(h, s)) return rc;
        int P = A;                                                       // device-side counts: the worst case sizes the launches
        if (!dyn) {
            P = *static_cast<volatile int32_t*>(h->cp_host);
            if (P < 0 || P > A) return fail(DESIRE_ERR_HIP, "present-agent scan returned a count out of range");
        }
        h->cp_P = dyn ? -1 : P; h->cp_enc = true; Ae = P;
        launch_fill_f32(W(h, "HxHy"), (size_t)A * 2 * H, 0.f, s); launch_fill_f32(W(h, "p_last"), (size_t)A * 2, 0.f, s);
        if (d.posterior) launch_fill_f32(W(h, "params"), (size_t)A * 2 * d.L, 0.f, s);
        if (P == 0) { HIPCHK(hipGetLastError()); return DESIRE_OK; }
        amap = static_cast<const int32_t*>(h->ws["cp_amap"].p);
        launch_gather_frames(dev_past, W(h, "cp_past"), amap, P, d.T_obs, d.mno, s, dynP);
        if (d.posterior) launch_gather_frames(dev_fut, W(h, "cp_fut"), amap, P, d.T_pred, d.mno, s, dynP);
        pastE = W(h, "cp_past"); futE = W(h, "cp_fut");
        HxE = W(h, "cp_HxHy"); plE = W(h, "cp_plast"); validE = static_cast<uint8_t*>(h->ws["cp_valid2"].p); paramsE = W(h, "cp_params");
    }
    EncArgs e{};
    e.n_scenes = enc_c ? 1 : d.n_scenes; e.mno = enc_c ? Ae : d.mno; e.sx = d.sx; e.sy = d.sy; e.H = H;
    e.frames = pastE; e.T = d.T_obs;
    e.wx_g = D(h, "enc_x/gk"); e.b_g = D(h, "enc_x/gb"); e.wx_c = D(h, "enc_x/ck"); e.b_c = D(h, "enc_x/cb");
    e.Whg = D4(h, "enc_x/Whg"); e.Whc = D4(h, "enc_x/Whc");
    e.out = HxE; e.ldo = 2 * H; e.p_last = plE; e.valid = validE;
    e.dyn = DynCount{dynP, 1, hintP};
    if (h->training) { e.sv_r = W(h, "ex_sv_r"); e.sv_u = W(h, "ex_sv_u"); e.sv_c = W(h, "ex_sv_c"); e.sv_h = W(h, "ex_sv_h"); e.sv_x = W(h, "ex_sv_x"); }
    const EncArgs ex = e;
    if (d.posterior) {
        e.frames = futE; e.T = d.T_pred;
        e.wx_g = D(h, "enc_y/gk"); e.b_g = D(h, "enc_y/gb"); e.wx_c = D(h, "enc_y/ck"); e.b_c = D(h, "enc_y/cb");
        e.Whg = D4(h, "enc_y/Whg"); e.Whc = D4(h, "enc_y/Whc");
        e.out = HxE + H; e.p_last = nullptr; e.valid = nullptr;
        if (h->training) { e.sv_r = W(h, "ey_sv_r"); e.sv_u = W(h, "ey_sv_u"); e.sv_c = W(h, "ey_sv_c"); e.sv_h = W(h, "ey_sv_h"); e.sv_x = W(h, "ey_sv_x"); }
    }
    if (d.bf16 == 1) {
        EncArgs e16 = ex;
        e16.Whg = D4(h, "enc_x/Whg16"); e16.Whc = D4(h, "enc_x/Whc16");
        { Timer t(h, s, "encoder_x"); launch_encoder_bf16(e16, s); }
        if (d.posterior) { e.Whg = D4(h, "enc_y/Whg16"); e.Whc = D4(h, "enc_y/Whc16"); Timer t(h, s, "encoder_y"); launch_encoder_bf16(e, s); }
    } else if (d.posterior) {      // the two encoders are independent and latency-bound: one launch
        Timer t(h, s, "encoder_xy"); launch_encoder_pair(ex, e, s);
    } else { Timer t(h, s, "encoder_x"); launch_encoder(ex, s); }
    if (enc_c) {          // back to the caller's layout for the IOC stage (absent agents: zeros, filled above)
        launch_scatter_agents(HxE, W(h, "HxHy"), amap, Ae, 2 * H, s, dynP);
        launch_scatter_agents(plE, W(h, "p_last"), amap, Ae, 2, s, dynP);
    } else if (compact_rows(h) || compact_ioc(h)) {
        // DESIRE_FLAG_COMPACT_IOC alone: the slot-class maps are built behind the encoder that writes `valid`; their sizes reach the host through a
        // mapped word while the CVAE encoder below keeps the device busy, and desire_ioc_refine waits on the event before it sizes its launches
        if (int rc = compact_setup(h)) return rc;
        if (int rc = compact_scans(h, s)) return rc;
    }
    if (d.posterior) {
        GemmArgs g{};
        g.A = HxE; g.lda = 2 * H; g.M = Ae; g.K = 2 * H; g.Bp = D4(h, "fc_c/W"); g.G = 2 * H / 8;
        g.NT = h->V / 32; g.out = W(h, "vae_in"); g.ldo = h->V; g.N = h->V; g.p0 = D(h, "fc_c/b");
        g.dyn = DynCount{dynP, 1, hintP}; g.M_hint = hintP;
        { Timer t(h, s, "fc_c"); launch_gemm_rows(g, EPI_BIAS_RELU, s); }
        ConvArgs c{};
        c.n = Ae; c.dyn = DynCount{dynP, 1, hintP};
        c.in = W(h, "vae_in"); c.out = W(h, "c1"); c.w_raw = D(h, "vae_enc/conv1/raw");
        c.scale = D(h, "vae_enc/conv1/scale"); c.shift = D(h, "vae_enc/conv1/shift");
        const bool pobn = d.bn_mode != 0;                 // batch statistics: linear conv epilogue, then a normalise + activate pass per layer
        auto norm = [&](const char* layer, float* x, int n, int P, int C, int sig) {     // 1: per sample (k_instnorm_act), 2: over the whole batch
            const float* ga = D(h, (std::string(layer) + "/gamma").c_str()); const float* be = D(h, (std::string(layer) + "/beta").c_str());
            if (h->training)            // the batch-statistics backward needs the pre-norm tensor: kept next to the activation
                launch_copy_f32(W(h, (std::string(layer).substr(std::string(layer).rfind('/') + 1) + "_pre").c_str()), x, (size_t)n * P * C, s);
            if (d.bn_mode == 2) launch_batchnorm_act(x, (size_t)n, P, C, ga, be, sig, W(h, "bn_part"), W(h, "bn_stat"), s);
            else launch_instnorm_act(x, n, P, C, ga, be, sig, s);
        };
        if (pobn) c.mode = 3;
        { Timer t(h, s, "conv1"); launch_conv1(c, s); if (pobn) norm("vae_enc/conv1", W(h, "c1"), Ae, 256, 32, 0); }
        c.in = W(h, "c1"); c.out = W(h, "c2"); c.Wp = D4(h, "vae_enc/conv2/W");
        c.scale = D(h, "vae_enc/conv2/scale"); c.shift = D(h, "vae_enc/conv2/shift");
        if (d.bf16 == 1) { c.Wp = D4(h, "vae_enc/conv2/W16"); Timer t(h, s, "conv2"); launch_conv2_bf16(c, s); }
        else { Timer t(h, s, "conv2"); launch_conv2(c, s); if (pobn) norm("vae_enc/conv2", W(h, "c2"), Ae, 64, 64, 0); }
        c.in = W(h, "c2"); c.out = W(h, "c3"); c.Wp = D4(h, "vae_enc/conv3/W");
        c.scale = D(h, "vae_enc/conv3/scale"); c.shift = D(h, "vae_enc/conv3/shift");
        if (d.bf16 == 1) { c.Wp = D4(h, "vae_enc/conv3/W16"); Timer t(h, s, "conv3"); launch_conv3_bf16(c, s); }
        else { Timer t(h, s, "conv3"); launch_conv3(c, s); if (pobn) norm("vae_enc/conv3", W(h, "c3"), Ae, 16, 128, 0); }
        g = GemmArgs{};
        g.A = W(h, "c3"); g.lda = 2048; g.M = Ae; g.K = 2048; g.Bp = D4(h, "vae_enc/fc/W"); g.G = 2048 / 8;
        g.NT = (2 * d.L + 31) / 32; g.out = paramsE; g.ldo = 2 * d.L; g.N = 2 * d.L; g.p0 = D(h, "vae_enc/fc/b");
        g.dyn = DynCount{dynP, 1, hintP}; g.M_hint = hintP;
        { Timer t(h, s, "vae_enc_fc"); launch_gemm_rows(g, EPI_BIAS, s); }
        if (enc_c) launch_scatter_agents(paramsE, W(h, "params"), amap, Ae, 2 * d.L, s, dynP);       // desire_losses / the reparam backward read them per agent
    }
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_sample(desire_handle* h, const float* dev_eps, float* dev_Yhat, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!dev_eps || !dev_Yhat) return fail(DESIRE_ERR_ARG, "null argument");
    const desire_dims& d = h->d;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int H = d.H;
    // per-row stages: all R = A*K rows, or (DESIRE_FLAG_COMPACT_ROWS) the K*P rows of the P present agents laid out as one pseudo-scene of P
    // slots (kernels_compact.hip) -- the kernels below are the same either way, they only see (R, K, mno) and the agent-level inputs
    int R = h->R, mno = d.mno;
    const float* HxS = W(h, "HxHy"); const float* plS = W(h, "p_last"); float* Yout = W(h, "Y0");
    const bool compact = compact_rows(h);
    const bool dyn = compact && compact_dyn(h);
    const int32_t* dynP = dyn ? static_cast<const int32_t*>(h->ws["cp_count"].p) : nullptr;      // device-side count: launches sized for P = A
    h->cp_last = compact;
    if (compact) {
        if (int rc = compact_wait(h, s)) return rc;
        int P = h->A;
        if (!dyn) {
            P = *static_cast<volatile int32_t*>(h->cp_host);
            if (P < 0 || P > h->A) return fail(DESIRE_ERR_HIP, "present-agent scan returned a count out of range");
        }
        h->cp_P = dyn ? -1 : P;
        R = P * d.K; mno = P;
        const size_t RT2 = (size_t)h->R * d.T_pred * 2;
        if (P == 0) {       // nothing present: every row is padding
            launch_fill_f32(W(h, "Y0"), RT2, 0.f, s); launch_fill_f32(dev_Yhat, RT2, 0.f, s);
            HIPCHK(hipGetLastError());
            return DESIRE_OK;
        }
        const int32_t* amap = static_cast<const int32_t*>(h->ws["cp_amap"].p);
        Timer t(h, s, "compact_gather");
        if (!h->cp_enc) {           // (an encoder stack that ran compact has left all three in place)
            launch_gather_agents(W(h, "HxHy"), W(h, "cp_HxHy"), amap, P, 2 * H, s, dynP);
            launch_gather_agents(W(h, "p_last"), W(h, "cp_plast"), amap, P, 2, s, dynP);
            if (d.posterior) launch_gather_agents(W(h, "params"), W(h, "cp_params"), amap, P, 2 * d.L, s, dynP);
        }
        HxS = W(h, "cp_HxHy"); plS = W(h, "cp_plast"); Yout = W(h, "cp_Y0");
    }
    if (compact) { Timer t(h, s, "reparam"); launch_reparam_c(W(h, "cp_params"), dev_eps, W(h, "z"), static_cast<const int32_t*>(h->ws["cp_amap"].p), mno, d.K, d.mno, d.L, d.posterior, s, dynP); }
    else { Timer t(h, s, "reparam"); launch_reparam(W(h, "params"), dev_eps, W(h, "z"), R, d.L, d.K, d.mno, d.posterior, s); }
    auto normd = [&](const char* layer, float* x, int P, int C, int sig) {          // batch statistics of the decoder layers (see desire_encode)
        const float* ga = D(h, (std::string(layer) + "/gamma").c_str()); const float* be = D(h, (std::string(layer) + "/beta").c_str());
        if (h->training)
            launch_copy_f32(W(h, (std::string(layer).substr(std::string(layer).rfind('/') + 1) + "_pre").c_str()), x, (size_t)R * P * C, s);
        if (d.bn_mode == 2) launch_batchnorm_act(x, (size_t)R, P, C, ga, be, sig, W(h, "bn_part"), W(h, "bn_stat"), s);
        else launch_instnorm_act(x, R, P, C, ga, be, sig, s);
    };
    GemmArgs g{};
    g.A = W(h, "z"); g.lda = d.L; g.M = R; g.K = d.L; g.Bp = D4(h, "vae_dec/deconv1/W"); g.G = d.L / 8;
    g.NT = 64; g.out = W(h, "d1"); g.ldo = 2048; g.N = 2048;
    g.p0 = D(h, "vae_dec/deconv1/scale"); g.p1 = D(h, "vae_dec/deconv1/shift"); g.chmod = 128;
    int hintS = 0;                                                       // count hint for this call's launches (see desire_encode)
    if (dyn) { const int hp = *static_cast<volatile int32_t*>(h->cp_host); hintS = (hp > 0 && hp <= h->A) ? hp : 0; }
    g.dyn = DynCount{dynP, d.K, hintS}; g.M_hint = hintS * d.K;
    // six-product sample generation (the fp32 kernels' accuracy class on the bf16 matrix pipe): dims.bf16 = 3, and dims.bf16 = 2 as well --
    // two-piece operands are an IOC-kernel matter (sample generation must not move Y0 by more than fp32 rounding)
    const bool x6gen = ((d.bf16 == 3 && !h->training) || (d.bf16 == 2 && (!h->training || (train_x3_mask(h) & 8)))) && d.bn_mode == 0 && !d.ref_compat;
    if (d.bf16 == 1 && d.L <= 512 && !(d.L & 15)) { g.Bp = D4(h, "vae_dec/deconv1/W16"); Timer t(h, s, "deconv1"); launch_deconv1_bf16(g, s); }
    else if (x6gen && rows_x6_supported(d.L, 64)) { g.Bp = D4(h, "vae_dec/deconv1/W6"); Timer t(h, s, "deconv1"); launch_deconv1_x6(g, s); }
    else if (d.bn_mode != 0) {
        Timer t(h, s, "deconv1"); launch_gemm_rows(g, EPI_NONE, s);
        normd("vae_dec/deconv1", W(h, "d1"), 16, 128, 0);
    }
    else { Timer t(h, s, "deconv1"); launch_gemm_rows(g, EPI_SCALE_SHIFT_ELU, s); }
    ConvArgs c{};
    c.n = R; c.dyn = DynCount{dynP, d.K, hintS};
    const bool pobn = d.bn_mode != 0;
    if (pobn) c.mode = 3;
    c.in = W(h, "d1"); c.out = W(h, "d2"); c.Wp = D4(h, "vae_dec/deconv2/W");
    c.scale = D(h, "vae_dec/deconv2/scale"); c.shift = D(h, "vae_dec/deconv2/shift");
    // dims.bf16 = 3: six-product forms of the two large transposed convolutions and of the decoder (frozen batch-norm, inference)
    if (d.bf16 == 1) { c.Wp = D4(h, "vae_dec/deconv2/W16"); Timer t(h, s, "deconv2"); launch_deconv2_bf16(c, s); }
    else if (x6gen) { c.Wp = D4(h, "vae_dec/deconv2/W6"); Timer t(h, s, "deconv2"); launch_deconv2_x6(c, s, (h->training && (d.flags & DESIRE_FLAG_TRAIN_FWD_3P)) ? 2 : 3); }
    else { Timer t(h, s, "deconv2"); launch_deconv2(c, s);
           if (pobn) normd("vae_dec/deconv2", W(h, "d2"), 64, 64, 0); }
    c.in = W(h, "d2"); c.out = W(h, "d3"); c.Wp = D4(h, "vae_dec/deconv3/W");
    c.scale = D(h, "vae_dec/deconv3/scale"); c.shift = D(h, "vae_dec/deconv3/shift");
    const bool fuse34 = d.bf16 == 1 && !(d.flags & DESIRE_FLAG_NO_FUSE34);       // bf16: deconv3+deconv4 in one kernel, d3 never written
    // (the six-product form of that fusion was measured and dropped: 15.4 ms against 11.9 + 2.5 for the two kernels -- the tap products cost
    //  the contracting waves more than the d3 pass did)
    if (fuse34) {
        c.Wp = D4(h, "vae_dec/deconv3/W16"); c.w_raw = D(h, "vae_dec/deconv4/W16"); c.out = W(h, "xhat");
        Timer t(h, s, "deconv34");
        launch_deconv34_bf16(c, D(h, "vae_dec/deconv4/scale"), D(h, "vae_dec/deconv4/shift"), s);
    } else {
        if (d.bf16 == 1) { c.Wp = D4(h, "vae_dec/deconv3/W16"); Timer t(h, s, "deconv3"); launch_deconv3_bf16(c, s); }
        else if (x6gen) { c.Wp = D4(h, "vae_dec/deconv3/W6"); Timer t(h, s, "deconv3"); launch_deconv3_x6(c, s, (h->training && (d.flags & DESIRE_FLAG_TRAIN_FWD_3P)) ? 2 : 3); }
        else { Timer t(h, s, "deconv3"); launch_deconv3(c, s);
               if (pobn) normd("vae_dec/deconv3", W(h, "d3"), 256, 32, 0); }
        c.in = W(h, "d3"); c.out = W(h, "xhat"); c.w_raw = D(h, "vae_dec/deconv4/raw");
        c.scale = D(h, "vae_dec/deconv4/scale"); c.shift = D(h, "vae_dec/deconv4/shift");
        { Timer t(h, s, "deconv4"); launch_deconv4(c, s);
          if (pobn) normd("vae_dec/deconv4", W(h, "xhat"), 1024, 1, 1); }
    }
    MaskArgs m{};
    m.xhat = W(h, "xhat"); m.R = R; m.V = h->V; m.H = H; m.Hl = h->Hl; m.K = d.K; m.mno = mno;
    m.Wp = D4(h, "mask/W"); m.bias = D(h, "mask/b"); m.Hx = HxS; m.ldhx = 2 * H; m.xz = W(h, "xz");
    m.dyn = DynCount{dynP, 1, hintS};
    if (h->training) m.sv_p = W(h, "mask_sv_p");
    if (d.bf16 == 1) { m.Wp = D4(h, "mask/W16"); Timer t(h, s, "mask_fc"); launch_mask_bf16(m, s); }
    else if (x6gen && (H == 64 || H == 128) && h->V % 128 == 0) { m.Wp = D4(h, "mask/W6"); Timer t(h, s, "mask_fc"); launch_mask_x6(m, s); }
    else { Timer t(h, s, "mask_fc"); launch_mask(m, s); }
    DecArgs a{};
    a.xz = W(h, "xz"); a.Hx = HxS; a.ldhx = 2 * H; a.p_last = plS;
    a.R = R; a.K = d.K; a.mno = mno; a.H = H; a.T = d.T_pred;
    a.Wxg = D4(h, "dec/Wxg"); a.Wxc = D4(h, "dec/Wxc"); a.Whg = D4(h, "dec/Whg"); a.Whc = D4(h, "dec/Whc");
    a.b_g = D(h, "dec/gb"); a.b_c = D(h, "dec/cb"); a.w_head = D(h, "head/w"); a.b_head = D(h, "head/b");
    a.Y = Yout; a.hdump = nullptr; a.dyn = DynCount{dynP, 1, hintS};
    if (d.ref_compat) { a.T = d.n_dec; a.hdump = W(h, "dec_states"); }       // model/model.py:280-285: 7 steps, the states are the output
    if (h->training) { a.hdump = W(h, "dec_sv_h"); a.sv_r = W(h, "dec_sv_r"); a.sv_u = W(h, "dec_sv_u"); a.sv_c = W(h, "dec_sv_c"); }
    if (d.bf16 == 1) {
        a.Whg = D4(h, "dec/Whg16"); a.Whc = D4(h, "dec/Whc16");
        Timer t(h, s, "decoder"); launch_decoder_bf16(a, s);
    } else
    if (x6gen && decoder_x6_supported(H)) {
        a.Whg = D4(h, "dec/Whg6"); a.Whc = D4(h, "dec/Whc6");
        Timer t(h, s, "decoder"); launch_decoder_x6(a, s, (h->training && (d.flags & DESIRE_FLAG_TRAIN_FWD_3P)) ? 2 : 3);
    } else
    { Timer t(h, s, "decoder"); launch_decoder(a, s); }
    if (d.ref_compat)      // model/model.py:286-289: each state [H] re-read as T_obs points (x, y) -> [A, n_dec, T_obs, 2]
        launch_copy_cols(dev_Yhat, W(h, "dec_states"), (size_t)R * d.n_dec, h->Hl, H, s);
    else if (compact) {     // back to the caller's row layout; rows of absent agents are zeros (the cost masks them, model/model.py:351-366)
        Timer t(h, s, "compact_scatter");
        const size_t RT2 = (size_t)h->R * d.T_pred * 2;
        launch_fill_f32(W(h, "Y0"), RT2, 0.f, s); launch_fill_f32(dev_Yhat, RT2, 0.f, s);
        launch_scatter_rows(Yout, W(h, "Y0"), dev_Yhat, static_cast<const int32_t*>(h->ws["cp_amap"].p), mno, d.K, d.mno, d.T_pred * 2, s, dynP);
    } else
        launch_copy_f32(dev_Yhat, W(h, "Y0"), (size_t)R * d.T_pred * 2, s);
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

// One IOC launch sequence over a VIEW of the handle's rows: the handle's own shape (row_off 0), or one slot class of DESIRE_FLAG_COMPACT_IOC --
// n_scenes windows of mno slots each with their own agent-level inputs; training-mode saves go to the view's row offset in the shared buffers.
struct IocView {
    int R, mno, n_scenes; float* Y; float* score; const float* Hx; int ldhx; const float* p_last; const uint8_t* valid; const int32_t* gos; size_t row_off;
    const int32_t* dynN = nullptr;   // the view's window count on the device (a slot class under device-side counts): R / n_scenes / ngrp above are the worst case
    int gpt = 0, ngrp = 0;       // padded tiles (slot classes that do not divide 32; kernels.h: IocArgs.gpt): groups per 32-row tile, real groups; R = tiles * 32
};
static int ioc_core(desire_handle* h, const IocView& v, hipStream_t s) {
    const desire_dims& d = h->d;
    IocArgs a{};
    a.Y = v.Y; a.score = v.score; a.Hx = v.Hx; a.ldhx = v.ldhx; a.p_last = v.p_last;
    a.valid = v.valid;
    a.R = v.R; a.K = d.K; a.mno = v.mno; a.H = d.H; a.T = d.T_pred; a.iters = d.iters;
    a.C = d.C; a.Gh = d.Gh; a.Gw = d.Gw; a.E_v = d.E_v; a.G = d.grid_size; a.nb_w = d.nb_w; a.nb_h = d.nb_h;
    a.grids = h->grids; a.grid_of_scene = v.gos;
    a.bin_tab = d.bin_mode == 1 ? W(h, "bin_tab") : nullptr;
    a.w_vel = D(h, "ioc/vel_w"); a.b_vel = D(h, "ioc/vel_b");
    a.Wsoc = D4(h, "ioc/Wsoc"); a.b_soc = D(h, "ioc/soc_b"); a.Wsoc_c = D4(h, "ioc/Wsoc_c");
    a.Wg = D4(h, "ioc/Wg"); a.Wc = D4(h, "ioc/Wc"); a.b_g = D(h, "ioc/gb"); a.b_c = D(h, "ioc/cb");
    a.w_score = D(h, "ioc/score_w"); a.b_score = D(h, "ioc/score_b");
    a.Wreg = D4(h, "ioc/Wreg"); a.b_reg = D(h, "ioc/reg_b"); a.NTreg = (2 * d.T_pred + 31) / 32;
    a.variant = d.ioc_form;
    a.gpt = v.gpt; a.ngrp = v.ngrp;
    a.dyn = DynCount{v.dynN, 1, 0};
    // bf16: one workgroup holds groups of up to 64 agents; 96 / 128 (and 64 when variant 4 / 6 asks for it) run the cluster form
    // split forms: groups of up to 32 agents on 32-row tiles (also the training-mode forward); inference on groups of 64 agents runs the
    // 64-row tile of kernels_x6r2.hip (one group per tile) in either piece count
    const bool wide64 = v.mno == 64 && !h->training && ioc_x6r2_supported(v.mno, d.H, d.grid_size * d.grid_size);
    const bool x3 = d.bf16 == 2 && (ioc_x3_supported(v.mno, d.H, d.grid_size * d.grid_size) || wide64 || (v.gpt > 0 && ioc_x3_supported(32, d.H, d.grid_size * d.grid_size)));
    const bool x6 = d.bf16 == 3 && (ioc_x3_supported(v.mno, d.H, d.grid_size * d.grid_size) || wide64);     // six-product form: inference only
    const bool cluster = d.bf16 == 1 ? (v.mno > 64 || (v.mno == 64 && (a.variant == 4 || a.variant == 6)))
                                : (!(x3 || x6) || h->training) && ioc_uses_cluster(v.mno, d.H, d.grid_size * d.grid_size, a.variant);
    if (cluster) {
        const size_t n_groups = (size_t)v.R / v.mno;
        if (!h->ws.count("hex")) {          // (sized for the handle's own shape: every view of it -- DESIRE_FLAG_COMPACT_IOC classes -- is smaller)
            if (h->ws["hex"].alloc((size_t)2 * h->R * d.H * sizeof(float)) || h->ws["grp_cnt"].alloc(((size_t)h->R / 32 + 1) * sizeof(int)) ||
                h->ws["ioc_err"].alloc(sizeof(int)))
                return fail(DESIRE_ERR_HIP, "hipMalloc failed for the cluster exchange buffers");
        }
        HIPCHK(hipMemsetAsync(h->ws["grp_cnt"].p, 0, n_groups * sizeof(int), s));
        HIPCHK(hipMemsetAsync(h->ws["ioc_err"].p, 0, sizeof(int), s));
        a.hex = W(h, "hex"); a.grp_cnt = static_cast<int*>(h->ws["grp_cnt"].p); a.err = static_cast<int*>(h->ws["ioc_err"].p);
    }
    // a handful of windows, fp32 inference: the bins of every tile split over several workgroups (k_ioc NSPL; dims.ioc_split = 1: off).
    // The members of a tile wait for each other, so the split is taken only when the whole launch is co-resident on THIS device
    // (occupancy x compute units, not a constant: a partition with fewer CUs falls back to the plain form).
    if (!cluster && d.bf16 == 0 && !h->training && d.ioc_split != 1 && a.variant == 0 && v.gpt == 0) {
        int nspl = ioc_bin_split(v.R, v.mno, d.H, d.grid_size * d.grid_size, d.iters);
        if (nspl > 1 && d.ioc_split > 1) nspl = std::min(nspl, d.ioc_split);
        const size_t tiles = ((size_t)v.R + 31) / 32, tiles_max = ((size_t)h->R + 31) / 32;
        while (nspl > 1 && (size_t)ioc_bin_split_capacity(a, nspl) < tiles * nspl) --nspl;
        if (nspl > 1) {
            if (!h->ws.count("hex_s") || !h->ws["hex_s"].p || !h->ws["cnt_s"].p) {
                if (h->ws["hex_s"].alloc(tiles_max * 2 * 4 * 32 * d.H * sizeof(float)) || h->ws["cnt_s"].alloc(tiles_max * sizeof(int)))
                    return fail(DESIRE_ERR_HIP, "hipMalloc failed for the bin-split exchange buffers");
            }
            // the error word is mapped host memory: no read-back (and no stream synchronisation) per call; a timed-out hand-off is
            // reported by the NEXT call on this handle.  Allocated and checked on its own (a failure here must not leave a later call
            // with exchange buffers and a null word); the kernels write it with system-scope atomics.
            if (!h->host_err) {
                if (hipHostMalloc(reinterpret_cast<void**>(&h->host_err), sizeof(int), hipHostMallocMapped) != hipSuccess || !h->host_err) {
                    h->host_err = nullptr;
                    return fail(DESIRE_ERR_HIP, "hipHostMalloc failed for the bin-split error word");
                }
                *h->host_err = 0;
            }
            if (*static_cast<volatile int*>(h->host_err)) {
                *h->host_err = 0;
                return fail(DESIRE_ERR_HIP, "bin-split IOC hand-off timed out in an earlier call (workgroups of a tile were not co-resident)");
            }
            // (a fill KERNEL, not hipMemsetAsync: memset nodes of a captured graph were seen to run out of order on replay -- section 6a --
            //  and a counter that still holds the previous pass's arrivals lets every member read its peers' slots before they are written)
            launch_fill_f32(W(h, "cnt_s"), tiles, 0.f, s);
            a.hex = W(h, "hex_s"); a.grp_cnt = static_cast<int*>(h->ws["cnt_s"].p); a.err = h->host_err;
            a.nspl = nspl;
        }
    }
#ifdef DESIRE_IOC_TIMING
    if (!h->ws.count("dbg")) { h->ws["dbg"].alloc(10 * sizeof(long long)); }
    a.dbg = static_cast<long long*>(h->ws["dbg"].p);
#endif
    if (h->training && d.bf16 != 1) {
        // training-mode forward: one launch per refinement pass, each keeping its own activations and the positions it ran on
        // (the pass's input is DETACHED where it enters the features -- cells, bins, velocity embedding -- and Y_p = Y_{p-1} + dY_p
        // carries the gradient: DESIGN.md section 8)
        const size_t RT = (size_t)v.R * d.T_pred, RTf = (size_t)(h->R + 128) * d.T_pred, ro = v.row_off * d.T_pred;     // a view's saves sit at its row offset
        a.iters = 1;
        for (int p = 0; p < d.iters; ++p) {
            const size_t po = (size_t)p * RTf + ro;
            launch_copy_f32(W(h, "ioc_Yin") + po * 2, v.Y, RT * 2, s);
            a.sv_x = W(h, "ioc_sv_x") + po * h->E; a.sv_r = W(h, "ioc_sv_r") + po * d.H;
            a.sv_u = W(h, "ioc_sv_u") + po * d.H; a.sv_c = W(h, "ioc_sv_c") + po * d.H;
            a.sv_h = W(h, "ioc_sv_h") + po * d.H;
            if (cluster && p > 0) HIPCHK(hipMemsetAsync(h->ws["grp_cnt"].p, 0, ((size_t)v.R / v.mno) * sizeof(int), s));
            if (x3) {       // split-bf16 operands; the saves are fp32 and the backward pass is the fp32 one
                a.Wsoc = D4(h, "ioc/Wsoc16"); a.Wg = D4(h, "ioc/Wg16"); a.Wc = D4(h, "ioc/Wc16"); a.Wreg = D4(h, "ioc/Wreg16");
                Timer t(h, s, "ioc"); launch_ioc_x3(a, s);
            } else
            { Timer t(h, s, "ioc"); launch_ioc(a, s); }
        }
    } else
    if (x3 || x6) {   // split-bf16 operands: fp32-equivalent results on the bf16 matrix pipe (shapes without that form run the fp32 kernels)
        a.Wsoc = D4(h, "ioc/Wsoc16"); a.Wg = D4(h, "ioc/Wg16"); a.Wc = D4(h, "ioc/Wc16"); a.Wreg = D4(h, "ioc/Wreg16");
        Timer t(h, s, "ioc");
        if (x6) launch_ioc_x6(a, s); else launch_ioc_x3(a, s);
    } else
    if (d.bf16 == 1) {
        if (h->training) return fail(DESIRE_ERR_STATE, "bf16 operands are inference-only");
        a.Wsoc = D4(h, "ioc/Wsoc16"); a.Wg = D4(h, "ioc/Wg16"); a.Wc = D4(h, "ioc/Wc16"); a.Wreg = D4(h, "ioc/Wreg16");
        Timer t(h, s, "ioc");
        if (cluster) { if (launch_ioc_bf16_cluster(a, s)) return fail(DESIRE_ERR_HIP, "bf16 cluster IOC: no resident grid for this shape"); }
        else launch_ioc_bf16(a, s);
    } else
    { Timer t(h, s, "ioc"); launch_ioc(a, s); }
#ifdef DESIRE_IOC_TIMING
    {
        long long host[10];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host, a.dbg, sizeof(host), hipMemcpyDeviceToHost);
        const char* n32[10] = {"P0 pos+clear", "P1 ev/es/masks", "build0+bar", "build(b+1)", "mma bin", "bin barrier",
                               "P3 e_r+bar", "P4 gates(2 mma)+ep+2bar", "P5 cand+ep+2bar", ""};
        const char* n16[10] = {"loop top", "P1 ev/es/masks", "barrier 1", "P2 pooling chain + e_r", "barrier 2", "P4 gates + r*h",
                               "barrier 3", "P5 cand + publish", "barrier 4", ""};
        const char* nx3[10] = {"step top (bar 4 wait)", "P1 ev/es/masks", "barrier 1", "P2 pooling chain", "exchange + e_r", "barrier 2",
                               "P4 gates + r*h", "barrier 3", "P5 cand + publish", "barrier 4"};
        const char* ncl[10] = {"step top: positions + clear + bar", "P1 ev/es/masks", "wait for the peers", "copy peers' Ht + bar", "P2 pooling chains",
                               "exchange + e_r", "barrier 2", "P4 gates + r*h + cand frags", "bar 3 + P5 cand + publish stores", "drain + arrive + bar"};
        const char** names = (x3 || x6) ? nx3 : d.bf16 == 1 ? (cluster ? ncl : n16) : n32;
        const int nk = (x3 || x6 || (d.bf16 == 1 && cluster)) ? 10 : 9;
        long long tot = 0; for (int k = 0; k < nk; ++k) tot += host[k];
        for (int k = 0; k < nk; ++k) fprintf(stderr, "[ioc timing] %-26s %12lld cyc  %5.1f%%\n", names[k], host[k], 100.0 * host[k] / (double)tot);
    }
#endif
    HIPCHK(hipGetLastError());
    if (cluster) {
        int err = 0;
        HIPCHK(hipMemcpyAsync(&err, h->ws["ioc_err"].p, sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (err) return fail(DESIRE_ERR_HIP, "IOC cluster hand-off timed out (workgroups of a group were not co-resident)");
    }
    return DESIRE_OK;
}

extern "C" int desire_ioc_refine(desire_handle* h, float* dev_Yhat, float* dev_score, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!dev_Yhat || !dev_score) return fail(DESIRE_ERR_ARG, "null argument");
    if (h->d.ref_compat) return fail(DESIRE_ERR_STATE, "ref_compat: the reference graph has no ranking/refinement module (model/model.py:312-313)");
    if (!h->grids_set) return fail(DESIRE_ERR_STATE, "scene grids not set (desire_set_scene_grids)");
    const desire_dims& d = h->d;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // step-wise form (one launch of the agent-sharded kernel per step, a single rank): scenes of 160 .. 256 agents (beyond the cluster
    // form's 128-bit neighbour masks) on any operands but plain bf16, and -- dims.bf16 = 2 / 3, inference -- H = 256 (BASELINE configs[3]:
    // no persistent split kernel: the bin-split accumulators do not fit eight waves' registers) with split operands instead of the fp32
    // fallback: 16.1 -> 9.1 ms (three products) / 12.7 ms (six) at configs[3]'s per-GPU shape.  (Groups of 96 / 128 agents at H <= 128
    // were measured too: 34.4 vs 34.9 ms with three products, SLOWER with six -- they keep the fp32 cluster kernel.)
    const int B_ = d.grid_size * d.grid_size;
    const bool split_mode = (d.bf16 == 2 || d.bf16 == 3) && !h->training;
    const bool split_served = ioc_x3_supported(d.mno, d.H, B_) || (d.mno == 64 && ioc_x6r2_supported(d.mno, d.H, B_));
    const bool stepwise = d.mno > 128 || (split_mode && !split_served && d.H == 256 && d.ioc_form == DESIRE_IOC_AUTO);
    if (stepwise) {
        if (h->training) return fail(DESIRE_ERR_STATE, "training supports up to 128 agents per scene");
        const size_t RH = (size_t)h->R * d.H;
        if ((!h->ws.count("stw_h") || !h->ws["stw_h"].p || !h->ws["stw_sc"].p) &&
            ((!h->ws["stw_h"].p && h->ws["stw_h"].alloc(2 * RH * sizeof(float))) || (!h->ws["stw_sc"].p && h->ws["stw_sc"].alloc((size_t)h->R * sizeof(float)))))
            return fail(DESIRE_ERR_HIP, "hipMalloc failed for the step-wise IOC state");
        float* hb[2] = {W(h, "stw_h"), W(h, "stw_h") + RH};
        const int NTs = d.H / 32, KXs = d.E_v + d.C + 2 * d.H;
        for (int it = 0; it < d.iters; ++it) {
            launch_hx_rows(hb[1], W(h, "HxHy"), 2 * d.H, d.n_scenes, d.K, d.mno, d.H, s);       // h_{-1} = Hx of the row's agent
            Timer tm(h, s, "ioc");                                                            // (one profile entry per pass, as for the persistent kernels)
            for (int t = 0; t < d.T_pred; ++t) {
                IocStepArgs q{};
                q.t = t; q.rank = 0; q.nranks = 1; q.m_loc = d.mno; q.n_scenes = d.n_scenes; q.K = d.K; q.R = h->R;
                q.H = d.H; q.T = d.T_pred; q.Gh = d.Gh; q.Gw = d.Gw; q.G = d.grid_size; q.nb_w = d.nb_w; q.nb_h = d.nb_h;
                q.Yall = dev_Yhat; q.plast_all = W(h, "p_last"); q.valid_all = static_cast<const uint8_t*>(h->ws["valid"].p); q.Hall = hb[(t + 1) & 1];
                q.st_h = hb[(t + 1) & 1]; q.st_h_out = hb[t & 1]; q.st_score = W(h, "stw_sc");
                q.grids = h->grids; q.grid_of_scene = static_cast<const int32_t*>(h->ws["grid_of_scene"].p);
                q.w_vel = D(h, "ioc/vel_w"); q.b_vel = D(h, "ioc/vel_b"); q.Wsoc = D4(h, "ioc/Wsoc"); q.b_soc = D(h, "ioc/soc_b");
                q.Wg = D4(h, "ioc/Wg"); q.Wc = D4(h, "ioc/Wc"); q.b_g = D(h, "ioc/gb"); q.b_c = D(h, "ioc/cb"); q.w_score = D(h, "ioc/score_w");
                q.bin_tab = d.bin_mode == 1 ? W(h, "bin_tab") : nullptr;
                if (split_mode) {
                    q.np = d.bf16 == 3 ? 3 : 2;
                    q.Wsoc = D4(h, "ioc/Wsoc16l"); q.Wg = D4(h, "ioc/Wg16"); q.Wc = D4(h, "ioc/Wc16");
                    q.plo_soc = (size_t)B_ * NTs * (d.H / 16) * 64; q.plo_g = (size_t)2 * NTs * (KXs / 16) * 64; q.plo_c = (size_t)NTs * (KXs / 16) * 64;
                }
                launch_ioc_step(q, s);
            }
            if (int rc = desire_ioc_finish(h, hb[(d.T_pred - 1) & 1], W(h, "stw_sc"), dev_Yhat, dev_score, stream)) return rc;
        }
        HIPCHK(hipGetLastError());
        return DESIRE_OK;
    }
    if (compact_ioc(h)) {
        // DESIRE_FLAG_COMPACT_IOC: one launch sequence per slot class over the windows seated in it; windows without a present agent are not run
        // (their rows keep the Y they came with and score 0)
        if (int rc = compact_wait(h, s)) return rc;
        const bool dyn = compact_dyn(h);
        const int32_t* cnt_dev = static_cast<const int32_t*>(h->ws["cp_count"].p) + 4;
        int m4[4];
        const int n_cls = compact_classes(h, m4);
        const int32_t* cnt = h->cp_host + 4;
        size_t aoff = 0, roff = 0, woff = 0;
        const size_t T2 = (size_t)d.T_pred * 2;
        launch_fill_f32(dev_score, (size_t)h->R, 0.f, s);
        h->ci_n = 0;
        for (int c = 0; c < n_cls; ++c) {
            // device-side counts: every class is launched for the worst case (all windows in it) at a static offset; an empty class's grids exit
            const int n_c = dyn ? d.n_scenes : static_cast<volatile const int32_t*>(cnt)[c], m_c = m4[c];
            const int32_t* dynN = dyn ? cnt_dev + c : nullptr;
            if (n_c < 0 || n_c > d.n_scenes) return fail(DESIRE_ERR_HIP, "slot-class scan returned a count out of range");
            if (n_c == 0) continue;
            const int32_t* cmap = static_cast<const int32_t*>(h->ws["ci_map"].p) + (size_t)c * h->A;
            const int32_t* win = static_cast<const int32_t*>(h->ws["ci_win"].p) + (size_t)c * d.n_scenes;
            const int gpt = (m_c <= 32 && 32 % m_c) ? 32 / m_c : 0, ngrp = n_c * d.K;              // padded tiles for a class that does not divide 32
            const int R_c = gpt ? ((ngrp + gpt - 1) / gpt) * 32 : n_c * d.K * m_c;
            IocView v{R_c, m_c, n_c, W(h, "ci_Y") + roff * T2, W(h, "ci_score") + roff, W(h, "ci_Hx") + aoff * 2 * d.H, 2 * d.H, W(h, "ci_pl") + aoff * 2,
                      static_cast<const uint8_t*>(h->ws["ci_valid"].p) + aoff, static_cast<const int32_t*>(h->ws["ci_gos"].p) + woff, roff};
            v.gpt = gpt; v.ngrp = ngrp; v.dynN = dynN;
            {
                Timer t(h, s, "ioc_repack");
                launch_cls_gather_agents(W(h, "HxHy"), 2 * d.H, W(h, "p_last"), static_cast<const int32_t*>(h->ws["grid_of_scene"].p), cmap, win, n_c, m_c,
                                         const_cast<float*>(v.Hx), const_cast<float*>(v.p_last), const_cast<uint8_t*>(v.valid), const_cast<int32_t*>(v.gos), s, dynN);
                launch_cls_rows(dev_Yhat, v.Y, cmap, n_c, m_c, d.K, d.mno, (int)T2, 0, s, gpt, dynN);
            }
            if (int rc = ioc_core(h, v, s)) return rc;
            {
                Timer t(h, s, "ioc_repack");
                launch_cls_rows(dev_Yhat, v.Y, cmap, n_c, m_c, d.K, d.mno, (int)T2, 1, s, gpt, dynN);
                launch_cls_rows(dev_score, v.score, cmap, n_c, m_c, d.K, d.mno, 1, 1, s, gpt, dynN);
            }
            h->ci_cls[h->ci_n] = c; h->ci_cnt[h->ci_n] = n_c; ++h->ci_n;
            aoff += (size_t)n_c * m_c; roff += (size_t)R_c; woff += (size_t)n_c;
        }
        h->ci_last = true;
        if (h->training && d.bf16 != 1) {
            launch_copy_f32(W(h, "Y_ref"), dev_Yhat, (size_t)h->R * d.T_pred * 2, s);
            launch_copy_f32(W(h, "score_sv"), dev_score, (size_t)h->R, s);
        }
        HIPCHK(hipGetLastError());
        return DESIRE_OK;
    }
    h->ci_last = false;
    IocView full{h->R, d.mno, d.n_scenes, dev_Yhat, dev_score, W(h, "HxHy"), 2 * d.H, W(h, "p_last"), static_cast<const uint8_t*>(h->ws["valid"].p),
                 static_cast<const int32_t*>(h->ws["grid_of_scene"].p), 0};
    if (int rc = ioc_core(h, full, s)) return rc;
    if (h->training && d.bf16 != 1) {
        launch_copy_f32(W(h, "Y_ref"), dev_Yhat, (size_t)h->R * d.T_pred * 2, s);
        launch_copy_f32(W(h, "score_sv"), dev_score, (size_t)h->R, s);
    }
    return DESIRE_OK;
}

extern "C" int desire_forward(desire_handle* h, const float* dev_past, const float* dev_fut, const float* dev_eps,
                              float* dev_Yhat, float* dev_score, void* stream) {
    if (int rc = desire_encode(h, dev_past, dev_fut, stream)) return rc;
    if (int rc = desire_sample(h, dev_eps, dev_Yhat, stream)) return rc;
    if (h && h->d.ref_compat) return DESIRE_OK;          // the reference graph ends at the decoder states (dev_score untouched)
    return desire_ioc_refine(h, dev_Yhat, dev_score, stream);
}

