// api.hip -- C ABI of libdesire_hip.so (include/desire_hip.h): handle, weight repacking into MFMA
// B-fragment order, workspace, and the launch sequence of the hot path.  Host code only.
#include "ctx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_err;
int desire_fail(int code, const std::string& msg) { g_err = msg; return code; }

extern "C" const char* desire_last_error(void) { return g_err.c_str(); }
extern "C" int desire_version(void) { return 5; }
extern "C" int desire_dims_size(void) { return (int)sizeof(desire_dims); }
#ifndef DESIRE_SRC_HASH
#define DESIRE_SRC_HASH "unstamped"
#endif
extern "C" const char* desire_build_hash(void) { return DESIRE_SRC_HASH; }

// Packed fragment order: out[((nt*G + g)*64 + lane)*4 + i] = W(k = 8g + 4*(lane>>5) + i, n = nt*32 + (lane&31))
std::vector<float> pack_b(int K, int N, const std::function<float(int, int)>& at) {
    const int G = (K + 7) / 8, NT = (N + 31) / 32;
    std::vector<float> out((size_t)NT * G * 64 * 4, 0.f);
    for (int nt = 0; nt < NT; ++nt)
        for (int g = 0; g < G; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    const int k = 8 * g + 4 * (lane >> 5) + i, n = nt * 32 + (lane & 31);
                    if (k < K && n < N) out[(((size_t)nt * G + g) * 64 + lane) * 4 + i] = at(k, n);
                }
    return out;
}

std::vector<float> pack_vals16(int K, int N, const std::function<int(int, int, int)>& kmap, const std::function<float(int, int)>& at) {
    const int G = (K + 15) / 16, NT = (N + 31) / 32;
    std::vector<float> o((size_t)NT * G * 64 * 8, 0.f);
    for (int nt = 0; nt < NT; ++nt)
        for (int g = 0; g < G; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int k = kmap(g, lane >> 5, e), n = nt * 32 + (lane & 31);
                    if (k >= 0 && k < K && n < N) o[(((size_t)nt * G + g) * 64 + lane) * 8 + e] = at(k, n);
                }
    return o;
}
std::vector<float> pack_b16(int K, int N, const std::function<int(int, int, int)>& kmap, const std::function<float(int, int)>& at) {
    const int G = (K + 15) / 16, NT = (N + 31) / 32;
    std::vector<uint16_t> o((size_t)NT * G * 64 * 8, 0);
    for (int nt = 0; nt < NT; ++nt)
        for (int g = 0; g < G; ++g)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int k = kmap(g, lane >> 5, e), n = nt * 32 + (lane & 31);
                    if (k >= 0 && k < K && n < N) o[(((size_t)nt * G + g) * 64 + lane) * 8 + e] = bf16_rne(at(k, n));
                }
    std::vector<float> out(o.size() / 2);
    std::memcpy(out.data(), o.data(), o.size() * 2);
    return out;
}

int desire_upload(desire_ctx* h, const std::string& name, const std::vector<float>& v) {
    if (h->pack_mode == 1) { h->captured[name] = v; return 0; }
    DevBuf& b = h->dev[name];
    if (b.p && b.bytes == v.size() * sizeof(float))          // same shape: refresh in place (pointers stay valid)
        return hipMemcpy(b.p, v.data(), b.bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
    b.release();
    if (b.alloc(v.size() * sizeof(float))) return -1;
    return hipMemcpy(b.p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}

static void embed_walk(const Embed& em, const std::function<void(size_t, size_t)>& f) {
    size_t cl = 0, cp = 0;
    for (auto& c : em.cols.seg) { cl += c.first; cp += c.second; }
    size_t rl = 0, rp = 0;
    for (auto& r : em.rows.seg) {
        for (int i = 0; i < r.first; ++i) {
            size_t ol = 0, op = 0;
            for (auto& c : em.cols.seg) {
                for (int j = 0; j < c.first; ++j) f((rl + i) * cl + ol + j, (rp + i) * cp + op + j);
                ol += c.first; op += c.second;
            }
        }
        rl += r.first; rp += r.second;
    }
}
std::vector<float> desire_embed(const desire_ctx* h, const std::string& name, const float* user) {
    auto it = h->emb.find(name);
    const size_t np = h->want.at(name);
    if (it == h->emb.end()) return std::vector<float>(user, user + np);
    std::vector<float> out(np, 0.f);
    embed_walk(it->second, [&](size_t il, size_t ip) { out[ip] = user[il]; });
    return out;
}
void desire_extract(const desire_ctx* h, const std::string& name, const float* phys, float* user) {
    auto it = h->emb.find(name);
    if (it == h->emb.end()) { std::memcpy(user, phys, h->want.at(name) * sizeof(float)); return; }
    embed_walk(it->second, [&](size_t il, size_t ip) { user[il] = phys[ip]; });
}

namespace {

void shapes(desire_ctx* h, int H, std::map<std::string, size_t>& s) {
    const desire_dims& d = h->d;
    const int L = d.L, V = h->V;
    const int E = d.E_v + d.C + H;
    auto gru = [&](const std::string& p, int n_in) {
        s[p + "/gates/kernel"] = (size_t)(n_in + H) * 2 * H;
        s[p + "/gates/bias"] = 2 * H;
        s[p + "/candidate/kernel"] = (size_t)(n_in + H) * H;
        s[p + "/candidate/bias"] = H;
    };
    auto bn = [&](const std::string& p, int c) {
        for (const char* n : {"beta", "gamma", "moving_mean", "moving_var"}) s[p + "/bn/" + n] = c;
    };
    gru("enc_x", 2); gru("enc_y", 2);
    s["fc_c/w"] = (size_t)2 * H * V; s["fc_c/b"] = V;
    struct CL { const char* n; int k, ci, co; };
    for (CL c : {CL{"conv1", 5, 1, 32}, CL{"conv2", 5, 32, 64}, CL{"conv3", 5, 64, 128}}) {
        const std::string p = std::string("vae_enc/") + c.n;
        s[p + "/w"] = (size_t)c.k * c.k * c.ci * c.co; s[p + "/b"] = c.co; bn(p, c.co);
    }
    s["vae_enc/fc/w"] = (size_t)2048 * 2 * L; s["vae_enc/fc/b"] = 2 * L;
    for (CL c : {CL{"deconv1", 4, L, 128}, CL{"deconv2", 5, 128, 64}, CL{"deconv3", 5, 64, 32}, CL{"deconv4", 5, 32, 1}}) {
        const std::string p = std::string("vae_dec/") + c.n;
        s[p + "/w"] = (size_t)c.k * c.k * c.ci * c.co; s[p + "/b"] = c.co; bn(p, c.co);
    }
    s["mask_fc/w"] = (size_t)V * H; s["mask_fc/b"] = H;
    gru("dec", H);
    s["head/w"] = 2 * H; s["head/b"] = 2;
    s["ioc/vel_fc/w"] = 2 * d.E_v; s["ioc/vel_fc/b"] = d.E_v;
    s["ioc/social_fc/w"] = (size_t)h->B * H * H; s["ioc/social_fc/b"] = H;
    gru("ioc", E);
    s["ioc/score/w"] = H; s["ioc/score/b"] = 1;
    s["ioc/reg/w"] = (size_t)H * 2 * d.T_pred; s["ioc/reg/b"] = 2 * d.T_pred;
    s["scene_cnn/conv1/w"] = 25 * 3 * 16; s["scene_cnn/conv1/b"] = 16;
    s["scene_cnn/conv2/w"] = 25 * 16 * 32; s["scene_cnn/conv2/b"] = 32;
    s["scene_cnn/conv3/w"] = (size_t)25 * 32 * d.C; s["scene_cnn/conv3/b"] = d.C;
    s["temporal/w"] = (size_t)d.T_obs * 2 * 100; s["temporal/b"] = 200;
    s["gauss_head/w"] = (size_t)H * 5; s["gauss_head/b"] = 5;       // sample()'s 5-wide output layer (model/model.py:315-321,445-449)
}

// logical -> physical embedding of every weight that has a hidden-width axis (ctx.h: Embed)
void embeddings(desire_ctx* h) {
    const desire_dims& d = h->d;
    const std::pair<int, int> Hs{h->Hl, d.H};
    auto fix = [](int n) { return std::pair<int, int>{n, n}; };
    auto& e = h->emb;
    auto gru = [&](const std::string& p, std::vector<std::pair<int, int>> in) {
        in.push_back(Hs);
        e[p + "/gates/kernel"] = Embed{{in}, {{Hs, Hs}}};
        e[p + "/gates/bias"] = Embed{{{fix(1)}}, {{Hs, Hs}}};
        e[p + "/candidate/kernel"] = Embed{{in}, {{Hs}}};
        e[p + "/candidate/bias"] = Embed{{{fix(1)}}, {{Hs}}};
    };
    gru("enc_x", {fix(2)}); gru("enc_y", {fix(2)}); gru("dec", {Hs}); gru("ioc", {fix(d.E_v + d.C), Hs});
    e["fc_c/w"] = Embed{{{Hs, Hs}}, {{fix(h->V)}}};
    e["mask_fc/w"] = Embed{{{fix(h->V)}}, {{Hs}}};
    e["mask_fc/b"] = Embed{{{fix(1)}}, {{Hs}}};
    e["head/w"] = Embed{{{Hs}}, {{fix(2)}}};
    e["ioc/social_fc/w"] = Embed{{std::vector<std::pair<int, int>>(h->B, Hs)}, {{Hs}}};
    e["ioc/social_fc/b"] = Embed{{{fix(1)}}, {{Hs}}};
    e["ioc/score/w"] = Embed{{{Hs}}, {{fix(1)}}};
    e["ioc/reg/w"] = Embed{{{Hs}}, {{fix(2 * d.T_pred)}}};
    e["gauss_head/w"] = Embed{{{Hs}}, {{fix(5)}}};
}

// frozen batch-norm + bias -> (scale, shift); float64 then one rounding (desire_amd/spec.py:fold_bn)
void fold_bn(desire_ctx* h, const std::string& p, std::vector<float>& scale, std::vector<float>& shift) {
    const auto& g = h->host_w.at(p + "/bn/gamma"); const auto& be = h->host_w.at(p + "/bn/beta");
    const auto& mu = h->host_w.at(p + "/bn/moving_mean"); const auto& var = h->host_w.at(p + "/bn/moving_var");
    const auto& b = h->host_w.at(p + "/b");
    scale.resize(g.size()); shift.resize(g.size());
    for (size_t i = 0; i < g.size(); ++i) {
        const double sc = (double)g[i] / std::sqrt((double)var[i] + 1e-3);
        scale[i] = (float)sc;
        shift[i] = (float)((double)be[i] + sc * ((double)b[i] - (double)mu[i]));
    }
}

int check_options(const desire_dims& d) {
    switch (d.ioc_form) {
        case DESIRE_IOC_AUTO: case DESIRE_IOC_TILE64: case DESIRE_IOC_CLUSTER: case DESIRE_IOC_CLUSTER_BINS: case DESIRE_IOC_COMPACT:
        case DESIRE_IOC_TRAIN_DENSE: case DESIRE_IOC_X6_TILE32: case DESIRE_IOC_X6_TILE64: break;
        default: return fail(DESIRE_ERR_ARG, "ioc_form must be one of DESIRE_IOC_* (include/desire_hip.h)");
    }
    if (d.ioc_split < 0 || d.ioc_split > 4) return fail(DESIRE_ERR_ARG, "ioc_split must be 0 (auto), 1 (never split: batch-size invariant results) or 2..4 (cap)");
    if (d.train_fp32_mask < 0 || d.train_fp32_mask > 15) return fail(DESIRE_ERR_ARG, "train_fp32_mask is a mask of bits 1, 2, 4, 8");
    if (d.flags & ~(DESIRE_FLAG_NO_FUSE34 | DESIRE_FLAG_TRAIN_FWD_3P | DESIRE_FLAG_COMPACT_ROWS | DESIRE_FLAG_COMPACT_IOC)) return fail(DESIRE_ERR_ARG, "unknown bit in flags (DESIRE_FLAG_*)");
    if ((d.flags & DESIRE_FLAG_COMPACT_IOC) && d.ref_compat) return fail(DESIRE_ERR_ARG, "DESIRE_FLAG_COMPACT_IOC: ref_compat has no IOC stage");
    if ((d.flags & DESIRE_FLAG_COMPACT_ROWS) && (d.bn_mode == 2 || d.ref_compat))
        return fail(DESIRE_ERR_ARG, "DESIRE_FLAG_COMPACT_ROWS: not with bn_mode = 2 (whole-batch statistics depend on the padding rows) or ref_compat");
    return 0;
}

int check_dims(const desire_dims& d) {
    if (d.S != 32) return fail(DESIRE_ERR_ARG, "S must be 32 (rnn_size=512): CVAE stack shapes, model/model.py:465-468");
    if (d.mno < 1 || d.mno > 256 || (d.mno <= 32 ? (32 % d.mno) : (d.mno % 32)))
        return fail(DESIRE_ERR_ARG, "mno must divide 32 or be a multiple of 32 up to 256 (above 128: inference, step-wise IOC)");
    if (d.mno > 128 && d.bf16 == 1) return fail(DESIRE_ERR_ARG, "more than 128 agents per scene run the fp32 step-wise IOC (bf16 = 0, 2 or 3)");
    if (d.H != 16 && d.H != 32 && d.H != 64 && d.H != 128 && d.H != 256)
        return fail(DESIRE_ERR_ARG, "H must be 16, 32 (run zero-padded on the 64-wide tile), 64, 128 or 256");

    if (d.L % 8 || d.L < 8) return fail(DESIRE_ERR_ARG, "L must be a positive multiple of 8");
    if (d.C != 32 || d.E_v != 16) return fail(DESIRE_ERR_ARG, "C=32 and E_v=16 are the instantiated IOC widths in this round");
    if (d.n_scenes < 1 || d.K < 1 || d.T_obs < 1 || d.T_pred < 1 || d.n_grids < 1 || d.iters < 1 || d.Gh < 1 || d.Gw < 1)
        return fail(DESIRE_ERR_ARG, "sizes must be >= 1");
    if (d.grid_size < 1 || d.grid_size > 6) return fail(DESIRE_ERR_ARG, "grid_size must be 1..6 (6 x 6 = the paper's 36 bins)");
    if (d.grid_size > 4 && d.H == 256) return fail(DESIRE_ERR_ARG, "grid_size 5..6 needs H <= 128 (LDS budget of the IOC tile)");
    if (d.bf16 < 0 || d.bf16 > 3)
        return fail(DESIRE_ERR_ARG, "bf16 must be 0 (fp32 operands), 1 (bf16 operands), 2 (split-bf16 operands: hi + lo, three products) or 3 (three "
                                    "bf16 pieces, six products: fp32-class accuracy)");
    if (d.bn_mode < 0 || d.bn_mode > 2) return fail(DESIRE_ERR_ARG, "bn_mode must be 0 (frozen statistics), 1 (per-object statistics) or 2 (whole-batch statistics)");
    if (d.bn_mode && d.bf16 == 1) return fail(DESIRE_ERR_ARG, "batch statistics (bn_mode 1 / 2) run on fp32 operands (bf16 = 0 or 2)");
    if (d.bin_mode != 0 && d.bin_mode != 1) return fail(DESIRE_ERR_ARG, "bin_mode must be 0 (rectangular) or 1 (log-polar)");
    if (d.bin_mode == 1 && (d.grid_size < 3 || !(d.nb_h > 0.f) || !(d.nb_w > d.nb_h)))
        return fail(DESIRE_ERR_ARG, "log-polar bins: grid_size >= 3 and 0 < nb_h (inner radius) < nb_w (outer radius)");
    if (!(d.nb_w > 0.f) || !(d.nb_h > 0.f)) return fail(DESIRE_ERR_ARG, "nb_w/nb_h must be > 0");
    if (d.ref_compat != 0 && d.ref_compat != 1) return fail(DESIRE_ERR_ARG, "ref_compat must be 0 or 1");
    if (int rc = check_options(d)) return rc;
    if (d.ref_compat) {
        if (d.K != 1 || !d.posterior || d.bn_mode != 1 || d.bf16 || d.n_dec < 1 || d.H != 2 * d.T_obs || d.T_pred != d.T_obs)
            return fail(DESIRE_ERR_ARG, "ref_compat (the reference graph as written, model/model.py:116-311) needs K = 1 (one eps per object, "
                                        ":262-263), posterior = 1, bn_mode = 1, bf16 = 0, H == 2*T_obs (:286-289), T_pred == T_obs and n_dec >= 1 (:280 runs 7)");
    } else if (d.n_dec != 0) return fail(DESIRE_ERR_ARG, "n_dec belongs to ref_compat (0 otherwise)");
    return 0;
}

}  // namespace

extern "C" int desire_set_option(desire_handle* h, const char* name, int32_t value) {
    if (!h || !name) return fail(DESIRE_ERR_ARG, "null argument");
    desire_dims d = h->d;
    const std::string nm(name);
    if (nm == "ioc_form") d.ioc_form = value;
    else if (nm == "ioc_split") d.ioc_split = value;
    else if (nm == "train_fp32_mask") d.train_fp32_mask = value;
    else if (nm == "flags") d.flags = value;
    else if (nm == "compact_min_rows") {       // DESIRE_FLAG_COMPACT_IOC: a slot class with fewer rows than this is folded into the next larger one (default 8192)
        if (value < 0) return fail(DESIRE_ERR_ARG, "compact_min_rows must be >= 0");
        h->ci_min_rows = value;
        return DESIRE_OK;
    }
    else return fail(DESIRE_ERR_ARG, "unknown option: " + nm + " (ioc_form, ioc_split, train_fp32_mask, flags, compact_min_rows)");
    if (int rc = check_options(d)) return rc;
    h->d = d;
    return DESIRE_OK;
}

extern "C" int desire_create(const desire_dims* dims, desire_handle** out) {
    if (!dims || !out) return fail(DESIRE_ERR_ARG, "null argument");
    if (int rc = check_dims(*dims)) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(DESIRE_ERR_NODEV, "no HIP device: libdesire_hip has no CPU path");
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(DESIRE_ERR_NODEV, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    desire_ctx* h = new desire_ctx();
    h->d = *dims;
    h->Hl = dims->H;
    if (dims->H < 64) h->d.H = 64;          // narrowest instantiated recurrent tile; the padding is exact (ctx.h: EmbedAxis)
    h->A = dims->n_scenes * dims->mno;
    h->R = h->A * dims->K;
    h->V = dims->S * dims->S;
    h->B = dims->grid_size * dims->grid_size;
    h->E = dims->E_v + dims->C + h->d.H;
    shapes(h, h->d.H, h->want);
    shapes(h, h->Hl, h->want_user);
    if (h->Hl != h->d.H) embeddings(h);
    const desire_dims& d = h->d;
    const size_t A = h->A, R = h->R, f = sizeof(float);
    struct WS { const char* n; size_t bytes; };
    const WS list[] = {
        {"HxHy", A * 2 * d.H * f}, {"p_last", A * 2 * f}, {"valid", A}, {"lmask", A}, {"nfut", A * f}, {"vae_in", A * h->V * f},
        {"c1", A * 8192 * f}, {"c2", A * 4096 * f}, {"c3", A * 2048 * f}, {"params", A * 2 * d.L * f},
        {"z", R * d.L * f}, {"d1", R * 2048 * f}, {"d2", R * 4096 * f}, {"d3", R * 8192 * f},
        {"xhat", R * 1024 * f}, {"xz", R * d.H * f}, {"Y0", R * (size_t)(d.n_dec > d.T_pred ? d.n_dec : d.T_pred) * 2 * f},
        {"dec_states", d.ref_compat ? R * (size_t)d.n_dec * d.H * f : 0},
        {"bn_part", d.bn_mode == 2 ? (size_t)512 * 128 * f : 0}, {"bn_stat", d.bn_mode == 2 ? (size_t)2 * 128 * f : 0},
        {"grid_of_scene", (size_t)d.n_scenes * sizeof(int32_t)},
        // desire_build_windows*: allocated here, not lazily, because a feeder thread may call the builder while the owner thread runs a forward on
        // the same handle (desire_amd/prefetch.py: DeviceWindowFeeder) -- the builder then touches these two buffers and nothing else of the handle
        {"bw_starts", (size_t)d.n_scenes * sizeof(int32_t)}, {"bw_err", sizeof(int32_t)},
    };
    for (const WS& w : list) {
        if (h->ws[w.n].alloc(w.bytes)) { desire_destroy(h); return fail(DESIRE_ERR_HIP, std::string("hipMalloc failed for ") + w.n); }
        (void)hipMemset(h->ws[w.n].p, 0, w.bytes);
    }
    if (d.bin_mode == 1) {
        // log-polar social bins: G rings with geometric radii between r_min = nb_h and r_max = nb_w, G equal sectors
        std::vector<float> tab(20, 0.f);
        const int G = d.grid_size;
        for (int k = 0; k < G; ++k) {
            const float t = (float)((double)d.nb_h * std::pow((double)d.nb_w / (double)d.nb_h, (double)(k + 1) / G));
            tab[k] = t * t;
            tab[8 + 2 * k] = (float)std::cos(2.0 * M_PI * k / G);
            tab[9 + 2 * k] = (float)std::sin(2.0 * M_PI * k / G);
        }
        h->bin_tab_host = tab;
        if (h->ws["bin_tab"].alloc(20 * f)) { desire_destroy(h); return fail(DESIRE_ERR_HIP, "hipMalloc failed for bin_tab"); }
        if (hipMemcpy(h->ws["bin_tab"].p, tab.data(), 20 * f, hipMemcpyHostToDevice) != hipSuccess) { desire_destroy(h); return fail(DESIRE_ERR_HIP, "bin table upload failed"); }
    }
    *out = h;
    return DESIRE_OK;
}

extern "C" int desire_get_bin_table(desire_handle* h, float* host_out20) {
    if (!h || !host_out20) return fail(DESIRE_ERR_ARG, "null argument");
    if (h->d.bin_mode != 1) return fail(DESIRE_ERR_STATE, "the rectangular grid has no table (dims.bin_mode = 0)");
    std::memcpy(host_out20, h->bin_tab_host.data(), 20 * sizeof(float));
    return DESIRE_OK;
}

extern "C" int desire_peer_close(desire_handle* h);
extern "C" int desire_destroy(desire_handle* h) {
    if (!h) return DESIRE_OK;
    if (h->host_err) { (void)hipHostFree(h->host_err); h->host_err = nullptr; }
    if (h->cp_host) { (void)hipHostFree(h->cp_host); h->cp_host = nullptr; }
    if (h->cp_ev) { (void)hipEventDestroy(h->cp_ev); h->cp_ev = nullptr; }
    (void)desire_peer_close(h);
    for (auto& kv : h->dev) kv.second.release();
    for (auto& kv : h->ws) kv.second.release();
    for (auto& p : h->prof) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (void* g : h->graphs) if (g) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(g));
    delete h;
    return DESIRE_OK;
}

extern "C" int desire_set_weight(desire_handle* h, const char* name, const float* host_data, size_t n) {
    if (!h || !name || !host_data) return fail(DESIRE_ERR_ARG, "null argument");
    auto it = h->want_user.find(name);
    if (it == h->want_user.end()) return fail(DESIRE_ERR_ARG, std::string("unknown weight: ") + name);
    if (it->second != n)
        return fail(DESIRE_ERR_ARG, std::string("weight ") + name + ": expected " + std::to_string(it->second) +
                                        " values, got " + std::to_string(n));
    h->host_w[name] = desire_embed(h, name, host_data);
    h->finalized = false;
    h->training = false;            // the optimiser's master copy is rebuilt by the next desire_set_training(h, 1)
    return DESIRE_OK;
}

extern "C" int desire_finalize_weights(desire_handle* h) {
    if (!h) return fail(DESIRE_ERR_ARG, "null handle");
    for (auto& kv : h->want)
        if (!h->host_w.count(kv.first)) return fail(DESIRE_ERR_STATE, "weight not set: " + kv.first);
    if (int rc = desire_pack_all(h)) return rc;
    h->finalized = true;
    return DESIRE_OK;
}

int desire_pack_all(desire_ctx* h) {
    const desire_dims& d = h->d;
    const int H = d.H, L = d.L, V = h->V, E = h->E, B = h->B;
    auto& hw = h->host_w;
    auto up = [&](const std::string& n, const std::vector<float>& v) { return desire_upload(h, n, v); };
    auto rowmajor = [](const std::vector<float>& w, int ldw, int k0) {
        return [&w, ldw, k0](int k, int n) { return w[(size_t)(k0 + k) * ldw + n]; };
    };
    int bad = 0;
    // GRUs: raw kernels/biases + packed sub-blocks
    for (const char* p : {"enc_x", "enc_y"}) {
        const std::string s(p);
        bad |= up(s + "/gk", hw[s + "/gates/kernel"]);   bad |= up(s + "/gb", hw[s + "/gates/bias"]);
        bad |= up(s + "/ck", hw[s + "/candidate/kernel"]); bad |= up(s + "/cb", hw[s + "/candidate/bias"]);
        bad |= up(s + "/Whg", pack_b(H, 2 * H, rowmajor(hw[s + "/gates/kernel"], 2 * H, 2)));
        bad |= up(s + "/Whc", pack_b(H, H, rowmajor(hw[s + "/candidate/kernel"], H, 2)));
    }
    for (const char* p : {"enc_x", "enc_y"}) {      // transposed h-blocks for the encoders' BPTT
        const std::string s(p);
        const auto& gk = hw[s + "/gates/kernel"]; const auto& ck = hw[s + "/candidate/kernel"];
        bad |= up(s + "/WgT_h", pack_b(2 * H, H, [&](int k, int n) { return gk[(size_t)(2 + n) * 2 * H + k]; }));
        bad |= up(s + "/WcT_h", pack_b(H, H, [&](int k, int n) { return ck[(size_t)(2 + n) * H + k]; }));
    }
    bad |= up("dec/gb", hw["dec/gates/bias"]); bad |= up("dec/cb", hw["dec/candidate/bias"]);
    bad |= up("dec/Wxg", pack_b(H, 2 * H, rowmajor(hw["dec/gates/kernel"], 2 * H, 0)));
    bad |= up("dec/Whg", pack_b(H, 2 * H, rowmajor(hw["dec/gates/kernel"], 2 * H, H)));
    bad |= up("dec/Wxc", pack_b(H, H, rowmajor(hw["dec/candidate/kernel"], H, 0)));
    bad |= up("dec/Whc", pack_b(H, H, rowmajor(hw["dec/candidate/kernel"], H, H)));
    {   // transposed blocks for the backward data-gradient contractions: B(k', n') = W[row0 + n'][k']
        const auto& gk = hw["dec/gates/kernel"]; const auto& ck = hw["dec/candidate/kernel"];
        bad |= up("dec/WgT_x", pack_b(2 * H, H, [&](int k, int n) { return gk[(size_t)n * 2 * H + k]; }));
        bad |= up("dec/WgT_h", pack_b(2 * H, H, [&](int k, int n) { return gk[(size_t)(H + n) * 2 * H + k]; }));
        bad |= up("dec/WcT_x", pack_b(H, H, [&](int k, int n) { return ck[(size_t)n * H + k]; }));
        bad |= up("dec/WcT_h", pack_b(H, H, [&](int k, int n) { return ck[(size_t)(H + n) * H + k]; }));
    }
    bad |= up("head/w", hw["head/w"]); bad |= up("head/b", hw["head/b"]);
    bad |= up("ioc/gb", hw["ioc/gates/bias"]); bad |= up("ioc/cb", hw["ioc/candidate/bias"]);
    bad |= up("ioc/Wg", pack_b(E + H, 2 * H, rowmajor(hw["ioc/gates/kernel"], 2 * H, 0)));
    bad |= up("ioc/Wc", pack_b(E + H, H, rowmajor(hw["ioc/candidate/kernel"], H, 0)));
    {   // transposed blocks for the IOC BPTT: B(k', n') = W[row0 + n'][k']
        const auto& gk = hw["ioc/gates/kernel"]; const auto& ck = hw["ioc/candidate/kernel"];
        const auto& wr = hw["ioc/reg/w"]; const auto& ws = hw["ioc/social_fc/w"];
        const int xr = d.E_v + d.C;                      // first e_r row of the GRU kernels
        const int T2 = 2 * d.T_pred, KR = (T2 + 7) / 8 * 8;
        bad |= up("ioc/WgT_h", pack_b(2 * H, H, [&](int k, int n) { return gk[(size_t)(E + n) * 2 * H + k]; }));
        bad |= up("ioc/WgT_er", pack_b(2 * H, H, [&](int k, int n) { return gk[(size_t)(xr + n) * 2 * H + k]; }));
        bad |= up("ioc/WgT_ev", pack_b(2 * H, 32, [&](int k, int n) { return n < d.E_v ? gk[(size_t)n * 2 * H + k] : 0.f; }));
        bad |= up("ioc/WcT_h", pack_b(H, H, [&](int k, int n) { return ck[(size_t)(E + n) * H + k]; }));
        bad |= up("ioc/WcT_er", pack_b(H, H, [&](int k, int n) { return ck[(size_t)(xr + n) * H + k]; }));
        bad |= up("ioc/WcT_ev", pack_b(H, 32, [&](int k, int n) { return n < d.E_v ? ck[(size_t)n * H + k] : 0.f; }));
        bad |= up("ioc/WrT", pack_b(KR, H, [&](int k, int n) { return k < T2 ? wr[(size_t)n * T2 + k] : 0.f; }));
        std::vector<float> all;
        for (int b = 0; b < B; ++b) {
            auto pk = pack_b(H, H, [&](int k, int n) { return ws[((size_t)b * H + n) * H + k]; });
            all.insert(all.end(), pk.begin(), pk.end());
        }
        bad |= up("ioc/WsT", all);
        {   // the same transposed blocks in 16x16x4 fragment order (row-compacted dpool of k_ioc_bwd): per bin, 16-column tile
            // ct, 16-k group g, lane (col = lane&15, q = lane>>4) holds WsT_b[16g + 4q + 0..3][16ct + col] = W_b[16ct + col][16g + 4q + ..]
            const int T16 = H / 16;
            std::vector<float> tc((size_t)B * H * H);
            for (int b = 0; b < B; ++b)
                for (int ct = 0; ct < T16; ++ct)
                    for (int g = 0; g < T16; ++g)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j)
                                tc[((((size_t)b * T16 + ct) * T16 + g) * 64 + lane) * 4 + j] =
                                    ws[((size_t)b * H + 16 * ct + (lane & 15)) * H + 16 * g + 4 * (lane >> 4) + j];
            bad |= up("ioc/WsT_c", tc);
        }
    }
    if (d.bf16 == 2 || d.bf16 == 3) {   // split-bf16 packs of the IOC kernel (kernels_x3.hip): [hi | lo], hi = bf16(w), lo = bf16(w - hi);
                                         // dims.bf16 = 3: [hi | mid | lo], one more piece of the remainder (w = hi + mid + lo exactly)
        const size_t np_default = d.bf16 == 3 ? 3 : 2;
        const auto& gk = hw["ioc/gates/kernel"]; const auto& ck = hw["ioc/candidate/kernel"];
        const auto& wr = hw["ioc/reg/w"]; const auto& ws = hw["ioc/social_fc/w"];
        auto lin = [](int g, int hi, int e) { return 16 * g + 8 * hi + e; };
        auto chain = [](int g, int hi, int e) { const int hb = g >> 1, r = 8 * (g & 1) + e; return 32 * hb + (r & 3) + 8 * (r >> 2) + 4 * hi; };
        // vals = one fp32 value per bf16 slot.  While the repack maps are being built (pack_mode 1: values are index codes) the
        // value list itself is captured under "<name>#x3": it IS the gather map of the hi half, and of the lo half
        auto up_split = [&](const std::string& name, const std::vector<float>& vals, size_t np_over = 0) {
            const size_t np = np_over ? np_over : np_default;
            if (h->pack_mode == 1) { h->captured[name + (np == 3 ? "#x6" : "#x3")] = vals; return 0; }
            const size_t n = vals.size();
            std::vector<uint16_t> o(np * n + (np * n & 1));
            for (size_t i = 0; i < n; ++i) {
                float r = vals[i];
                for (size_t pc = 0; pc < np; ++pc) {
                    o[pc * n + i] = bf16_rne(r);
                    r -= bf16_to_f32(o[pc * n + i]);              // exact in fp32
                }
            }
            std::vector<float> out(o.size() / 2);
            std::memcpy(out.data(), o.data(), out.size() * 4);
            return up(name, out);
        };
        bad |= up_split("ioc/Wg16", pack_vals16(E + H, 2 * H, lin, [&](int k, int n) { return gk[(size_t)k * 2 * H + n]; }));
        bad |= up_split("ioc/Wc16", pack_vals16(E + H, H, lin, [&](int k, int n) { return ck[(size_t)k * H + n]; }));
        bad |= up_split("ioc/Wreg16", pack_vals16(H, 2 * d.T_pred, lin, [&](int k, int n) { return wr[(size_t)k * 2 * d.T_pred + n]; }));
        std::vector<float> all;
        for (int b = 0; b < B; ++b) {
            const auto pv = pack_vals16(H, H, chain, [&](int k, int n) { return ws[((size_t)b * H + k) * H + n]; });
            all.insert(all.end(), pv.begin(), pv.end());
        }
        bad |= up_split("ioc/Wsoc16", all);
        if (d.mno > 128 || d.H == 256) {     // shapes served by the step-wise split kernel (k_ioc_step<.., NP>): the pooled operand is a plain
            std::vector<float> alll;          // fp32 tile there, so the social weights are wanted in plain k order as well
            for (int b = 0; b < B; ++b) {
                const auto pv = pack_vals16(H, H, lin, [&](int k, int n) { return ws[((size_t)b * H + k) * H + n]; });
                alll.insert(alll.end(), pv.begin(), pv.end());
            }
            bad |= up_split("ioc/Wsoc16l", alll);
        }
        if (d.bf16 == 2) {   // training under dims.bf16 = 2: the two large data-gradient convolutions of the CVAE decoder (kernels_bwd_x3.hip)
            auto taps16 = [&](const std::vector<float>& wt, int CI, int CO) {       // as pack_taps(.., false): w[tap][ci][co]
                std::vector<float> out;
                for (int tap = 0; tap < 25; ++tap) {
                    const float* base = wt.data() + (size_t)tap * CI * CO;
                    const auto pv = pack_vals16(CI, CO, lin, [&](int k, int n) { return base[(size_t)k * CO + n]; });
                    out.insert(out.end(), pv.begin(), pv.end());
                }
                return out;
            };
            {   // transposed blocks of the IOC BPTT (k_ioc_bwd_x3): n-tiles [h columns | e_r columns | one e_v tile], B(k', n') = W[row0 + n'][k']
                const int xr = d.E_v + d.C;
                bad |= up_split("ioc/WcT16", pack_vals16(H, 2 * H + 32, lin, [&](int k, int n) {
                    return n < H ? ck[(size_t)(E + n) * H + k] : n < 2 * H ? ck[(size_t)(xr + n - H) * H + k] : (n - 2 * H < d.E_v ? ck[(size_t)(n - 2 * H) * H + k] : 0.f); }));
                bad |= up_split("ioc/WgT16", pack_vals16(2 * H, 2 * H + 32, lin, [&](int k, int n) {
                    return n < H ? gk[(size_t)(E + n) * 2 * H + k] : n < 2 * H ? gk[(size_t)(xr + n - H) * 2 * H + k] : (n - 2 * H < d.E_v ? gk[(size_t)(n - 2 * H) * 2 * H + k] : 0.f); }));
                std::vector<float> allT;
                for (int b = 0; b < B; ++b) {
                    const auto pv = pack_vals16(H, H, lin, [&](int k, int n) { return ws[((size_t)b * H + n) * H + k]; });
                    allT.insert(allT.end(), pv.begin(), pv.end());
                }
                bad |= up_split("ioc/WsT16", allT);
            }
            bad |= up_split("vae_dec/deconv3/Wbwd16", taps16(hw["vae_dec/deconv3/w"], 32, 64));
            bad |= up_split("vae_dec/deconv2/Wbwd16", taps16(hw["vae_dec/deconv2/w"], 64, 128));
        }
        {   // three-piece packs of the sample-generation kernels (kernels_x6.hip): decoder h-blocks, deconv2 / deconv3 taps.  dims.bf16 = 3:
            // inference; dims.bf16 = 2: the training-mode forward (sample generation stays in the fp32 kernels' accuracy class there too)
            const auto& dg = hw["dec/gates/kernel"]; const auto& dc = hw["dec/candidate/kernel"];
            bad |= up_split("dec/Whg6", pack_vals16(H, 2 * H, lin, [&](int k, int n) { return dg[(size_t)(H + k) * 2 * H + n]; }), 3);
            bad |= up_split("dec/Whc6", pack_vals16(H, H, lin, [&](int k, int n) { return dc[(size_t)(H + k) * H + n]; }), 3);
            auto taps6 = [&](const std::vector<float>& wt, int CI, int CO) {        // transposed conv weights [tap][co][ci]
                std::vector<float> out;
                for (int tap = 0; tap < 25; ++tap) {
                    const float* base = wt.data() + (size_t)tap * CI * CO;
                    const auto pv = pack_vals16(CI, CO, lin, [&](int k, int n) { return base[(size_t)n * CI + k]; });
                    out.insert(out.end(), pv.begin(), pv.end());
                }
                return out;
            };
            {
                const auto& w1 = hw["vae_dec/deconv1/w"]; const auto& wm = hw["mask_fc/w"];
                bad |= up_split("vae_dec/deconv1/W6", pack_vals16(L, 2048, lin, [&](int k, int n) { return w1[(size_t)n * L + k]; }), 3);
                bad |= up_split("mask/W6", pack_vals16(V, H, lin, [&](int k, int n) { return wm[(size_t)k * H + n]; }), 3);
            }
            bad |= up_split("vae_dec/deconv2/W6", taps6(hw["vae_dec/deconv2/w"], 128, 64), 3);
            bad |= up_split("vae_dec/deconv3/W6", taps6(hw["vae_dec/deconv3/w"], 64, 32), 3);
        }
    }
    if (d.bf16 == 1) {   // bf16 operand packs of the IOC kernel (kernels_bf16.hip)
        const auto& gk = hw["ioc/gates/kernel"]; const auto& ck = hw["ioc/candidate/kernel"];
        const auto& wr = hw["ioc/reg/w"]; const auto& ws = hw["ioc/social_fc/w"];
        auto lin = [](int g, int hi, int e) { return 16 * g + 8 * hi + e; };
        // chain order: k-slot (hi, e) of group g = 2*hb + g2 holds hidden 32*hb + rowmap(8*g2 + e, hi), the accumulator
        // row a lane of the pooling MFMA owns (rowmap(r, hi) = (r&3) + 8*(r>>2) + 4*hi)
        auto chain = [](int g, int hi, int e) { const int hb = g >> 1, r = 8 * (g & 1) + e; return 32 * hb + (r & 3) + 8 * (r >> 2) + 4 * hi; };
        bad |= up("ioc/Wg16", pack_b16(E + H, 2 * H, lin, [&](int k, int n) { return gk[(size_t)k * 2 * H + n]; }));
        bad |= up("ioc/Wc16", pack_b16(E + H, H, lin, [&](int k, int n) { return ck[(size_t)k * H + n]; }));
        bad |= up("ioc/Wreg16", pack_b16(H, 2 * d.T_pred, lin, [&](int k, int n) { return wr[(size_t)k * 2 * d.T_pred + n]; }));
        std::vector<float> all;
        for (int b = 0; b < B; ++b) {
            auto pk = pack_b16(H, H, chain, [&](int k, int n) { return ws[((size_t)b * H + k) * H + n]; });
            all.insert(all.end(), pk.begin(), pk.end());
        }
        bad |= up("ioc/Wsoc16", all);
        auto taps16 = [&](const std::vector<float>& wt, int CI, int CO) {       // transposed conv weights [tap][co][ci]
            std::vector<float> out;
            for (int tap = 0; tap < 25; ++tap) {
                const float* base = wt.data() + (size_t)tap * CI * CO;
                auto pk = pack_b16(CI, CO, lin, [&](int k, int n) { return base[(size_t)n * CI + k]; });
                out.insert(out.end(), pk.begin(), pk.end());
            }
            return out;
        };
        for (const char* pfx : {"enc_x", "enc_y"}) {
            const std::string sp(pfx);
            const auto& eg = hw[sp + "/gates/kernel"]; const auto& ec = hw[sp + "/candidate/kernel"];
            bad |= up(sp + "/Whg16", pack_b16(H, 2 * H, lin, [&](int k, int n) { return eg[(size_t)(2 + k) * 2 * H + n]; }));
            bad |= up(sp + "/Whc16", pack_b16(H, H, lin, [&](int k, int n) { return ec[(size_t)(2 + k) * H + n]; }));
        }
        {
            const auto& dg = hw["dec/gates/kernel"]; const auto& dc = hw["dec/candidate/kernel"];
            bad |= up("dec/Whg16", pack_b16(H, 2 * H, lin, [&](int k, int n) { return dg[(size_t)(H + k) * 2 * H + n]; }));
            bad |= up("dec/Whc16", pack_b16(H, H, lin, [&](int k, int n) { return dc[(size_t)(H + k) * H + n]; }));
        }
        {
            const auto& w1 = hw["vae_dec/deconv1/w"]; const auto& wm = hw["mask_fc/w"];
            bad |= up("vae_dec/deconv1/W16", pack_b16(L, 2048, lin, [&](int k, int n) { return w1[(size_t)n * L + k]; }));
            bad |= up("mask/W16", pack_b16(V, H, lin, [&](int k, int n) { return wm[(size_t)k * H + n]; }));
        }
        {   // forward conv weights [tap][ci][co]
            auto fwd16 = [&](const std::vector<float>& wt, int CI, int CO) {
                std::vector<float> out;
                for (int tap = 0; tap < 25; ++tap) {
                    const float* base = wt.data() + (size_t)tap * CI * CO;
                    auto pk = pack_b16(CI, CO, lin, [&](int k, int n) { return base[(size_t)k * CO + n]; });
                    out.insert(out.end(), pk.begin(), pk.end());
                }
                return out;
            };
            bad |= up("vae_enc/conv2/W16", fwd16(hw["vae_enc/conv2/w"], 32, 64));
            bad |= up("vae_enc/conv3/W16", fwd16(hw["vae_enc/conv3/w"], 64, 128));
        }
        bad |= up("vae_dec/deconv2/W16", taps16(hw["vae_dec/deconv2/w"], 128, 64));
        bad |= up("vae_dec/deconv3/W16", taps16(hw["vae_dec/deconv3/w"], 64, 32));
        {   // deconv4 as "tap products": A[m = tap][k = channel, chain order] = w4[tap][0][channel]
            const auto& w4 = hw["vae_dec/deconv4/w"];
            bad |= up("vae_dec/deconv4/W16", pack_b16(32, 32, chain, [&](int k, int n) { return n < 25 ? w4[(size_t)n * 32 + k] : 0.f; }));
        }
    }
    bad |= up("ioc/vel_w", hw["ioc/vel_fc/w"]); bad |= up("ioc/vel_b", hw["ioc/vel_fc/b"]);
    {
        std::vector<float> all;
        for (int b = 0; b < B; ++b) {
            auto pk = pack_b(H, H, rowmajor(hw["ioc/social_fc/w"], H, b * H));
            all.insert(all.end(), pk.begin(), pk.end());
        }
        bad |= up("ioc/Wsoc", all);
    }
    {   // the same weights for the row-compacted pooling (16x16x4 MFMA tiles, kernels_rnn.hip k_ioc<..., CP>): per bin, per
        // 16-column tile ct and 16-k group g, lane (col = lane&15, q = lane>>4) holds W_b[16g + 4q + 0..3][16ct + col]
        const auto& ws = hw["ioc/social_fc/w"];
        const int T16 = H / 16;
        std::vector<float> all((size_t)B * H * H);
        for (int b = 0; b < B; ++b)
            for (int ct = 0; ct < T16; ++ct)
                for (int g = 0; g < T16; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 4; ++j)
                            all[((((size_t)b * T16 + ct) * T16 + g) * 64 + lane) * 4 + j] =
                                ws[((size_t)b * H + 16 * g + 4 * (lane >> 4) + j) * H + 16 * ct + (lane & 15)];
        bad |= up("ioc/Wsoc_c", all);
    }
    bad |= up("ioc/soc_b", hw["ioc/social_fc/b"]);
    bad |= up("ioc/score_w", hw["ioc/score/w"]); bad |= up("ioc/score_b", hw["ioc/score/b"]);
    bad |= up("ioc/Wreg", pack_b(H, 2 * d.T_pred, rowmajor(hw["ioc/reg/w"], 2 * d.T_pred, 0)));
    bad |= up("ioc/reg_b", hw["ioc/reg/b"]);
    // dense layers
    bad |= up("fc_c/W", pack_b(2 * H, V, rowmajor(hw["fc_c/w"], V, 0)));  bad |= up("fc_c/b", hw["fc_c/b"]);
    bad |= up("vae_enc/fc/W", pack_b(2048, 2 * L, rowmajor(hw["vae_enc/fc/w"], 2 * L, 0)));
    bad |= up("vae_enc/fc/b", hw["vae_enc/fc/b"]);
    bad |= up("mask/W", pack_b(V, H, rowmajor(hw["mask_fc/w"], H, 0)));  bad |= up("mask/b", hw["mask_fc/b"]);
    // conv stack: folded batch-norm + packed taps
    std::vector<float> sc, sh;
    for (const char* n : {"vae_enc/conv1", "vae_enc/conv2", "vae_enc/conv3", "vae_dec/deconv1", "vae_dec/deconv2",
                          "vae_dec/deconv3", "vae_dec/deconv4"}) {
        fold_bn(h, n, sc, sh);
        bad |= up(std::string(n) + "/scale", sc); bad |= up(std::string(n) + "/shift", sh);
        if (d.bn_mode != 0) { bad |= up(std::string(n) + "/gamma", hw[std::string(n) + "/bn/gamma"]); bad |= up(std::string(n) + "/beta", hw[std::string(n) + "/bn/beta"]); }
    }
    bad |= up("vae_enc/conv1/raw", hw["vae_enc/conv1/w"]);
    bad |= up("vae_dec/deconv4/raw", hw["vae_dec/deconv4/w"]);
    auto pack_taps = [&](const std::vector<float>& w, int CI, int CO, bool transposed) {
        std::vector<float> all;   // forward conv: w[tap][ci][co]; transposed conv: w[tap][co][ci]
        for (int tap = 0; tap < 25; ++tap) {
            const float* base = w.data() + (size_t)tap * CI * CO;
            auto pk = pack_b(CI, CO, [&](int k, int n) { return transposed ? base[(size_t)n * CI + k] : base[(size_t)k * CO + n]; });
            all.insert(all.end(), pk.begin(), pk.end());
        }
        return all;
    };
    bad |= up("vae_enc/conv2/W", pack_taps(hw["vae_enc/conv2/w"], 32, 64, false));
    bad |= up("vae_enc/conv3/W", pack_taps(hw["vae_enc/conv3/w"], 64, 128, false));
    bad |= up("vae_dec/deconv2/W", pack_taps(hw["vae_dec/deconv2/w"], 128, 64, true));
    bad |= up("vae_dec/deconv3/W", pack_taps(hw["vae_dec/deconv3/w"], 64, 32, true));
    {   // deconv1 as GEMM: B(k = ci, n = (ky*4+kx)*128 + co) = w[n*L + k]
        const auto& w1 = hw["vae_dec/deconv1/w"];
        bad |= up("vae_dec/deconv1/W", pack_b(L, 2048, [&](int k, int n) { return w1[(size_t)n * L + k]; }));
    }
    for (const char* n : {"scene_cnn/conv1/w", "scene_cnn/conv1/b", "scene_cnn/conv2/w", "scene_cnn/conv2/b",
                          "scene_cnn/conv3/w", "scene_cnn/conv3/b", "temporal/w", "temporal/b", "gauss_head/w", "gauss_head/b"})
        bad |= up(n, hw[n]);
    {   // operands of the backward data-gradient passes (the forward kernels run with swapped roles)
        const auto& wm = hw["mask_fc/w"]; const auto& wfc = hw["vae_enc/fc/w"]; const auto& wcc = hw["fc_c/w"];
        bad |= up("mask/WT", pack_b(H, V, [&](int k, int n) { return wm[(size_t)n * H + k]; }));
        bad |= up("vae_enc/fc/WT", pack_b(2 * L, 2048, [&](int k, int n) { return wfc[(size_t)n * 2 * L + k]; }));
        bad |= up("fc_c/WT", pack_b(V, 2 * H, [&](int k, int n) { return wcc[(size_t)n * V + k]; }));
        bad |= up("vae_dec/deconv1/WT", pack_b(2048, L, rowmajor(hw["vae_dec/deconv1/w"], L, 0)));
        bad |= up("vae_dec/deconv3/Wbwd", pack_taps(hw["vae_dec/deconv3/w"], 32, 64, false));   // [tap][co=32][ci=64] as conv 32->64
        bad |= up("vae_dec/deconv2/Wbwd", pack_taps(hw["vae_dec/deconv2/w"], 64, 128, false));  // [tap][co=64][ci=128] as conv 64->128
        bad |= up("vae_enc/conv3/Wbwd", pack_taps(hw["vae_enc/conv3/w"], 128, 64, true));      // [tap][ci=64][co=128] as deconv 128->64
        bad |= up("vae_enc/conv2/Wbwd", pack_taps(hw["vae_enc/conv2/w"], 64, 32, true));       // [tap][ci=32][co=64] as deconv 64->32
    }
    if (bad) return fail(DESIRE_ERR_HIP, "weight upload failed");
    if (h->pack_mode == 0) HIPCHK(hipDeviceSynchronize());
    return DESIRE_OK;
}

extern "C" int desire_set_scene_grids(desire_handle* h, const float* dev_grids, const int32_t* host_grid_of_scene) {
    if (!h || !dev_grids || !host_grid_of_scene) return fail(DESIRE_ERR_ARG, "null argument");
    for (int i = 0; i < h->d.n_scenes; ++i)
        if (host_grid_of_scene[i] < 0 || host_grid_of_scene[i] >= h->d.n_grids)
            return fail(DESIRE_ERR_ARG, "grid_of_scene entry out of range");
    HIPCHK(hipMemcpy(h->ws["grid_of_scene"].p, host_grid_of_scene, h->d.n_scenes * sizeof(int32_t), hipMemcpyHostToDevice));
    h->grids = dev_grids;
    h->grids_set = true;
    return DESIRE_OK;
}

int desire_ready(desire_handle* h) {
    if (!h) return fail(DESIRE_ERR_ARG, "null handle");
    if (!h->finalized) return fail(DESIRE_ERR_STATE, "weights not finalized (desire_finalize_weights)");
    return 0;
}

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
int compact_classes(const desire_ctx* h, int* m4) {         // slot classes: 8, 16, 32 below the handle's own mno, then mno itself
    int n = 0;
    for (int m : {8, 16, 32}) if (m < h->d.mno) m4[n++] = m;
    m4[n++] = h->d.mno;
    for (int i = n; i < 4; ++i) m4[i] = h->d.mno;
    return n;
}
int compact_setup(desire_ctx* h) {
    const desire_dims& d = h->d;
    const size_t A = h->A, R = h->R, f = sizeof(float);
    struct WS { const char* n; size_t bytes; };
    const WS list[] = {{"cp_amap", A * sizeof(int32_t)}, {"cp_inv", A * sizeof(int32_t)}, {"cp_count", 8 * sizeof(int32_t)}, {"cp_HxHy", A * 2 * d.H * f},
                       {"cp_plast", A * 2 * f}, {"cp_params", A * 2 * d.L * f}, {"cp_Y0", R * (size_t)d.T_pred * 2 * f}};
    const WS list_ioc[] = {{"ci_win", 4 * (size_t)d.n_scenes * sizeof(int32_t)}, {"ci_map", 4 * A * sizeof(int32_t)}, {"ci_Hx", A * 2 * d.H * f}, {"ci_pl", A * 2 * f},
                           {"ci_valid", A}, {"ci_gos", (size_t)d.n_scenes * sizeof(int32_t)}, {"ci_Y", R * (size_t)d.T_pred * 2 * f}, {"ci_score", R * f}};
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
// waits (once per desire_encode) for the scans' counts to reach the host
static int compact_wait(desire_ctx* h, hipStream_t s) {
    if (!h->cp_pending) return fail(DESIRE_ERR_STATE, "DESIRE_FLAG_COMPACT_*: desire_encode comes first (it builds the present-agent maps)");
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
    EncArgs e{};
    e.n_scenes = d.n_scenes; e.mno = d.mno; e.sx = d.sx; e.sy = d.sy; e.H = H;
    e.frames = dev_past; e.T = d.T_obs;
    e.wx_g = D(h, "enc_x/gk"); e.b_g = D(h, "enc_x/gb"); e.wx_c = D(h, "enc_x/ck"); e.b_c = D(h, "enc_x/cb");
    e.Whg = D4(h, "enc_x/Whg"); e.Whc = D4(h, "enc_x/Whc");
    e.out = W(h, "HxHy"); e.ldo = 2 * H; e.p_last = W(h, "p_last"); e.valid = static_cast<uint8_t*>(h->ws["valid"].p);
    if (h->training) { e.sv_r = W(h, "ex_sv_r"); e.sv_u = W(h, "ex_sv_u"); e.sv_c = W(h, "ex_sv_c"); e.sv_h = W(h, "ex_sv_h"); e.sv_x = W(h, "ex_sv_x"); }
    const EncArgs ex = e;
    if (d.posterior) {
        e.frames = dev_fut; e.T = d.T_pred;
        e.wx_g = D(h, "enc_y/gk"); e.b_g = D(h, "enc_y/gb"); e.wx_c = D(h, "enc_y/ck"); e.b_c = D(h, "enc_y/cb");
        e.Whg = D4(h, "enc_y/Whg"); e.Whc = D4(h, "enc_y/Whc");
        e.out = W(h, "HxHy") + H; e.p_last = nullptr; e.valid = nullptr;
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
    if (compact_rows(h) || compact_ioc(h)) {
        // present-row compaction (DESIRE_FLAG_COMPACT_ROWS): the map of the agents present at the last observed frame, built right behind the
        // encoder that writes `valid`; its size reaches the host through a mapped word while the CVAE encoder below keeps the device busy, and
        // desire_sample waits on the event before it sizes its launches.
        if (int rc = compact_setup(h)) return rc;
        launch_present_scan(static_cast<const uint8_t*>(h->ws["valid"].p), A, static_cast<int32_t*>(h->ws["cp_amap"].p), static_cast<int32_t*>(h->ws["cp_inv"].p),
                            static_cast<int32_t*>(h->ws["cp_count"].p), h->cp_host, s);
        if (compact_ioc(h)) {
            int m4[4];
            const int n_cls = compact_classes(h, m4);
            launch_class_scan(static_cast<const uint8_t*>(h->ws["valid"].p), d.n_scenes, d.mno, n_cls, m4, d.K, h->ci_min_rows, static_cast<int32_t*>(h->ws["ci_win"].p),
                              static_cast<int32_t*>(h->ws["ci_map"].p), static_cast<int32_t*>(h->ws["cp_count"].p) + 4, h->cp_host + 4, s);
        }
        HIPCHK(hipEventRecord(h->cp_ev, s));
        h->cp_pending = true;
    }
    if (d.posterior) {
        GemmArgs g{};
        g.A = W(h, "HxHy"); g.lda = 2 * H; g.M = A; g.K = 2 * H; g.Bp = D4(h, "fc_c/W"); g.G = 2 * H / 8;
        g.NT = h->V / 32; g.out = W(h, "vae_in"); g.ldo = h->V; g.N = h->V; g.p0 = D(h, "fc_c/b");
        { Timer t(h, s, "fc_c"); launch_gemm_rows(g, EPI_BIAS_RELU, s); }
        ConvArgs c{};
        c.n = A;
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
        { Timer t(h, s, "conv1"); launch_conv1(c, s); if (pobn) norm("vae_enc/conv1", W(h, "c1"), A, 256, 32, 0); }
        c.in = W(h, "c1"); c.out = W(h, "c2"); c.Wp = D4(h, "vae_enc/conv2/W");
        c.scale = D(h, "vae_enc/conv2/scale"); c.shift = D(h, "vae_enc/conv2/shift");
        if (d.bf16 == 1) { c.Wp = D4(h, "vae_enc/conv2/W16"); Timer t(h, s, "conv2"); launch_conv2_bf16(c, s); }
        else { Timer t(h, s, "conv2"); launch_conv2(c, s); if (pobn) norm("vae_enc/conv2", W(h, "c2"), A, 64, 64, 0); }
        c.in = W(h, "c2"); c.out = W(h, "c3"); c.Wp = D4(h, "vae_enc/conv3/W");
        c.scale = D(h, "vae_enc/conv3/scale"); c.shift = D(h, "vae_enc/conv3/shift");
        if (d.bf16 == 1) { c.Wp = D4(h, "vae_enc/conv3/W16"); Timer t(h, s, "conv3"); launch_conv3_bf16(c, s); }
        else { Timer t(h, s, "conv3"); launch_conv3(c, s); if (pobn) norm("vae_enc/conv3", W(h, "c3"), A, 16, 128, 0); }
        g = GemmArgs{};
        g.A = W(h, "c3"); g.lda = 2048; g.M = A; g.K = 2048; g.Bp = D4(h, "vae_enc/fc/W"); g.G = 2048 / 8;
        g.NT = (2 * d.L + 31) / 32; g.out = W(h, "params"); g.ldo = 2 * d.L; g.N = 2 * d.L; g.p0 = D(h, "vae_enc/fc/b");
        { Timer t(h, s, "vae_enc_fc"); launch_gemm_rows(g, EPI_BIAS, s); }
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
    h->cp_last = compact;
    if (compact) {
        if (int rc = compact_wait(h, s)) return rc;
        const int P = *static_cast<volatile int32_t*>(h->cp_host);
        if (P < 0 || P > h->A) return fail(DESIRE_ERR_HIP, "present-agent scan returned a count out of range");
        h->cp_P = P;
        R = P * d.K; mno = P;
        const size_t RT2 = (size_t)h->R * d.T_pred * 2;
        if (P == 0) {       // nothing present: every row is padding
            launch_fill_f32(W(h, "Y0"), RT2, 0.f, s); launch_fill_f32(dev_Yhat, RT2, 0.f, s);
            HIPCHK(hipGetLastError());
            return DESIRE_OK;
        }
        const int32_t* amap = static_cast<const int32_t*>(h->ws["cp_amap"].p);
        Timer t(h, s, "compact_gather");
        launch_gather_agents(W(h, "HxHy"), W(h, "cp_HxHy"), amap, P, 2 * H, s);
        launch_gather_agents(W(h, "p_last"), W(h, "cp_plast"), amap, P, 2, s);
        if (d.posterior) launch_gather_agents(W(h, "params"), W(h, "cp_params"), amap, P, 2 * d.L, s);
        HxS = W(h, "cp_HxHy"); plS = W(h, "cp_plast"); Yout = W(h, "cp_Y0");
    }
    if (compact) { Timer t(h, s, "reparam"); launch_reparam_c(W(h, "cp_params"), dev_eps, W(h, "z"), static_cast<const int32_t*>(h->ws["cp_amap"].p), mno, d.K, d.mno, d.L, d.posterior, s); }
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
    c.n = R;
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
    if (h->training) m.sv_p = W(h, "mask_sv_p");
    if (d.bf16 == 1) { m.Wp = D4(h, "mask/W16"); Timer t(h, s, "mask_fc"); launch_mask_bf16(m, s); }
    else if (x6gen && (H == 64 || H == 128) && h->V % 128 == 0) { m.Wp = D4(h, "mask/W6"); Timer t(h, s, "mask_fc"); launch_mask_x6(m, s); }
    else { Timer t(h, s, "mask_fc"); launch_mask(m, s); }
    DecArgs a{};
    a.xz = W(h, "xz"); a.Hx = HxS; a.ldhx = 2 * H; a.p_last = plS;
    a.R = R; a.K = d.K; a.mno = mno; a.H = H; a.T = d.T_pred;
    a.Wxg = D4(h, "dec/Wxg"); a.Wxc = D4(h, "dec/Wxc"); a.Whg = D4(h, "dec/Whg"); a.Whc = D4(h, "dec/Whc");
    a.b_g = D(h, "dec/gb"); a.b_c = D(h, "dec/cb"); a.w_head = D(h, "head/w"); a.b_head = D(h, "head/b");
    a.Y = Yout; a.hdump = nullptr;
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
        launch_scatter_rows(Yout, W(h, "Y0"), dev_Yhat, static_cast<const int32_t*>(h->ws["cp_amap"].p), mno, d.K, d.mno, d.T_pred * 2, s);
    } else
        launch_copy_f32(dev_Yhat, W(h, "Y0"), (size_t)R * d.T_pred * 2, s);
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

// One IOC launch sequence over a VIEW of the handle's rows: the handle's own shape (row_off 0), or one slot class of DESIRE_FLAG_COMPACT_IOC --
// n_scenes windows of mno slots each with their own agent-level inputs; training-mode saves go to the view's row offset in the shared buffers.
struct IocView {
    int R, mno, n_scenes; float* Y; float* score; const float* Hx; int ldhx; const float* p_last; const uint8_t* valid; const int32_t* gos; size_t row_off;
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
    // bf16: one workgroup holds groups of up to 64 agents; 96 / 128 (and 64 when variant 4 / 6 asks for it) run the cluster form
    // split forms: groups of up to 32 agents on 32-row tiles (also the training-mode forward); inference on groups of 64 agents runs the
    // 64-row tile of kernels_x6r2.hip (one group per tile) in either piece count
    const bool wide64 = v.mno == 64 && !h->training && ioc_x6r2_supported(v.mno, d.H, d.grid_size * d.grid_size);
    const bool x3 = d.bf16 == 2 && (ioc_x3_supported(v.mno, d.H, d.grid_size * d.grid_size) || wide64);
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
    if (!cluster && d.bf16 == 0 && !h->training && d.ioc_split != 1 && a.variant == 0) {
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
        const size_t RT = (size_t)v.R * d.T_pred, RTf = (size_t)h->R * d.T_pred, ro = v.row_off * d.T_pred;     // a view's saves sit at its row offset
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
        int m4[4];
        const int n_cls = compact_classes(h, m4);
        const int32_t* cnt = h->cp_host + 4;
        size_t aoff = 0, roff = 0, woff = 0;
        const size_t T2 = (size_t)d.T_pred * 2;
        launch_fill_f32(dev_score, (size_t)h->R, 0.f, s);
        h->ci_n = 0;
        for (int c = 0; c < n_cls; ++c) {
            const int n_c = static_cast<volatile const int32_t*>(cnt)[c], m_c = m4[c];
            if (n_c < 0 || n_c > d.n_scenes) return fail(DESIRE_ERR_HIP, "slot-class scan returned a count out of range");
            if (n_c == 0) continue;
            const int32_t* cmap = static_cast<const int32_t*>(h->ws["ci_map"].p) + (size_t)c * h->A;
            const int32_t* win = static_cast<const int32_t*>(h->ws["ci_win"].p) + (size_t)c * d.n_scenes;
            const int R_c = n_c * d.K * m_c;
            IocView v{R_c, m_c, n_c, W(h, "ci_Y") + roff * T2, W(h, "ci_score") + roff, W(h, "ci_Hx") + aoff * 2 * d.H, 2 * d.H, W(h, "ci_pl") + aoff * 2,
                      static_cast<const uint8_t*>(h->ws["ci_valid"].p) + aoff, static_cast<const int32_t*>(h->ws["ci_gos"].p) + woff, roff};
            {
                Timer t(h, s, "ioc_repack");
                launch_cls_gather_agents(W(h, "HxHy"), 2 * d.H, W(h, "p_last"), static_cast<const int32_t*>(h->ws["grid_of_scene"].p), cmap, win, n_c, m_c,
                                         const_cast<float*>(v.Hx), const_cast<float*>(v.p_last), const_cast<uint8_t*>(v.valid), const_cast<int32_t*>(v.gos), s);
                launch_cls_rows(dev_Yhat, v.Y, cmap, n_c, m_c, d.K, d.mno, (int)T2, 0, s);
            }
            if (int rc = ioc_core(h, v, s)) return rc;
            {
                Timer t(h, s, "ioc_repack");
                launch_cls_rows(dev_Yhat, v.Y, cmap, n_c, m_c, d.K, d.mno, (int)T2, 1, s);
                launch_cls_rows(dev_score, v.score, cmap, n_c, m_c, d.K, d.mno, 1, 1, s);
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

extern "C" int desire_read_buffer(desire_handle* h, const char* name, float* host_out, size_t n, void* stream) {
    if (!h || !name || !host_out) return fail(DESIRE_ERR_ARG, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIPCHK(hipStreamSynchronize(s));
    const desire_dims& d = h->d;
    const size_t A = h->A, R = h->R, f = sizeof(float);
    const std::string nm(name);
    auto strided = [&](const float* src, size_t cols, size_t pitch_cols, size_t rows) -> int {
        if (n != rows * cols) return fail(DESIRE_ERR_ARG, nm + ": expected " + std::to_string(rows * cols) + " values");
        HIPCHK(hipMemcpy2D(host_out, cols * f, src, pitch_cols * f, cols * f, rows, hipMemcpyDeviceToHost));
        return 0;
    };
    if (nm == "Hx") return strided(W(h, "HxHy"), h->Hl, 2 * d.H, A);
    if (nm == "Hy") return strided(W(h, "HxHy") + d.H, h->Hl, 2 * d.H, A);
    if (nm == "xz") return strided(W(h, "xz"), h->Hl, d.H, R);
    if (nm == "z_mean") return strided(W(h, "params"), d.L, 2 * d.L, A);
    if (nm == "z_log_sigma_sq") return strided(W(h, "params") + d.L, d.L, 2 * d.L, A);
    struct P { const char* n; size_t cnt; };
    const P plain[] = {{"vae_in", A * h->V}, {"c1", A * 8192}, {"c2", A * 4096}, {"c3", A * 2048}, {"z", R * d.L},
                       {"d1", R * 2048}, {"d2", R * 4096}, {"d3", R * 8192}, {"xhat", R * 1024},
                       {"Y0", R * d.T_pred * 2}, {"p_last", A * 2}};
    for (const P& p : plain)
        if (nm == p.n) {
            if (n != p.cnt) return fail(DESIRE_ERR_ARG, nm + ": expected " + std::to_string(p.cnt) + " values");
            HIPCHK(hipMemcpy(host_out, W(h, p.n), p.cnt * f, hipMemcpyDeviceToHost));
            return DESIRE_OK;
        }
    {   // any other workspace buffer by its internal name (training-mode saves and gradient streams; the caller knows the layout)
        auto it = h->ws.find(nm);
        if (it != h->ws.end() && it->second.p) {
            if (n * f > it->second.bytes) return fail(DESIRE_ERR_ARG, nm + ": holds " + std::to_string(it->second.bytes / f) + " values");
            HIPCHK(hipMemcpy(host_out, it->second.p, n * f, hipMemcpyDeviceToHost));
            return DESIRE_OK;
        }
    }
    return fail(DESIRE_ERR_ARG, "unknown buffer: " + nm);
}

// ---- hipGraph capture of any sequence of desire_* calls made on `stream` (launch-bound shapes: small batches, the
// training step's ~100 launches, the agent-sharded IOC loop).  Everything the library enqueues is stream-ordered and
// pointer-stable, so a captured sequence can be replayed as long as the caller keeps the same device buffers.
// Not capturable: calls that synchronise or copy to the host (desire_train_loss, desire_read_buffer, desire_get_*, the
// cluster-form IOC's error read-back) and desire_adam_step (its bias-corrected step size is a per-call kernel argument).
extern "C" int desire_graph_begin(desire_handle* h, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!stream) return fail(DESIRE_ERR_ARG, "graph capture needs an explicit (non-default) stream");
    if (h->profiling) return fail(DESIRE_ERR_STATE, "switch profiling off before capturing (event records are not part of the graph)");
    HIPCHK(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
    return DESIRE_OK;
}

extern "C" int desire_graph_end(desire_handle* h, void* stream, int32_t* graph_id) {
    if (!h || !stream || !graph_id) return fail(DESIRE_ERR_ARG, "null argument");
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamEndCapture(static_cast<hipStream_t>(stream), &g));
    hipGraphExec_t ex = nullptr;
    const hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(DESIRE_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
    h->graphs.push_back(ex);
    *graph_id = (int32_t)h->graphs.size() - 1;
    return DESIRE_OK;
}

extern "C" int desire_graph_launch(desire_handle* h, int32_t graph_id, void* stream) {
    if (!h) return fail(DESIRE_ERR_ARG, "null handle");
    if (graph_id < 0 || graph_id >= (int32_t)h->graphs.size() || !h->graphs[graph_id]) return fail(DESIRE_ERR_ARG, "unknown graph id");
    HIPCHK(hipGraphLaunch(static_cast<hipGraphExec_t>(h->graphs[graph_id]), static_cast<hipStream_t>(stream)));
    return DESIRE_OK;
}

extern "C" int desire_device_buffer(desire_handle* h, const char* name, void** dev_ptr, size_t* bytes) {
    if (!h || !name || !dev_ptr || !bytes) return fail(DESIRE_ERR_ARG, "null argument");
    auto it = h->ws.find(name);
    if (it == h->ws.end()) return fail(DESIRE_ERR_ARG, std::string("unknown buffer: ") + name);
    *dev_ptr = it->second.p; *bytes = it->second.bytes;
    return DESIRE_OK;
}

// ---- agent-sharded IOC over PEER buffers: no collective and no host in the step loop ------------------------
// Every rank owns one exchange region (desire_peer_export: allocated uncached / fine-grained like RCCL's own buffers, exported as a
// hipIpcMemHandle) holding a progress counter and its OWN block of what desire_ioc_step takes as gathered arrays: presence flags, last
// observed positions, decoded positions, and two parities of its hidden-state rows.  desire_peer_open maps the others' regions (over
// xGMI when they live on another GPU; the same HBM when two ranks share a device, which is how the one-box test runs it).  One pass
// (desire_ioc_peer_pass) is then a fixed, stream-ordered sequence of ordinary launches -- wait(previous pass done) / publish / flag,
// T x { wait(peers at step t) / k_ioc_step reading the peers' blocks in place / flag }, finish -- with a ONE-WAVE wait kernel between
// steps (kernels_rnn.hip: k_peer_wait); everything is a kernel, so the pass can be captured with desire_graph_begin / _end.
// Hazards: step t reads parity (t - 1) & 1 of every rank and writes parity t & 1 of its own; a rank overwrites a parity only after
// all peers have flagged the step that read it, which is exactly the wait the data dependence needs anyway.
namespace {
struct PeerLayout { size_t valid, plast, Y, H0, H1, total; };
PeerLayout peer_layout(const desire_ctx* h) {
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t A = h->A, R = h->R, T = h->d.T_pred, H = h->d.H;
    PeerLayout l;
    l.valid = 256; l.plast = l.valid + up(A); l.Y = l.plast + up(A * 2 * 4); l.H0 = l.Y + up(R * T * 2 * 4); l.H1 = l.H0 + up(R * H * 4);
    l.total = l.H1 + up(R * H * 4);
    return l;
}
}  // namespace

extern "C" int desire_peer_export(desire_handle* h, uint8_t* handle_out64, size_t* bytes_out) {
    if (int rc = desire_ready(h)) return rc;
    if (!handle_out64) return fail(DESIRE_ERR_ARG, "null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    const PeerLayout l = peer_layout(h);
    if (!h->peer_region) {
        void* p = nullptr;
        if (hipExtMallocWithFlags(&p, l.total, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHK(hipMalloc(&p, l.total));                       // (a device without fine-grained allocations: same-device peers only)
        }
        HIPCHK(hipMemset(p, 0, l.total));
        const unsigned done0 = (unsigned)(h->d.T_pred + 1);       // "pass 0 complete": the first pass's pre-publish wait passes
        HIPCHK(hipMemcpy(p, &done0, sizeof(done0), hipMemcpyHostToDevice));
        h->peer_region = p; h->peer_bytes = l.total;
        if (hipHostMalloc(reinterpret_cast<void**>(&h->peer_err), sizeof(int), hipHostMallocMapped) != hipSuccess || !h->peer_err) {
            h->peer_err = nullptr;
            return fail(DESIRE_ERR_HIP, "hipHostMalloc failed for the peer error word");
        }
        *h->peer_err = 0;
        for (const char* nm : {"peer_epoch", "peer_score", "peer_hT"}) h->ws[nm].release();          // (export after a close: no leak)
        if (h->ws["peer_epoch"].alloc(sizeof(unsigned)) || h->ws["peer_score"].alloc((size_t)h->R * sizeof(float)) ||
            h->ws["peer_hT"].alloc((size_t)h->R * h->d.H * sizeof(float)))
            return fail(DESIRE_ERR_HIP, "hipMalloc failed for the peer buffers");
        HIPCHK(hipMemset(h->ws["peer_epoch"].p, 0, sizeof(unsigned)));
    }
    hipIpcMemHandle_t hd;
    HIPCHK(hipIpcGetMemHandle(&hd, h->peer_region));
    std::memcpy(handle_out64, &hd, 64);
    if (bytes_out) *bytes_out = l.total;
    return DESIRE_OK;
}

static int peer_attach(desire_handle* h, int32_t rank, int32_t nranks, int32_t peer, void* region, bool mapped) {
    h->peer_rank = rank; h->peer_nranks = nranks;
    h->peer_base[peer] = region; h->peer_mapped[peer] = mapped;
    bool all = true;
    for (int r = 0; r < nranks; ++r) all = all && h->peer_base[r];
    if (all) h->peer_ready = true;
    return DESIRE_OK;
}
static int peer_check(desire_handle* h, int32_t rank, int32_t nranks, int32_t peer) {
    if (int rc = desire_ready(h)) return rc;
    if (!h->peer_region) return fail(DESIRE_ERR_STATE, "desire_peer_export first");
    if (nranks < 1 || nranks > 8 || rank < 0 || rank >= nranks || peer < 0 || peer >= nranks) return fail(DESIRE_ERR_ARG, "bad rank / peer (at most 8 ranks)");
    if ((long)h->d.mno * nranks > 256) return fail(DESIRE_ERR_ARG, "agent-sharded IOC: at most 256 agents per scene over all ranks");
    if (h->peer_nranks && (h->peer_nranks != nranks || h->peer_rank != rank)) return fail(DESIRE_ERR_STATE, "peer set already opened with another rank / size");
    return DESIRE_OK;
}

extern "C" int desire_peer_open(desire_handle* h, int32_t rank, int32_t nranks, int32_t peer, const uint8_t* handle64) {
    if (int rc = peer_check(h, rank, nranks, peer)) return rc;
    if (peer == rank) return peer_attach(h, rank, nranks, peer, h->peer_region, false);
    if (!handle64) return fail(DESIRE_ERR_ARG, "null handle");
    hipIpcMemHandle_t hd;
    std::memcpy(&hd, handle64, 64);
    void* p = nullptr;
    HIPCHK(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
    return peer_attach(h, rank, nranks, peer, p, true);
}

// Ranks that live in the SAME process (one process driving several handles / devices with peer access enabled): the peer's region by
// its device pointer (desire_peer_region of the peer's handle) -- hipIpc handles cannot be opened by the process that exported them.
extern "C" int desire_peer_region(desire_handle* h, void** dev_region, size_t* bytes) {
    if (!h || !dev_region) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->peer_region) return fail(DESIRE_ERR_STATE, "desire_peer_export first");
    *dev_region = h->peer_region;
    if (bytes) *bytes = h->peer_bytes;
    return DESIRE_OK;
}
extern "C" int desire_peer_open_ptr(desire_handle* h, int32_t rank, int32_t nranks, int32_t peer, void* dev_region) {
    if (int rc = peer_check(h, rank, nranks, peer)) return rc;
    if (peer == rank) return peer_attach(h, rank, nranks, peer, h->peer_region, false);
    if (!dev_region) return fail(DESIRE_ERR_ARG, "null region");
    return peer_attach(h, rank, nranks, peer, dev_region, false);
}

extern "C" int desire_peer_close(desire_handle* h) {
    if (!h) return fail(DESIRE_ERR_ARG, "null handle");
    bool any = h->peer_region != nullptr || h->peer_err != nullptr;
    for (int r = 0; r < 8; ++r) any = any || h->peer_base[r] != nullptr;
    if (!any) return DESIRE_OK;          // a handle that never used peer buffers: nothing to wait for (no device-wide stall in desire_destroy)
    (void)hipDeviceSynchronize();
    for (int r = 0; r < 8; ++r) {
        if (h->peer_mapped[r] && h->peer_base[r]) (void)hipIpcCloseMemHandle(h->peer_base[r]);
        h->peer_base[r] = nullptr; h->peer_mapped[r] = false;
    }
    if (h->peer_region) { (void)hipFree(h->peer_region); h->peer_region = nullptr; }
    if (h->peer_err) { (void)hipHostFree(h->peer_err); h->peer_err = nullptr; }
    h->peer_ready = false; h->peer_nranks = 0; h->peer_rank = -1;
    return DESIRE_OK;
}

// The mapped error word of the peer exchange, for a caller that HAS synchronised the stream its pass ran on: 0 = every wait of the passes
// enqueued so far was satisfied, 1 = a bounded wait gave up (the results of that pass are not to be used).  Reading clears nothing: the next
// desire_ioc_peer_pass still fails with DESIRE_ERR_HIP and resets the word.
extern "C" int desire_peer_status(desire_handle* h, int32_t* timed_out) {
    if (!h || !timed_out) return fail(DESIRE_ERR_ARG, "null argument");
    *timed_out = h->peer_err ? (*static_cast<volatile int*>(h->peer_err) != 0 ? 1 : 0) : 0;
    return DESIRE_OK;
}

extern "C" int desire_ioc_peer_pass(desire_handle* h, float* dev_Y, float* dev_score, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!dev_Y || !dev_score) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->peer_ready) return fail(DESIRE_ERR_STATE, "desire_peer_export + desire_peer_open for every rank first");
    if (!h->grids_set) return fail(DESIRE_ERR_STATE, "desire_set_scene_grids first");
    const desire_dims& d = h->d;
    if (d.bf16 == 1) return fail(DESIRE_ERR_STATE, "agent-sharded IOC runs on fp32 operands");
    if (*static_cast<volatile int*>(h->peer_err)) {
        *h->peer_err = 0;
        return fail(DESIRE_ERR_HIP, "peer exchange timed out in an earlier pass (a rank never reached the step the others waited for)");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const PeerLayout l = peer_layout(h);
    const int G = h->peer_nranks, T = d.T_pred;
    const unsigned pp = (unsigned)(T + 2);
    PeerFlags flags{};
    for (int r = 0; r < G; ++r) flags.f[r] = static_cast<const unsigned*>(h->peer_base[r]);
    unsigned* epoch = static_cast<unsigned*>(h->ws["peer_epoch"].p);
    char* mine = static_cast<char*>(h->peer_region);
    float* Hpar[2] = {reinterpret_cast<float*>(mine + l.H0), reinterpret_cast<float*>(mine + l.H1)};
    for (int it = 0; it < d.iters; ++it) {
        launch_peer_epoch(epoch, s);
        launch_peer_wait(flags, G, epoch, pp, (unsigned)-1, h->peer_err, s);          // every peer has finished the previous pass: nobody reads my region
        launch_peer_publish(static_cast<const uint8_t*>(h->ws["valid"].p), W(h, "p_last"), dev_Y, W(h, "HxHy"), 2 * d.H,
                            reinterpret_cast<uint8_t*>(mine + l.valid), reinterpret_cast<float*>(mine + l.plast), reinterpret_cast<float*>(mine + l.Y),
                            Hpar[1], d.n_scenes, d.K, d.mno, T, d.H, s);              // h_{-1} goes to parity 1 (= (0 - 1) & 1)
        launch_peer_set(reinterpret_cast<unsigned*>(mine), epoch, pp, 1u, s);
        for (int t = 0; t < T; ++t) {
            launch_peer_wait(flags, G, epoch, pp, (unsigned)(t + 1), h->peer_err, s);
            IocStepArgs a{};
            a.t = t; a.rank = h->peer_rank; a.nranks = G; a.m_loc = d.mno; a.n_scenes = d.n_scenes; a.K = d.K; a.R = h->R;
            a.H = d.H; a.T = T; a.Gh = d.Gh; a.Gw = d.Gw; a.G = d.grid_size; a.nb_w = d.nb_w; a.nb_h = d.nb_h;
            a.peer = 1;
            for (int r = 0; r < G; ++r) {
                const char* b = static_cast<const char*>(h->peer_base[r]);
                a.vp[r] = reinterpret_cast<const uint8_t*>(b + l.valid); a.plp[r] = reinterpret_cast<const float*>(b + l.plast);
                a.Yp[r] = reinterpret_cast<const float*>(b + l.Y); a.Hp[r] = reinterpret_cast<const float*>(b + (((t + 1) & 1) ? l.H1 : l.H0));
            }
            a.st_h = Hpar[(t + 1) & 1]; a.st_h_out = Hpar[t & 1]; a.st_score = W(h, "peer_score");
            a.st_h_copy = (t == T - 1) ? W(h, "peer_hT") : nullptr;
            a.grids = h->grids; a.grid_of_scene = static_cast<const int32_t*>(h->ws["grid_of_scene"].p);
            a.w_vel = D(h, "ioc/vel_w"); a.b_vel = D(h, "ioc/vel_b"); a.Wsoc = D4(h, "ioc/Wsoc"); a.b_soc = D(h, "ioc/soc_b");
            a.Wg = D4(h, "ioc/Wg"); a.Wc = D4(h, "ioc/Wc"); a.b_g = D(h, "ioc/gb"); a.b_c = D(h, "ioc/cb"); a.w_score = D(h, "ioc/score_w");
            a.bin_tab = d.bin_mode == 1 ? W(h, "bin_tab") : nullptr;
            { Timer tm(h, s, "ioc_step"); launch_ioc_step(a, s); }
            launch_peer_set(reinterpret_cast<unsigned*>(mine), epoch, pp, (unsigned)(t + 2), s);
        }
        if (int rc = desire_ioc_finish(h, W(h, "peer_hT"), W(h, "peer_score"), dev_Y, dev_score, stream)) return rc;
    }
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

// ---- agent-sharded IOC (one step per call; the caller all-gathers hidden states between steps) ----
extern "C" int desire_ioc_step(desire_handle* h, int32_t t, int32_t rank, int32_t nranks, const float* dev_Yall,
                               const float* dev_plast_all, const uint8_t* dev_valid_all, const float* dev_Hall,
                               float* dev_h_state, float* dev_score_state, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    const desire_dims& d = h->d;
    if (!dev_Yall || !dev_plast_all || !dev_valid_all || !dev_Hall || !dev_h_state || !dev_score_state) return fail(DESIRE_ERR_ARG, "null argument");
    if (!h->grids_set) return fail(DESIRE_ERR_STATE, "desire_set_scene_grids first");
    if (t < 0 || t >= d.T_pred || nranks < 1 || rank < 0 || rank >= nranks) return fail(DESIRE_ERR_ARG, "bad step / rank");
    if ((long)d.mno * nranks > 256) return fail(DESIRE_ERR_ARG, "agent-sharded IOC: at most 256 agents per scene over all ranks");
    if (d.bf16 == 1) return fail(DESIRE_ERR_STATE, "agent-sharded IOC runs on fp32 operands");
    IocStepArgs a{};
    a.t = t; a.rank = rank; a.nranks = nranks; a.m_loc = d.mno; a.n_scenes = d.n_scenes; a.K = d.K; a.R = h->R;
    a.H = d.H; a.T = d.T_pred; a.Gh = d.Gh; a.Gw = d.Gw; a.G = d.grid_size; a.nb_w = d.nb_w; a.nb_h = d.nb_h;
    a.Yall = dev_Yall; a.plast_all = dev_plast_all; a.valid_all = dev_valid_all; a.Hall = dev_Hall;
    a.st_h = dev_h_state; a.st_h_out = dev_h_state; a.st_score = dev_score_state;
    a.grids = h->grids; a.grid_of_scene = static_cast<const int32_t*>(h->ws["grid_of_scene"].p);
    a.w_vel = D(h, "ioc/vel_w"); a.b_vel = D(h, "ioc/vel_b"); a.Wsoc = D4(h, "ioc/Wsoc"); a.b_soc = D(h, "ioc/soc_b");
    a.Wg = D4(h, "ioc/Wg"); a.Wc = D4(h, "ioc/Wc"); a.b_g = D(h, "ioc/gb"); a.b_c = D(h, "ioc/cb"); a.w_score = D(h, "ioc/score_w");
    a.bin_tab = d.bin_mode == 1 ? W(h, "bin_tab") : nullptr;
    hipStream_t s = static_cast<hipStream_t>(stream);
    { Timer tm(h, s, "ioc_step"); launch_ioc_step(a, s); }
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_ioc_finish(desire_handle* h, const float* dev_h_state, const float* dev_score_state, float* dev_Y,
                                 float* dev_score, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    const desire_dims& d = h->d;
    if (!dev_h_state || !dev_score_state || !dev_Y || !dev_score) return fail(DESIRE_ERR_ARG, "null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int T2 = 2 * d.T_pred;
    if (!h->ws.count("ioc_dY") && h->ws["ioc_dY"].alloc((size_t)h->R * T2 * sizeof(float))) return fail(DESIRE_ERR_HIP, "hipMalloc failed");
    GemmArgs g{};
    g.A = dev_h_state; g.lda = d.H; g.M = h->R; g.K = d.H; g.Bp = D4(h, "ioc/Wreg"); g.G = d.H / 8; g.NT = (T2 + 31) / 32;
    g.out = W(h, "ioc_dY"); g.ldo = T2; g.N = T2; g.p0 = D(h, "ioc/reg_b");
    launch_gemm_rows(g, EPI_BIAS, s);
    launch_ioc_finish(dev_Y, W(h, "ioc_dY"), dev_score_state, D(h, "ioc/score_b"), dev_score, h->R, d.T_pred, s);
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_neighbor_bins(desire_handle* h, const float* dev_pos, const uint8_t* dev_valid,
                                    int32_t* dev_bins, int32_t n_groups, void* stream) {
    if (!h || !dev_pos || !dev_valid || !dev_bins || n_groups < 0) return fail(DESIRE_ERR_ARG, "bad argument");
    if (n_groups == 0) return DESIRE_OK;
    launch_neighbor_bins(dev_pos, dev_valid, dev_bins, n_groups, h->d.mno, h->d.nb_w, h->d.nb_h, h->d.grid_size,
                         h->d.bin_mode == 1 ? W(h, "bin_tab") : nullptr, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_scene_cells(desire_handle* h, const float* dev_pos, int32_t* dev_cells, int32_t n, void* stream) {
    if (!h || !dev_pos || !dev_cells || n < 0) return fail(DESIRE_ERR_ARG, "bad argument");
    if (n == 0) return DESIRE_OK;
    launch_scene_cells(dev_pos, dev_cells, n, h->d.Gh, h->d.Gw, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_scene_cnn(desire_handle* h, const float* dev_image, int32_t Hi, int32_t Wi, float* dev_grids, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!dev_image || !dev_grids) return fail(DESIRE_ERR_ARG, "null argument");
    const desire_dims& d = h->d;
    if (Hi != 4 * d.Gh || Wi != 4 * d.Gw) return fail(DESIRE_ERR_ARG, "scene image must be [n_grids, 4*Gh, 4*Gw, 3]");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t n1 = (size_t)d.n_grids * (Hi / 2) * (Wi / 2) * 16, n2 = (size_t)d.n_grids * d.Gh * d.Gw * 32;
    if (!h->ws.count("scnn1")) {
        if (h->ws["scnn1"].alloc(n1 * sizeof(float)) || h->ws["scnn2"].alloc(n2 * sizeof(float)))
            return fail(DESIRE_ERR_HIP, "hipMalloc failed for the scene CNN workspace");
    }
    { Timer t(h, s, "scene_cnn");
      launch_conv_direct(dev_image, D(h, "scene_cnn/conv1/w"), D(h, "scene_cnn/conv1/b"), W(h, "scnn1"), d.n_grids, Hi, Wi, 3, 16, 2, 1, s);
      launch_conv_direct(W(h, "scnn1"), D(h, "scene_cnn/conv2/w"), D(h, "scene_cnn/conv2/b"), W(h, "scnn2"), d.n_grids, Hi / 2, Wi / 2, 16, 32, 2, 1, s);
      launch_conv_direct(W(h, "scnn2"), D(h, "scene_cnn/conv3/w"), D(h, "scene_cnn/conv3/b"), dev_grids, d.n_grids, d.Gh, d.Gw, 32, d.C, 1, 0, s); }
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_losses(desire_handle* h, const float* dev_fut, const float* dev_Yhat, float* dev_kld,
                             float* dev_recon, float* dev_cost, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!dev_fut || !dev_Yhat || !dev_kld || !dev_recon || !dev_cost) return fail(DESIRE_ERR_ARG, "null argument");
    const desire_dims& d = h->d;
    if (!d.posterior) return fail(DESIRE_ERR_STATE, "losses need the posterior path (dims.posterior = 1)");
    if (d.ref_compat) return fail(DESIRE_ERR_STATE, "ref_compat has no trajectory head: the reference's cost has undefined inputs (model/model.py:342)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_loss_mask(static_cast<const uint8_t*>(h->ws["valid"].p), dev_fut, static_cast<uint8_t*>(h->ws["lmask"].p), W(h, "nfut"),
                     d.n_scenes, d.mno, d.T_pred, s);
    launch_losses(W(h, "params"), dev_Yhat, dev_fut, static_cast<const uint8_t*>(h->ws["lmask"].p), W(h, "nfut"), dev_kld, dev_recon,
                  dev_cost, d.n_scenes, d.mno, d.K, d.T_pred, d.L, d.sx, d.sy, s);
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_temporal_conv(desire_handle* h, const float* dev_past, float* dev_rho, void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!dev_past || !dev_rho) return fail(DESIRE_ERR_ARG, "null argument");
    const desire_dims& d = h->d;
    launch_temporal_conv(dev_past, D(h, "temporal/w"), D(h, "temporal/b"), dev_rho, d.n_scenes, d.T_obs, d.mno,
                         static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_feature_pooling(desire_handle* h, const float* dev_Yhat, const float* dev_rho, float* dev_out, void* stream) {
    if (!h || !dev_Yhat || !dev_rho || !dev_out) return fail(DESIRE_ERR_ARG, "null argument");
    const desire_dims& d = h->d;
    // ref_compat: dev_Yhat = output_states [A, n_dec, T_obs, 2] -> [A, n_dec*T_obs, 200] (model/model.py:291-311 over the 7 states)
    launch_feature_pooling(dev_Yhat, dev_rho, dev_out, h->R, d.ref_compat ? d.n_dec * d.T_obs : d.T_pred, d.K, d.mno, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_build_windows(desire_handle* h, const float* dev_frames, int32_t n_frames, int32_t mno_in,
                                    const int32_t* host_starts, int32_t n_windows, float* dev_past, float* dev_fut, void* stream) {
    return desire_build_windows_la(h, dev_frames, n_frames, mno_in, host_starts, n_windows, 0, dev_past, dev_fut, stream);
}

extern "C" int desire_build_windows_la(desire_handle* h, const float* dev_frames, int32_t n_frames, int32_t mno_in,
                                     const int32_t* host_starts, int32_t n_windows, int32_t lookahead, float* dev_past, float* dev_fut,
                                     void* stream) {
    if (!h || !dev_frames || !host_starts || !dev_past || !dev_fut) return fail(DESIRE_ERR_ARG, "null argument");
    if (lookahead != 0 && lookahead != 1) return fail(DESIRE_ERR_ARG, "lookahead must be 0 or 1");
    const desire_dims& d = h->d;
    if (n_windows < 1 || n_windows > d.n_scenes) return fail(DESIRE_ERR_ARG, "n_windows must be 1..n_scenes");
    if (mno_in < 1 || n_frames < d.T_obs + d.T_pred) return fail(DESIRE_ERR_ARG, "video shorter than one window");
    for (int i = 0; i < n_windows; ++i)
        if (host_starts[i] < 0 || host_starts[i] + d.T_obs + d.T_pred > n_frames)
            return fail(DESIRE_ERR_ARG, "window start out of range");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // (ws.at, not ws[]: no insertion into the handle's map from this call -- it may run on a feeder thread, see desire_create)
    HIPCHK(hipMemcpyAsync(h->ws.at("bw_starts").p, host_starts, n_windows * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->ws.at("bw_err").p, 0, sizeof(int32_t), s));
    launch_build_windows(dev_frames, n_frames, mno_in, static_cast<const int32_t*>(h->ws.at("bw_starts").p), n_windows, d.T_obs,
                         d.T_pred, d.mno, dev_past, dev_fut, static_cast<int32_t*>(h->ws.at("bw_err").p), lookahead, s);
    int32_t err = 0;
    HIPCHK(hipMemcpyAsync(&err, h->ws.at("bw_err").p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (err & 2) return fail(DESIRE_ERR_ARG, "a window holds more unique ids than max_num_obj slots (utils/data_loader.py:227 IndexError)");
    if (err & 4) return fail(DESIRE_ERR_ARG, "a track id occurs twice in one frame of a window (utils/data_loader.py:224-229 ValueError)");
    if (err & 1) return fail(DESIRE_ERR_ARG, "track id outside [0, 65536)");
    return DESIRE_OK;
}

extern "C" int desire_gaussian_sample(desire_handle* h, const float* dev_params, const float* dev_normals, float* dev_out,
                                      int32_t n, void* stream) {
    if (!h || !dev_params || !dev_normals || !dev_out || n < 0) return fail(DESIRE_ERR_ARG, "bad argument");
    if (n == 0) return DESIRE_OK;
    launch_gaussian_sample(dev_params, dev_normals, dev_out, n, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

// sample()'s autoregressive rollout (model/model.py:623-688): warm-up over the observed frames with the X-encoder GRU (the
// reference's loop :623-632 carrying `states`), then `num` prediction steps, each: 5-wide Gaussian head on the state (:651,
// 661-663) -> draw (:665) -> clip (:666-669) -> feed the drawn position back as the next input (:680-681).
extern "C" int desire_rollout(desire_handle* h, const float* dev_past, const float* dev_normals, int32_t num, float* dev_out,
                              void* stream) {
    if (int rc = desire_ready(h)) return rc;
    if (!dev_past || !dev_normals || !dev_out) return fail(DESIRE_ERR_ARG, "null argument");
    if (num < 1) return fail(DESIRE_ERR_ARG, "num must be >= 1");
    const desire_dims& d = h->d;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!h->ws.count("roll_h") && h->ws["roll_h"].alloc((size_t)h->A * d.H * sizeof(float))) return fail(DESIRE_ERR_HIP, "hipMalloc failed");
    EncArgs e{};
    e.n_scenes = d.n_scenes; e.mno = d.mno; e.sx = d.sx; e.sy = d.sy; e.H = d.H;
    e.frames = dev_past; e.T = d.T_obs;
    e.wx_g = D(h, "enc_x/gk"); e.b_g = D(h, "enc_x/gb"); e.wx_c = D(h, "enc_x/ck"); e.b_c = D(h, "enc_x/cb");
    e.Whg = D4(h, "enc_x/Whg"); e.Whc = D4(h, "enc_x/Whc");
    e.out = W(h, "roll_h"); e.ldo = d.H;
    e.n_roll = num; e.w5 = D(h, "gauss_head/w"); e.b5 = D(h, "gauss_head/b"); e.normals = dev_normals; e.roll_out = dev_out;
    { Timer t(h, s, "rollout"); launch_encoder(e, s); }
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_ade_fde(desire_handle* h, const float* dev_Yhat, const float* dev_fut, float* dev_out, void* stream) {
    if (!h || !dev_Yhat || !dev_fut || !dev_out) return fail(DESIRE_ERR_ARG, "null argument");
    const desire_dims& d = h->d;
    launch_ade_fde(dev_Yhat, dev_fut, dev_out, d.n_scenes, d.mno, d.K, d.T_pred, d.sx, d.sy, static_cast<hipStream_t>(stream));
    HIPCHK(hipGetLastError());
    return DESIRE_OK;
}

extern "C" int desire_set_profiling(desire_handle* h, int enable) {
    if (!h) return fail(DESIRE_ERR_ARG, "null handle");
    h->profiling = enable != 0;
    return DESIRE_OK;
}

extern "C" int desire_get_profile(desire_handle* h, float* host_ms, const char** host_names, int32_t* count) {
    if (!h || !count) return fail(DESIRE_ERR_ARG, "null argument");
    const int cap = *count;
    int n = 0;
    for (auto& p : h->prof) {
        if (n >= cap) break;
        HIPCHK(hipEventSynchronize(p.e1));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, p.e0, p.e1));
        if (host_ms) host_ms[n] = ms;
        if (host_names) host_names[n] = p.name.c_str();
        ++n;
    }
    *count = n;
    if (!host_ms && !host_names) return DESIRE_OK;      // count query only
    h->prof_name_store.clear();
    for (auto& p : h->prof) h->prof_name_store.push_back(p.name);
    if (host_names) for (int i = 0; i < n; ++i) host_names[i] = h->prof_name_store[i].c_str();
    for (auto& p : h->prof) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    h->prof.clear();
    return DESIRE_OK;
}
