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
    else if (nm == "compact_host_counts") {    // DESIRE_FLAG_COMPACT_*, inference: 1 = read the scans' counts back and size the launches exactly (one host wait per
        h->cp_host_counts = value != 0;        // desire_encode, not capturable -- what training always does); 0 (default) = device-side counts (kernels.h: DynCount)
        h->cp_pending = false; h->cp_enc = false;
        return DESIRE_OK;
    }
    else if (nm == "compact_min_rows") {       // DESIRE_FLAG_COMPACT_IOC: a slot class with fewer rows than this is folded into the next larger one (default 8192)
        if (value < 0) return fail(DESIRE_ERR_ARG, "compact_min_rows must be >= 0");
        h->ci_min_rows = value;
        return DESIRE_OK;
    }
    else return fail(DESIRE_ERR_ARG, "unknown option: " + nm + " (ioc_form, ioc_split, train_fp32_mask, flags, compact_min_rows, compact_host_counts)");
    if (int rc = check_options(d)) return rc;
    if ((d.flags ^ h->d.flags) & (DESIRE_FLAG_COMPACT_ROWS | DESIRE_FLAG_COMPACT_IOC)) {
        h->cp_pending = false; h->cp_enc = false;       // the maps of the last desire_encode were built for the other setting: a new desire_encode comes first
    }
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
        // the same handle (desire_amd/prefetch.py: DeviceWindowFeeder) -- the builder then touches these two buffers (through h->bw_starts / h->bw_err,
        // never through the map: other calls insert into it) and nothing else of the handle
        {"bw_starts", (size_t)d.n_scenes * sizeof(int32_t)}, {"bw_err", sizeof(int32_t)},
    };
    for (const WS& w : list) {
        if (h->ws[w.n].alloc(w.bytes)) { desire_destroy(h); return fail(DESIRE_ERR_HIP, std::string("hipMalloc failed for ") + w.n); }
        (void)hipMemset(h->ws[w.n].p, 0, w.bytes);
    }
    h->bw_starts = static_cast<int32_t*>(h->ws.at("bw_starts").p); h->bw_err = static_cast<int32_t*>(h->ws.at("bw_err").p);
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
