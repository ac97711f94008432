// api_pack.hip -- desire_pack_all: every packed / folded device tensor rebuilt from the handle's host weights (MFMA B-fragment order for fp32,
// bf16 and split-bf16 operands, folded batch-norm, transposed copies for the backward pass).  Host code only; split out of api.hip in round 5.
#include "ctx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
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

}  // namespace

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

