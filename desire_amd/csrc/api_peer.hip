// api_peer.hip -- the agent-sharded IOC entry points: desire_ioc_step / desire_ioc_finish (caller-side all-gather) and the peer-buffer exchange
// (desire_peer_*, desire_ioc_peer_pass).  Host code only; split out of api.hip in round 5.
#include "ctx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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

