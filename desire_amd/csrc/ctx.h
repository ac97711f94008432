// ctx.h -- the handle behind desire_handle* and the small host helpers shared by api.hip (inference ABI) and
// train.hip (training ABI).  Host code only.
#pragma once
#include "../../include/desire_hip.h"
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

int desire_fail(int code, const std::string& msg);            // sets the thread-local last-error text
#define fail desire_fail
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess)                                                                       \
            return fail(DESIRE_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));            \
    } while (0)

struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    int alloc(size_t b) {
        bytes = b;
        hipError_t e = hipMalloc(&p, b ? b : 4);
        return e == hipSuccess ? 0 : -1;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; }
    float* f() const { return static_cast<float*>(p); }
};

struct Prof { std::string name; hipEvent_t e0, e1; };

struct WSlot { size_t off; size_t n; };                       // a named tensor inside the flat training buffers

// One axis of a weight's natural layout as a list of (logical, physical) segment lengths: when the caller's hidden width
// is below the narrowest instantiated recurrent tile (dims.H = 16 or 32, the reference's own default d_dim = 16,
// train.py:85), the kernels run at a physical width of 64 with the extra hidden units' weights, biases and initial state
// exactly zero -- such a unit stays at 0 for ever (c = tanh(0) = 0, h' = u*0 + (1-u)*0) and feeds nothing, and the
// added products are exact zeros, so the logical units' values do not change.
struct EmbedAxis { std::vector<std::pair<int, int>> seg; };
struct Embed { EmbedAxis rows, cols; };

struct desire_ctx {
    desire_dims d;                                           // d.H is the PHYSICAL hidden width the kernels run at
    int Hl = 0;                                              // logical hidden width = dims.H as given to desire_create
    int A, R, V, B, E;
    std::map<std::string, size_t> want_user;                 // name -> element count in the caller's (logical) layout
    std::map<std::string, Embed> emb;                        // weights whose logical layout differs from the physical one
    std::map<std::string, std::vector<float>> host_w;       // raw weights as set
    std::map<std::string, size_t> want;                      // name -> element count
    std::map<std::string, DevBuf> dev;                       // raw / packed / folded device tensors
    std::map<std::string, DevBuf> ws;                        // workspace
    bool finalized = false;
    std::vector<void*> graphs;                               // instantiated hipGraphExec_t of desire_graph_end
    std::vector<float> bin_tab_host;                         // log-polar bin table (20 floats) when dims.bin_mode == 1
    const float* grids = nullptr;
    bool grids_set = false;
    bool profiling = false;
    // peer exchange of the agent-sharded IOC (desire_peer_*): this rank's region, the mapped regions of the others, a mapped host error word
    void* peer_region = nullptr; size_t peer_bytes = 0; void* peer_base[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool peer_mapped[8] = {false, false, false, false, false, false, false, false}; int peer_rank = -1, peer_nranks = 0; bool peer_ready = false;
    int* peer_err = nullptr;
    float head_loss_w = 0.f;                                 // desire_set_head_loss: weight of the Gaussian-head NLL term in the training loss
    int* host_err = nullptr;                                 // mapped host word the bin-split IOC's bounded spins report into (checked by the next call)
    // present-row compaction (DESIRE_FLAG_COMPACT_ROWS, kernels_compact.hip): mapped host word the scan kernel reports the present-agent count
    // into, the event behind it, the count of the last desire_sample (-1: none yet)
    int32_t* cp_host = nullptr; hipEvent_t cp_ev = nullptr; bool cp_pending = false; int cp_P = -1;
    int ci_n = 0, ci_cls[4] = {0, 0, 0, 0}, ci_cnt[4] = {0, 0, 0, 0}; bool ci_last = false; int ci_min_rows = 8192;     // DESIRE_FLAG_COMPACT_IOC: the classes the last IOC stage ran (class index, windows)
    bool cp_enc = false;                                     // the last desire_encode ran its stack on the present agents only (saves in compact agent order)
    bool cp_host_counts = false;                             // desire_set_option("compact_host_counts", 1): inference reads the counts back like training does (A/B)
    bool cp_last = false;                                    // the last desire_sample ran compacted (desire_backward follows it, not the flag)
    // desire_build_windows*: plain pointers, cached at creation -- a feeder thread may run the builder while the owner thread runs a forward or a backward on
    // the same handle (desire_amd/prefetch.py: DeviceWindowFeeder), and those insert workspace entries lazily: the builder must not walk the map
    int32_t* bw_starts = nullptr; int32_t* bw_err = nullptr;
    std::vector<Prof> prof;
    std::vector<std::string> prof_name_store;
    // ---- training (train.hip) ----
    bool training = false;
    std::map<std::string, WSlot> slots;                      // natural-layout offsets in Wflat / Gflat / Mflat / Vflat
    size_t n_params = 0;
    const float* last_eps = nullptr;                         // inputs of the last training-mode forward
    int pack_mode = 0;                                       // 1: desire_upload captures instead of uploading (index pass)
    std::map<std::string, std::vector<float>> captured;
    int adam_t = 0;                                          // Adam step counter
    int n_seg = 0;                                           // repack segments (train.hip)
    int n_seg16 = 0;                                         // split [hi | lo] bf16 packs among them (dims.bf16 = 2)
};

struct Timer {
    desire_ctx* h; hipStream_t s; bool on;
    Timer(desire_ctx* h_, hipStream_t s_, const char* name) : h(h_), s(s_), on(h_->profiling) {
        if (!on) return;
        Prof p; p.name = name;
        (void)hipEventCreate(&p.e0); (void)hipEventCreate(&p.e1);
        (void)hipEventRecord(p.e0, s);
        h->prof.push_back(p);
    }
    ~Timer() { if (on) (void)hipEventRecord(h->prof.back().e1, s); }
};

// which parts of the training step use split operands under dims.bf16 = 2 (1: weight-gradient reductions, 2: data-gradient
// convolutions, 4: IOC BPTT, 8: six-product sample generation in the forward pass): everything dims.train_fp32_mask does not hold back
inline int train_x3_mask(const desire_ctx* h) { return 15 & ~h->d.train_fp32_mask; }

inline const float* D(desire_ctx* h, const char* name) { return h->dev.at(name).f(); }
inline const float4* D4(desire_ctx* h, const char* name) { return reinterpret_cast<const float4*>(h->dev.at(name).f()); }
inline float* W(desire_ctx* h, const char* name) { return h->ws.at(name).f(); }

// Packed fragment order: out[((nt*G + g)*64 + lane)*4 + i] = W(k = 8g + 4*(lane>>5) + i, n = nt*32 + (lane&31))
std::vector<float> pack_b(int K, int N, const std::function<float(int, int)>& at);
// bf16 fragment order (v_mfma_f32_32x32x16_bf16): out16[((nt*G + g)*64 + lane)*8 + e] = bf16(W(k = kmap(g, lane>>5, e), n = nt*32 + (lane&31))),
// G = ceil(K/16); returned as floats holding two bf16 bit patterns each (so the float upload path carries it)
std::vector<float> pack_b16(int K, int N, const std::function<int(int, int, int)>& kmap, const std::function<float(int, int)>& at);
// the same element order with the fp32 VALUES kept (one float per bf16 slot, 0 where the pack pads): source of the split
// [hi | lo] packs of kernels_x3.hip, and -- run over index-coded weights -- of their device repack map (train.hip)
std::vector<float> pack_vals16(int K, int N, const std::function<int(int, int, int)>& kmap, const std::function<float(int, int)>& at);
inline uint16_t bf16_rne(float f) {
    uint32_t u; std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t b) { const uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; }
// logical (caller) layout <-> physical layout of one named weight (identity when the weight has no Embed entry)
std::vector<float> desire_embed(const desire_ctx* h, const std::string& name, const float* user);
void desire_extract(const desire_ctx* h, const std::string& name, const float* phys, float* user);
int desire_upload(desire_ctx* h, const std::string& name, const std::vector<float>& v);
int desire_ready(desire_handle* h);
bool compact_rows(const desire_ctx* h);                        // DESIRE_FLAG_COMPACT_ROWS set
bool compact_ioc(const desire_ctx* h);                         // DESIRE_FLAG_COMPACT_IOC set and the shape is served
bool compact_dyn(const desire_ctx* h);                         // compacted launches take their counts from device words (inference, frozen batch-norm): no host wait
bool compact_padded_ok(const desire_ctx* h);                   // the padded-tile IOC kernels serve this handle (slot class 10)
int compact_classes(const desire_ctx* h, int* m4);             // its slot classes (ascending, the handle's mno last): returns how many
int compact_setup(desire_ctx* h);                              // its buffers, event and mapped count word (idempotent)
int desire_pack_all(desire_ctx* h);                            // (re)builds every packed / folded device tensor from host_w
