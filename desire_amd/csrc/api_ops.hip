// api_ops.hip -- the stand-alone ops of the ABI: integer paths, scene CNN, losses, the reference's literal tensors, window builder, Gaussian head,
// rollout, ADE / FDE.  Host code only; split out of api.hip in round 5.
#include "ctx.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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
    // (plain pointers cached by desire_create, not the handle's map: this call may run on a feeder thread while another call inserts into the map)
    HIPCHK(hipMemcpyAsync(h->bw_starts, host_starts, n_windows * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(h->bw_err, 0, sizeof(int32_t), s));
    launch_build_windows(dev_frames, n_frames, mno_in, h->bw_starts, n_windows, d.T_obs,
                         d.T_pred, d.mno, dev_past, dev_fut, h->bw_err, lookahead, s);
    int32_t err = 0;
    HIPCHK(hipMemcpyAsync(&err, h->bw_err, sizeof(int32_t), hipMemcpyDeviceToHost, s));
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

