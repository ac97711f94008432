/* desire_hip.h -- C ABI of libdesire_hip.so: DESIRE sample-generation + ranking/refinement hot
 * path on MI355X (gfx950).
 *
 * The reference (tdavchev/DESIRE) has NO FFI / plugin / operator interface: its hot path is
 * one Python function that unrolls a per-object loop into a TF1 graph
 * (/root/reference/model/model.py:79-403, loop at :211) behind two Python classes
 * (DESIREModel model/model.py:29, DataLoader utils/data_loader.py:20).  This ABI is therefore
 * the boundary a maintainer would bind with ctypes from model/model.py; each entry point
 * below names the reference lines whose work it replaces.  See INTEGRATION.md for the stub.
 *
 * Conventions
 *   - plain C, no torch types; every pointer named dev_* is a DEVICE pointer owned by the
 *     caller (e.g. torch tensor .data_ptr()), every pointer named host_* is host memory.
 *   - all tensors fp32, C-contiguous, layouts given per argument.
 *   - row index r = (scene*K + k)*mno + slot   (R = n_scenes*K*mno rows);
 *     agent index a = scene*mno + slot          (A = n_scenes*mno agents).
 *   - every call returns 0 on success, <0 on error; desire_last_error() gives the text.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are
 *     stream-ordered and asynchronous; one handle per device, not thread-safe per handle.
 *   - there is no CPU path: on a machine without a gfx950 device desire_create fails.
 */
#ifndef DESIRE_HIP_H
#define DESIRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DESIRE_OK 0
#define DESIRE_ERR_ARG (-1)     /* bad argument / dims */
#define DESIRE_ERR_STATE (-2)   /* weights missing, grids missing, ... */
#define DESIRE_ERR_HIP (-3)     /* a HIP runtime call failed */
#define DESIRE_ERR_NODEV (-4)   /* no gfx950 device */

/* Mirrors desire_amd/spec.py:Dims.  Field meaning and the reference args they come from:
 * mno=max_num_obj, T_obs/T_pred (ref: one seq_length), H=d_dim, L=latent_size,
 * S=int(sqrt(2*rnn_size)) (model/model.py:44-60); grid_size (train.py:71-72);
 * nb_w/nb_h = neighborhood_size (train.py:68-70) in normalised units; sx/sy pixel scale. */
typedef struct desire_dims {
    int32_t n_scenes, mno, K, T_obs, T_pred;
    int32_t H, L, S, C, Gh, Gw, n_grids;
    int32_t grid_size, E_v, iters, posterior;
    float nb_w, nb_h, sx, sy;
    int32_t bin_mode;      /* social pooling layout: 0 = rectangular grid_size x grid_size window of nb_w x nb_h (Social-LSTM /
                              the reference's flags, train.py:68-72); 1 = log-polar (the paper's): grid_size rings with
                              geometric radii between nb_h (inner radius) and nb_w (outer radius) x grid_size sectors */
    int32_t bn_mode;       /* batch normalisation of the CVAE conv stacks: 0 = frozen moving statistics folded into the kernels
                              (default); 1 = "per-object": the reference's literal graph -- phase=train on a batch of one
                              object (model/model.py:453-462,471-481), i.e. per-sample per-channel moments over the layer's
                              pixels.  2 = whole-batch statistics: the same phase=train moments taken over everything one call
                              batches (per channel over all samples and pixels) -- what prettytensor's default does when objects ARE
                              batched.  Modes 1 and 2: fp32 operands; mode 1 also trains (desire_backward goes through the per-sample
                              normalisation); mode 2 trains too (desire_backward takes the gradient means over every sample of the call; with
                              several data-parallel ranks each rank normalises over ITS share of the batch). */
    int32_t bf16;          /* 0: fp32 matrix operands (default); 1: bf16 operands / fp32 accumulate + fp32 state
                              for the recurrent IOC kernel (BASELINE configs[2]); inference only.  2: SPLIT bf16 operands --
                              every fp32 operand enters the bf16 matrix pipe as hi + lo (hi = bf16(x), lo = bf16(x - hi)) and a
                              product is three bf16 MFMAs (hi.hi + lo.hi + hi.lo, fp32 accumulate): results agree with the fp32
                              kernels to ~1e-5 relative at 3/16 of their matrix time: the IOC kernel, and in a training step the IOC
                              BPTT, the large weight-gradient reductions and data-gradient convolutions (gradients stay within 2e-4 of
                              float64 autograd); sample generation (decoder, deconv2, deconv3) runs the six-product kernels of mode 3,
                              because two-piece operands there would move the sampled positions.  Kernels without a split form run the
                              fp32 kernels, so 2 is always at least as accurate as it claims.  3: THREE bf16 pieces per operand
                              (x = hi + mid + lo exactly) and six products per fp32 product -- every term down to 2^-16 |a b| -- i.e.
                              the accuracy class of the fp32 fmaf chain itself at 6/16 of its matrix time; inference only, same
                              shapes as 2 (others run the fp32 kernels). */
    int32_t ref_compat;    /* 1: the reference graph AS WRITTEN (model/model.py:116-311) instead of the frozen spec: the GRU
                              decoder runs n_dec steps (7, :280) and every output state [H] is re-read as T_obs (x, y) points
                              (:286-289, needs H == 2*T_obs); one eps per object (K = 1, :262-263); target window = input
                              shifted one frame (T_pred == T_obs, utils/data_loader.py:206-207); per-object batch-norm
                              (bn_mode = 1); no head, no IOC (:312-313).  desire_sample / desire_forward then write
                              dev_Yhat [A, n_dec, T_obs, 2] = output_states; desire_ioc_refine is an error.  sx = sy = 1
                              reproduces the raw-pixel inputs of :216-231. */
    int32_t n_dec;         /* ref_compat only: decoder steps (the reference hard-codes 7); 0 otherwise */
    /* ---- behavioural switches (round 4: these were environment variables read inside the library; a C caller could neither see nor
     * set them).  All zero = the measured winners.  desire_set_option changes them on a live handle. ---- */
    int32_t ioc_form;      /* which form of the IOC kernel serves the shape: DESIRE_IOC_* below.  Every form computes the same function;
                              forms differ in fp32 summation order only (<= 2e-6 on trajectories), except DESIRE_IOC_COMPACT, which also
                              executes fewer products (exact zeros skipped). */
    int32_t ioc_split;     /* the BIN-SPLIT regime of the fp32 inference IOC kernel.  A call with few 32-row tiles (tiles * n <= the
                              workgroups the device keeps resident, n = 2..4) spreads the social bins of every tile over n workgroups that
                              exchange partial sums once per step: one window 3.5 -> 2.4 ms.  CONSEQUENCE: with the default (0 = auto)
                              the fp32 result of a window depends on HOW MANY windows share the call -- the partial sums of the social
                              embedding are regrouped, a difference of <= 2e-6 in trajectories and scores (inside every gate of
                              tests/, but not bit-identical across batch sizes).  1 = never split: results of a window are bit-identical
                              whatever the batch (the rounds 1-2 behaviour).  n > 1 = at most n workgroups per tile. */
    int32_t train_fp32_mask; /* dims.bf16 = 2 training step: bit mask of the parts that stay on fp32 operands instead of split-bf16 ones
                              (1 weight-gradient reductions, 2 data-gradient convolutions, 4 IOC BPTT, 8 six-product sample generation
                              in the forward pass); 0 = everything that has a split form uses it. */
    int32_t flags;         /* DESIRE_FLAG_* below */
} desire_dims;

#define DESIRE_IOC_AUTO 0          /* the measured winner per shape */
#define DESIRE_IOC_TILE64 2        /* 64-row tiles also for groups of <= 32 agents (fp32 and bf16 operands) */
#define DESIRE_IOC_CLUSTER 4       /* groups of 64 agents through the cluster form (one group = two workgroups exchanging hidden states
                                      through global memory; the default for 96 / 128 agents); bf16 operands: with column-split pooling */
#define DESIRE_IOC_CLUSTER_BINS 6  /* bf16 operands, 64 agents: the cluster form with the pooling split over bins (what 96 / 128 run) */
#define DESIRE_IOC_COMPACT 8       /* row-compacted social pooling (fp32 inference, groups of <= 32 agents, H <= 128): the pooling MFMAs
                                      run on the rows that have a neighbour in the bin only */
#define DESIRE_IOC_TRAIN_DENSE 9   /* training-mode forward with the dense pooling (default there: the compacted one) */
#define DESIRE_IOC_X6_TILE32 13    /* dims.bf16 = 3: the 32-row / three-image six-product kernel whatever the launch size */
#define DESIRE_IOC_X6_TILE64 14    /* dims.bf16 = 3: the 64-row six-product kernel whatever the launch size (default: launches of >= 256 tiles) */
#define DESIRE_FLAG_NO_FUSE34 1    /* dims.bf16 = 1: deconv3 and deconv4 as separate kernels (default: fused, d3 never written) */
#define DESIRE_FLAG_TRAIN_FWD_3P 2 /* dims.bf16 = 2, training mode: the forward pass's sample generation (GRU decoder, deconv2, deconv3) with TWO-piece
                                      operands (three bf16 products per fp32 product, the first two pieces of the same packs) instead of three pieces /
                                      six products: 67.6 instead of 70.0 ms per 81 920-sample step, every weight gradient within 5e-4 of float64
                                      autograd instead of 2e-4 (Y0 moves by ~1e-5, and the step's rounding class becomes that of the IOC kernels) */

#define DESIRE_FLAG_COMPACT_ROWS 4  /* PRESENT-ROW COMPACTION.  The loader pads every window to max_num_obj slots (utils/data_loader.py:209-229) and the reference
                                      masks id-0 objects in the cost only (model/model.py:351-366); with this bit the per-row sample-generation stages
                                      (reparameterisation, deconv1..4, mask fc, GRU decoder -- and in training their saves and their whole backward) run on
                                      the K rows of the agents PRESENT at the last observed frame only.  Rows of present agents are bit-identical to the
                                      uncompacted path (every stage is row-independent); the GRU encoders and the CVAE encoder run on the present agents too (their
                                      windows gathered into one pseudo-scene; "Hx" / "Hy" / "z_mean" of absent agents read back as zeros; not while
                                      desire_set_head_loss is on, whose term also counts objects that left before the last observed frame); rows of absent
                                      agents come back as zeros in dev_Yhat / "Y0" instead
                                      of the decode of an all-zero track.  The IOC stage keeps its scene-shaped tiles.  COUNTS: in inference with frozen
                                      batch-norm (bn_mode 0) the host never learns how many agents are present -- every compacted launch is sized for the
                                      worst case and reads its count from a device word the scan kernels wrote (round 6), so there is NO host wait and a compacted
                                      desire_forward can be captured in a hipGraph and replayed on other data.  In training, with bn_mode 1, or after
                                      desire_set_option(h, "compact_host_counts", 1) the count is read back through a mapped word instead (one host wait per
                                      desire_encode, launches sized exactly, not capturable).  Not with bn_mode = 2 (whole-batch statistics would
                                      change) or ref_compat.  The intermediates "vae_in", "z", "d1".."d3", "xhat", "xz" of desire_read_buffer are then in the COMPACT
                                      row order r' = k*P + a' (P = present agents, a' = rank of the agent among them). */

#define DESIRE_FLAG_COMPACT_IOC 8   /* IOC SLOT CLASSES.  The IOC kernels tile rows by whole (scene, k) groups of mno slots, so a window with 9 of 32 slots present
                                      still costs 32 rows per sample.  With this bit every window is re-seated in the smallest slot class m in {8, 10, 16, 32, 64, 96 (the three largest below mno; 10 = three groups per padded 32-row tile, where the padded-tile kernels serve the handle: inference, fp32 or split operands, H <= 128), mno}
                                      that holds its present agents (compacted to the front, order kept), the windows of a class run as one pseudo-batch on the
                                      same kernels, and windows without a present agent are not run (their dev_Yhat rows are left as they came, score 0); rows of
                                      absent agents inside a window likewise keep their incoming dev_Yhat and get score 0.  A class too small to fill the device
                                      (< 8192 rows) is folded into the next larger one, so -- like dims.ioc_split -- the summation order of a window's social
                                      pooling, i.e. the last bits of its result (<= 2e-6 on trajectories), depends on what else is in the batch.  Counts as for
                                      DESIRE_FLAG_COMPACT_ROWS (device-side in inference: every class is launched for the worst case at a static offset of the
                                      class buffers and an empty class's grids exit); shapes on the step-wise IOC (mno > 128; split operands at H = 256)
                                      ignore the bit. */

typedef struct desire_ctx desire_handle;

const char* desire_last_error(void);
int desire_version(void);
/* sizeof(desire_dims) as THIS library was built: a host compiled against another revision of this header (the struct has grown by appending
 * fields) compares it with its own sizeof before the first desire_create and refuses to run on a mismatch. */
int desire_dims_size(void);
/* First 16 hex digits of the sha256 over every source file under csrc, this header and the compile flags the library was built from (desire_amd/_build.py:
 * source_hash): equal to the tree's value <=> the loaded .so is the tree's code.  "unstamped" for a build that bypassed _build.py. */
const char* desire_build_hash(void);
/* Changes one of the behavioural switches of desire_dims on a live handle: name = "ioc_form", "ioc_split", "train_fp32_mask" or
 * "flags"; takes effect at the next call (a flag that changes what desire_encode prepares -- DESIRE_FLAG_COMPACT_* -- at the next
 * desire_encode).  "compact_min_rows" (not a desire_dims field): the fold threshold of DESIRE_FLAG_COMPACT_IOC.  "compact_host_counts" (not a
 * desire_dims field): 1 = inference reads the compaction counts back like training does (see DESIRE_FLAG_COMPACT_ROWS), 0 = device-side counts (default).
 * Unknown name or value out of range: DESIRE_ERR_ARG. */
int desire_set_option(desire_handle* h, const char* name, int32_t value);

/* Replaces DESIREModel.__init__/build_model graph construction (model/model.py:36-77). */
int desire_create(const desire_dims* dims, desire_handle** out);
int desire_destroy(desire_handle* h);

/* Weights by name (names/shapes: desire_amd/spec.py:weight_shapes; the reference's own names
 * are define_weights model/model.py:420-451 plus the implicit TF/prettytensor scopes).
 * host_data is n fp32 values in the natural (TF) layout; the library repacks GEMM operands
 * into MFMA B-fragment order and uploads.  desire_finalize_weights folds frozen batch-norm
 * (model/model.py:457-462) into per-channel scale/shift and verifies every tensor is set. */
int desire_set_weight(desire_handle* h, const char* name, const float* host_data, size_t n);
int desire_finalize_weights(desire_handle* h);

/* Scene feature grids rho(I) (absent in the reference, model/model.py:312-313; the `grid`
 * argument of sample(), :613).  dev_grids [n_grids, Gh, Gw, C]; host_grid_of_scene [n_scenes].
 * The device buffer is referenced, not copied: keep it alive while the handle uses it. */
int desire_set_scene_grids(desire_handle* h, const float* dev_grids, const int32_t* host_grid_of_scene);

/* Encoders + CVAE encoder (model/model.py:233-259,471-492).
 * dev_past [n_scenes, T_obs, mno, 3] and dev_fut [n_scenes, T_pred, mno, 3] are stacks of the
 * loader's windows (id, x_px, y_px; utils/data_loader.py:212-229).  dev_fut may be NULL when
 * dims.posterior == 0. */
int desire_encode(desire_handle* h, const float* dev_past, const float* dev_fut, void* stream);

/* Reparameterise + CVAE decoder + mask fc + GRU decoder (model/model.py:260-289).
 * dev_eps [R, L]; dev_Yhat [R, T_pred, 2] out (normalised coordinates). */
int desire_sample(desire_handle* h, const float* dev_eps, float* dev_Yhat, void* stream);

/* IOC scoring + regression refinement, dims.iters passes (paper; model/model.py:312-313).
 * dev_Yhat [R, T_pred, 2] in/out; dev_score [R] out.  Scenes of up to 32 agents: one persistent workgroup per 32-row tile; 64 / 96 /
 * 128: the cluster form (workgroups of a group exchanging hidden states inside one launch); 160 .. 256 (mno a multiple of 32): one
 * launch per step of the agent-sharded kernel with a single rank (fp32 operands, inference only) -- same results, more launches. */
int desire_ioc_refine(desire_handle* h, float* dev_Yhat, float* dev_score, void* stream);

/* encode + sample + ioc_refine, the unit bench.py times. */
int desire_forward(desire_handle* h, const float* dev_past, const float* dev_fut, const float* dev_eps,
                   float* dev_Yhat, float* dev_score, void* stream);

/* Intermediates kept in the handle's workspace, for parity tests:
 * "Hx" [A,H], "Hy" [A,H], "vae_in" [A,V], "z_mean" [A,L], "z_log_sigma_sq" [A,L], "z" [R,L],
 * "d1" [R,2048], "d2" [R,4096], "d3" [R,8192], "xhat" [R,1024], "xz" [R,H], "Y0" [R,T_pred,2].
 * Any other name is looked up among the handle's internal workspace buffers (training-mode saves and gradient streams, e.g.
 * "ioc_sv_h" [R,T_pred,H]; layouts as in csrc/train.hip) -- a debugging aid, not a stable interface.
 * Copies n floats to host_out after synchronising the stream. */
int desire_read_buffer(desire_handle* h, const char* name, float* host_out, size_t n, void* stream);

/* Integer paths, exposed for bit-exact tests (the same device functions the IOC kernel uses).
 * dev_pos [n_groups, mno, 2] normalised; dev_valid [n_groups, mno] uint8;
 * dev_bins [n_groups, mno, mno] int32 out (-1 = not pooled). */
/* The log-polar bin table of a bin_mode = 1 handle: out[0..grid_size-1] squared ring radii, out[8+2k], out[9+2k] = (cos, sin)
 * of sector boundary k -- the exact fp32 constants the kernels compare against (the oracle takes them as input). */
int desire_get_bin_table(desire_handle* h, float* host_out20);
int desire_neighbor_bins(desire_handle* h, const float* dev_pos, const uint8_t* dev_valid,
                         int32_t* dev_bins, int32_t n_groups, void* stream);
/* dev_pos [n, 2] -> dev_cells [n, 2] = (cy, cx). */
int desire_scene_cells(desire_handle* h, const float* dev_pos, int32_t* dev_cells, int32_t n, void* stream);

/* Scene-context CNN rho(I) (paper; the reference has no image input, model/model.py:312-313):
 * dev_image [n_grids, Hi=4*Gh, Wi=4*Gw, 3] -> dev_grids [n_grids, Gh, Gw, C] (then pass dev_grids to
 * desire_set_scene_grids).  conv5x5/16/s2+ReLU, conv5x5/32/s2+ReLU, conv5x5/C/s1, TF SAME padding. */
int desire_scene_cnn(desire_handle* h, const float* dev_image, int32_t Hi, int32_t Wi, float* dev_grids, void* stream);

/* Train-path scalars after desire_forward (posterior mode): dev_kld [A] (model/model.py:587-589 per agent),
 * dev_recon [A] = mean_k mean_{t present} ||Y_gt - Yhat_k|| (paper; the reference's NLL has undefined inputs, :342),
 * dev_cost [2] = {mean of recon+kld over the agents that count, #such agents}.  Masking rule of :351-366,374-376: an object
 * counts when it exists (id != 0 at the last observed frame) and exists in the target; with a multi-frame target that is
 * per frame -- a target frame whose id is 0 carries no ground truth and is skipped, an object absent from every target
 * frame does not count at all.  The training loss, its gradients and desire_ade_fde use the same rule. */
int desire_losses(desire_handle* h, const float* dev_fut, const float* dev_Yhat, float* dev_kld, float* dev_recon,
                  float* dev_cost, void* stream);

/* The reference's temporal convolution O1 (model/model.py:116-133; channels are (id, x), its quirk) ->
 * dev_rho [A, 200], and feature pooling O11 (:291-311) -> dev_out [R, T_pred, 200].  No consumer exists in
 * the reference; exposed as ops for callers that want the literal tensors. */
int desire_temporal_conv(desire_handle* h, const float* dev_past, float* dev_rho, void* stream);
int desire_feature_pooling(desire_handle* h, const float* dev_Yhat, const float* dev_rho, float* dev_out, void* stream);

/* ---- rows "next" of the scope table ----
 * N1: device-side window + slot builder = DataLoader.next_batch's inner loop (utils/data_loader.py:203-229).
 * dev_frames [n_frames, mno_in, 3] is one preprocessed video (frames_from_csv layout); host_starts[i] is the
 * first frame of window i; each window spans T_obs+T_pred consecutive frames; slot = rank of the id among the
 * window's sorted unique ids (np.unique semantics, 0 included when padding exists).  Outputs
 * dev_past [n_windows, T_obs, mno, 3], dev_fut [n_windows, T_pred, mno, 3].  Returns DESIRE_ERR_ARG where the
 * reference raises IndexError (:227: more unique ids than slots) or ValueError (:224-229: an id twice in one frame).
 * Synchronises the stream (error word read-back). */
int desire_build_windows(desire_handle* h, const float* dev_frames, int32_t n_frames, int32_t mno_in,
                         const int32_t* host_starts, int32_t n_windows, float* dev_past, float* dev_fut, void* stream);
/* The same with lookahead = 1: the slots are ranked over ONE MORE frame (f0 + T_obs + T_pred, when the video has it), which is
 * what DataLoader(seq_length = T_obs + T_pred).next_batch does -- it takes np.unique over seq_length + 1 frames because its target
 * is the window shifted by a frame (utils/data_loader.py:203-209) -- so the device windows equal the x of that loader exactly,
 * including its IndexError / ValueError conditions on the extra frame.  lookahead = 0 is desire_build_windows (= a loader with
 * seq_length = T_obs + T_pred - 1, x plus the last target frame). */
int desire_build_windows_la(desire_handle* h, const float* dev_frames, int32_t n_frames, int32_t mno_in,
                          const int32_t* host_starts, int32_t n_windows, int32_t lookahead, float* dev_past, float* dev_fut, void* stream);
/* N3: bivariate-Gaussian head of sample() (model/model.py:552-565,595-611,661-669): dev_params [n,5] raw head
 * outputs, dev_normals [n,2] ~ N(0,1); dev_out [n,2] sample clipped to <= 1.0. */
int desire_gaussian_sample(desire_handle* h, const float* dev_params, const float* dev_normals, float* dev_out,
                           int32_t n, void* stream);
/* N3: sample()'s autoregressive rollout (model/model.py:623-688) in one launch: warm-up over dev_past [n_scenes, T_obs, mno, 3]
 * with the X-encoder GRU (the loop :623-632), then `num` prediction steps: the 5-wide output layer "gauss_head/w|b"
 * (the reference's output_w / output_b, :315-321,445-449) reads (mux, muy, log sx, log sy, corr) off the state, a position is
 * drawn from that bivariate Gaussian with the caller's dev_normals [num, A, 2] (:661-665), clipped to <= 1.0 (:666-669) and fed
 * back as the next input (:680-681).  dev_out [num, A, 2], normalised units.  Objects with id 0 are stepped like any other
 * (the reference does the same and carries the id over, :680). */
int desire_rollout(desire_handle* h, const float* dev_past, const float* dev_normals, int32_t num, float* dev_out, void* stream);
/* N4: evaluation harness: dev_out [A,4] = (ADE mean-of-K, FDE mean-of-K, ADE best-of-K, FDE best-of-K) over the target frames
 * the object is present in (FDE: the last such frame); zeros for an object absent from every target frame. */
int desire_ade_fde(desire_handle* h, const float* dev_Yhat, const float* dev_fut, float* dev_out, void* stream);

/* ---- hipGraph capture: desire_graph_begin(h, stream); any stream-ordered desire_* calls on that stream (desire_forward,
 * desire_backward, desire_clip_grads, desire_ioc_step ...) ; desire_graph_end -> graph id; desire_graph_launch replays them with
 * the SAME device pointers.  For launch-bound shapes (small batches, the training step, the agent-sharded IOC loop).  Calls
 * that synchronise or read back to the host, and desire_adam_step (per-call step size), cannot be captured.  `stream` must be an
 * explicit non-default stream. */
int desire_graph_begin(desire_handle* h, void* stream);
int desire_graph_end(desire_handle* h, void* stream, int32_t* graph_id);
int desire_graph_launch(desire_handle* h, int32_t graph_id, void* stream);

/* ---- agent-sharded IOC (BASELINE north_star / SURVEY.md 8(e) E1: "agents shard across the GPUs with an RCCL all-gather
 * only for the social-pooling neighbour exchange").  The handle of rank g is created with mno = the slots it owns
 * (m_loc); every scene then has nranks * m_loc agents.  Everything before the IOC is per-agent and runs unchanged
 * (desire_encode / desire_sample); one IOC time step is one desire_ioc_step call, and the CALLER all-gathers the hidden
 * states in between (desire_amd/dist.py: ShardedIoc does it with torch.distributed = RCCL).  Gathered layouts, rank-major:
 *   Yall      [nranks][n_scenes*K][m_loc][T_pred][2]   decoded positions of every rank   (gathered once per pass)
 *   plast_all [nranks][n_scenes][m_loc][2]             last observed positions           (gathered once)
 *   valid_all [nranks][n_scenes][m_loc]                presence flags, uint8              (gathered once)
 *   Hall      [nranks][n_scenes*K*m_loc][H]            h_{t-1} of every rank              (gathered before every step)
 * dev_h_state [R_loc, H] (in: h_{t-1} of the local rows, out: h_t) and dev_score_state [R_loc] are the caller-held
 * recurrent state; desire_ioc_finish applies the regression head (Y += dY) and writes the scores.
 * desire_device_buffer exposes the handle's workspace tensors ("HxHy" [A,2H], "p_last" [A,2], "valid" [A] uint8,
 * "Y0" [R,T,2]) so they can be gathered without a host round trip. */
int desire_device_buffer(desire_handle* h, const char* name, void** dev_ptr, size_t* bytes);
int desire_ioc_step(desire_handle* h, int32_t t, int32_t rank, int32_t nranks, const float* dev_Yall,
                    const float* dev_plast_all, const uint8_t* dev_valid_all, const float* dev_Hall,
                    float* dev_h_state, float* dev_score_state, void* stream);
int desire_ioc_finish(desire_handle* h, const float* dev_h_state, const float* dev_score_state, float* dev_Y,
                      float* dev_score, void* stream);
/* The same agent-sharded pass WITHOUT a collective and without the host in the step loop: every rank owns an exchange region (its
 * blocks of the four gathered arrays above, two parities of hidden states, one progress counter), exported as a 64-byte
 * hipIpcMemHandle by desire_peer_export and mapped by the other ranks with desire_peer_open(h, rank, nranks, peer, handle) -- over xGMI
 * when the peer is another GPU, the same HBM when two ranks share a device.  How the handles travel is the caller's business
 * (desire_amd/dist.py: PeerShardedIoc uses torch.distributed.all_gather_object once); open the own rank with a NULL handle.  After
 * every rank has opened every peer (barrier), desire_ioc_peer_pass(h, dev_Y inout [R_loc, T_pred, 2], dev_score out [R_loc], stream)
 * enqueues the whole refinement -- publish, then per step: one-wave wait on the peers' counters, the step kernel reading their hidden
 * states IN PLACE, counter bump -- dims.iters times, with nothing but kernel launches (capturable by desire_graph_*).  A peer that
 * never arrives makes the bounded wait give up (a few seconds); the NEXT call then fails with DESIRE_ERR_HIP.  Results are bit-identical to
 * the desire_ioc_step loop.  desire_peer_close unmaps / frees (desire_destroy calls it). */
int desire_peer_export(desire_handle* h, uint8_t* handle_out64, size_t* bytes_out);
int desire_peer_open(desire_handle* h, int32_t rank, int32_t nranks, int32_t peer, const uint8_t* handle64);
int desire_ioc_peer_pass(desire_handle* h, float* dev_Y, float* dev_score, void* stream);
int desire_peer_close(desire_handle* h);
/* Whether a bounded wait of the passes enqueued so far has given up (*timed_out = 1): meaningful once the caller has synchronised the stream
 * the pass ran on -- a pass is stream-ordered, so this is the earliest point at which ITS outcome is known; without the query a caller only
 * learns of a timed-out pass from the next desire_ioc_peer_pass.  Does not clear the condition. */
int desire_peer_status(desire_handle* h, int32_t* timed_out);
/* Ranks inside ONE process (one process driving several devices with peer access enabled) attach each other's regions by device
 * pointer: desire_peer_region gives a handle's own region, desire_peer_open_ptr takes the peer's (hipIpc handles cannot be opened by
 * the process that exported them).  Every rank's pass must then run on a HARDWARE queue of its own: a pass parks a one-wave wait
 * kernel on its stream until the peers catch up, and two streams that the runtime multiplexes onto one queue (it maps a process's
 * streams of ONE device onto 4 queues) would wait for each other until the time-out.  Streams of different devices never share a
 * queue; several ranks on one device belong in separate processes (tests/test_gpu_peer_ioc.py). */
int desire_peer_region(desire_handle* h, void** dev_region, size_t* bytes);
int desire_peer_open_ptr(desire_handle* h, int32_t rank, int32_t nranks, int32_t peer, void* dev_region);

/* ---- training (the reference builds tf.gradients(cost) + Adam and never runs them: model/model.py:388-403) ----
 * desire_set_training(h,1) allocates the activation-save and gradient buffers; a desire_forward made afterwards keeps
 * what backward needs.  desire_backward computes d(loss)/d(weight) for the loss of DESIGN.md section 8 into ONE flat
 * fp32 device buffer (natural TF layouts; desire_grad_buffer exposes it so a multi-GPU caller can all-reduce it).
 * Frozen batch-norm parameters are constants (no gradient).  desire_set_training(h,0) hands the trained master weights back to
 * the handle (desire_get_weight returns them, inference keeps using the device operands the last desire_adam_step rebuilt);
 * enabling training again starts from them with the Adam moments at zero (desire_adam_state to carry moments across). */
int desire_set_training(desire_handle* h, int enable);
/* The reference's OWN loss for its 5-wide output layer (model/model.py:315-366: output_w / output_b on the recurrent state, get_coef
 * :552-565, -log max(N(next position | mux, muy, sx, sy, rho), 1e-20) :494-550, id == 0 masking :351-366, mean :374-376) as an extra
 * term of the training loss: weight x mean over the counted (object, observed frame) pairs, teacher-forced over the X encoder's
 * observed steps (target of frame t = the position in frame t + 1; frame T_obs is the first future frame).  desire_backward then
 * also fills the gradients of "gauss_head/w|b" -- the head desire_rollout / sample() read, which no other term reaches -- and adds
 * the term's gradient to the X encoder's, step by step.  weight = 0 (default) switches the term off. */
int desire_set_head_loss(desire_handle* h, float weight);
int desire_backward(desire_handle* h, const float* dev_past, const float* dev_fut, const float* dev_eps, void* stream);
int desire_get_grad(desire_handle* h, const char* name, float* host_out, size_t n, void* stream);
int desire_grad_buffer(desire_handle* h, float** dev_ptr, size_t* n);
/* Loss terms of the last training-mode forward, means over present agents:
 * host_out5 = {recon, kld, cross_entropy, regression, n_present};  loss = sum of the first four. */
int desire_train_loss(desire_handle* h, const float* dev_fut, float* host_out5, void* stream);
/* The same values written to a DEVICE buffer dev_out8 [8] = {recon, kld, cross_entropy, regression, n_present, nll_head, grad_norm,
 * n_head}, stream-ordered and without synchronising: a training loop that reads the loss one step late (desire_amd/train.py) never
 * stalls the host behind the step it has just enqueued.  nll_head / n_head: the weighted Gaussian-head term of the last
 * desire_backward and the (object, frame) pairs it averaged over (0 while that loss is off); grad_norm: the pre-clip global norm of
 * the last desire_clip_grads. */
int desire_train_loss_async(desire_handle* h, const float* dev_fut, float* dev_out8, void* stream);
/* tf.clip_by_global_norm (model/model.py:390) on the flat gradient buffer, in place; host_norm_out (optional, may be
 * NULL: then no synchronisation) receives the pre-clip norm. */
int desire_clip_grads(desire_handle* h, float max_norm, float* host_norm_out, void* stream);
/* One tf.train.AdamOptimizer update (model/model.py:394; lr_t = lr*sqrt(1-b2^t)/(1-b1^t)) of the device master weights
 * from the flat gradient buffer, followed by a device-side rebuild of every packed operand.  No host round trip. */
int desire_adam_step(desire_handle* h, float lr, float beta1, float beta2, float eps, void* stream);
/* Optimiser state for checkpoints (the reference's Saver stores every global variable, Adam slots included, train.py:114): the
 * moments are the workspace tensors "Mflat" / "Vflat" (desire_device_buffer; the gradient buffer's flat layout), the step
 * counter of the bias correction is read (set = 0) or written (set = 1) through *step. */
int desire_adam_state(desire_handle* h, int32_t* step, int set);
/* Current value of a weight (the trained one in training mode), natural TF layout. */
int desire_get_weight(desire_handle* h, const char* name, float* host_out, size_t n, void* stream);

/* Per-kernel GPU time measured with hipEvents on the launch stream (enabled by
 * desire_set_profiling(h,1); adds two event records per kernel, nothing else).  Entries accumulate
 * over calls; desire_get_profile synchronises on them, copies up to *count (in: capacity) entries
 * to host_ms[i] / host_names[i], sets *count, and clears the list.  With host_ms == host_names ==
 * NULL it only reports the pending entry count. */
int desire_set_profiling(desire_handle* h, int enable);
int desire_get_profile(desire_handle* h, float* host_ms, const char** host_names, int32_t* count);

#ifdef __cplusplus
}
#endif
#endif /* DESIRE_HIP_H */
