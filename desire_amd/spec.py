"""Frozen dimension / weight contract of the DESIRE hot path (host side, numpy only).

The reference never finishes its graph, so the arithmetic this build
commits to is written down here and in DESIGN.md section 2 ("frozen spec").  Reference anchors:

* args fields and defaults ............ /root/reference/train.py:30-88, model/model.py:44-60
* weight names/shapes the ref defines .. model/model.py:420-451 (temporal_w, w_hidden_enc1,
                                          w_post_vae) + implicit TF/prettytensor scopes
                                          model/model.py:126,136,143,233,238,257,279
* CVAE layer stack ..................... model/model.py:453-492
* the IOC / scene / social block ....... absent in the reference (model/model.py:312-313);
                                          paper-defined, frozen in DESIGN.md section 2.

Nothing here touches the GPU.  `Dims` is mirrored 1:1 by `struct desire_dims` in
include/desire_hip.h.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

BN_EPS = 1e-3  # variance_epsilon, model/model.py:460,479


IOC_AUTO, IOC_TILE64, IOC_CLUSTER, IOC_CLUSTER_BINS, IOC_COMPACT, IOC_TRAIN_DENSE, IOC_X6_TILE32, IOC_X6_TILE64 = 0, 2, 4, 6, 8, 9, 13, 14   # desire_hip.h: DESIRE_IOC_*
FLAG_NO_FUSE34 = 1
FLAG_COMPACT_ROWS = 4     # per-row sample-generation stages on the rows of present agents only (include/desire_hip.h)
FLAG_COMPACT_IOC = 8      # IOC stage on slot classes: windows re-seated in the smallest of {8, 16, 32, mno} slots that holds their present agents
FLAG_TRAIN_FWD_3P = 2     # dims.bf16 = 2 training: two-piece operands in the forward's sample generation (include/desire_hip.h)


@dataclass(frozen=True)
class Dims:
    """Static sizes of one forward call.  Row index r = (scene*K + k)*mno + slot."""

    n_scenes: int = 1      # windows per call (DataLoader batch entries), each its own scene
    mno: int = 32          # agent slots per window (max_num_obj): divides 32, or a multiple of 32 up to 256 (above 128: inference, step-wise IOC)
    K: int = 20            # samples per agent
    T_obs: int = 8
    T_pred: int = 40
    H: int = 128           # GRU hidden = args.d_dim (model/model.py:51,137)
    L: int = 128           # latent_size
    S: int = 32            # CVAE image side = int(sqrt(2*rnn_size)) (model/model.py:57-58)
    C: int = 32            # scene feature channels
    Gh: int = 64           # scene grid rows
    Gw: int = 64           # scene grid cols
    n_grids: int = 1       # distinct scene feature grids resident
    grid_size: int = 4     # social grid side (train.py:71-72), B = grid_size**2
    E_v: int = 16          # velocity embedding width
    iters: int = 1         # IOC refinement iterations
    posterior: int = 1     # 1: z = mu + sigma*eps (needs future), 0: z = eps (prior)
    nb_w: float = 0.25     # social window width  (normalised units, = neighborhood_size*sx)
    nb_h: float = 0.25     # social window height (normalised units)
    sx: float = 1.0        # pixel -> normalised scale, x
    sy: float = 1.0        # pixel -> normalised scale, y
    bin_mode: int = 0      # social bins: 0 rectangular window nb_w x nb_h; 1 log-polar (rings nb_h .. nb_w x sectors)
    bn_mode: int = 0       # CVAE batch-norm: 0 frozen moving statistics; 1 per-object statistics (the reference's batch of one); 2 whole-batch statistics
    bf16: int = 0          # 1: bf16 MFMA operands (fp32 accumulate / state), inference only; 2: split-bf16 operands (hi + lo, three bf16
                           #    MFMAs per product): fp32-equivalent results (~1e-5) from the bf16 matrix pipe where a kernel has that form;
                           #    3: three bf16 pieces, six MFMAs per product: fp32-class accuracy (inference only)
    ref_compat: int = 0    # 1: the reference graph as written (model/model.py:116-311): n_dec decoder states re-read as T_obs points
    n_dec: int = 0         # ref_compat only: decoder steps (the reference hard-codes 7, model/model.py:280)
    # behavioural switches (include/desire_hip.h: desire_dims; all zero = the measured winners; Handle.set_option changes them live)
    ioc_form: int = 0      # IOC_* below: which form of the IOC kernel serves the shape
    ioc_split: int = 0     # bin-split regime of the fp32 inference IOC kernel: 0 auto (a window's result then depends on the batch it is in, <= 2e-6),
                           #    1 never (bit-identical across batch sizes), 2..4 cap on the workgroups per tile
    train_fp32_mask: int = 0   # dims.bf16 = 2 training: parts kept on fp32 operands (1 weight gradients, 2 data-gradient convs, 4 IOC BPTT, 8 sample generation)
    flags: int = 0         # FLAG_NO_FUSE34 = 1: bf16 deconv3 / deconv4 as separate kernels; FLAG_TRAIN_FWD_3P = 2: see the header

    @property
    def A(self) -> int:
        return self.n_scenes * self.mno

    @property
    def R(self) -> int:
        return self.A * self.K

    @property
    def V(self) -> int:
        return self.S * self.S

    @property
    def B(self) -> int:
        return self.grid_size * self.grid_size

    @property
    def E(self) -> int:
        """IOC GRU input width: [velocity embed | scene feature | social embed]."""
        return self.E_v + self.C + self.H

    def validate(self) -> None:
        if self.S != 32:
            raise ValueError("CVAE stack closes only for S=32 (rnn_size=512), model/model.py:465-468")
        if not (1 <= self.mno <= 256) or (32 % self.mno if self.mno <= 32 else self.mno % 32):
            raise ValueError("mno must divide 32 or be a multiple of 32 up to 256 (host pads max_num_obj up; above 128: inference only)")
        if self.mno > 128 and self.bf16 == 1:
            raise ValueError("more than 128 agents per scene run the fp32 step-wise IOC (bf16 = 0, 2 or 3)")
        if self.L % 8 or self.C % 8 or self.E_v % 8:
            raise ValueError("L%8, C%8, E_v%8 required by the MFMA tiling")
        if self.H not in (16, 32, 64, 128, 256):
            raise ValueError("H must be 16, 32 (run zero-padded on the 64-wide recurrent tile), 64, 128 or 256")
        if self.ref_compat:
            if not (self.K == 1 and self.posterior and self.bn_mode == 1 and not self.bf16 and self.n_dec >= 1
                    and self.H == 2 * self.T_obs and self.T_pred == self.T_obs):
                raise ValueError("ref_compat (model/model.py:116-311): K=1, posterior, bn_mode=1, fp32, H == 2*T_obs, T_pred == T_obs, n_dec >= 1")
        elif self.n_dec:
            raise ValueError("n_dec belongs to ref_compat")
        if self.C != 32 or self.E_v != 16:
            raise ValueError("C=32, E_v=16 are the instantiated IOC widths in this round")
        if min(self.n_scenes, self.K, self.T_obs, self.T_pred, self.n_grids, self.iters) < 1:
            raise ValueError("sizes must be >= 1")

    def replace(self, **kw) -> "Dims":
        return dataclasses.replace(self, **kw)


def flops_per_sample(d: Dims) -> float:
    """Algorithmic FLOPs per agent-trajectory-sample (multiply-add = 2),
    re-derived for this spec's dims.  Used by bench.py for the MFMA roofline."""
    H, L, V, Tp, To, K = d.H, d.L, d.V, d.T_pred, d.T_obs, d.K
    f_dec_cvae = 2.0 * (16 * L * 128 + 16 * 128 * 25 * 64 + 64 * 64 * 25 * 32 + 256 * 32 * 25 * 1)
    f_mask = 2.0 * V * H
    f_dec_gru = Tp * (6.0 * H * 2 * H + 4 * H)
    f_ioc = d.iters * (Tp * (6.0 * H * (d.E + H) + 2.0 * d.B * H * H + 2 * H + 2 * 2 * d.E_v)
                       + 2.0 * H * 2 * Tp)
    f_enc_cvae = 2.0 * (256 * 25 * 32 + 64 * 25 * 32 * 64 + 16 * 25 * 64 * 128 + 2048 * 2 * L)
    per_agent = To * 6.0 * H * (2 + H) + 4.0 * H * V
    if d.posterior:
        per_agent += Tp * 6.0 * H * (2 + H) + f_enc_cvae
    return f_dec_cvae + f_mask + f_dec_gru + f_ioc + per_agent / K


# ------------------------------------------------------------------------------------------------
# weight registry: name -> shape.  All fp32.  GRU layout follows TF-1.x GRUCell:
#   gates/kernel [(in+H), 2H] (columns r|u), gates/bias [2H] (init 1.0),
#   candidate/kernel [(in+H), H], candidate/bias [H]; rows ordered [input ; state].
# conv kernels HWIO; transposed-conv kernels [kh, kw, out, in] (conv2d_transpose layout,
# utils/convolutional_vae_util.py:83).  bn = (beta, gamma, moving_mean, moving_var).
# ------------------------------------------------------------------------------------------------
def weight_shapes(d: Dims) -> Dict[str, Tuple[int, ...]]:
    H, L, V = d.H, d.L, d.V
    s: Dict[str, Tuple[int, ...]] = {}

    def gru(prefix: str, n_in: int) -> None:
        s[prefix + "/gates/kernel"] = (n_in + H, 2 * H)
        s[prefix + "/gates/bias"] = (2 * H,)
        s[prefix + "/candidate/kernel"] = (n_in + H, H)
        s[prefix + "/candidate/bias"] = (H,)

    def bn(prefix: str, c: int) -> None:
        for n in ("beta", "gamma", "moving_mean", "moving_var"):
            s[prefix + "/bn/" + n] = (c,)

    gru("enc_x", 2)                      # model/model.py:136-141,233-236
    gru("enc_y", 2)                      # model/model.py:143-148,238-241
    s["fc_c/w"] = (2 * H, V)             # w_hidden_enc1, model/model.py:434-437
    s["fc_c/b"] = (V,)
    for name, kk, ci, co in (("conv1", 5, 1, 32), ("conv2", 5, 32, 64), ("conv3", 5, 64, 128)):
        s[f"vae_enc/{name}/w"] = (kk, kk, ci, co)     # model/model.py:484-486
        s[f"vae_enc/{name}/b"] = (co,)
        bn(f"vae_enc/{name}", co)
    s["vae_enc/fc/w"] = (2048, 2 * L)    # model/model.py:488
    s["vae_enc/fc/b"] = (2 * L,)
    for name, kk, co, ci in (("deconv1", 4, 128, L), ("deconv2", 5, 64, 128),
                             ("deconv3", 5, 32, 64), ("deconv4", 5, 1, 32)):
        s[f"vae_dec/{name}/w"] = (kk, kk, co, ci)     # model/model.py:465-468
        s[f"vae_dec/{name}/b"] = (co,)
        bn(f"vae_dec/{name}", co)
    s["mask_fc/w"] = (V, H)              # w_post_vae, model/model.py:439-443
    s["mask_fc/b"] = (H,)
    gru("dec", H)                        # scope hidden_states, model/model.py:279-285
    s["head/w"] = (H, 2)                 # the ref's commented-out output layer, :315-321,445-449
    s["head/b"] = (2,)
    # ---- IOC block (paper-defined) ----
    s["ioc/vel_fc/w"] = (2, d.E_v)
    s["ioc/vel_fc/b"] = (d.E_v,)
    s["ioc/social_fc/w"] = (d.B * H, H)
    s["ioc/social_fc/b"] = (H,)
    gru("ioc", d.E)
    s["ioc/score/w"] = (H, 1)
    s["ioc/score/b"] = (1,)
    s["ioc/reg/w"] = (H, 2 * d.T_pred)
    s["ioc/reg/b"] = (2 * d.T_pred,)
    # ---- scene-context CNN rho(I) (paper; absent in the reference): image [4Gh,4Gw,3] -> [Gh,Gw,C] ----
    s["scene_cnn/conv1/w"] = (5, 5, 3, 16)
    s["scene_cnn/conv1/b"] = (16,)
    s["scene_cnn/conv2/w"] = (5, 5, 16, 32)
    s["scene_cnn/conv2/b"] = (32,)
    s["scene_cnn/conv3/w"] = (5, 5, 32, d.C)
    s["scene_cnn/conv3/b"] = (d.C,)
    # ---- temporal convolution O1 (model/model.py:116-133, weights :427-431) ----
    s["temporal/w"] = (1, d.T_obs, 2, 100)
    s["temporal/b"] = (200,)
    # ---- sample()'s 5-wide output layer (mux, muy, log sx, log sy, corr): the reference's commented-out output_w / output_b
    # (model/model.py:315-321,445-449), read by the autoregressive rollout (:643-681).  LAST on purpose: init_weights draws
    # in this order, so everything above keeps the values the committed goldens were made with. ----
    s["gauss_head/w"] = (H, 5)
    s["gauss_head/b"] = (5,)
    return s


def init_weights(d: Dims, seed: int = 0, ref_init: bool = False) -> Dict[str, np.ndarray]:
    """Deterministic random-init weights.

    ref_init=True reproduces the reference's literal initialisers where it states them
    (N(0,1) for fc_c / mask_fc, model/model.py:434-443) -- those saturate every downstream
    nonlinearity, so the default uses fan-in scaled normals (what xavier_init does for the
    prettytensor layers, utils/convolutional_vae_util.py:86-90) everywhere.  GRU gate bias is
    1.0 as in TF GRUCell (bias_start=1.0)."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for name, shape in weight_shapes(d).items():
        if name.endswith("gates/bias"):
            a = np.ones(shape, np.float32)
        elif name.endswith("/bn/gamma"):
            a = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("/bn/moving_var"):
            a = (0.5 + rng.random(shape)).astype(np.float32)
        elif name.endswith("/bn/beta") or name.endswith("/bn/moving_mean"):
            a = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("/b") or name.endswith("/bias"):
            a = (0.05 * rng.standard_normal(shape)).astype(np.float32)
        elif ref_init and name in ("fc_c/w", "mask_fc/w"):
            a = rng.standard_normal(shape).astype(np.float32)
        else:
            if name == "temporal/w":
                fan_in = shape[1]
            elif len(shape) == 4:        # conv / deconv: fan-in = kh*kw*in
                fan_in = shape[0] * shape[1] * (shape[3] if "vae_dec" in name else shape[2])
            else:
                fan_in = shape[0]
            a = (rng.standard_normal(shape) / np.sqrt(max(fan_in, 1))).astype(np.float32)
        w[name] = np.ascontiguousarray(a)
    return w


def fold_bn(w: Dict[str, np.ndarray], prefix: str) -> Tuple[np.ndarray, np.ndarray]:
    """Frozen (inference-phase) batch-norm + layer bias folded to per-channel (scale, shift):
    y = scale * conv + shift, scale = gamma/sqrt(var+eps), shift = beta + scale*(b - mean).
    prettytensor batch_normalize with scale_after_normalization=True, variance_epsilon=1e-3
    (model/model.py:457-462).  Computed in float64 then rounded once, identically on every
    caller (oracle and product both call this function)."""
    g = w[prefix + "/bn/gamma"].astype(np.float64)
    beta = w[prefix + "/bn/beta"].astype(np.float64)
    mean = w[prefix + "/bn/moving_mean"].astype(np.float64)
    var = w[prefix + "/bn/moving_var"].astype(np.float64)
    b = w[prefix + "/b"].astype(np.float64)
    scale = g / np.sqrt(var + BN_EPS)
    shift = beta + scale * (b - mean)
    return scale.astype(np.float32), shift.astype(np.float32)
