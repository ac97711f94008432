"""DESIREModel -- host-side mirror of the reference's model/model.py:29-688 public surface, backed
by libdesire_hip.so.

Kept from the reference:
  * constructor takes the argparse Namespace of train.py:30-88 (seq_length, d_dim, rnn_size,
    latent_size, max_num_obj, learning_rate, grad_clip, neighborhood_size, grid_size, ...);
  * attribute names input_data / target_data / cost / learning_rate / gru_states / final_states /
    final_output exist (here: the last fed arrays / last results, not TF tensors);
  * sample(sess, traj, grid, dimensions, true_traj, num) -> ndarray [obs+num, max_num_obj, 3]
    (model/model.py:613-688).

New (the reference has one seq_length, K hard-coded to 7, no IOC): args.pred_length, args.num_samples,
args.ioc_iters, args.img_width/img_height, args.bf16 (True/1: bf16 matrix operands, inference only; 2 or "x3": split-bf16 operands, fp32-equivalent), and forward() that returns
all K refined samples + scores.

PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .spec import Dims, init_weights, weight_shapes


def _next_divisor_of_64(n: int) -> int:
    """Pad max_num_obj up to a slot count the IOC tiling accepts: a divisor of 32, or a multiple of 32 up to 256 (above 128 the IOC pass
    runs step-wise and training is refused)."""
    for m in (1, 2, 4, 8, 16, 32, 64, 96, 128, 160, 192, 224, 256):
        if m >= n:
            return m
    raise ValueError("max_num_obj > 256 is not supported")


def dims_from_args(args, n_scenes: int, posterior: bool = True, ref_compat: bool = False) -> Dims:
    S = int(np.sqrt(2 * args.rnn_size))                       # model/model.py:57-58
    mno = _next_divisor_of_64(int(args.max_num_obj))
    if ref_compat:
        # the reference graph as written (model/model.py:116-311): raw pixels (:216-231), one eps per object (:262-263), 7 decoder
        # states re-read as seq_length points (:280-289), batch-norm in train phase on a batch of one object (:453,471)
        return Dims(n_scenes=n_scenes, mno=mno, K=1, T_obs=int(args.seq_length), T_pred=int(args.seq_length), H=int(args.d_dim),
                    L=int(args.latent_size), S=S, grid_size=int(getattr(args, "grid_size", 4)), posterior=1, sx=1.0, sy=1.0,
                    bn_mode=1, ref_compat=1, n_dec=int(getattr(args, "n_dec", 7)))
    w_img = float(getattr(args, "img_width", 2048.0))
    h_img = float(getattr(args, "img_height", 2048.0))
    nb = float(getattr(args, "neighborhood_size", 32))
    return Dims(
        n_scenes=n_scenes, mno=mno, K=int(getattr(args, "num_samples", 20)),
        T_obs=int(args.seq_length), T_pred=int(getattr(args, "pred_length", None) or args.seq_length),
        H=int(args.d_dim), L=int(args.latent_size), S=S,
        C=int(getattr(args, "scene_channels", 32)), Gh=int(getattr(args, "scene_grid", 64)),
        Gw=int(getattr(args, "scene_grid", 64)), n_grids=int(getattr(args, "n_grids", 1)),
        grid_size=int(getattr(args, "grid_size", 4)), E_v=16, iters=int(getattr(args, "ioc_iters", 1)),
        posterior=int(posterior), nb_w=nb / w_img, nb_h=nb / h_img, sx=1.0 / w_img, sy=1.0 / h_img,
        bin_mode=int(getattr(args, "social_layout", "rect") == "logpolar"),
        bn_mode={"frozen": 0, "per_object": 1, "batch": 2}[getattr(args, "batch_norm", "frozen")],
        bf16=_operand_mode(getattr(args, "bf16", False)),
        flags=_flags_from_args(args, {"frozen": 0, "per_object": 1, "batch": 2}[getattr(args, "batch_norm", "frozen")]))                          # DESIRE_FLAG_* bits: padding skipped unless --keep_padding / an explicit args.dims_flags


def _flags_from_args(args, bn_mode: int = 0) -> int:
    """desire_dims.flags from the argparse Namespace.  PADDING IS SKIPPED BY DEFAULT (round 6): the loader pads every window to max_num_obj slots
    (utils/data_loader.py:209-229; train.py:74 default 60, SDD frames hold ~8 objects) and the reference masks id-0 objects in the cost only
    (model/model.py:351-366), so DESIRE_FLAG_COMPACT_ROWS | DESIRE_FLAG_COMPACT_IOC -- present rows bit-identical, absent rows zeros -- is on unless
      * args.keep_padding is set (train.py --keep_padding), or
      * args.dims_flags is given (an int: taken literally, plus train.py's --two_piece_forward / --skip_padding bits), or
      * the batch-norm mode is 'batch' (whole-batch statistics depend on the padding rows: the library refuses the combination)."""
    from .spec import FLAG_COMPACT_IOC, FLAG_COMPACT_ROWS, FLAG_TRAIN_FWD_3P
    given = getattr(args, "dims_flags", None)
    f = int(given or 0)
    if getattr(args, "two_piece_forward", False):
        f |= FLAG_TRAIN_FWD_3P
    if getattr(args, "skip_padding", False) or (given is None and not getattr(args, "keep_padding", False) and bn_mode != 2):
        f |= FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC
    if getattr(args, "keep_padding", False):
        f &= ~(FLAG_COMPACT_ROWS | FLAG_COMPACT_IOC)
    return f


def _operand_mode(v) -> int:
    """args.bf16 -> dims.bf16: False/0 fp32 operands, True/1 bf16 operands, 2 / "x3" / "split" split-bf16 operands (three products),
    3 / "x6" three bf16 pieces (six products, fp32-class accuracy)."""
    if isinstance(v, str):
        return {"": 0, "0": 0, "f32": 0, "1": 1, "bf16": 1, "2": 2, "x3": 2, "split": 2, "3": 3, "x6": 3}[v.lower()]
    return int(v) if int(v) in (0, 1, 2, 3) else 1


class PendingLoss(object):
    """Loss terms of a training step that has been ENQUEUED, not waited for.  get() blocks on the step's event (at most once) and returns
    the same dict train_step(sync=True) returns."""

    def __init__(self, host8, event, dev8):
        self._host8, self._event, self._dev8, self._terms = host8, event, dev8, None

    def done(self) -> bool:
        return self._terms is not None or self._event.query()

    def get(self) -> Dict[str, float]:
        if self._terms is None:
            self._event.synchronize()
            self._terms = _lib.Handle.loss_terms(self._host8.tolist())
            self._dev8 = None
        return self._terms


class DESIREModel(object):
    """Drop-in for model/model.py:29 DESIREModel(args)."""

    def __init__(self, args, weights: Optional[Dict[str, np.ndarray]] = None, seed: int = 0):
        import torch
        if not torch.cuda.is_available():
            raise _lib.DesireError("DESIREModel needs an MI355X: there is no CPU path (oracle/ is test-only)")
        self.args = args
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.learning_rate = float(args.learning_rate)
        self.max_num_obj = int(args.max_num_obj)
        self.seq_length = int(args.seq_length)
        self.batch_size = int(getattr(args, "batch_size", 1))
        self._weights = weights
        self._head_given = bool(weights) and "gauss_head/w" in weights      # (restore() clears it when it had to fill the head in)
        self._seed = seed
        self._handles: Dict[Tuple[int, int, int], _lib.Handle] = {}
        self._trained = None                 # the handle train_step updates (weights + Adam moments live on the device)
        self._version = 0                    # bumped by every optimiser step
        self._weights_ver = 0                # version self._weights (host copy) corresponds to
        self._grids = None
        self._grid_of_scene = None
        # reference attribute names (model/model.py:62-75)
        self.input_data = None
        self.target_data = None
        self.cost = None
        self.gru_states = None
        self.final_states = None
        self.final_output = None

    # ---- plumbing -------------------------------------------------------------------------------
    def _weights_for(self, d: Dims) -> Dict[str, np.ndarray]:
        """self._weights in the shapes `d` wants.  Only ref_compat handles can differ from the model's own dims (they force K = 1 and
        T_pred = T_obs, so the IOC regression head [H, 2*T_pred] -- which ref_compat never runs -- is cut / zero-padded)."""
        want = weight_shapes(d)
        out = {}
        for k, v in self._weights.items():
            shp = want.get(k)
            v = np.asarray(v)
            if shp is None or tuple(v.shape) == tuple(shp):
                out[k] = v
                continue
            if not d.ref_compat or v.ndim != len(shp):
                raise ValueError("weight %s has shape %s, this call needs %s" % (k, v.shape, shp))
            fit = np.zeros(shp, np.float32)
            sl = tuple(slice(0, min(a, b)) for a, b in zip(v.shape, shp))
            fit[sl] = v[sl]
            out[k] = fit
        return out

    def _handle(self, n_scenes: int, posterior: bool, ref_compat: bool = False) -> _lib.Handle:
        key = (n_scenes, int(posterior), int(ref_compat))
        if key not in self._handles:
            d = dims_from_args(self.args, n_scenes, posterior, ref_compat)
            h = _lib.Handle(d)
            if self._weights is None:
                # always drawn for the model's OWN dims (a first call through forward_ref_compat must not size them for its T_pred = T_obs)
                self._weights = init_weights(dims_from_args(self.args, n_scenes, posterior, False), self._seed)
            h.set_weights(self._weights_for(d) if ref_compat else self._weights)
            h._wver = getattr(self, "_version", 0) if getattr(self, "_weights_ver", 0) == getattr(self, "_version", 0) else -1
            self._handles[key] = h
        h = self._handles[key]
        trained = getattr(self, "_trained", None)
        if trained is not None and h is not trained and getattr(h, "_wver", 0) != self._version:
            # the optimiser updates the weights inside the handle it trains on; any other handle (another batch size, the
            # prior path) gets the current values before it runs
            h.set_weights(self._weights_for(h.dims) if h.dims.ref_compat else self.sync_weights())
            h._wver = self._version
        return h

    def set_scene_grids(self, grids: np.ndarray, grid_of_scene: Sequence[int]) -> None:
        """grids [n_grids, Gh, Gw, C] scene features rho(I); grid_of_scene[i] = grid index of window i."""
        self._grids = self.torch.as_tensor(np.ascontiguousarray(grids, np.float32), device=self.device)
        self._grid_of_scene = np.asarray(grid_of_scene, np.int32)

    def _pad_windows(self, batch: Sequence[np.ndarray], mno: int):
        x = np.stack([np.asarray(b) for b in batch]).astype(np.float32)      # [n, T, MNO, 3]
        if x.ndim != 4 or x.shape[3] != 3:
            raise ValueError("windows must be [T, MNO, 3] arrays (id, x, y), got %s" % (x.shape[1:],))
        if x.shape[2] > mno:
            raise ValueError("windows hold %d object slots, the model was built for max_num_obj=%d" % (x.shape[2], self.max_num_obj))
        if x.shape[2] < mno:
            x = np.concatenate([x, np.zeros(x.shape[:2] + (mno - x.shape[2], 3), np.float32)], axis=2)
        return self.torch.as_tensor(np.ascontiguousarray(x), device=self.device)

    # ---- the hot path ---------------------------------------------------------------------------
    def forward(self, x_batch: Sequence[np.ndarray], y_batch: Optional[Sequence[np.ndarray]] = None,
                eps: Optional[np.ndarray] = None, seed: int = 0):
        """x_batch: loader windows [T_obs, MNO, 3] (DataLoader.next_batch x); y_batch: future windows
        [T_pred, MNO, 3] or None (prior sampling).  Returns (Yhat [n, K, mno, T_pred, 2] normalised,
        score [n, K, mno]) as torch tensors on the GPU."""
        posterior = y_batch is not None
        d = self._handle(len(x_batch), posterior).dims
        past = self._pad_windows(x_batch, d.mno)
        fut = self._pad_windows(y_batch, d.mno) if posterior else None
        out = self.forward_device(past, fut, eps, seed)
        self.input_data, self.target_data = x_batch, y_batch
        return out

    def forward_device(self, past, fut=None, eps=None, seed: int = 0):
        """forward() on windows that are already in HBM: past [n, T_obs, mno, 3], fut [n, T_pred, mno, 3] or None -- float32 device
        tensors in the loader's layout with the slot axis padded to the model's mno (what desire_amd.prefetch's feeders and
        forward_from_video produce).  Nothing here touches the host except the launches themselves."""
        torch = self.torch
        n = int(past.shape[0])
        posterior = fut is not None
        h = self._handle(n, posterior)
        d = h.dims
        if tuple(past.shape[1:]) != (d.T_obs, d.mno, 3) or (posterior and tuple(fut.shape[1:]) != (d.T_pred, d.mno, 3)):
            raise ValueError("window lengths must be (seq_length, pred_length) and the slot axis max_num_obj padded to %d" % d.mno)
        if eps is None:
            g = torch.Generator(device=self.device).manual_seed(seed)
            eps_t = torch.randn((d.R, d.L), generator=g, device=self.device, dtype=torch.float32)
        elif torch.is_tensor(eps):
            eps_t = eps.reshape(d.R, d.L)
        else:
            eps_t = torch.as_tensor(np.ascontiguousarray(eps, np.float32), device=self.device).reshape(d.R, d.L)
        if self._grids is None:
            self._grids = torch.zeros((d.n_grids, d.Gh, d.Gw, d.C), device=self.device)
            self._grid_of_scene = np.zeros(n, np.int32)
        gos = self._grid_of_scene if len(self._grid_of_scene) == n else np.resize(self._grid_of_scene, n)
        h.set_scene_grids(self._grids.data_ptr(), gos)
        Y = torch.empty((n, d.K, d.mno, d.T_pred, 2), device=self.device, dtype=torch.float32)
        score = torch.empty((n, d.K, d.mno), device=self.device, dtype=torch.float32)
        stream = torch.cuda.current_stream().cuda_stream
        h.forward(past.data_ptr(), fut.data_ptr() if posterior else 0, eps_t.data_ptr(), Y.data_ptr(),
                  score.data_ptr(), stream)
        self._keep = (past, fut, eps_t)            # keep inputs alive until the stream has consumed them
        self.final_output, self.final_states = Y, score
        if posterior:                              # train-path scalars (model/model.py:339-376): cost = mean(recon + kld)
            self._kld = torch.empty(d.A, device=self.device)
            self._recon = torch.empty(d.A, device=self.device)
            self._cost = torch.empty(2, device=self.device)
            h.losses(fut.data_ptr(), Y.data_ptr(), self._kld.data_ptr(), self._recon.data_ptr(), self._cost.data_ptr(), stream)
            self.cost = self._cost[0]              # 0-dim device tensor; float(model.cost) synchronises
        else:
            self.cost = None
        return Y, score

    def forward_ref_compat(self, x_batch: Sequence[np.ndarray], y_batch: Sequence[np.ndarray],
                           eps: Optional[np.ndarray] = None, seed: int = 0) -> Dict[str, "object"]:
        """The reference graph AS WRITTEN (model/model.py:116-311; dims.ref_compat) on loader batches: x_batch / y_batch are
        DataLoader.next_batch's x and y (y = x shifted one frame, utils/data_loader.py:206-207), [seq_length, MNO, 3] each, raw
        pixels.  Needs d_dim == 2*seq_length (:286-289).  eps [n, mno, L]: ONE draw per object (:262-263).  Returns device
        tensors: rho [n, mno, 200] (O1, :116-133), Hx / Hy [n, mno, H], output_states [n, mno, n_dec, seq_length, 2] (:280-289)
        and feature_pooling [n, mno, n_dec, seq_length, 200] (O11, :291-311).  The reference's graph stops there (:312-313)."""
        torch = self.torch
        n = len(x_batch)
        h = self._handle(n, True, ref_compat=True)
        d = h.dims
        past, fut = self._pad_windows(x_batch, d.mno), self._pad_windows(y_batch, d.mno)
        if past.shape[1] != d.T_obs or fut.shape[1] != d.T_obs:
            raise ValueError("ref_compat windows are seq_length frames each")
        if eps is None:
            g = torch.Generator(device=self.device).manual_seed(seed)
            eps_t = torch.randn((d.A, d.L), generator=g, device=self.device, dtype=torch.float32)
        else:
            eps_t = torch.as_tensor(np.ascontiguousarray(eps, np.float32), device=self.device).reshape(d.A, d.L)
        stream = torch.cuda.current_stream().cuda_stream
        states = torch.empty((n, d.mno, d.n_dec, d.T_obs, 2), device=self.device)
        rho = torch.empty((n, d.mno, 200), device=self.device)
        fp = torch.empty((n, d.mno, d.n_dec, d.T_obs, 200), device=self.device)
        h.forward(past.data_ptr(), fut.data_ptr(), eps_t.data_ptr(), states.data_ptr(), 0, stream)
        h.temporal_conv(past.data_ptr(), rho.data_ptr(), stream)
        h.feature_pooling(states.data_ptr(), rho.data_ptr(), fp.data_ptr(), stream)
        self._keep = (past, fut, eps_t)
        self.input_data, self.target_data = x_batch, y_batch
        HxHy = h.device_tensor("HxHy").view(n, d.mno, 2, -1)[..., : d.H]
        self.final_output = states
        return {"rho": rho, "Hx": HxHy[:, :, 0], "Hy": HxHy[:, :, 1], "output_states": states, "feature_pooling": fp}

    # ---- training (train.py:140-181 runs only `cost`; the Adam op of model/model.py:386-403 is never applied) ----
    def train_step(self, x_batch: Sequence[np.ndarray], y_batch: Sequence[np.ndarray], eps: Optional[np.ndarray] = None,
                   seed: int = 0, group=None, sync: bool = True):
        """One optimiser step on a batch of loader windows: forward (posterior path), backward, gradient mean over
        the data-parallel ranks (RCCL all-reduce of ONE flat buffer when torch.distributed is initialised),
        clip_by_global_norm(args.grad_clip), Adam(args.learning_rate).  Returns the loss terms of DESIGN.md section 8
        evaluated BEFORE the update (what `sess.run([cost, train_op])` would have returned); with sync=False a PendingLoss
        whose .get() returns them later (see train_step_device)."""
        d = self._handle(len(x_batch), True).dims
        out = self.train_step_device(self._pad_windows(x_batch, d.mno), self._pad_windows(y_batch, d.mno), eps, seed, group, sync)
        self.input_data, self.target_data = x_batch, y_batch
        return out

    def train_step_device(self, past, fut, eps=None, seed: int = 0, group=None, sync: bool = True):
        """train_step on windows already in HBM (forward_device's layout).  sync=False: nothing waits for the GPU -- the loss terms
        go to a device buffer (desire_train_loss_async), a pinned copy is enqueued, and the returned PendingLoss reads them when asked,
        normally one step later (desire_amd/train.py): the host runs ahead of the device and the loader thread is never starved by a
        read-back in the middle of every step."""
        from .dist import allreduce_mean_
        torch = self.torch
        h = self._handle(int(past.shape[0]), True)
        prev = getattr(self, "_trained", None)
        if prev is not None and prev is not h:
            raise ValueError("train_step was called with %d windows after training with %d: the Adam moments live in the handle "
                             "of one batch size (keep the batch size fixed, as DataLoader.next_batch does)"
                             % (int(past.shape[0]), prev.dims.n_scenes))
        if not getattr(h, "_training_on", False):
            h.set_training(True)
            h._training_on = True
            pending = self.__dict__.pop("_opt_pending", None)
            if pending is not None:                   # resumed run: Adam moments and step counter of the checkpoint
                h.set_opt_state(pending)
            self._configure_head_loss(h)
        self.forward_device(past, fut, eps, seed)
        past, fut, eps_t = self._keep
        stream = torch.cuda.current_stream().cuda_stream
        h.backward(past.data_ptr(), fut.data_ptr(), eps_t.data_ptr(), stream)
        allreduce_mean_(h.grad_tensor(), group)
        clip = float(getattr(self.args, "grad_clip", 0.0) or 0.0)
        if clip > 0:
            h.clip_grads(clip, stream=stream)
        if sync:
            terms = h.train_loss(fut.data_ptr(), stream)
        else:
            dev8 = torch.empty(8, device=self.device, dtype=torch.float32)
            h.train_loss_async(fut.data_ptr(), dev8.data_ptr(), stream)
            host8 = torch.empty(8, dtype=torch.float32).pin_memory()
            host8.copy_(dev8, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            terms = PendingLoss(host8, ev, dev8)
        h.adam_step(self.learning_rate, stream=stream)
        self._trained = h
        self._version = getattr(self, "_version", 0) + 1
        h._wver = self._version
        return terms

    def _configure_head_loss(self, h) -> None:
        """Hook of the Gaussian-head term (args.head_loss_weight; see train_step / DESIGN.md section 12)."""
        lam = float(getattr(self.args, "head_loss_weight", 0.0) or 0.0)
        if lam > 0.0 and hasattr(h, "set_head_loss"):
            h.set_head_loss(lam)
            self._head_given = True                   # the head is trained from now on: sample(mode="rollout") reads learned weights

    def sync_weights(self) -> Dict[str, np.ndarray]:
        """Pull the trained weights back from the device; other handles (batch sizes / prior path) are rebuilt lazily."""
        h = getattr(self, "_trained", None)
        if h is not None and getattr(self, "_weights_ver", 0) != self._version:
            self._weights = {k: h.get_weight(k, np.shape(v)) for k, v in self._weights.items()}
            self._weights_ver = self._version
        return self._weights

    def forward_from_video(self, frames, starts: Sequence[int], posterior: bool = True, eps=None, seed: int = 0):
        """Device-side batching: `frames` [F, max_num_obj, 3] is one preprocessed video
        (DataLoader.data[i]); the windows starting at `starts` are cut and slot-assigned on the GPU exactly like the x of
        DataLoader(seq_length = seq_length + pred_length).next_batch (slots ranked over the window plus the loader's one
        look-ahead frame), then run through the hot path without touching the host again."""
        torch = self.torch
        n = len(starts)
        h = self._handle(n, posterior)
        d = h.dims
        fr = torch.as_tensor(np.ascontiguousarray(np.asarray(frames), np.float32), device=self.device)
        past = torch.empty((n, d.T_obs, d.mno, 3), device=self.device)
        fut = torch.empty((n, d.T_pred, d.mno, 3), device=self.device)
        stream = torch.cuda.current_stream().cuda_stream
        h.build_windows(fr.data_ptr(), fr.shape[0], fr.shape[1], starts, past.data_ptr(), fut.data_ptr(), stream, lookahead=1)
        if eps is None:
            g = torch.Generator(device=self.device).manual_seed(seed)
            eps_t = torch.randn((d.R, d.L), generator=g, device=self.device, dtype=torch.float32)
        else:
            eps_t = torch.as_tensor(np.ascontiguousarray(eps, np.float32), device=self.device).reshape(d.R, d.L)
        if self._grids is None:
            self._grids = torch.zeros((d.n_grids, d.Gh, d.Gw, d.C), device=self.device)
            self._grid_of_scene = np.zeros(n, np.int32)
        gos = self._grid_of_scene if len(self._grid_of_scene) == n else np.resize(self._grid_of_scene, n)
        h.set_scene_grids(self._grids.data_ptr(), gos)
        Y = torch.empty((n, d.K, d.mno, d.T_pred, 2), device=self.device, dtype=torch.float32)
        score = torch.empty((n, d.K, d.mno), device=self.device, dtype=torch.float32)
        h.forward(past.data_ptr(), fut.data_ptr() if posterior else 0, eps_t.data_ptr(), Y.data_ptr(), score.data_ptr(), stream)
        self._keep = (fr, past, fut, eps_t)
        self.final_output, self.final_states = Y, score
        return Y, score, past, fut

    def evaluate(self, Y, fut_windows) -> np.ndarray:
        """[A, 4] = (ADE mean-of-K, FDE mean-of-K, ADE best-of-K, FDE best-of-K), normalised units (N4 harness).
        `fut_windows`: list of loader windows [T_pred, MNO, 3] or a device tensor [n, T_pred, mno, 3]."""
        torch = self.torch
        n = Y.shape[0]
        h = self._handle(n, True)
        d = h.dims
        fut = fut_windows if torch.is_tensor(fut_windows) else self._pad_windows(fut_windows, d.mno)
        out = torch.empty((d.A, 4), device=self.device)
        h.ade_fde(Y.data_ptr(), fut.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out.cpu().numpy()

    # ---- checkpoints (train.py:114,197-206 saves TF checkpoints; here: a named fp32 archive) -----------------
    def save(self, path: str) -> None:
        from .formats import save_weights
        if self._weights is None:
            raise ValueError("no weights yet: run forward() once or pass weights=")
        blob = dict(self.sync_weights())
        # whether gauss_head/* holds trained (or caller-supplied) values rather than the random init: an archive always carries the head, so
        # key presence says nothing -- restore() reads this marker to pick sample()'s default mode
        blob["meta/head_trained"] = np.asarray([1.0 if getattr(self, "_head_given", False) else 0.0], np.float32)
        h = getattr(self, "_trained", None)
        if h is not None:                             # the reference's Saver keeps the optimiser slots too (train.py:114)
            for k, v in h.opt_state().items():
                blob["opt/" + k] = v
        save_weights(path, blob)

    @classmethod
    def restore(cls, args, path: str) -> "DESIREModel":
        from .formats import load_weights
        blob = load_weights(path)
        opt = {k[4:]: blob.pop(k) for k in list(blob) if k.startswith("opt/")}
        meta = {k[5:]: blob.pop(k) for k in list(blob) if k.startswith("meta/")}
        # archives of rounds 2-4 carry no marker: the rule they were written under applies (a head that is in the archive counts as given), so a
        # checkpoint trained with --head_loss_weight > 0 keeps sample()'s default mode 'rollout' (INTEGRATION.md "Checkpoints")
        head_trained = bool(float(np.asarray(meta["head_trained"]).reshape(-1)[0])) if "head_trained" in meta else ("gauss_head/w" in blob)
        # archives written before an auxiliary weight existed (the sample() head "gauss_head/*" came with round 2): complete them
        # with the values init_weights draws, so an older checkpoint still loads
        d0 = dims_from_args(args, 1, True)
        missing = [k for k in weight_shapes(d0) if k not in blob]
        if missing:
            fresh = init_weights(d0, 0)
            unknown = [k for k in missing if not k.startswith("gauss_head/")]
            if unknown:
                raise ValueError("checkpoint %s lacks weights %s" % (path, unknown))
            for k in missing:
                blob[k] = fresh[k]
            if opt:                                   # moments of the flat buffer no longer line up: Adam restarts from zero
                opt = {}
        m = cls(args, weights=blob)
        m._head_given = head_trained and not missing      # NOT key presence: save() writes the head whether or not anything ever trained it
        if opt:
            m._opt_pending = opt                      # applied when training starts (the moments live in the training handle)
        return m

    # ---- reference-shaped sampling API (model/model.py:613-688) ------------------------------------
    def _sample_model(self, obs_len: int, dimensions) -> "DESIREModel":
        """The model sample() runs on: same weights, observation length = the trajectory's, frame size = `dimensions`.
        Cached per (obs_len, dimensions); refreshed when the optimiser has moved the weights since (train_step keeps them on
        the device: sync_weights pulls them back)."""
        key = (int(obs_len), None if dimensions is None else (float(dimensions[0]), float(dimensions[1])))
        cache = self.__dict__.setdefault("_sample_models", {})
        wts = self.sync_weights()
        if wts is None:
            self._handle(1, False)                       # first use: draws the initial weights
            wts = self._weights
        sub, ver = cache.get(key, (None, -1))
        if sub is None or ver != self._version:
            args = SimpleNamespace(**vars(self.args))
            args.seq_length = int(obs_len)
            args.pred_length = int(getattr(self.args, "pred_length", None) or self.args.seq_length)
            if key[1] is not None:
                args.img_width, args.img_height = key[1]
            w = dict(wts)
            tw = np.asarray(w["temporal/w"])
            if tw.shape[1] != obs_len:                   # O1's window is seq_length wide; sample() does not use it
                fit = np.zeros((1, obs_len, 2, 100), np.float32)
                n = min(obs_len, tw.shape[1])
                fit[:, :n] = tw[:, :n]
                w["temporal/w"] = fit
            if sub is None:
                sub = DESIREModel(args, w, self._seed)
            else:
                sub._weights = w
                for hd in sub._handles.values():
                    hd.set_weights(w)
            cache[key] = (sub, self._version)
        return sub

    def sample(self, sess, traj, grid, dimensions, true_traj, num=10, mode: Optional[str] = None, normals=None, seed: int = 0):
        """traj [obs, MNO, 3] observed frames; returns [obs+num, MNO, 3]: the observed frames followed by `num` predicted
        frames in pixel units, ids carried over from the last observed frame (model/model.py:680-688).  `sess` is ignored;
        `dimensions` = (width, height) of the frame in pixels (default: args.img_width / img_height).

        mode None (default): "rollout" -- the reference's own sample() semantics -- whenever the 5-wide head "gauss_head/w|b" it reads
        is TRAINED (args.head_loss_weight > 0 adds the reference's Gaussian NLL, model/model.py:494-550, to train_step's loss) or was
        supplied (weights= / a checkpoint that holds it); otherwise "ioc", the path the default loss trains, with a one-time warning
        that says so.  The mode is never inferred from `normals`: they belong to the rollout, and passing them to "ioc" is an error.

        mode "rollout" is the reference's loop (:623-688) on the device in one launch: warm-up over the observed
        frames, then per step the 5-wide Gaussian head "gauss_head/w|b" -> a draw (`normals` [num, MNO, 2] ~ N(0,1), or
        torch's generator seeded with `seed`) -> clip to <= 1.0 in normalised units (:666-669) -> fed back as the next input.
        Like the reference, objects with id 0 are stepped too and keep id 0.  `true_traj` only feeds the reference's cost
        print-outs (:649,684-685) and is unused.
        mode "ioc": the top-scored IOC-refined sample of the frozen-spec forward (prior path); `grid` may be a [Gh,Gw,C]
        scene feature grid; num <= pred_length."""
        torch = self.torch
        traj = np.asarray(traj, np.float64)
        sub = self._sample_model(traj.shape[0], dimensions)
        out = np.zeros((traj.shape[0] + num, traj.shape[1], 3))
        out[: traj.shape[0]] = traj
        m = traj.shape[1]
        out[traj.shape[0]:, :, 0] = traj[-1, :, 0]
        if mode is None:
            mode = "rollout" if getattr(self, "_head_given", False) else "ioc"
            if mode == "ioc" and not getattr(self, "_warned_default_mode", False):
                import warnings
                self._warned_default_mode = True
                warnings.warn("sample(): the reference's rollout reads the Gaussian output layer gauss_head/w|b, which this model has neither "
                              "been given nor trained (train with args.head_loss_weight > 0); the default therefore returns the top-scored "
                              "IOC-refined sample (mode='ioc').  Pass mode='rollout' to run the reference's loop on the untrained head.",
                              stacklevel=2)
        if normals is not None and mode != "rollout":
            raise ValueError("`normals` are the rollout's Gaussian draws: pass mode='rollout' with them (the mode is not inferred)")
        if mode == "rollout":
            if not getattr(self, "_head_given", False):
                import warnings
                warnings.warn("sample(mode='rollout') reads gauss_head/w|b, which train_step trains only with args.head_loss_weight > 0; this "
                              "model's head holds its random initial values (train it, pass weights= with a trained head, or use mode='ioc')",
                              stacklevel=2)
            h = sub._handle(1, False)
            d = h.dims
            past = sub._pad_windows([traj], d.mno)
            if normals is None:
                g = torch.Generator(device=self.device).manual_seed(seed)
                nrm = torch.randn((num, d.A, 2), generator=g, device=self.device, dtype=torch.float32)
            else:
                nrm = torch.zeros((num, d.A, 2), device=self.device)
                nrm[:, :m] = torch.as_tensor(np.asarray(normals, np.float32), device=self.device)
            pos = torch.empty((num, d.A, 2), device=self.device)
            h.rollout(past.data_ptr(), nrm.data_ptr(), int(num), pos.data_ptr(), torch.cuda.current_stream().cuda_stream)
            p = pos.cpu().numpy()[:, :m]
            out[traj.shape[0]:, :, 1] = p[..., 0] / d.sx
            out[traj.shape[0]:, :, 2] = p[..., 1] / d.sy
            return out
        if mode != "ioc":
            raise ValueError("sample(mode=...): 'rollout' (the reference's loop) or 'ioc' (top-scored refined sample)")
        t_pred = int(getattr(self.args, "pred_length", None) or self.args.seq_length)
        if num > t_pred:
            raise ValueError("num=%d exceeds the model's pred_length=%d (the IOC regression head is sized by it)"
                             % (num, t_pred))
        if grid is not None and np.ndim(grid) == 3:
            sub.set_scene_grids(np.asarray(grid, np.float32)[None], [0])
        Y, score = sub.forward([traj], None, seed=seed)
        d = sub._handle(1, False).dims
        best = score[0].argmax(dim=0)                                        # [mno]
        idx = best.view(1, -1, 1, 1).expand(1, d.mno, d.T_pred, 2)
        top = torch.gather(Y[0], 0, idx)[0].cpu().numpy()[:, :num]           # [mno, num, 2]
        out[traj.shape[0]:, :, 1] = (top[:m, :, 0] / d.sx).T
        out[traj.shape[0]:, :, 2] = (top[:m, :, 1] / d.sy).T
        out[traj.shape[0]:][:, traj[-1, :, 0] == 0] = 0
        return out
