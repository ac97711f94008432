"""ctypes binding of libdesire_hip.so (include/desire_hip.h).  This is the ONLY compute path of
the package: if the library is missing or no gfx950 device is present, calls raise -- there is
no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

import numpy as np

from .spec import Dims

LIB_PATH = os.environ.get("DESIRE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdesire_hip.so")

EXPORTS = [
    "desire_last_error", "desire_version", "desire_dims_size", "desire_build_hash", "desire_create", "desire_destroy", "desire_set_weight",
    "desire_finalize_weights", "desire_set_scene_grids", "desire_encode", "desire_sample",
    "desire_ioc_refine", "desire_forward", "desire_read_buffer", "desire_neighbor_bins",
    "desire_scene_cells", "desire_set_profiling", "desire_get_profile", "desire_scene_cnn", "desire_losses",
    "desire_temporal_conv", "desire_feature_pooling", "desire_build_windows", "desire_gaussian_sample", "desire_ade_fde",
    "desire_set_training", "desire_backward", "desire_get_grad", "desire_grad_buffer",
    "desire_train_loss", "desire_adam_step", "desire_get_weight", "desire_clip_grads",
    "desire_device_buffer", "desire_ioc_step", "desire_ioc_finish", "desire_get_bin_table",
    "desire_graph_begin", "desire_graph_end", "desire_graph_launch", "desire_rollout", "desire_build_windows_la", "desire_adam_state",
    "desire_set_option", "desire_train_loss_async", "desire_set_head_loss",
    "desire_peer_export", "desire_peer_open", "desire_ioc_peer_pass", "desire_peer_close", "desire_peer_region", "desire_peer_open_ptr", "desire_peer_status",
]


class DesireDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("n_scenes", "mno", "K", "T_obs", "T_pred", "H", "L", "S", "C", "Gh", "Gw", "n_grids",
                 "grid_size", "E_v", "iters", "posterior")] + \
               [(n, C.c_float) for n in ("nb_w", "nb_h", "sx", "sy")] + [(n, C.c_int32) for n in ("bin_mode", "bn_mode", "bf16", "ref_compat", "n_dec", "ioc_form", "ioc_split", "train_fp32_mask", "flags")]

    @classmethod
    def from_dims(cls, d: Dims) -> "DesireDims":
        return cls(d.n_scenes, d.mno, d.K, d.T_obs, d.T_pred, d.H, d.L, d.S, d.C, d.Gh, d.Gw, d.n_grids,
                   d.grid_size, d.E_v, d.iters, d.posterior, d.nb_w, d.nb_h, d.sx, d.sy, int(getattr(d, "bin_mode", 0)),
                   int(getattr(d, "bn_mode", 0)), int(getattr(d, "bf16", 0)), int(getattr(d, "ref_compat", 0)), int(getattr(d, "n_dec", 0)),
                   int(getattr(d, "ioc_form", 0)), int(getattr(d, "ioc_split", 0)), int(getattr(d, "train_fp32_mask", 0)), int(getattr(d, "flags", 0)))


class DesireError(RuntimeError):
    pass


_lib = None


def load() -> C.CDLL:
    """Load the shared library (building is __graft_entry__.build()'s job, never done here)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own HIP / HSA runtime libraries.  The process must end up with ONE runtime: importing torch first
    # makes the dynamic loader resolve this library's libamdhip64 / libhsa-runtime64 against what torch already mapped; the other
    # load order leaves two runtimes in the process and the second one to initialise sees no device.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise DesireError(
            "libdesire_hip.so not found at %s -- run `python -c 'import __graft_entry__ as g; g.build()'`; "
            "desire_amd has no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, f32p = C.c_void_p, C.c_int32, C.c_void_p
    lib.desire_last_error.restype = C.c_char_p
    lib.desire_build_hash.restype = C.c_char_p
    lib.desire_create.argtypes = [C.POINTER(DesireDims), C.POINTER(vp)]
    lib.desire_destroy.argtypes = [vp]
    lib.desire_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.desire_set_weight.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_size_t]
    lib.desire_finalize_weights.argtypes = [vp]
    lib.desire_set_scene_grids.argtypes = [vp, f32p, C.POINTER(C.c_int32)]
    lib.desire_encode.argtypes = [vp, f32p, f32p, vp]
    lib.desire_sample.argtypes = [vp, f32p, f32p, vp]
    lib.desire_ioc_refine.argtypes = [vp, f32p, f32p, vp]
    lib.desire_forward.argtypes = [vp, f32p, f32p, f32p, f32p, f32p, vp]
    lib.desire_read_buffer.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_size_t, vp]
    lib.desire_neighbor_bins.argtypes = [vp, f32p, vp, vp, i32, vp]
    lib.desire_scene_cells.argtypes = [vp, f32p, vp, i32, vp]
    lib.desire_scene_cnn.argtypes = [vp, f32p, i32, i32, f32p, vp]
    lib.desire_losses.argtypes = [vp, f32p, f32p, f32p, f32p, f32p, vp]
    lib.desire_temporal_conv.argtypes = [vp, f32p, f32p, vp]
    lib.desire_feature_pooling.argtypes = [vp, f32p, f32p, f32p, vp]
    lib.desire_build_windows.argtypes = [vp, f32p, i32, i32, C.POINTER(C.c_int32), i32, f32p, f32p, vp]
    lib.desire_build_windows_la.argtypes = [vp, f32p, i32, i32, C.POINTER(C.c_int32), i32, i32, f32p, f32p, vp]
    lib.desire_gaussian_sample.argtypes = [vp, f32p, f32p, f32p, i32, vp]
    lib.desire_ade_fde.argtypes = [vp, f32p, f32p, f32p, vp]
    lib.desire_rollout.argtypes = [vp, f32p, f32p, i32, f32p, vp]
    lib.desire_set_training.argtypes = [vp, C.c_int]
    lib.desire_backward.argtypes = [vp, f32p, f32p, f32p, vp]
    lib.desire_get_grad.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_size_t, vp]
    lib.desire_grad_buffer.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.desire_train_loss.argtypes = [vp, f32p, C.POINTER(C.c_float), vp]
    lib.desire_train_loss_async.argtypes = [vp, f32p, f32p, vp]
    lib.desire_set_head_loss.argtypes = [vp, C.c_float]
    lib.desire_peer_export.argtypes = [vp, C.c_char_p, C.POINTER(C.c_size_t)]
    lib.desire_peer_open.argtypes = [vp, i32, i32, i32, C.c_char_p]
    lib.desire_ioc_peer_pass.argtypes = [vp, f32p, f32p, vp]
    lib.desire_peer_close.argtypes = [vp]
    lib.desire_peer_status.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.desire_peer_region.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.desire_peer_open_ptr.argtypes = [vp, i32, i32, i32, vp]
    lib.desire_adam_step.argtypes = [vp, C.c_float, C.c_float, C.c_float, C.c_float, vp]
    lib.desire_adam_state.argtypes = [vp, C.POINTER(C.c_int32), C.c_int]
    lib.desire_get_weight.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float), C.c_size_t, vp]
    lib.desire_clip_grads.argtypes = [vp, C.c_float, C.POINTER(C.c_float), vp]
    lib.desire_get_bin_table.argtypes = [vp, C.POINTER(C.c_float)]
    lib.desire_graph_begin.argtypes = [vp, vp]
    lib.desire_graph_end.argtypes = [vp, vp, C.POINTER(C.c_int32)]
    lib.desire_graph_launch.argtypes = [vp, i32, vp]
    lib.desire_device_buffer.argtypes = [vp, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.desire_ioc_step.argtypes = [vp, i32, i32, i32, f32p, f32p, vp, f32p, f32p, f32p, vp]
    lib.desire_ioc_finish.argtypes = [vp, f32p, f32p, f32p, f32p, vp]
    lib.desire_set_profiling.argtypes = [vp, C.c_int]
    lib.desire_get_profile.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]
    for n in EXPORTS:
        if n not in ("desire_last_error", "desire_build_hash"):
            getattr(lib, n).restype = C.c_int
    if lib.desire_dims_size() != C.sizeof(DesireDims):          # this binding and the library disagree about desire_dims: refuse to run
        raise DesireError("libdesire_hip.so was built with sizeof(desire_dims) = %d, this binding has %d: rebuild (python -c 'import "
                          "__graft_entry__ as g; g.build()')" % (lib.desire_dims_size(), C.sizeof(DesireDims)))
    _lib = lib
    return lib


def _chk(rc: int) -> None:
    if rc != 0:
        raise DesireError("libdesire_hip error %d: %s" % (rc, load().desire_last_error().decode()))


class Handle:
    """Thin owner of one desire_handle*.  All device pointers are raw integers (torch .data_ptr())."""

    def __init__(self, dims: Dims):
        dims.validate()
        self.dims = dims
        self.lib = load()
        self._h = C.c_void_p()
        cd = DesireDims.from_dims(dims)
        _chk(self.lib.desire_create(C.byref(cd), C.byref(self._h)))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.desire_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name: str, value: int) -> None:
        """One of the behavioural switches of desire_dims ("ioc_form", "ioc_split", "train_fp32_mask", "flags") on the live handle."""
        _chk(self.lib.desire_set_option(self._h, name.encode(), int(value)))
        if hasattr(self.dims, name):                      # ("compact_min_rows" is a handle option, not a desire_dims field)
            self.dims = self.dims.replace(**{name: int(value)})

    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        for name, arr in weights.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            _chk(self.lib.desire_set_weight(self._h, name.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size))
        _chk(self.lib.desire_finalize_weights(self._h))

    def set_scene_grids(self, dev_grids_ptr: int, grid_of_scene) -> None:
        g = np.ascontiguousarray(grid_of_scene, dtype=np.int32)
        if g.size != self.dims.n_scenes:
            raise DesireError("grid_of_scene must have n_scenes entries")
        _chk(self.lib.desire_set_scene_grids(self._h, dev_grids_ptr, g.ctypes.data_as(C.POINTER(C.c_int32))))

    def encode(self, past_ptr: int, fut_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_encode(self._h, past_ptr, fut_ptr or None, stream or None))

    def sample(self, eps_ptr: int, yhat_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_sample(self._h, eps_ptr, yhat_ptr, stream or None))

    def ioc_refine(self, yhat_ptr: int, score_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_ioc_refine(self._h, yhat_ptr, score_ptr, stream or None))

    def forward(self, past_ptr: int, fut_ptr: int, eps_ptr: int, yhat_ptr: int, score_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_forward(self._h, past_ptr, fut_ptr or None, eps_ptr, yhat_ptr, score_ptr or None, stream or None))

    def read_buffer(self, name: str, shape: Tuple[int, ...], stream: int = 0) -> np.ndarray:
        out = np.empty(shape, np.float32)
        _chk(self.lib.desire_read_buffer(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size,
                                         stream or None))
        return out

    def neighbor_bins(self, pos_ptr: int, valid_ptr: int, bins_ptr: int, n_groups: int, stream: int = 0) -> None:
        _chk(self.lib.desire_neighbor_bins(self._h, pos_ptr, valid_ptr, bins_ptr, n_groups, stream or None))

    def scene_cells(self, pos_ptr: int, cells_ptr: int, n: int, stream: int = 0) -> None:
        _chk(self.lib.desire_scene_cells(self._h, pos_ptr, cells_ptr, n, stream or None))

    def scene_cnn(self, image_ptr: int, Hi: int, Wi: int, grids_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_scene_cnn(self._h, image_ptr, Hi, Wi, grids_ptr, stream or None))

    def losses(self, fut_ptr: int, yhat_ptr: int, kld_ptr: int, recon_ptr: int, cost_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_losses(self._h, fut_ptr, yhat_ptr, kld_ptr, recon_ptr, cost_ptr, stream or None))

    def temporal_conv(self, past_ptr: int, rho_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_temporal_conv(self._h, past_ptr, rho_ptr, stream or None))

    def feature_pooling(self, yhat_ptr: int, rho_ptr: int, out_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_feature_pooling(self._h, yhat_ptr, rho_ptr, out_ptr, stream or None))

    def build_windows(self, frames_ptr: int, n_frames: int, mno_in: int, starts, past_ptr: int, fut_ptr: int, stream: int = 0,
                      lookahead: int = 0) -> None:
        st = np.ascontiguousarray(starts, dtype=np.int32)
        _chk(self.lib.desire_build_windows_la(self._h, frames_ptr, n_frames, mno_in, st.ctypes.data_as(C.POINTER(C.c_int32)),
                                            st.size, int(lookahead), past_ptr, fut_ptr, stream or None))

    def gaussian_sample(self, params_ptr: int, normals_ptr: int, out_ptr: int, n: int, stream: int = 0) -> None:
        _chk(self.lib.desire_gaussian_sample(self._h, params_ptr, normals_ptr, out_ptr, n, stream or None))

    def rollout(self, past_ptr: int, normals_ptr: int, num: int, out_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_rollout(self._h, past_ptr, normals_ptr, num, out_ptr, stream or None))

    def ade_fde(self, yhat_ptr: int, fut_ptr: int, out_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_ade_fde(self._h, yhat_ptr, fut_ptr, out_ptr, stream or None))

    def set_training(self, on: bool) -> None:
        _chk(self.lib.desire_set_training(self._h, int(on)))

    def backward(self, past_ptr: int, fut_ptr: int, eps_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_backward(self._h, past_ptr, fut_ptr, eps_ptr, stream or None))

    def get_grad(self, name: str, shape: Tuple[int, ...], stream: int = 0) -> np.ndarray:
        out = np.empty(shape, np.float32)
        _chk(self.lib.desire_get_grad(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, stream or None))
        return out

    def grad_buffer(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.lib.desire_grad_buffer(self._h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def graph_begin(self, stream: int) -> None:
        _chk(self.lib.desire_graph_begin(self._h, stream))

    def graph_end(self, stream: int) -> int:
        gid = C.c_int32(-1)
        _chk(self.lib.desire_graph_end(self._h, stream, C.byref(gid)))
        return int(gid.value)

    def graph_launch(self, graph_id: int, stream: int) -> None:
        _chk(self.lib.desire_graph_launch(self._h, graph_id, stream))

    def bin_table(self) -> np.ndarray:
        out = np.zeros(20, np.float32)
        _chk(self.lib.desire_get_bin_table(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def device_tensor(self, name: str, dtype: str = "<f4"):
        """A workspace tensor of the handle ("HxHy", "p_last", "valid", "Y0", ...) as a flat zero-copy torch tensor."""
        import torch
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.lib.desire_device_buffer(self._h, name.encode(), C.byref(p), C.byref(n)))
        cnt = int(n.value) // (1 if dtype == "|u1" else 4)

        class _Dev:
            __cuda_array_interface__ = {"shape": (cnt,), "typestr": dtype, "data": (int(p.value), False), "version": 2}
        return torch.as_tensor(_Dev(), device="cuda")

    def ioc_step(self, t: int, rank: int, nranks: int, Yall_ptr: int, plast_all_ptr: int, valid_all_ptr: int, Hall_ptr: int,
                 h_state_ptr: int, score_state_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_ioc_step(self._h, t, rank, nranks, Yall_ptr, plast_all_ptr, valid_all_ptr, Hall_ptr, h_state_ptr,
                                      score_state_ptr, stream or None))

    def ioc_finish(self, h_state_ptr: int, score_state_ptr: int, Y_ptr: int, score_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_ioc_finish(self._h, h_state_ptr, score_state_ptr, Y_ptr, score_ptr, stream or None))

    def grad_tensor(self):
        """The flat gradient buffer as a zero-copy torch tensor (for torch.distributed.all_reduce over RCCL)."""
        import torch
        p, n = self.grad_buffer()

        class _Dev:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (p, False), "version": 2}
        return torch.as_tensor(_Dev(), device="cuda")

    def clip_grads(self, max_norm: float, want_norm: bool = False, stream: int = 0):
        out = C.c_float(0.0)
        _chk(self.lib.desire_clip_grads(self._h, C.c_float(max_norm), C.byref(out) if want_norm else None, stream or None))
        return float(out.value) if want_norm else None

    def train_loss(self, fut_ptr: int, stream: int = 0) -> Dict[str, float]:
        out = (C.c_float * 5)()
        _chk(self.lib.desire_train_loss(self._h, fut_ptr, out, stream or None))                 # (synchronises the stream)
        return self.loss_terms(self.read_buffer("loss_out", (8,), stream))                      # + the Gaussian-head term and the clip norm

    def peer_export(self) -> bytes:
        """This rank's exchange region as a 64-byte hipIpcMemHandle (allocated on the first call)."""
        buf = C.create_string_buffer(64)
        n = C.c_size_t(0)
        _chk(self.lib.desire_peer_export(self._h, buf, C.byref(n)))
        return bytes(buf.raw)

    def peer_open(self, rank: int, nranks: int, peer: int, handle: bytes = None) -> None:
        _chk(self.lib.desire_peer_open(self._h, rank, nranks, peer, None if peer == rank else bytes(handle)))

    def peer_region(self) -> int:
        p, n = C.c_void_p(), C.c_size_t(0)
        _chk(self.lib.desire_peer_region(self._h, C.byref(p), C.byref(n)))
        return int(p.value)

    def peer_open_ptr(self, rank: int, nranks: int, peer: int, region_ptr: int = 0) -> None:
        _chk(self.lib.desire_peer_open_ptr(self._h, rank, nranks, peer, region_ptr or None))

    def ioc_peer_pass(self, y_ptr: int, score_ptr: int, stream: int = 0) -> None:
        _chk(self.lib.desire_ioc_peer_pass(self._h, y_ptr, score_ptr, stream or None))

    def peer_timed_out(self) -> bool:
        """After synchronising the stream a peer pass ran on: did one of its bounded waits give up (results unusable)?"""
        v = C.c_int32(0)
        _chk(self.lib.desire_peer_status(self._h, C.byref(v)))
        return bool(v.value)

    def peer_close(self) -> None:
        _chk(self.lib.desire_peer_close(self._h))

    def set_head_loss(self, weight: float) -> None:
        """Weight of the reference's Gaussian-NLL term for the 5-wide output layer (model/model.py:494-550) in the training loss."""
        _chk(self.lib.desire_set_head_loss(self._h, C.c_float(float(weight))))

    def train_loss_async(self, fut_ptr: int, out8_ptr: int, stream: int = 0) -> None:
        """The loss terms of the last training-mode forward into a device buffer of 8 floats, no synchronisation (see loss_terms)."""
        _chk(self.lib.desire_train_loss_async(self._h, fut_ptr, out8_ptr, stream or None))

    @staticmethod
    def loss_terms(v8) -> Dict[str, float]:
        v8 = [float(x) for x in list(v8)[:8]]
        r = dict(zip(("recon", "kld", "ce", "reg", "n_present", "nll_head", "grad_norm", "n_head"), v8))
        r["loss"] = r["recon"] + r["kld"] + r["ce"] + r["reg"] + r["nll_head"]
        return r

    def adam_step(self, lr: float = 0.005, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, stream: int = 0) -> None:
        _chk(self.lib.desire_adam_step(self._h, C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps), stream or None))

    def opt_state(self) -> Dict[str, np.ndarray]:
        """Adam moments (flat, the gradient buffer's layout) and step counter of a training handle."""
        t = C.c_int32(0)
        _chk(self.lib.desire_adam_state(self._h, C.byref(t), 0))
        return {"m": self.device_tensor("Mflat").cpu().numpy(), "v": self.device_tensor("Vflat").cpu().numpy(),
                "t": np.array([t.value], np.int64)}

    def set_opt_state(self, st: Dict[str, np.ndarray]) -> None:
        import torch
        m, v = self.device_tensor("Mflat"), self.device_tensor("Vflat")
        if st["m"].size != m.numel() or st["v"].size != v.numel():
            raise DesireError("optimiser state of %d values does not fit this model (%d)" % (st["m"].size, m.numel()))
        m.copy_(torch.as_tensor(np.ascontiguousarray(st["m"], np.float32), device=m.device))
        v.copy_(torch.as_tensor(np.ascontiguousarray(st["v"], np.float32), device=v.device))
        t = C.c_int32(int(np.asarray(st["t"]).reshape(-1)[0]))
        _chk(self.lib.desire_adam_state(self._h, C.byref(t), 1))

    def get_weight(self, name: str, shape: Tuple[int, ...], stream: int = 0) -> np.ndarray:
        out = np.empty(shape, np.float32)
        _chk(self.lib.desire_get_weight(self._h, name.encode(), out.ctypes.data_as(C.POINTER(C.c_float)), out.size, stream or None))
        return out

    def set_profiling(self, on: bool) -> None:
        _chk(self.lib.desire_set_profiling(self._h, int(on)))

    def get_profile(self) -> List[Tuple[str, float]]:
        n0 = C.c_int32(1 << 30)
        _chk(self.lib.desire_get_profile(self._h, None, None, C.byref(n0)))
        cap = max(int(n0.value), 1)
        ms = (C.c_float * cap)()
        names = (C.c_char_p * cap)()
        n = C.c_int32(cap)
        _chk(self.lib.desire_get_profile(self._h, ms, names, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]
