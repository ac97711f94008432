"""On-disk formats either side of the hot path.

* Reference formats, read AND written unchanged:
    - annotations CSV, 4 rows x N columns: frame id, track id, x centre, y centre
      (scripts/preprocess.py:30-34 writes it, utils/data_loader.py:98 reads it);
    - trajectories.cpkl: pickle protocol 2 tuple (all_frame_data, frame_list_data, num_obj_data)
      (utils/data_loader.py:148-151,161-167).
* New: `DSRTRJ1`, a memory-mappable binary container of the same content -- the loader maps a video
  without parsing (np.memmap), so opening 42 SDD videos costs milliseconds instead of the 5.4 s/video
  CSV parse of the reference.  Layout (little endian):
      magic  8 bytes  b"DSRTRJ1\\0"
      u32 n_videos, u32 max_num_obj
      per video: u64 n_frames, u64 data_offset (bytes from file start, 64-byte aligned)
      per video payload: float32 [n_frames, max_num_obj, 3]  = (id, x_px, y_px), zero rows = absent
  SDD centres are multiples of 0.5 and ids < 2^24, so float32 is lossless here (checked on write).
* Weight checkpoints: a TF-checkpoint-free named fp32 archive (.npz), names = desire_amd.spec.weight_shapes.
"""
from __future__ import annotations

import pickle
import struct
from typing import Dict, List, Sequence

import numpy as np

MAGIC = b"DSRTRJ1\0"


def write_traj_bin(path: str, videos: Sequence[np.ndarray]) -> None:
    videos = [np.asarray(v) for v in videos]
    if not videos:
        raise ValueError("no videos")
    mno = videos[0].shape[1]
    for v in videos:
        if v.ndim != 3 or v.shape[1] != mno or v.shape[2] != 3:
            raise ValueError("every video must be [frames, max_num_obj, 3]")
        if not np.array_equal(v.astype(np.float32).astype(v.dtype), v):
            raise ValueError("values are not exactly representable in float32")
    head = len(MAGIC) + 8 + 16 * len(videos)
    offsets, off = [], (head + 63) // 64 * 64
    for v in videos:
        offsets.append(off)
        off = (off + v.shape[0] * mno * 3 * 4 + 63) // 64 * 64
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<II", len(videos), mno))
        for v, o in zip(videos, offsets):
            fh.write(struct.pack("<QQ", v.shape[0], o))
        for v, o in zip(videos, offsets):
            fh.seek(o)
            fh.write(np.ascontiguousarray(v, dtype="<f4").tobytes())
        fh.truncate(off)


def read_traj_bin(path: str) -> List[np.memmap]:
    with open(path, "rb") as fh:
        if fh.read(len(MAGIC)) != MAGIC:
            raise ValueError("%s is not a DSRTRJ1 file" % path)
        n, mno = struct.unpack("<II", fh.read(8))
        table = [struct.unpack("<QQ", fh.read(16)) for _ in range(n)]
    return [np.memmap(path, dtype="<f4", mode="r", offset=o, shape=(f, mno, 3)) for f, o in table]


def read_cpkl(path: str):
    """The reference's trajectories.cpkl tuple (utils/data_loader.py:161-167)."""
    with open(path, "rb") as fh:
        raw = pickle.load(fh, encoding="latin1")
    if not (isinstance(raw, tuple) and len(raw) == 3):
        raise ValueError("not a (all_frame_data, frame_list_data, num_obj_data) pickle")
    return raw


def write_cpkl(path: str, videos: Sequence[np.ndarray], frame_lists=None, num_obj=None) -> None:
    videos = [np.asarray(v, np.float64) for v in videos]
    frame_lists = frame_lists or [list(map(float, range(len(v)))) for v in videos]
    num_obj = num_obj or [[int((fr[:, 0] != 0).sum()) for fr in v] for v in videos]
    with open(path, "wb") as fh:
        pickle.dump((videos, frame_lists, num_obj), fh, protocol=2)


def cpkl_to_bin(cpkl_path: str, bin_path: str) -> None:
    write_traj_bin(bin_path, read_cpkl(cpkl_path)[0])


def save_weights(path: str, weights: Dict[str, np.ndarray]) -> None:
    np.savez(path, **{k.replace("/", "__"): (np.asarray(v) if k == "opt/t" else np.asarray(v, np.float32)) for k, v in weights.items()})


def load_weights(path: str) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {k.replace("__", "/"): (np.asarray(z[k]) if k == "opt__t" else np.ascontiguousarray(z[k], np.float32)) for k in z.files}
