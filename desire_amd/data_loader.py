"""DataLoader -- same constructor, attributes and batch layout as the reference's
utils/data_loader.py:20-266, rebuilt around vectorised numpy (the reference spends 5.4 s per
video in a Python triple loop).

Kept contract (checked against tests/golden/loader_*.npz, produced by the reference itself):

* CSV input: 4 rows x N columns = frame id, track id, x centre, y centre
  (scripts/preprocess.py:30-34; read at utils/data_loader.py:98-134).
* per video a float64 array (num_frames, max_num_obj, 3) of [id, x, y] rows, objects of a
  frame in CSV column order, zero rows = absent (:113,140-141); ValueError when a frame holds
  more than max_num_obj objects (:140).
* pickle protocol 2 tuple (all_frame_data, frame_list_data, num_obj_data) at
  data/trajectories.cpkl (:148-151).
* num_batches = 2 * floor(sum floor(frames/(T+2)) / batch_size) (:177-183).
* next_batch(random_update=True) -> (x, y, d): lists of len batch_size; x[i], y[i] float64
  [seq_length, max_num_obj, 3]; slot = rank of the id in np.unique(ids of the T+1 window
  frames) (0 included when padding is present, :209,218-229); y = x shifted one frame; pointer
  advance randint(1,T) or T (:235-238); IndexError when a window has more unique ids than
  slots (:227).

Differences (documented, not silent): directories are walked in sorted order (the reference
uses raw os.walk order, :88), `data_dir` is a keyword, and nothing is printed.
"""
from __future__ import annotations

import os
import pickle
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np


def frames_from_csv(data: np.ndarray, max_num_obj: int, fix_id0: bool = False) -> Tuple[np.ndarray, List[float], List[int]]:
    """Vectorised utils/data_loader.py:98-146 for one video.  data [4, N].
    fix_id0=True renames track id 0 (every SDD video has one) to max_id+1, so it is no longer mistaken for
    padding and silently dropped by next_batch (utils/data_loader.py:221-222); default keeps the reference."""
    frames, ids, xs, ys = data[0], data[1], data[2], data[3]
    if fix_id0 and ids.size:
        ids = np.where(ids == 0, ids.max() + 1, ids)
    frame_list = np.unique(frames)
    fidx = np.searchsorted(frame_list, frames)
    order = np.argsort(fidx, kind="stable")           # CSV column order inside each frame
    f_sorted = fidx[order]
    counts = np.bincount(f_sorted, minlength=frame_list.size)
    if counts.size and counts.max() > max_num_obj:
        raise ValueError(
            "could not broadcast input array from shape (%d,3) into shape (%d,3)"
            % (int(counts.max()), max_num_obj))
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
    rank = np.arange(order.size) - starts[f_sorted]
    # the reference takes the FIRST x,y of an id inside the frame (:133-134): map duplicates
    key = f_sorted.astype(np.int64) * (int(ids.max()) + 2 if ids.size else 1) + ids[order].astype(np.int64)
    uniq_keys, inv = np.unique(key, return_inverse=True)
    first_pos = np.full(uniq_keys.size, order.size, np.int64)
    np.minimum.at(first_pos, inv, np.arange(order.size))
    src = order[first_pos[inv]]
    out = np.zeros((frame_list.size, max_num_obj, 3))
    out[f_sorted, rank, 0] = ids[order]
    out[f_sorted, rank, 1] = xs[src]
    out[f_sorted, rank, 2] = ys[src]
    return out, frame_list.tolist(), counts.tolist()


def window_to_slots(window: np.ndarray, seq_length: int, max_num_obj: int) -> Tuple[np.ndarray, np.ndarray]:
    """Vectorised utils/data_loader.py:205-229.  window [T+1, MNO, 3] -> (source, target)."""
    window = np.asarray(window, np.float64)
    ids = window[:, :, 0]
    uniq = np.unique(ids)
    slot = np.searchsorted(uniq, ids)
    src = np.zeros((seq_length, max_num_obj, 3))
    tgt = np.zeros((seq_length, max_num_obj, 3))
    nz = ids != 0
    # The reference fails in two ways while it fills the slots (utils/data_loader.py:215-229), in loop order (frame, slot,
    # source before target): IndexError when a PRESENT object's slot is >= max_num_obj, ValueError when an id occurs twice
    # in one frame (a (k,3) block cannot be assigned to one (3,) slot).  Same exception, decided by the same first event.
    events = []
    for which, fr in ((0, slice(0, seq_length)), (1, slice(1, seq_length + 1))):
        m = nz[fr]
        if not m.any():
            continue
        tt = np.broadcast_to(np.arange(seq_length)[:, None], m.shape)[m]
        ss = slot[fr][m]
        over = ss >= max_num_obj
        if over.any():
            k = np.lexsort((ss[over], tt[over]))[0]
            events.append((int(tt[over][k]), int(ss[over][k]), which, "index"))
        key = tt.astype(np.int64) * (len(uniq) + 1) + ss
        uk, cnt = np.unique(key, return_counts=True)
        if (cnt > 1).any():
            k0 = int(uk[cnt > 1].min())
            events.append((k0 // (len(uniq) + 1), k0 % (len(uniq) + 1), which, "dup:%d" % int(cnt[uk == k0][0])))
    if events:
        ev = min(events)
        if ev[3] == "index":
            raise IndexError("index %d is out of bounds for axis 1 with size %d" % (ev[1], max_num_obj))
        raise ValueError("could not broadcast input array from shape (%s,3) into shape (3,)" % ev[3][4:])
    t_idx = np.broadcast_to(np.arange(seq_length + 1)[:, None], ids.shape)
    m = nz[:seq_length]
    src[t_idx[:seq_length][m], slot[:seq_length][m]] = window[:seq_length][m]
    m = nz[1:]
    tgt[t_idx[:seq_length][m], slot[1:][m]] = window[1:][m]
    return src, tgt


def windows_to_slots_batch(windows: np.ndarray, seq_length: int, max_num_obj: int, out_src: Optional[np.ndarray] = None,
                           out_tgt: Optional[np.ndarray] = None, chunk: int = 8) -> Optional[Tuple[np.ndarray, Optional[np.ndarray]]]:
    """window_to_slots for a whole batch (the loader-in-the-loop path: a few numpy passes per CHUNK of windows instead of ~40 small calls
    per window; utils/data_loader.py:205-229 for every window).  windows [n, T+1, MNO, 3] -> (source, target) [n, T, MNO, 3], written
    into out_src / out_tgt when given (any float dtype: the prefetcher hands pinned float32 staging buffers; out_tgt=False skips the
    target), else fresh float64 arrays.  Returns None when the batch needs the per-window path: an id that is not a small non-negative
    integer, or a window in which the reference would raise (more unique ids than slots, an id twice in a frame) -- the caller then
    walks the windows one by one and gets the reference's exception at the reference's window.  Slot = rank of the id among the
    window's unique ids (0 included when present), from a presence bitmap and its prefix sum, as the device builder does
    (csrc/kernels_aux.hip).  Chunks of 8 windows keep every temporary below the allocator's mmap threshold: fresh multi-megabyte
    temporaries cost a page fault per 4 KB on first touch, which was 5x the arithmetic."""
    w = np.asarray(windows, np.float64)
    n, t1, m = w.shape[0], w.shape[1], w.shape[2]
    T = seq_length
    want_tgt = out_tgt is not False
    src = np.zeros((n, T, max_num_obj, 3)) if out_src is None else out_src
    tgt = (np.zeros((n, T, max_num_obj, 3)) if out_tgt is None else out_tgt) if want_tgt else None
    src2, tgt2 = src.reshape(-1, 3), (tgt.reshape(-1, 3) if want_tgt else None)
    w2 = w.reshape(-1, 3)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        k = c1 - c0
        ids = w[c0:c1, :, :, 0]
        ii = ids.astype(np.int64)
        if not np.isfinite(ids).all() or (ii != ids).any() or ii.min() < 0 or ii.max() > 65535:
            return None
        top = int(ii.max()) + 1
        flat = ii.reshape(k, -1)
        rows = np.arange(k)[:, None]
        seen = np.zeros((k, top), np.bool_)
        seen[rows, flat] = True
        rank = np.cumsum(seen, axis=1, dtype=np.int64) - 1               # rank[w, id] = #unique ids of window w below id
        slot = rank[rows, flat].reshape(k, t1, m)
        present = ii != 0
        if (present & (slot >= max_num_obj)).any():
            return None                                                  # IndexError in the reference (:227)
        srt = np.sort(ii, axis=2)
        if ((srt[:, :, 1:] == srt[:, :, :-1]) & (srt[:, :, 1:] != 0)).any():
            return None                                                  # ValueError in the reference (:224-229)
        if out_src is not None:
            src[c0:c1] = 0
        if want_tgt and out_tgt is not None:
            tgt[c0:c1] = 0
        wi, ti, mi = np.nonzero(present)
        lin = ((wi + c0) * t1 + ti) * m + mi                              # row of w2
        dst = ((wi + c0) * T + ti) * max_num_obj + slot[wi, ti, mi]       # row of src2 for frame ti; frame ti feeds target row ti - 1
        s_ok = ti < T
        src2[dst[s_ok]] = w2[lin[s_ok]]
        if want_tgt:
            t_ok = ti >= 1
            tgt2[dst[t_ok] - max_num_obj] = w2[lin[t_ok]]
    return src, tgt


class DataLoader(object):
    """Drop-in for utils/data_loader.py:20 (same positional arguments and defaults)."""

    def __init__(self, batch_size=50, seq_length=5, max_num_obj=40, leave_dataset=1,
                 preprocess=False, data_dir: str = "data/",
                 frames: Optional[Sequence[np.ndarray]] = None, traj_bin: Optional[str] = None,
                 fix_id0: bool = False):
        self.leave_dataset = leave_dataset
        self.data_dir = data_dir
        self.frame_pointer = 0
        self.dataset_pointer = 0
        self.max_num_obj = max_num_obj
        self.batch_size = batch_size
        self.seq_length = seq_length
        self.fix_id0 = fix_id0
        if traj_bin is not None:                     # memory-mapped DSRTRJ1 container (desire_amd/formats.py)
            from .formats import read_traj_bin
            frames = read_traj_bin(traj_bin)
            if frames[0].shape[1] != max_num_obj:
                raise ValueError("traj_bin holds max_num_obj=%d, loader asked for %d" % (frames[0].shape[1], max_num_obj))
        if frames is not None:                       # already-preprocessed (frames, MNO, 3) arrays
            self.raw_data = ([f if isinstance(f, np.memmap) else np.asarray(f, np.float64) for f in frames],
                             [list(range(len(f))) for f in frames],
                             [[int((fr[:, 0] != 0).sum()) for fr in f] for f in frames])
            self._index()
        else:
            data_file = os.path.join(self.data_dir, "trajectories.cpkl")
            if preprocess or not os.path.exists(data_file) or self._has_csv():
                self.frame_preprocess(data_file)
            self.load_preprocessed(data_file)
        self.reset_batch_pointer()

    def _csv_paths(self) -> List[str]:
        paths = []
        for subdir, dirs, files in os.walk(self.data_dir):
            dirs.sort()
            for f in sorted(files):
                if f == "annotations_processed.csv":
                    paths.append(os.path.join(subdir, f))
        return paths

    def _has_csv(self) -> bool:
        return len(self._csv_paths()) > 0

    def frame_preprocess(self, data_file):
        all_frame_data, frame_list_data, num_obj_data = [], [], []
        for path in self._csv_paths()[: self.leave_dataset]:   # flag used as a COUNT (:91)
            data = np.genfromtxt(path, delimiter=",")
            arr, fl, no = frames_from_csv(data, self.max_num_obj, self.fix_id0)
            all_frame_data.append(arr)
            frame_list_data.append(fl)
            num_obj_data.append(no)
        # written beside the target and renamed into place: several ranks preprocess the same directory at start-up
        # (train.py under torchrun), and a reader must never see a half-written pickle
        tmp = "%s.%d.tmp" % (data_file, os.getpid())
        with open(tmp, "wb") as fh:
            pickle.dump((all_frame_data, frame_list_data, num_obj_data), fh, protocol=2)
        os.replace(tmp, data_file)

    def load_preprocessed(self, data_file):
        with open(data_file, "rb") as fh:
            self.raw_data = pickle.load(fh)
        self._index()

    def _index(self):
        self.data = self.raw_data[0]
        self.frame_list = self.raw_data[1]
        self.num_obj_list = self.raw_data[2]
        counter = 0
        for all_frame_data in self.data:
            counter += int(len(all_frame_data) / (self.seq_length + 2))
        self.num_batches = int(counter / self.batch_size) * 2

    def next_batch(self, random_update=True):
        """utils/data_loader.py:185-247.  The pointer walk (and its random.randint draws) runs first, in the reference's order; the
        windows are then cut and slot-assigned in ONE batched pass (windows_to_slots_batch).  If that pass declines -- a window in
        which the reference would raise, or ids that are not small integers -- the walk is rewound and redone window by window, so
        the exception, its type and the pointer state it leaves behind are the reference's."""
        state = (self.dataset_pointer, self.frame_pointer, random.getstate() if random_update else None)
        picks, dval = self._walk(random_update)
        out = self._fill(picks, None, None)
        if out is None:
            self.dataset_pointer, self.frame_pointer = state[0], state[1]
            if random_update:
                random.setstate(state[2])
            return self._next_batch_scalar(random_update)
        return list(out[0]), list(out[1]), dval

    def _walk(self, random_update):
        """The pointer walk of utils/data_loader.py:190-247 without the window work: [(video, first frame)] and d."""
        picks, dval = [], []
        guard = 0
        while len(picks) < self.batch_size:
            current_data = self.data[self.dataset_pointer]
            idx = self.frame_pointer
            if idx + self.seq_length < current_data.shape[0]:
                picks.append((self.dataset_pointer, idx))
                if random_update:
                    self.frame_pointer += random.randint(1, self.seq_length)
                else:
                    self.frame_pointer += self.seq_length
                dval.append(self.dataset_pointer)
                guard = 0
            else:
                self.tick_batch_pointer()
                guard += 1
                if guard > len(self.data):
                    raise RuntimeError("no video holds seq_length+1 frames")  # the ref spins forever
        return picks, dval

    def _fill(self, picks, out_x, out_y):
        T = self.seq_length
        win = getattr(self, "_cut", None)                        # persistent cut buffer: no fresh 19 MB allocation per batch
        if win is None or win.shape[0] != len(picks):
            win = self._cut = np.empty((len(picks), T + 1, self.max_num_obj, 3))
        for i, (v, idx) in enumerate(picks):
            win[i] = self.data[v][idx:idx + T + 1]
        return windows_to_slots_batch(win, T, self.max_num_obj, out_x, out_y)

    def next_batch_into(self, out_x: np.ndarray, out_y=False, random_update=True):
        """next_batch() writing its x (and, with out_y an array, its y) straight into caller-owned [batch_size, seq_length, max_num_obj, 3]
        buffers of any float dtype -- the prefetcher's pinned float32 staging -- instead of returning fresh float64 lists.  Same pointer
        walk, same random draws, same windows, same exceptions as next_batch (one rounding to the buffer's dtype).  Returns d."""
        state = (self.dataset_pointer, self.frame_pointer, random.getstate() if random_update else None)
        picks, dval = self._walk(random_update)
        if self._fill(picks, out_x, out_y) is None:
            self.dataset_pointer, self.frame_pointer = state[0], state[1]
            if random_update:
                random.setstate(state[2])
            x, y, dval = self._next_batch_scalar(random_update)
            out_x[...] = np.stack(x)
            if out_y is not False:
                out_y[...] = np.stack(y)
        return dval

    def _next_batch_scalar(self, random_update=True):
        """The per-window walk (the reference's own loop structure): exceptions surface at the window, and with the pointer state, the
        reference would have."""
        x_batch, y_batch, dval = [], [], []
        i = 0
        guard = 0
        while i < self.batch_size:
            current_data = self.data[self.dataset_pointer]
            idx = self.frame_pointer
            if idx + self.seq_length < current_data.shape[0]:
                window = current_data[idx:idx + self.seq_length + 1]
                src, tgt = window_to_slots(window, self.seq_length, self.max_num_obj)
                x_batch.append(src)
                y_batch.append(tgt)
                if random_update:
                    self.frame_pointer += random.randint(1, self.seq_length)
                else:
                    self.frame_pointer += self.seq_length
                dval.append(self.dataset_pointer)
                i += 1
                guard = 0
            else:
                self.tick_batch_pointer()
                guard += 1
                if guard > len(self.data):
                    raise RuntimeError("no video holds seq_length+1 frames")  # the ref spins forever
        return x_batch, y_batch, dval

    def tick_batch_pointer(self):
        self.dataset_pointer += 1
        self.frame_pointer = 0
        if self.dataset_pointer >= len(self.data):
            self.dataset_pointer = 0

    def reset_batch_pointer(self):
        self.dataset_pointer = 0
        self.frame_pointer = 0
