"""Loader in the loop (/root/reference/train.py:131-183 is the serial original: next_batch -> feed -> sess.run, one
after the other, every step).

`WindowFeeder` runs the DataLoader on a background thread and keeps the GPU fed:

    loader.next_batch_into(pinned float32 staging)      host thread, numpy (releases the GIL in its array passes)
      -> cudaMemcpyAsync on a COPY stream               pinned -> device, overlaps the compute stream's kernels
      -> past / future split on the copy stream         two small device copies into contiguous [n, T_obs, mno, 3] / [n, T_pred, mno, 3]
      -> event                                          the consumer's stream waits on it; no host synchronisation anywhere

with `depth` batches in flight (double buffering by default).  A slot is reused only after the consumer has recorded that its step no
longer reads it (`Batch.release()` records an event on the compute stream; the copy stream waits on it before overwriting).  The feeder
is the ONLY consumer of the loader and of the `random` module's state while it runs, so it produces exactly the batches the serial loop
would (tests/test_prefetch.py, tests/test_gpu_prefetch.py).

`DeviceWindowFeeder` is the same interface over the device-side window builder (`desire_build_windows_la`): the preprocessed videos are
resident in HBM, the host only walks the loader's pointers (DataLoader._walk: same random draws) and ships the window starts.
"""
from __future__ import annotations

import queue
import threading
from typing import Iterator, List, Optional, Sequence, Tuple

import numpy as np


class Batch(object):
    """One step's input, already on the device.  `past` [n, T_obs, mno, 3], `fut` [n, T_pred, mno, 3] float32 (loader layout: id, x_px,
    y_px); `d` = next_batch's video indices; `epoch`, `index` = position in the schedule.  wait(): make the current stream wait for
    the copies; release(): tell the feeder the current stream's work enqueued so far is the last reader of these buffers."""

    def __init__(self, feeder, slot, past, fut, d, epoch, index, ready):
        self._feeder, self._slot = feeder, slot
        self.past, self.fut, self.d, self.epoch, self.index, self._ready = past, fut, d, epoch, index, ready

    def wait(self, stream=None) -> None:
        if self._ready is not None:
            (stream or self._feeder.torch.cuda.current_stream()).wait_event(self._ready)

    def release(self, stream=None) -> None:
        self._feeder._release(self._slot, stream)


class _FeederBase(object):
    def __init__(self, device, depth: int):
        import torch
        self.torch = torch
        self.device = device
        self.depth = max(1, int(depth))
        self._q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        self._free: "queue.Queue" = queue.Queue()
        self._stop = threading.Event()
        self._err: Optional[BaseException] = None
        self._thread: Optional[threading.Thread] = None
        self._cuda = device is not None and torch.device(device).type == "cuda"
        if self._cuda:                                   # an explicit index: the producer thread has to select the device itself
            dv = torch.device(device)
            device = self.device = torch.device("cuda", torch.cuda.current_device() if dv.index is None else dv.index)
        self._copy_stream = torch.cuda.Stream(device=device) if self._cuda else None
        self._done_ev = {}

    def _fence_allocation(self) -> None:
        """The device buffers were zero-filled on the CONSTRUCTING thread's current stream; the copy stream is independent of it, so
        without this its first uploads can be overtaken by those fills (seen: the first batches of a run arrived as zeros)."""
        if self._cuda:
            ev = self.torch.cuda.Event()
            ev.record(self.torch.cuda.current_stream())
            self._copy_stream.wait_event(ev)

    # -- consumer side --------------------------------------------------------------------------
    def __iter__(self) -> Iterator[Batch]:
        self.start()
        while True:
            item = self._q.get()
            if item is None:
                if self._err is not None:
                    raise self._err
                return
            yield item

    def start(self) -> None:
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, name="desire-window-feeder", daemon=True)
            self._thread.start()

    def close(self) -> None:
        self._stop.set()
        while self._thread is not None and self._thread.is_alive():
            try:                                        # unblock a producer waiting on a full queue / an empty free list
                self._q.get_nowait()
            except queue.Empty:
                pass
            self._free.put(None)
            self._thread.join(timeout=0.05)
        self._thread = None

    def _release(self, slot: int, stream=None) -> None:
        if self._cuda:
            ev = self.torch.cuda.Event()
            ev.record(stream or self.torch.cuda.current_stream())
            self._done_ev[slot] = ev
        self._free.put(slot)

    # -- producer side --------------------------------------------------------------------------
    def _run(self) -> None:
        try:
            if self._cuda:
                self.torch.cuda.set_device(self.device)
            self._produce()
        except BaseException as e:                       # noqa: BLE001 -- re-raised in the consumer
            self._err = e
        finally:
            self._q.put(None)

    def _take_slot(self) -> Optional[int]:
        slot = self._free.get()
        if slot is None or self._stop.is_set():
            return None
        ev = self._done_ev.pop(slot, None)
        if ev is not None:
            self._copy_stream.wait_event(ev)             # the step that read this slot has finished with it (device-side wait)
        return slot

    def _produce(self) -> None:
        raise NotImplementedError


class WindowFeeder(_FeederBase):
    """Host loader -> pinned staging -> copy stream -> device.  `loader`: desire_amd.data_loader.DataLoader whose windows are
    seq_length = t_obs + t_pred frames; `num_epochs` x (reset_batch_pointer, num_batches x next_batch) is the reference loop's schedule
    (train.py:122-131); `shard = (rank, world)` keeps this rank's block of every batch (dist.shard_batch's partition); `mno` pads the slot
    axis up to the model's tile (zeros = absent).  `max_batches` stops early (train.py --max_steps)."""

    def __init__(self, loader, t_obs: int, t_pred: int, device=None, depth: int = 2, num_epochs: int = 1, shard: Tuple[int, int] = (0, 1),
                 mno: Optional[int] = None, random_update: bool = True, max_batches: int = 0):
        super().__init__(device, depth)
        if loader.seq_length != t_obs + t_pred:
            raise ValueError("the loader window must be t_obs + t_pred frames")
        torch = self.torch
        self.loader, self.t_obs, self.t_pred = loader, int(t_obs), int(t_pred)
        self.num_epochs, self.random_update, self.max_batches = int(num_epochs), bool(random_update), int(max_batches)
        n, T, m_in = loader.batch_size, loader.seq_length, loader.max_num_obj
        self.mno = int(mno or m_in)
        if self.mno < m_in:
            raise ValueError("mno %d is smaller than the loader's max_num_obj %d" % (self.mno, m_in))
        rank, world = shard
        from .dist import shard_windows
        self.lo, self.hi = shard_windows(n, rank, world)
        k = self.hi - self.lo
        self._stage, self._full, self._past, self._fut = [], [], [], []
        self._staged_ready = [None] * (self.depth + 1)   # per slot: the event behind the last copy out of its pinned staging buffer
        for _ in range(self.depth + 1):
            st = torch.zeros((n, T, m_in, 3), dtype=torch.float32)
            if self._cuda:
                st = st.pin_memory()
            self._stage.append(st)
            dev = device if self._cuda else "cpu"
            self._full.append(torch.zeros((k, T, m_in, 3), dtype=torch.float32, device=dev))
            self._past.append(torch.zeros((k, self.t_obs, self.mno, 3), dtype=torch.float32, device=dev))
            self._fut.append(torch.zeros((k, self.t_pred, self.mno, 3), dtype=torch.float32, device=dev))
        for i in range(self.depth + 1):
            self._free.put(i)
        self._fence_allocation()

    def _produce(self) -> None:
        torch = self.torch
        m_in = self.loader.max_num_obj
        count = 0
        for epoch in range(self.num_epochs):
            self.loader.reset_batch_pointer()
            for b in range(self.loader.num_batches):
                slot = self._take_slot()
                if slot is None:
                    return
                st = self._stage[slot]
                # the H2D copy that last read this staging buffer may still be queued (the consumer's release only makes the COPY STREAM wait; a
                # consumer that never host-syncs can run ahead of the GPU): the host must not rewrite the buffer before that copy has finished
                prev = self._staged_ready[slot]
                if prev is not None:
                    prev.synchronize()
                d = self.loader.next_batch_into(st.numpy(), False, self.random_update)
                ready = None
                if self._cuda:
                    with torch.cuda.stream(self._copy_stream):
                        self._full[slot].copy_(st[self.lo:self.hi], non_blocking=True)
                        self._past[slot][:, :, :m_in].copy_(self._full[slot][:, :self.t_obs])
                        self._fut[slot][:, :, :m_in].copy_(self._full[slot][:, self.t_obs:])
                        ready = torch.cuda.Event()
                        ready.record(self._copy_stream)
                        self._staged_ready[slot] = ready
                else:                                    # CPU (tests of the schedule and of the batches; no GPU here)
                    self._past[slot][:, :, :m_in].copy_(st[self.lo:self.hi, :self.t_obs])
                    self._fut[slot][:, :, :m_in].copy_(st[self.lo:self.hi, self.t_obs:])
                self._q.put(Batch(self, slot, self._past[slot], self._fut[slot], d[self.lo:self.hi], epoch, b, ready))
                count += 1
                if self._stop.is_set() or (self.max_batches and count >= self.max_batches):
                    return


class DeviceWindowFeeder(_FeederBase):
    """Windows cut and slot-assigned ON the device (desire_build_windows_la, bit-exact with the loader's x): the videos are uploaded once,
    per batch the host walks the loader's pointers (same random draws as next_batch) and ships n window starts.  One builder call per
    video that contributes windows to the batch.  `handle`: a _lib.Handle whose dims give T_obs / T_pred / mno."""

    def __init__(self, loader, handle, device, depth: int = 2, num_epochs: int = 1, random_update: bool = True, max_batches: int = 0):
        super().__init__(device, depth)
        torch = self.torch
        d = handle.dims
        if loader.seq_length != d.T_obs + d.T_pred:
            raise ValueError("the loader window must be T_obs + T_pred frames")
        self.loader, self.handle = loader, handle
        self.num_epochs, self.random_update, self.max_batches = int(num_epochs), bool(random_update), int(max_batches)
        self.videos = [torch.as_tensor(np.ascontiguousarray(np.asarray(v), np.float32), device=device) for v in loader.data]
        n = loader.batch_size
        self._past = [torch.zeros((n, d.T_obs, d.mno, 3), dtype=torch.float32, device=device) for _ in range(self.depth + 1)]
        self._fut = [torch.zeros((n, d.T_pred, d.mno, 3), dtype=torch.float32, device=device) for _ in range(self.depth + 1)]
        for i in range(self.depth + 1):
            self._free.put(i)
        self._fence_allocation()

    def _produce(self) -> None:
        torch = self.torch
        count = 0
        for epoch in range(self.num_epochs):
            self.loader.reset_batch_pointer()
            for b in range(self.loader.num_batches):
                slot = self._take_slot()
                if slot is None:
                    return
                picks, dval = self.loader._walk(self.random_update)
                with torch.cuda.stream(self._copy_stream):
                    st = self._copy_stream.cuda_stream
                    i = 0
                    while i < len(picks):                # runs of consecutive windows from one video -> one builder call each
                        j = i
                        while j < len(picks) and picks[j][0] == picks[i][0]:
                            j += 1
                        v = self.videos[picks[i][0]]
                        starts = [p[1] for p in picks[i:j]]
                        self.handle.build_windows(v.data_ptr(), v.shape[0], v.shape[1], starts, self._past[slot][i:j].data_ptr(),
                                                  self._fut[slot][i:j].data_ptr(), st, lookahead=1)
                        i = j
                    ready = torch.cuda.Event()
                    ready.record(self._copy_stream)
                self._q.put(Batch(self, slot, self._past[slot], self._fut[slot], dval, epoch, b, ready))
                count += 1
                if self._stop.is_set() or (self.max_batches and count >= self.max_batches):
                    return


def serial_batches(loader, t_obs: int, num_epochs: int = 1, shard: Tuple[int, int] = (0, 1), random_update: bool = True,
                   max_batches: int = 0) -> Iterator[Tuple[np.ndarray, np.ndarray, List[int], int, int]]:
    """The schedule WindowFeeder follows, produced serially with next_batch() (the reference loop's order: train.py:122-183): yields
    (past [k, t_obs, M, 3], fut [k, T - t_obs, M, 3], d, epoch, index) float32 -- what tests compare the feeders against."""
    from .dist import shard_windows
    lo, hi = shard_windows(loader.batch_size, *shard)
    count = 0
    for epoch in range(num_epochs):
        loader.reset_batch_pointer()
        for b in range(loader.num_batches):
            x, _, d = loader.next_batch(random_update)
            xs = np.stack(x[lo:hi]).astype(np.float32)
            yield xs[:, :t_obs], xs[:, t_obs:], d[lo:hi], epoch, b
            count += 1
            if max_batches and count >= max_batches:
                return
